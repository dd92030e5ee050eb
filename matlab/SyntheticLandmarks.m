classdef SyntheticLandmarks < handle
    % The object behind Landmark('SYNTHETIC').landmarkObj: what RANSAC is to Landmark('RANSAC'), for a world whose landmarks
    % are known.  "laserdata" is a k-by-3 matrix [world_id, range, bearing_deg] of the landmarks in sight (what a simulator or
    % ekf_slam_amd/world.py::World.observe produces), not a ROS LaserScan.
    %   .landmark            struct array as RANSAC.m:238-241 builds it: loc (1x2, world frame), observe (count), index, fresh
    %   getLandmark(d, x)    each row: loc = x(1:2) + range * [cosd; sind](bearing + x(3)); a world_id seen for the first time
    %                        becomes a new entry with index = largest index so far + 1, a known one has its loc refreshed and its
    %                        count raised; returns [range, bearing_deg, index], rows sorted by index (EKF_SLAM.m:107-123 walks them
    %                        in that order and, with known correspondence, corrects landmark ii of row ii)
    %   plot(x, observed)    nothing to draw (EKF_SLAM.m:167 calls it)
    % The Python twin, tested on the GPU against the oracle's: ekf_slam_amd/world.py::SyntheticLandmark.
    properties
        landmark = struct('loc', {}, 'observe', {}, 'index', {}, 'fresh', {});
    end
    properties (Access = private)
        world_ids = zeros(1, 0);          % world_ids(k): the simulator's id of landmark(k)
    end
    methods
        function observed_LL = getLandmark(h, laserdata, x)
            k = size(laserdata, 1);
            observed_LL = zeros(k, 3);
            for ii = 1:k
                wid = laserdata(ii, 1); r = laserdata(ii, 2); b = laserdata(ii, 3);
                loc = [x(1) + r * cosd(b + x(3)), x(2) + r * sind(b + x(3))];
                at = find(h.world_ids == wid, 1);
                if isempty(at)
                    if isempty(h.landmark), nxt = 1; else, nxt = max([h.landmark.index]) + 1; end
                    at = numel(h.landmark) + 1;
                    h.landmark(at).loc = loc;
                    h.landmark(at).observe = 1;
                    h.landmark(at).index = nxt;
                    h.landmark(at).fresh = 0;
                    h.world_ids(at) = wid;
                else
                    h.landmark(at).loc = loc;
                    h.landmark(at).observe = h.landmark(at).observe + 1;
                end
                observed_LL(ii, :) = [r, b, h.landmark(at).index];
            end
            observed_LL = sortrows(observed_LL, 3);
        end
        function plot(~, ~, ~)
        end
    end
end
