classdef SyntheticSLAM < handle
    % The reference's SLAM facade (SLAM.m:17-68, 105-116) for a machine without ROS: same properties a caller touches (LM, slam,
    % algorithmName, u), same methods (predict, measure, plot, runSlam), the two rossubscribers replaced by a FEED -- a function handle
    % that returns one iteration's odometry increment and "laser data":
    %       [u, laserdata] = feed()        u = [delta_D, delta_theta_deg]  (what SLAM.m:108-110 computes from two odometry poses)
    %                                      laserdata = k-by-3 [world_id, range, bearing_deg]   (SyntheticLandmarks.m)
    % and Landmark('RANSAC') by Landmark('SYNTHETIC') (any method Landmark.m knows can be passed).  The reference's own SLAM.m keeps
    % working beside this file: it constructs EKF_SLAM / EKF_SLAM_UC, which are the GPU-backed classes of this directory.
    %   s = SyntheticSLAM('EKF_SLAM_UC', @myfeed);  for k = 1:1000, s.runSlam(); end;  s.slam.x(1:3)
    % Python twin, run on the GPU against the oracle: ekf_slam_amd/slam.py::SLAM.
    properties
        LM;
        slam;
        algorithmName;
        feed;
        u;
    end
    methods
        function h = SyntheticSLAM(inputString, feed, landmark_method, varargin)
            if nargin < 3, landmark_method = 'SYNTHETIC'; end
            h.algorithmName = inputString;
            h.feed = feed;
            h.u = [0; 0; 0];                                   % (SLAM.m:30)
            switch h.algorithmName
                case 'EKF_SLAM'
                    h.slam = EKF_SLAM(varargin{:});            % (varargin: capacity, tile, ... as matlab/EKF_SLAM.m takes them)
                case 'EKF_SLAM_UC'
                    h.slam = EKF_SLAM_UC(varargin{:});
                otherwise
                    h.slam = [];                               % (SLAM.m:37-39: silently nothing)
            end
            h.LM = Landmark(landmark_method);
        end
        function predict(h, u)
            if ~isempty(h.slam), h.slam.predict(u); end
        end
        function measure(h, laserdata, u)
            if ~isempty(h.slam), h.slam.measure(laserdata, u, h.LM); end
        end
        function plot(h)
            if ~isempty(h.slam), h.slam.plot(h.LM); end
        end
        function runSlam(h)
            % one SLAM iteration: predict, then measure (SLAM.m:105-116)
            [un, laserdata] = h.feed();
            h.u = un;
            h.slam.predict(h.u);
            h.slam.measure(laserdata, h.u, h.LM);
        end
    end
end
