classdef Landmark < handle
    % Landmark source selector with the reference's surface (Landmark.m:12-33): Landmark(method) builds the source object in
    % .landmarkObj, getLandmark(laserdata, x) returns the observed landmark list [range, bearing_deg, index] the EKF classes consume
    % (EKF_SLAM.m:102).  Beside the reference's 'RANSAC' (its own RANSAC.m, which needs ROS LaserScan objects and the Symbolic
    % Toolbox: out of scope here, INTEGRATION.md) there is 'SYNTHETIC': landmarks of a known world (SyntheticLandmarks.m), so
    % that the SLAM loop -- SLAM.m:105-116: predict, measure, plot -- runs on a machine without a robot.  Same duck type either way:
    % .landmarkObj.landmark is the struct array of RANSAC.m:238-241 (loc, observe, index, fresh) that measure() indexes
    % (EKF_SLAM.m:111,119; EKF_SLAM_UC.m:123).
    properties
        landmarkObj;
        method;
    end
    methods
        function h = Landmark(method)
            h.method = method;
            switch method
                case 'SYNTHETIC'
                    h.landmarkObj = SyntheticLandmarks();
                case 'RANSAC'
                    h.landmarkObj = RANSAC();                     % the reference's class, if it is on the path
                otherwise
                    warning('Improper landmark recognition method.');   % (Landmark.m:19-21: falls back to RANSAC)
                    h.method = 'RANSAC';
                    h.landmarkObj = RANSAC();
            end
        end
        function observed_LL = getLandmark(h, laserdata, x)
            observed_LL = h.landmarkObj.getLandmark(laserdata, x);
        end
    end
end
