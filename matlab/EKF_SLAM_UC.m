classdef EKF_SLAM_UC < EKF_SLAM
    % Drop-in for the reference's EKF_SLAM_UC (unknown correspondence): Rc = [.1,5], owns a Correspondence,
    % measure() associates every observation on the device (Correspondence.m:28-88) before append / correct.
    properties
        correspondence = Correspondence(1e-11, 1e9, 'EKF_SLAM_UC');
    end
    methods
        function h = EKF_SLAM_UC(varargin)
            h@EKF_SLAM(varargin{:});
            h.Rc = [.1, 5];
        end
    end
    methods (Access = protected)
        function m = abiMode(~), m = 1; end               % EKF_MODE_UC
    end
end
