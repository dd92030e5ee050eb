classdef EKF_SLAM_UC < EKF_SLAM
    % Drop-in for the reference's EKF_SLAM_UC (unknown correspondence): Rc = [.1,5], owns a Correspondence,
    % measure() associates every observation (Correspondence.m:28-88) before append / correct -- inside ekf_measure,
    % with the s_cost / s_thresh of h.correspondence forwarded before each call.
    properties
        correspondence = Correspondence(1e-11, 1e9, 'EKF_SLAM_UC');
    end
    methods
        function h = EKF_SLAM_UC(varargin)
            h@EKF_SLAM(varargin{:});
            h.Rc = [.1, 5];
        end
        function pushParams(h)   % the UC class keeps its association constants in the Correspondence it owns (EKF_SLAM_UC.m:16)
            ekfslam_mex('set_params', h.hnd, h.C, double(h.Rc(:)), h.correspondence.s_cost, h.correspondence.s_thresh, 0);
        end
    end
    methods (Access = protected)
        function m = abiMode(~), m = 1; end               % EKF_MODE_UC
    end
end
