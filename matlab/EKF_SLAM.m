classdef EKF_SLAM < handle
    % Drop-in for the reference's EKF_SLAM (known correspondence) over libekfslam (MEX -> C ABI -> HIP).
    % Same constructor, properties (x P Q s C Rc s_cost s_thresh landmark_list observed) and methods
    % (predict, f, append, measure); x, P, Q, s live on the GPU and are fetched on access.
    % NOT RUN in this repository's image (no MATLAB there) -- see INTEGRATION.md.
    properties (Dependent)
        x; P; Q; s;
    end
    properties
        C = 0.2; Rc = [.01, 5]; s_cost = 1e-11; s_thresh = 1e9;
        landmark_list; observed;
    end
    properties (Access = protected)
        hnd;
    end
    methods
        function h = EKF_SLAM(capacity)
            if nargin < 1, capacity = 1024; end
            h.hnd = ekfslam_mex('create', h.abiMode(), capacity);
        end
        function delete(h), ekfslam_mex('destroy', h.hnd); end
        function v = get.x(h), v = ekfslam_mex('get_x', h.hnd); end
        function v = get.P(h), v = ekfslam_mex('get_P', h.hnd); end
        function v = get.Q(h), v = ekfslam_mex('get_Q', h.hnd); end
        function v = get.s(h), v = ekfslam_mex('get_s', h.hnd); end
        function pushParams(h)   % forward the (re-assignable) tunables before each call that uses them
            ekfslam_mex('set_params', h.hnd, h.C, double(h.Rc(:)), h.s_cost, h.s_thresh, 0);
        end
        function predict(h, u), h.pushParams(); ekfslam_mex('predict', h.hnd, double(u(:))); end
        function [x_new, F] = f(~, x, u), [x_new, F] = ekfslam_mex('f', [], double(x), double(u(:))); end
        function append(h, u, R, landmarkPos, signature)
            ekfslam_mex('append', h.hnd, double(u(:)), double(R), double(landmarkPos(:)), double(signature));
        end
        function measure(h, laserData, u, landmark_list)
            observed_LL = landmark_list.getLandmark(laserData, h.x);
            h.observed = observed_LL;
            if ~isempty(observed_LL)
                h.pushParams();
                lm = landmark_list.landmarkObj.landmark;
                ekfslam_mex('measure', h.hnd, double(observed_LL), double(u(:)), double([lm.index]'), ...
                            double(reshape([lm.loc], 2, [])'));
            end
        end
        function B = covarianceBlock(h, r0, c0, nr, nc)   % what plot() reads: P(r0:r0+nr-1, c0:c0+nc-1)
            B = ekfslam_mex('get_P_block', h.hnd, r0, c0, nr, nc);
        end
    end
    methods (Access = protected)
        function m = abiMode(~), m = 0; end               % EKF_MODE_KNOWN
    end
end
