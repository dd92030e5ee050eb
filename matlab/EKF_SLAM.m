classdef EKF_SLAM < handle
    % Drop-in for the reference's EKF_SLAM (known correspondence) over libekfslam (MEX -> C ABI -> HIP).
    % Same constructor, properties (x P Q s C Rc s_cost s_thresh landmark_list observed -- all assignable, as in
    % the reference, EKF_SLAM.m:5-22) and methods (predict, f, append, measure, plot); x, P, s live on the GPU and are
    % fetched / stored on access.
    % NOT RUN under MATLAB in this repository's image (no MATLAB there) -- see INTEGRATION.md.
    properties (Dependent)
        x; P; Q; s;
    end
    properties
        C = 0.2; Rc = [.01, 5]; s_cost = 1e-11; s_thresh = 1e9;
        landmark_list; observed;
    end
    properties (Access = protected)
        hnd;
        Qassigned = [];     % a Q the caller assigned; predict() recomputes Q (EKF_SLAM.m:43-44), which drops it
    end
    methods
        function h = EKF_SLAM(capacity)
            if nargin < 1, capacity = 1024; end
            h.hnd = ekfslam_mex('create', h.abiMode(), capacity);
        end
        function delete(h), ekfslam_mex('destroy', h.hnd); end
        function v = get.x(h), v = ekfslam_mex('get_x', h.hnd); end
        function v = get.P(h), v = ekfslam_mex('get_P', h.hnd); end
        function v = get.s(h), v = ekfslam_mex('get_s', h.hnd); end
        function v = get.Q(h)
            if isempty(h.Qassigned), v = ekfslam_mex('get_Q', h.hnd); else, v = h.Qassigned; end
        end
        % assignment: x first (it fixes the number of landmarks), then s / P of matching size
        function set.x(h, v), ekfslam_mex('set_x', h.hnd, double(v(:))); end
        function set.P(h, v), ekfslam_mex('set_P', h.hnd, double(v)); end
        function set.s(h, v), ekfslam_mex('set_s', h.hnd, double(v(:))); end
        function set.Q(h, v), h.Qassigned = v; end
        function pushParams(h)   % forward the (re-assignable) tunables before each call that uses them
            ekfslam_mex('set_params', h.hnd, h.C, double(h.Rc(:)), h.s_cost, h.s_thresh, 0);
        end
        function predict(h, u)
            h.pushParams(); h.Qassigned = [];
            ekfslam_mex('predict', h.hnd, double(u(:)));
        end
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
        function B = covarianceBlock(h, r0, c0, nr, nc)   % P(r0:r0+nr-1, c0:c0+nc-1) without moving the rest of P
            B = ekfslam_mex('get_P_block', h.hnd, r0, c0, nr, nc);
        end
        function plot(h, landmark_list)
            % Same figure content as the reference's plot (robot, landmarks, the landmark source's own overlay, one
            % covariance ellipse per 2x2 diagonal block) from TWO device reads: x and the diagonal blocks
            % (the reference indexes the full h.P, EKF_SLAM.m:180,205 -- 3.2 GB at 10k landmarks).
            xs = h.x;
            B = reshape(ekfslam_mex('get_P_diag_blocks', h.hnd), 2, 2, []);
            hold on;
            if exist('drawRobot', 'file'), drawRobot(xs(1), xs(2), xs(3), 0.25); end
            if numel(xs) > 3, scatter(xs(4:2:end), xs(5:2:end), 'blue', 'x'); end
            if nargin > 1 && ~isempty(landmark_list)
                landmark_list.landmarkObj.plot(xs, h.observed);
            end
            EKF_SLAM.ellipse(B(:, :, 1), xs(1:2), 0.25);
            for k = 2:size(B, 3)
                EKF_SLAM.ellipse(B(:, :, k), xs(2 * k:2 * k + 1), 0.50);
            end
            hold off;
        end
    end
    methods (Static)
        function ellipse(Sigma, mu, shrink)
            % boundary of { mu + shrink * 2 * sqrt(chi2) * Sigma^(1/2) * [cos t; sin t] }
            if any(isnan(Sigma(:))), return; end            % block held by another shard
            [V, D] = eig((Sigma + Sigma') / 2);
            t = linspace(-pi, pi, 629);
            pts = V * (2 * sqrt(2.2788 * max(D, 0))) * [cos(t); sin(t)] * shrink;
            plot(pts(1, :) + mu(1), pts(2, :) + mu(2));
        end
    end
    methods (Access = protected)
        function m = abiMode(~), m = 0; end               % EKF_MODE_KNOWN
    end
end
