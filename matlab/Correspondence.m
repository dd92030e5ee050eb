classdef Correspondence
    % Drop-in for the reference's Correspondence value class; estimateCorrespondence runs on the GPU through a
    % scratch handle loaded with the caller's x, P, s and THIS object's s_cost / s_thresh (Correspondence.m:71,78).
    properties
        s_cost; s_thresh; method;
    end
    methods
        function h = Correspondence(cost, thresh, method)
            h.s_cost = cost; h.s_thresh = thresh; h.method = method;
            if ~strcmp(method, 'EKF_SLAM_UC')
                warning('Improper method specified. Using ML as default.');
                h.method = 'ML';
            end
        end
        function [newLL, index] = estimateCorrespondence(h, z, R, x, P, s)
            N = (length(x) - 3) / 2;
            tmp = ekfslam_mex('create', 1, max(N, 1));
            cleanup = onCleanup(@() ekfslam_mex('destroy', tmp));
            % C and Rc play no part in the association; w_pos = 0 is the reference's live likelihood (Correspondence.m:75)
            ekfslam_mex('set_params', tmp, 0.2, [.1; 5], double(h.s_cost), double(h.s_thresh), 0);
            ekfslam_mex('set_state', tmp, double(x(:)), double(P), double(s(:)));
            [newLL, index] = ekfslam_mex('associate', tmp, double(z(:)), double(R));
        end
    end
end
