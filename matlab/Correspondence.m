classdef Correspondence
    % Drop-in for the reference's Correspondence value class; estimateCorrespondence runs on the GPU through a
    % scratch handle loaded with the caller's x, P, s.
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
        function [newLL, index] = estimateCorrespondence(~, z, R, x, P, s)
            N = (length(x) - 3) / 2;
            tmp = ekfslam_mex('create', 1, max(N, 1));
            cleanup = onCleanup(@() ekfslam_mex('destroy', tmp));
            ekfslam_mex('set_state', tmp, double(x), double(P), double(s));
            [newLL, index] = ekfslam_mex('associate', tmp, double(z(:)), double(R));
        end
    end
end
