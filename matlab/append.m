function [x, P] = append(x, P, u, idx, R, pos)
% Drop-in for the reference's free function append(x,P,u,idx,R,pos): grows the state only if the map holds
% fewer than idx landmarks; the arithmetic runs on the GPU through a scratch handle.
N = (length(x) - 3) / 2;
if N < idx
    tmp = ekfslam_mex('create', 0, N + 1);
    cleanup = onCleanup(@() ekfslam_mex('destroy', tmp));
    ekfslam_mex('set_state', tmp, double(x), double(P), zeros(N, 1));
    ekfslam_mex('append', tmp, double(u(:)), double(R), double(pos(:)), 0);
    x = ekfslam_mex('get_x', tmp);
    P = ekfslam_mex('get_P', tmp);
end
end
