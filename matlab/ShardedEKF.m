classdef ShardedEKF < handle
    % ONE filter whose covariance is split over several GPUs, driven by one MATLAB thread (INTEGRATION.md, section 4):
    % shard r lives on device devices(r); x, s, the robot rows of P are replicated, tile (I,J) of the landmark block is held
    % by shard mod(I+J, world).  predict / append go to every shard; a correction is begin (every shard extracts its part of
    % the landmark's row-panel) -> 'exchange_local' (device-to-device copies) -> finish (every shard solves and downdates
    % its tiles).  Results equal the single-GPU filter bit for bit (tests/test_sharded_gpu.py drives the same C ABI calls).
    % NOT RUN under MATLAB in this repository's image -- the gateway commands used here are exercised under the MEX mock.
    properties (SetAccess = protected)
        hnd;                  % uint64 column, one handle per shard
    end
    methods
        function g = ShardedEKF(mode, capacity, devices, tile, batch, storage, pass_arith)
            % storage: 0 = F64 tiles (default), 1 = F32 tiles with every solve in F64; pass_arith: 1 = the pass over F32 tiles in F32
            % arithmetic on the matrix pipe ("F32 mixed precision with F64 innovation solve"; include/ekfslam.h, cfg.pass_arith), 2 = the
            % same in split arithmetic (three bfloat16 pieces per float operand, bf16 matrix pipe: EKF_ARITH_SPLIT3)
            if nargin < 4, tile = 0; end
            if nargin < 5, batch = 1; end
            if nargin < 6, storage = 0; end
            if nargin < 7, pass_arith = 0; end
            w = numel(devices);
            g.hnd = zeros(w, 1, 'uint64');
            for r = 1:w
                g.hnd(r) = ekfslam_mex('create', mode, capacity, tile, batch, devices(r), r - 1, w, storage, pass_arith);
            end
        end
        function delete(g)
            for r = 1:numel(g.hnd), ekfslam_mex('destroy', g.hnd(r)); end
        end
        function setParams(g, C, Rc, s_cost, s_thresh, w_pos)
            for r = 1:numel(g.hnd), ekfslam_mex('set_params', g.hnd(r), C, double(Rc(:)), s_cost, s_thresh, w_pos); end
        end
        function predict(g, u)
            for r = 1:numel(g.hnd), ekfslam_mex('predict', g.hnd(r), double(u(:))); end
        end
        function append(g, u, R, landmarkPos, signature)
            for r = 1:numel(g.hnd)
                ekfslam_mex('append', g.hnd(r), double(u(:)), double(R), double(landmarkPos(:)), double(signature));
            end
        end
        function correct(g, z, R, idx, next_idx)   % correction body of measure() for landmark idx (1-based)
            % next_idx (optional): the landmark the NEXT correct() will name -- this correction's pass over P then also extracts that
            % landmark's row-panel (ekf_hint_next; result-neutral, saves the next step's extraction launch when batch == 1)
            if nargin > 4 && ~isempty(next_idx)
                for r = 1:numel(g.hnd), ekfslam_mex('hint_next', g.hnd(r), next_idx); end
            end
            for r = 1:numel(g.hnd), ekfslam_mex('correct_begin', g.hnd(r), double(z(:)), double(R), idx); end
            ekfslam_mex('exchange_local', g.hnd);
            for r = 1:numel(g.hnd), ekfslam_mex('correct_finish', g.hnd(r)); end
        end
        function [newLL, index] = associate(g, z, R)
            % estimateCorrespondence with the position cost in the likelihood (w_pos ~= 0): every shard scores the landmarks
            % whose diagonal block it holds, the candidates are exchanged, every shard takes the same arg-min.  With the
            % reference's signature-only likelihood ekfslam_mex('associate', g.hnd(1), z, R) on any one shard is enough.
            for r = 1:numel(g.hnd), ekfslam_mex('associate_begin', g.hnd(r), double(z(:)), double(R)); end
            ekfslam_mex('exchange_local', g.hnd);
            for r = 1:numel(g.hnd), [newLL, index] = ekfslam_mex('associate_finish', g.hnd(r)); end
        end
        function flush(g)
            for r = 1:numel(g.hnd), ekfslam_mex('flush', g.hnd(r)); end
        end
        function v = x(g), v = ekfslam_mex('get_x', g.hnd(1)); end       % replicated: any shard
        function v = s(g), v = ekfslam_mex('get_s', g.hnd(1)); end
    end
end
