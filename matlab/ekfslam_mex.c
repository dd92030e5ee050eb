/*
 * MEX gateway: MATLAB -> C ABI (include/ekfslam.h) -> HIP.   NOT COMPILED AGAINST MATLAB IN THIS REPOSITORY'S IMAGE
 * (no MATLAB / MathWorks mex.h there); build on a MATLAB host with
 *     mex -I../include ekfslam_mex.c -L../ekf_slam_amd -lekfslam
 * What IS run here: tests/test_mex_gateway_cpu.py compiles this file against a small mock of the documented MEX C API and a
 * recording stand-in for libekfslam, and drives every command with the argument shapes the .m classes pass.
 *
 * One entry point, string command first:   out = ekfslam_mex('command', handle, args...)
 * The handle travels as a uint64 scalar.  MATLAB's 1-based landmark indices are converted to the ABI's
 * 0-based ones here.  Any non-zero status becomes mexErrMsgIdAndTxt('ekfslam:status', ...), so the .m classes
 * see MATLAB errors exactly where the reference's own code would raise them.
 *
 * Commands that take NO single handle ('create', 'f', and 'exchange_local', whose second argument is a uint64 VECTOR of handles
 * that it validates itself) are dispatched before anything looks at prhs[1] as a handle; every other command goes through
 * handle_of(), which rejects an empty / non-uint64 / null handle with a MATLAB error instead of dereferencing it.
 */
#include <stdint.h>
#include <string.h>

#include "ekfslam.h"
#include "mex.h"

static void need(int nrhs, int want, const char *cmd) {
    if (nrhs < want) mexErrMsgIdAndTxt("ekfslam:usage", "'%s' needs %d arguments, got %d", cmd, want, nrhs);
}

static ekf_handle *handle_of(int nrhs, const mxArray *prhs[]) {
    if (nrhs < 2 || !prhs[1] || mxGetClassID(prhs[1]) != mxUINT64_CLASS || mxGetNumberOfElements(prhs[1]) != 1 ||
        !mxGetData(prhs[1]))
        mexErrMsgIdAndTxt("ekfslam:handle", "second argument must be the uint64 handle returned by 'create'");
    ekf_handle *h = (ekf_handle *)(uintptr_t)(*(const uint64_t *)mxGetData(prhs[1]));
    if (!h) mexErrMsgIdAndTxt("ekfslam:handle", "null handle (already destroyed?)");
    return h;
}

static void check(ekf_handle *h, int32_t rc) {
    if (rc != EKF_OK)
        mexErrMsgIdAndTxt("ekfslam:status", "%s: %s", ekf_status_string(rc), h ? ekf_last_error(h) : "");
}

static int64_t nstate(ekf_handle *h) { int64_t N; check(h, ekf_num_landmarks(h, &N)); return 3 + 2 * N; }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[32];
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("ekfslam:usage", "command string expected");

    /* ---- commands without a handle ---- */
    if (!strcmp(cmd, "create")) {                 /* h = ekfslam_mex('create', mode, capacity [, tile [, batch [, device, rank, world [, storage [, pass_arith]]]]]) */
        ekf_config cfg;
        ekf_handle *h = NULL;
        need(nrhs, 3, cmd);
        check(NULL, ekf_config_default(&cfg, (int32_t)mxGetScalar(prhs[1])));
        cfg.capacity_landmarks = (int64_t)mxGetScalar(prhs[2]);
        if (nrhs > 3) cfg.tile = (int32_t)mxGetScalar(prhs[3]);
        if (nrhs > 4) cfg.batch = (int32_t)mxGetScalar(prhs[4]);      /* deferred downdate, same results */
        if (nrhs > 5) {                                               /* one shard of a filter split over several GPUs */
            need(nrhs, 8, cmd);
            cfg.device = (int32_t)mxGetScalar(prhs[5]);
            cfg.rank = (int32_t)mxGetScalar(prhs[6]);
            cfg.world = (int32_t)mxGetScalar(prhs[7]);
        }
        if (nrhs > 8) cfg.storage = (int32_t)mxGetScalar(prhs[8]);    /* EKF_STORE_*: 1 = float tiles (BASELINE configs[4]) */
        if (nrhs > 9) cfg.pass_arith = (int32_t)mxGetScalar(prhs[9]); /* EKF_ARITH_*: 1 = the pass over float tiles in F32 arithmetic, 2 = in split arithmetic */
        int32_t rc = ekf_create(&cfg, &h);
        if (rc != EKF_OK) {
            char msg[256];
            strncpy(msg, h ? ekf_last_error(h) : "", sizeof msg - 1); msg[sizeof msg - 1] = 0;
            if (h) ekf_destroy(h);
            mexErrMsgIdAndTxt("ekfslam:status", "%s: %s", ekf_status_string(rc), msg);
        }
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)h;
        mexLock();
        return;
    }
    if (!strcmp(cmd, "f")) {                      /* [x_new, F] = ekfslam_mex('f', [], x, u): pure host function, prhs[1] unused */
        need(nrhs, 4, cmd);
        mwSize n = mxGetNumberOfElements(prhs[2]);
        if (n < 3 || mxGetNumberOfElements(prhs[3]) < 2) mexErrMsgIdAndTxt("ekfslam:usage", "f: x needs >= 3 and u 2 elements");
        plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);
        mxArray *F = mxCreateDoubleMatrix(n, n, mxREAL);
        check(NULL, ekf_motion_model(mxGetPr(prhs[2]), (int64_t)n, mxGetPr(prhs[3]), mxGetPr(plhs[0]), mxGetPr(F)));
        if (nlhs > 1) plhs[1] = F; else mxDestroyArray(F);
        return;
    }

    if (!strcmp(cmd, "exchange_local")) {         /* ekfslam_mex('exchange_local', handles): all shards of ONE filter, uint64 vector,
                                                     each between the same *_begin and *_finish (include/ekfslam.h, transport (c)) */
        ekf_handle *hs[64];
        need(nrhs, 2, cmd);
        if (!prhs[1] || mxGetClassID(prhs[1]) != mxUINT64_CLASS || !mxGetData(prhs[1]))
            mexErrMsgIdAndTxt("ekfslam:handle", "exchange_local: a uint64 vector of handles expected");
        const mwSize w = mxGetNumberOfElements(prhs[1]);
        if (w < 1 || w > 64) mexErrMsgIdAndTxt("ekfslam:usage", "exchange_local: between 1 and 64 handles");
        for (mwSize r = 0; r < w; ++r) {
            hs[r] = (ekf_handle *)(uintptr_t)((const uint64_t *)mxGetData(prhs[1]))[r];
            if (!hs[r]) mexErrMsgIdAndTxt("ekfslam:handle", "exchange_local: null handle in the vector");
        }
        check(hs[0], ekf_exchange_local(hs, (int32_t)w));
        return;
    }

    /* ---- everything below operates on a live handle ---- */
    ekf_handle *h = handle_of(nrhs, prhs);
    if (!strcmp(cmd, "destroy")) { ekf_destroy(h); mexUnlock(); return; }
    if (!strcmp(cmd, "set_params")) {             /* (h, C, Rc, s_cost, s_thresh, w_pos) */
        need(nrhs, 7, cmd);
        check(h, ekf_set_params(h, mxGetScalar(prhs[2]), mxGetPr(prhs[3]), mxGetScalar(prhs[4]), mxGetScalar(prhs[5]), mxGetScalar(prhs[6])));
        return;
    }
    if (!strcmp(cmd, "predict")) { need(nrhs, 3, cmd); check(h, ekf_predict(h, mxGetPr(prhs[2]))); return; }
    if (!strcmp(cmd, "append"))  { need(nrhs, 6, cmd); check(h, ekf_append(h, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetScalar(prhs[5]))); return; }
    if (!strcmp(cmd, "correct")) { need(nrhs, 5, cmd); check(h, ekf_correct(h, mxGetPr(prhs[2]), mxGetPr(prhs[3]), (int64_t)mxGetScalar(prhs[4]) - 1)); return; }
    if (!strcmp(cmd, "associate")) {              /* [newLL, index] = ... (z 1x3, R 2x2) ; index 1-based */
        int32_t is_new; int64_t idx;
        need(nrhs, 4, cmd);
        check(h, ekf_associate(h, mxGetPr(prhs[2]), mxGetPr(prhs[3]), &is_new, &idx, NULL, NULL));
        plhs[0] = mxCreateLogicalScalar(is_new != 0);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)(idx + 1));
        return;
    }
    /* sharded handles driven by one host thread: begin on every shard, 'exchange_local', finish on every shard */
    if (!strcmp(cmd, "correct_begin")) { need(nrhs, 5, cmd); check(h, ekf_correct_begin(h, mxGetPr(prhs[2]), mxGetPr(prhs[3]), (int64_t)mxGetScalar(prhs[4]) - 1)); return; }
    if (!strcmp(cmd, "correct_finish")) { check(h, ekf_correct_finish(h)); return; }
    if (!strcmp(cmd, "hint_next")) { need(nrhs, 3, cmd); check(h, ekf_hint_next(h, (int64_t)mxGetScalar(prhs[2]) - 1)); return; }   /* idx 1-based */
    if (!strcmp(cmd, "associate_begin")) {        /* (h, z 1x3, R 2x2): candidates of this shard into its send area */
        need(nrhs, 4, cmd);
        check(h, ekf_associate_begin(h, mxGetPr(prhs[2]), mxGetPr(prhs[3]), 0));
        return;
    }
    if (!strcmp(cmd, "associate_finish")) {       /* [newLL, index] = ...; index 1-based */
        int32_t is_new; int64_t idx;
        check(h, ekf_associate_finish(h, &is_new, &idx, NULL, NULL));
        plhs[0] = mxCreateLogicalScalar(is_new != 0);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)(idx + 1));
        return;
    }
    if (!strcmp(cmd, "flush")) { check(h, ekf_flush(h)); return; }
    if (!strcmp(cmd, "measure")) {                /* (h, observed_LL m x 3, u, lm_index L x 1, lm_loc L x 2) */
        need(nrhs, 6, cmd);
        check(h, ekf_measure(h, mxGetPr(prhs[2]), (int64_t)mxGetM(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]),
                             mxGetPr(prhs[5]), (int64_t)mxGetNumberOfElements(prhs[4])));
        return;
    }
    if (!strcmp(cmd, "get_x")) { int64_t n = nstate(h); plhs[0] = mxCreateDoubleMatrix(1, (mwSize)n, mxREAL); check(h, ekf_get_x(h, mxGetPr(plhs[0]))); return; }
    if (!strcmp(cmd, "get_P")) { int64_t n = nstate(h); plhs[0] = mxCreateDoubleMatrix((mwSize)n, (mwSize)n, mxREAL); check(h, ekf_get_P(h, mxGetPr(plhs[0]))); return; }
    if (!strcmp(cmd, "get_s")) { int64_t n = (nstate(h) - 3) / 2; plhs[0] = mxCreateDoubleMatrix((mwSize)n, 1, mxREAL); check(h, ekf_get_s(h, mxGetPr(plhs[0]))); return; }
    if (!strcmp(cmd, "get_Q")) { int64_t n = nstate(h); double q[9]; check(h, ekf_get_Q(h, q));
        plhs[0] = mxCreateDoubleMatrix((mwSize)n, (mwSize)n, mxREAL);      /* zeros(size(P)) with the 3x3 block */
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) mxGetPr(plhs[0])[c * n + r] = q[3 * c + r];
        return; }
    if (!strcmp(cmd, "get_P_block")) {            /* (h, r0, c0, nr, nc), 1-based corner */
        need(nrhs, 6, cmd);
        mwSize nr = (mwSize)mxGetScalar(prhs[4]), nc = (mwSize)mxGetScalar(prhs[5]);
        plhs[0] = mxCreateDoubleMatrix(nr, nc, mxREAL);
        check(h, ekf_get_P_block(h, (int64_t)mxGetScalar(prhs[2]) - 1, (int64_t)mxGetScalar(prhs[3]) - 1, nr, nc, mxGetPr(plhs[0])));
        return;
    }
    if (!strcmp(cmd, "get_P_diag_blocks")) {      /* 2 x 2 x (N+1): P(1:2,1:2), then every landmark's block -- what plot() reads */
        int64_t nb = (nstate(h) - 3) / 2 + 1;
        plhs[0] = mxCreateDoubleMatrix(4, (mwSize)nb, mxREAL);             /* the .m side reshapes to 2 x 2 x nb */
        check(h, ekf_get_P_diag_blocks(h, mxGetPr(plhs[0])));
        return;
    }
    if (!strcmp(cmd, "set_state")) {              /* (h, x, P, s) */
        need(nrhs, 5, cmd);
        check(h, ekf_set_x(h, mxGetPr(prhs[2]), (int64_t)mxGetNumberOfElements(prhs[2])));
        check(h, ekf_set_s(h, mxGetPr(prhs[4]), (int64_t)mxGetNumberOfElements(prhs[4])));
        check(h, ekf_set_P(h, mxGetPr(prhs[3]), (int64_t)mxGetM(prhs[3])));
        return;
    }
    /* the reference's x, P, s are plain assignable properties (EKF_SLAM.m:6-9) */
    if (!strcmp(cmd, "set_x")) { need(nrhs, 3, cmd); check(h, ekf_set_x(h, mxGetPr(prhs[2]), (int64_t)mxGetNumberOfElements(prhs[2]))); return; }
    if (!strcmp(cmd, "set_P")) { need(nrhs, 3, cmd); check(h, ekf_set_P(h, mxGetPr(prhs[2]), (int64_t)mxGetM(prhs[2]))); return; }
    if (!strcmp(cmd, "set_s")) { need(nrhs, 3, cmd); check(h, ekf_set_s(h, mxGetPr(prhs[2]), (int64_t)mxGetNumberOfElements(prhs[2]))); return; }
    mexErrMsgIdAndTxt("ekfslam:usage", "unknown command '%s'", cmd);
}
