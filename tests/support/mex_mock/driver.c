/* TEST SUPPORT: drives matlab/ekfslam_mex.c through the MEX mock with the argument shapes the .m classes in matlab/ pass (one call per
 * command / shape) and prints what reached the C ABI (abi_stub.c) and what came back.  tests/test_mex_gateway_cpu.py reads
 * the transcript.  A crash here (e.g. a NULL handle dereferenced) fails the test by the process dying. */
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "ekfslam.h"
#include "mex_mock.h"

void stub_fail_next(ekf_handle *h);

static mxArray *out[4];

/* returns 0 on success, 1 if the gateway raised a MATLAB error */
static int call(int nlhs, int nrhs, ...) {
    const mxArray *prhs[12];
    va_list ap;
    va_start(ap, nrhs);
    for (int i = 0; i < nrhs; ++i) prhs[i] = va_arg(ap, const mxArray *);
    va_end(ap);
    char cmd[32] = "?";
    mxGetString(prhs[0], cmd, sizeof cmd);
    for (int i = 0; i < 4; ++i) out[i] = NULL;
    if (setjmp(mock_err_jmp)) { printf("MEX %s nrhs=%d -> ERROR %s | %s\n", cmd, nrhs, mock_err_id, mock_err_msg); return 1; }
    mexFunction(nlhs, out, nrhs, prhs);
    printf("MEX %s nrhs=%d -> ok", cmd, nrhs);
    for (int i = 0; i < 4 && out[i]; ++i) {
        printf(" out%d=%zux%zu%s[", i, mxGetM(out[i]), mxGetN(out[i]), mock_is_logical(out[i]) ? "L" : "");
        const size_t ne = mxGetNumberOfElements(out[i]);
        if (mxGetClassID(out[i]) == mxDOUBLE_CLASS) for (size_t k = 0; k < ne && k < 6; ++k) printf("%s%g", k ? "," : "", mxGetPr(out[i])[k]);
        else printf("handle");
        printf("]");
    }
    printf("\n");
    return 0;
}

#define S(x) mock_string(x)
#define D1(v) mock_double(1, 1, (const double[]){ v })

int main(void) {
    const double u[2] = { 0.1, 3.0 }, R[4] = { 0.05, 0, 0, 250 }, pos[2] = { 4, 5 }, z2[2] = { 5, 50 }, z3[3] = { 5, 50, 7 };
    const double x5[5] = { 1, 2, 3, 4, 5 }, P5[25] = { 9 }, s1[1] = { 42 }, Rc[2] = { .1, 5 };
    const double obs[6] = { 5, 6, 40, 50, 1, 2 };           /* 2 x 3 column-major: rows [5 40 1], [6 50 2] */
    const double lidx[3] = { 1, 2, 3 }, lloc[6] = { 10, 11, 12, 20, 21, 22 };
    mxArray *empty = mock_double(0, 0, NULL);

    /* commands that take no handle */
    if (call(1, 3, S("create"), D1(0), D1(16))) return 2;
    mxArray *h = out[0];
    printf("LOCKS %d\n", mock_lock_count);
    call(2, 4, S("f"), empty, mock_double(1, 5, x5), mock_double(2, 1, u));           /* EKF_SLAM.f passes [] as the handle */
    call(1, 4, S("f"), empty, mock_double(1, 5, x5), mock_double(2, 1, u));
    call(1, 5, S("create"), D1(1), D1(8), D1(64), D1(4));
    mxArray *h2 = out[0];
    call(1, 3, S("create"), D1(1), D1(666));                                          /* ekf_create fails: error, handle freed */
    call(1, 3, S("create"), D1(7), D1(8));                                            /* bad mode */

    /* the hot path */
    call(0, 7, S("set_params"), h, D1(0.2), mock_double(2, 1, Rc), D1(1e-11), D1(1e9), D1(0));
    call(0, 3, S("predict"), h, mock_double(2, 1, u));
    call(0, 6, S("append"), h, mock_double(2, 1, u), mock_double(2, 2, R), mock_double(2, 1, pos), D1(1));
    call(0, 5, S("correct"), h, mock_double(2, 1, z2), mock_double(2, 2, R), D1(1));  /* 1-based -> idx0 = 0 */
    call(2, 4, S("associate"), h, mock_double(3, 1, z3), mock_double(2, 2, R));       /* stub says idx0 = 6 -> 7 */
    call(0, 6, S("measure"), h, mock_double(2, 3, obs), mock_double(2, 1, u), mock_double(3, 1, lidx), mock_double(3, 2, lloc));
    /* state access */
    call(1, 2, S("get_x"), h); call(1, 2, S("get_P"), h); call(1, 2, S("get_s"), h); call(1, 2, S("get_Q"), h);
    call(1, 6, S("get_P_block"), h, D1(4), D1(4), D1(2), D1(2));                      /* 1-based corner -> r0 = c0 = 3 */
    call(1, 2, S("get_P_diag_blocks"), h);
    call(0, 5, S("set_state"), h, mock_double(5, 1, x5), mock_double(5, 5, P5), mock_double(1, 1, s1));
    call(0, 3, S("set_x"), h, mock_double(5, 1, x5));
    call(0, 3, S("set_P"), h, mock_double(5, 5, P5));
    call(0, 3, S("set_s"), h, mock_double(1, 1, s1));
    /* a failing ABI call becomes a MATLAB error carrying ekf_last_error */
    stub_fail_next((ekf_handle *)(uintptr_t)*(uint64_t *)mxGetData(h));
    call(0, 3, S("predict"), h, mock_double(2, 1, u));
    /* misuse: must be MATLAB errors, never a dereference */
    call(0, 3, S("predict"), empty, mock_double(2, 1, u));
    call(0, 3, S("predict"), D1(12345), mock_double(2, 1, u));                        /* a double is not a handle */
    call(0, 3, S("predict"), mock_uint64(0), mock_double(2, 1, u));
    call(1, 1, S("get_x"));
    call(0, 2, S("predict"), h);                                                      /* too few arguments */
    call(0, 2, S("no_such_command"), h);
    /* one filter on two GPUs, one host thread (matlab/ShardedEKF.m): begin on every shard, exchange_local, finish on every shard */
    call(1, 8, S("create"), D1(1), D1(32), D1(0), D1(1), D1(0), D1(0), D1(2));
    mxArray *g0 = out[0];
    call(1, 8, S("create"), D1(1), D1(32), D1(0), D1(1), D1(1), D1(1), D1(2));
    mxArray *g1 = out[0];
    const uint64_t hv[2] = { *(uint64_t *)mxGetData(g0), *(uint64_t *)mxGetData(g1) }, hbad[2] = { hv[0], 0 };
    mxArray *gv = mock_uint64_vec(2, hv);
    call(0, 3, S("hint_next"), g0, D1(7));
    call(0, 3, S("hint_next"), g1, D1(7));
    call(0, 5, S("correct_begin"), g0, mock_double(2, 1, z2), mock_double(2, 2, R), D1(3));
    call(0, 5, S("correct_begin"), g1, mock_double(2, 1, z2), mock_double(2, 2, R), D1(3));
    call(0, 2, S("exchange_local"), gv);
    call(0, 2, S("correct_finish"), g0);
    call(0, 2, S("correct_finish"), g1);
    call(0, 4, S("associate_begin"), g0, mock_double(3, 1, z3), mock_double(2, 2, R));
    call(0, 4, S("associate_begin"), g1, mock_double(3, 1, z3), mock_double(2, 2, R));
    call(0, 2, S("exchange_local"), gv);
    call(2, 2, S("associate_finish"), g0);
    call(2, 2, S("associate_finish"), g1);
    call(0, 2, S("flush"), g0);
    call(0, 2, S("exchange_local"), mock_double(2, 1, u));                            /* not a handle vector */
    call(0, 2, S("exchange_local"), mock_uint64_vec(2, hbad));                        /* a null handle inside */
    call(0, 2, S("exchange_local"), empty);
    call(1, 10, S("create"), D1(0), D1(64), D1(256), D1(32), D1(0), D1(0), D1(1), D1(1), D1(1));   /* float tiles, F32-arithmetic pass */
    call(0, 2, S("destroy"), out[0]);
    call(1, 6, S("create"), D1(1), D1(32), D1(0), D1(1), D1(0));                      /* device without rank / world */
    call(0, 2, S("destroy"), g1);
    call(0, 2, S("destroy"), g0);
    call(0, 2, S("destroy"), h2);
    call(0, 2, S("destroy"), h);
    printf("LOCKS %d\nMISUSE %d\n", mock_lock_count, mock_misuse);
    return 0;
}
