/* TEST SUPPORT: driver-side view of the MEX mock (mex_mock.c). */
#ifndef EKF_TEST_MEX_MOCK_H
#define EKF_TEST_MEX_MOCK_H
#include <setjmp.h>
#include <stdint.h>
#include "mex.h"
extern jmp_buf mock_err_jmp;
extern char mock_err_id[64], mock_err_msg[512];
extern int mock_lock_count, mock_misuse;
mxArray *mock_double(size_t m, size_t n, const double *v);
mxArray *mock_string(const char *s);
mxArray *mock_uint64(uint64_t v);
mxArray *mock_uint64_vec(size_t n, const uint64_t *v);
int mock_is_logical(const mxArray *a);
#endif
