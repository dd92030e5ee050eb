/* TEST SUPPORT: a few dozen lines standing in for the MEX C-API subset of ../mex_api_subset/mex.h, so that
 * matlab/ekfslam_mex.c can be EXECUTED on a machine without MATLAB.  Semantics follow the public API documentation where the
 * gateway depends on them: an empty array has no data (mxGetPr / mxGetData return NULL), mxGetScalar of an empty array is an
 * error in MATLAB (here: recorded as a mock failure), mexErrMsgIdAndTxt does not return (longjmp to the driver). */
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"
#include "mex_mock.h"

struct mxArray_tag { mxClassID cls; size_t m, n; void *data; char *str; int logical; };

jmp_buf mock_err_jmp;
char mock_err_id[64], mock_err_msg[512];
int mock_lock_count = 0;
int mock_misuse = 0;

mxArray *mock_double(size_t m, size_t n, const double *v) {
    mxArray *a = calloc(1, sizeof *a);
    a->cls = mxDOUBLE_CLASS; a->m = m; a->n = n;
    if (m * n != 0) { a->data = calloc(m * n, sizeof(double)); if (v) memcpy(a->data, v, m * n * sizeof(double)); }
    return a;
}
mxArray *mock_string(const char *s) { mxArray *a = calloc(1, sizeof *a); a->cls = mxUNKNOWN_CLASS; a->str = malloc(strlen(s) + 1); strcpy(a->str, s); a->m = 1; a->n = strlen(s); return a; }
mxArray *mock_uint64(uint64_t v) { mxArray *a = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL); *(uint64_t *)a->data = v; return a; }
mxArray *mock_uint64_vec(size_t n, const uint64_t *v) {
    mxArray *a = mxCreateNumericMatrix(n, 1, mxUINT64_CLASS, mxREAL);
    for (size_t i = 0; i < n; ++i) ((uint64_t *)a->data)[i] = v[i];
    return a;
}
int mock_is_logical(const mxArray *a) { return a->logical; }

mxClassID mxGetClassID(const mxArray *pa) { return pa->cls; }
double *mxGetPr(const mxArray *pa) { return pa->cls == mxDOUBLE_CLASS ? (double *)pa->data : NULL; }
void *mxGetData(const mxArray *pa) { return pa->data; }
double mxGetScalar(const mxArray *pa) {
    if (!pa->data || pa->m * pa->n == 0) { mock_misuse++; return 0.0; }      /* MATLAB: error / undefined on an empty array */
    if (pa->cls == mxUINT64_CLASS) return (double)*(uint64_t *)pa->data;
    return *(double *)pa->data;
}
size_t mxGetNumberOfElements(const mxArray *pa) { return pa->m * pa->n; }
size_t mxGetM(const mxArray *pa) { return pa->m; }
size_t mxGetN(const mxArray *pa) { return pa->n; }
int mxGetString(const mxArray *pa, char *buf, mwSize buflen) {
    if (!pa->str || strlen(pa->str) + 1 > buflen) return 1;
    strcpy(buf, pa->str);
    return 0;
}
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) { (void)flag; return mock_double(m, n, NULL); }
mxArray *mxCreateDoubleScalar(double value) { return mock_double(1, 1, &value); }
mxArray *mxCreateLogicalScalar(bool value) { double v = value; mxArray *a = mock_double(1, 1, &v); a->logical = 1; return a; }
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag) {
    (void)flag;
    mxArray *a = calloc(1, sizeof *a);
    a->cls = classid; a->m = m; a->n = n;
    if (m * n != 0) a->data = calloc(m * n, 8);
    return a;
}
void mxDestroyArray(mxArray *pa) { if (pa) { free(pa->data); free(pa->str); free(pa); } }
void mexErrMsgIdAndTxt(const char *identifier, const char *err_msg, ...) {
    va_list ap;
    va_start(ap, err_msg);
    vsnprintf(mock_err_msg, sizeof mock_err_msg, err_msg, ap);
    va_end(ap);
    snprintf(mock_err_id, sizeof mock_err_id, "%s", identifier);
    longjmp(mock_err_jmp, 1);
}
void mexLock(void) { mock_lock_count++; }
void mexUnlock(void) { mock_lock_count--; }
