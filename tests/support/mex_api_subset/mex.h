/* NOT MathWorks' mex.h.  Declarations of the handful of documented MEX C-API entry points matlab/ekfslam_mex.c uses, written
 * from the public API documentation, so that tests/test_mex_gateway_cpu.py can type-check the gateway (gcc -fsyntax-only)
 * against include/ekfslam.h on a machine without MATLAB, and run it against the mock in tests/support/mex_mock/ (which
 * implements exactly these declarations; it pins the gateway's own logic, nothing about MATLAB). */
#ifndef EKF_TEST_MEX_API_SUBSET_H
#define EKF_TEST_MEX_API_SUBSET_H
#include <stdbool.h>
#include <stddef.h>

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxUNKNOWN_CLASS = 0, mxDOUBLE_CLASS = 6, mxUINT64_CLASS = 15 } mxClassID;

mxClassID mxGetClassID(const mxArray *pa);
double *mxGetPr(const mxArray *pa);
void *mxGetData(const mxArray *pa);
double mxGetScalar(const mxArray *pa);
size_t mxGetNumberOfElements(const mxArray *pa);
size_t mxGetM(const mxArray *pa);
size_t mxGetN(const mxArray *pa);
int mxGetString(const mxArray *pa, char *buf, mwSize buflen);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double value);
mxArray *mxCreateLogicalScalar(bool value);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
void mxDestroyArray(mxArray *pa);
void mexErrMsgIdAndTxt(const char *identifier, const char *err_msg, ...);
void mexLock(void);
void mexUnlock(void);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#endif
