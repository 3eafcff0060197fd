/* mex.h -- a small MOCK of MATLAB's MEX API, just enough to compile and run matlab/nagp_mex.c without MATLAB
 * (no MATLAB / Octave in the build image).  Test infrastructure only; the real header comes with MATLAB. */
#ifndef NAGP_MOCK_MEX_H
#define NAGP_MOCK_MEX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef size_t mwSize;
typedef enum { mxDOUBLE_CLASS = 1, mxINT32_CLASS, mxINT64_CLASS, mxSTRUCT_CLASS, mxCHAR_CLASS, mxCELL_CLASS } mxClassID;
typedef enum { mxREAL = 0 } mxComplexity;
typedef struct mxArray_tag mxArray;

int mxIsStruct(const mxArray* a);
int mxIsDouble(const mxArray* a);
int mxIsComplex(const mxArray* a);
int mxIsInt32(const mxArray* a);
int mxIsInt64(const mxArray* a);
int mxIsEmpty(const mxArray* a);
int mxIsChar(const mxArray* a);
int mxIsCell(const mxArray* a);
int mxGetString(const mxArray* a, char* buf, mwSize buflen); /* 0 = ok, 1 = truncated (as MATLAB's) */
mxArray* mxGetCell(const mxArray* a, mwSize index);
void mxSetCell(mxArray* a, mwSize index, mxArray* value);
mxArray* mxCreateCellMatrix(mwSize m, mwSize n);
size_t mxGetN(const mxArray* a);
mxArray* mxGetField(const mxArray* a, size_t index, const char* name);
double mxGetScalar(const mxArray* a);
double* mxGetPr(const mxArray* a);
void* mxGetData(const mxArray* a);
size_t mxGetM(const mxArray* a);
size_t mxGetNumberOfElements(const mxArray* a);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity c);
void* mxCalloc(size_t n, size_t size);   /* memory MATLAB releases by itself when the MEX function is left, also through an error */
void mxFree(void* p);
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);

/* what a gateway exports */
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

/* constructors for the driver (not part of the real API) */
mxArray* mock_numeric(mxClassID cls, size_t m, size_t n, const void* data); /* copies */
mxArray* mock_struct(void);
void mock_set(mxArray* s, const char* name, mxArray* v);
mxArray* mock_scalar(double v);
mxArray* mock_string(const char* s);
#ifdef __cplusplus
}
#endif
#endif
