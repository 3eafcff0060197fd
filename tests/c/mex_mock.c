/* mex_mock.c -- implementation of the mock MEX API of tests/c/mex.h (test infrastructure) */
#include "mex.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXF 32
struct mxArray_tag {
  mxClassID cls;
  size_t m, n;       /* n = product of the trailing dimensions */
  void* data;
  int nf;
  char fname[MAXF][32];
  mxArray* fval[MAXF];
};

static size_t esize(mxClassID c) { return c == mxINT32_CLASS ? 4 : (c == mxCHAR_CLASS ? 1 : 8); }   /* cells: 8 = a pointer */
int mxIsStruct(const mxArray* a) { return a && a->cls == mxSTRUCT_CLASS; }
int mxIsDouble(const mxArray* a) { return a && a->cls == mxDOUBLE_CLASS; }
int mxIsComplex(const mxArray* a) { (void)a; return 0; }
int mxIsInt32(const mxArray* a) { return a && a->cls == mxINT32_CLASS; }
int mxIsInt64(const mxArray* a) { return a && a->cls == mxINT64_CLASS; }
int mxIsEmpty(const mxArray* a) { return !a || a->m * a->n == 0; }
int mxIsChar(const mxArray* a) { return a && a->cls == mxCHAR_CLASS; }
int mxIsCell(const mxArray* a) { return a && a->cls == mxCELL_CLASS; }
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
  const size_t n = a->m * a->n;
  if (a->cls != mxCHAR_CLASS || buflen == 0) return 1;
  if (n + 1 > buflen) { memcpy(buf, a->data, buflen - 1); buf[buflen - 1] = 0; return 1; }
  memcpy(buf, a->data, n); buf[n] = 0; return 0;
}
mxArray* mxGetCell(const mxArray* a, mwSize index) { return (a && a->cls == mxCELL_CLASS && index < a->m * a->n) ? ((mxArray**)a->data)[index] : NULL; }
void mxSetCell(mxArray* a, mwSize index, mxArray* value) { if (a && a->cls == mxCELL_CLASS && index < a->m * a->n) ((mxArray**)a->data)[index] = value; }
size_t mxGetN(const mxArray* a) { return a->n; }
mxArray* mxGetField(const mxArray* a, size_t index, const char* name) {
  int i;
  if (!a || a->cls != mxSTRUCT_CLASS || index != 0) return NULL;
  for (i = 0; i < a->nf; ++i)
    if (!strcmp(a->fname[i], name)) return a->fval[i];
  return NULL;
}
double mxGetScalar(const mxArray* a) {
  if (a->cls == mxDOUBLE_CLASS) return ((double*)a->data)[0];
  if (a->cls == mxINT32_CLASS) return (double)((int32_t*)a->data)[0];
  return (double)((int64_t*)a->data)[0];
}
double* mxGetPr(const mxArray* a) { return (double*)a->data; }
void* mxGetData(const mxArray* a) { return a->data; }
size_t mxGetM(const mxArray* a) { return a->m; }
size_t mxGetNumberOfElements(const mxArray* a) { return a->m * a->n; }
mxArray* mock_numeric(mxClassID cls, size_t m, size_t n, const void* data) {
  mxArray* a = (mxArray*)calloc(1, sizeof *a);
  a->cls = cls; a->m = m; a->n = n;
  a->data = calloc((m * n) > 0 ? m * n : 1, esize(cls));
  if (data && (m * n) > 0) memcpy(a->data, data, m * n * esize(cls));
  return a;
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) { (void)c; return mock_numeric(mxDOUBLE_CLASS, m, n, NULL); }
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c) { (void)c; return mock_numeric(cls, m, n, NULL); }
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity c) {
  size_t n = 1; mwSize i;
  (void)c;
  for (i = 1; i < ndim; ++i) n *= dims[i];
  return mock_numeric(cls, dims[0], n, NULL);
}
mxArray* mock_struct(void) { mxArray* a = (mxArray*)calloc(1, sizeof *a); a->cls = mxSTRUCT_CLASS; a->m = a->n = 1; return a; }
void mock_set(mxArray* s, const char* name, mxArray* v) {
  if (s->nf >= MAXF) { fprintf(stderr, "mock struct full\n"); exit(3); }
  strncpy(s->fname[s->nf], name, 31); s->fval[s->nf++] = v;
}
mxArray* mock_scalar(double v) { return mock_numeric(mxDOUBLE_CLASS, 1, 1, &v); }
mxArray* mock_string(const char* s) { return mock_numeric(mxCHAR_CLASS, 1, strlen(s), s); }
mxArray* mxCreateCellMatrix(mwSize m, mwSize n) { return mock_numeric(mxCELL_CLASS, m, n, NULL); }
void* mxCalloc(size_t n, size_t size) { return calloc(n ? n : 1, size ? size : 1); }
void mxFree(void* p) { free(p); }
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
  va_list ap;
  fprintf(stderr, "MEX error %s: ", id);
  va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
  fprintf(stderr, "\n");
  exit(2);      /* MATLAB would unwind to the prompt */
}
