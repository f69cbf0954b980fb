/* Test-only implementation of tests/mexstub/mex.h plus the call harness Python drives (see mex.h for what this is and is not). */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

struct mxArray_tag {
  mxClassID cls;
  mwSize ndim;
  mwSize dims[3];
  void* data;       /* numeric / char (as uint16 code units) payload */
  mxArray** items;  /* cell elements, or struct fields: items[elem * nfields + field] */
  int nfields;
  char** names;
};

static size_t elem_size(mxClassID c) {
  switch (c) {
    case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
    case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxINT16_CLASS: case mxUINT16_CLASS: case mxCHAR_CLASS: return 2;
    case mxINT8_CLASS: case mxUINT8_CLASS: case mxLOGICAL_CLASS: return 1;
    default: return 0;
  }
}

static mwSize numel(const mxArray* a) {
  mwSize n = 1;
  for (mwSize i = 0; i < a->ndim; ++i) n *= a->dims[i];
  return n;
}

static mxArray* new_array(mxClassID cls, mwSize ndim, const mwSize* dims) {
  mxArray* a = (mxArray*)calloc(1, sizeof *a);
  a->cls = cls;
  a->ndim = ndim < 2 ? 2 : ndim;
  a->dims[0] = a->dims[1] = a->dims[2] = 1;
  for (mwSize i = 0; i < ndim && i < 3; ++i) a->dims[i] = dims[i];
  size_t es = elem_size(cls);
  if (es) a->data = calloc(numel(a) ? numel(a) : 1, es);
  return a;
}

mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity c) { (void)c; return new_array(cls, ndim, dims); }
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c) { mwSize d[2] = {m, n}; (void)c; return new_array(cls, 2, d); }
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, c); }
mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); ((double*)a->data)[0] = v; return a; }
mxArray* mxCreateString(const char* s) {
  mwSize d[2] = {1, strlen(s)};
  mxArray* a = new_array(mxCHAR_CLASS, 2, d);
  for (mwSize i = 0; i < d[1]; ++i) ((uint16_t*)a->data)[i] = (unsigned char)s[i];
  return a;
}
mxArray* mxCreateCellMatrix(mwSize m, mwSize n) {
  mwSize d[2] = {m, n};
  mxArray* a = new_array(mxCELL_CLASS, 2, d);
  a->items = (mxArray**)calloc(numel(a) ? numel(a) : 1, sizeof(mxArray*));
  return a;
}
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names) {
  mwSize d[2] = {m, n};
  mxArray* a = new_array(mxSTRUCT_CLASS, 2, d);
  a->nfields = nfields;
  a->names = (char**)calloc((size_t)nfields ? (size_t)nfields : 1, sizeof(char*));
  for (int i = 0; i < nfields; ++i) a->names[i] = strdup(names[i]);
  a->items = (mxArray**)calloc((numel(a) ? numel(a) : 1) * (size_t)(nfields ? nfields : 1), sizeof(mxArray*));
  return a;
}
void mxDestroyArray(mxArray* a) {
  if (!a) return;
  if (a->items) {
    mwSize n = numel(a) * (a->cls == mxSTRUCT_CLASS ? (mwSize)a->nfields : 1);
    for (mwSize i = 0; i < n; ++i) mxDestroyArray(a->items[i]);
    free(a->items);
  }
  for (int i = 0; i < a->nfields; ++i) free(a->names[i]);
  free(a->names);
  free(a->data);
  free(a);
}

static jmp_buf g_jmp;
static int g_active;
static char g_err[1024];
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
  char msg[900];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof msg, fmt, ap);
  va_end(ap);
  snprintf(g_err, sizeof g_err, "%s: %s", id, msg);
  if (g_active) longjmp(g_jmp, 1);
  fprintf(stderr, "mexErrMsgIdAndTxt outside a call: %s\n", g_err);
  abort();
}

double mxGetScalar(const mxArray* a) {
  if (!a || !a->data || numel(a) == 0) mexErrMsgIdAndTxt("stub:scalar", "mxGetScalar of an empty or non-numeric array");
  switch (a->cls) {
    case mxDOUBLE_CLASS: return ((double*)a->data)[0];
    case mxSINGLE_CLASS: return ((float*)a->data)[0];
    case mxINT8_CLASS: return ((int8_t*)a->data)[0];
    case mxUINT8_CLASS: case mxLOGICAL_CLASS: return ((uint8_t*)a->data)[0];
    case mxINT16_CLASS: return ((int16_t*)a->data)[0];
    case mxUINT16_CLASS: case mxCHAR_CLASS: return ((uint16_t*)a->data)[0];
    case mxINT32_CLASS: return ((int32_t*)a->data)[0];
    case mxUINT32_CLASS: return ((uint32_t*)a->data)[0];
    case mxINT64_CLASS: return (double)((int64_t*)a->data)[0];
    case mxUINT64_CLASS: return (double)((uint64_t*)a->data)[0];
    default: mexErrMsgIdAndTxt("stub:scalar", "mxGetScalar of class %d", (int)a->cls);
  }
  return 0;
}
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
  if (!a || a->cls != mxCHAR_CLASS || buflen == 0) return 1;
  mwSize n = numel(a);
  int trunc = n + 1 > buflen;
  if (trunc) n = buflen - 1;
  for (mwSize i = 0; i < n; ++i) buf[i] = (char)((uint16_t*)a->data)[i];
  buf[n] = 0;
  return trunc;
}
static int field_index(const mxArray* a, const char* name) {
  for (int i = 0; i < a->nfields; ++i)
    if (!strcmp(a->names[i], name)) return i;
  return -1;
}
mxArray* mxGetField(const mxArray* a, mwIndex i, const char* name) {
  if (!a || a->cls != mxSTRUCT_CLASS || i >= numel(a)) return NULL;
  int f = field_index(a, name);
  return f < 0 ? NULL : a->items[i * (mwSize)a->nfields + (mwSize)f];
}
void mxSetField(mxArray* a, mwIndex i, const char* name, mxArray* v) {
  int f = field_index(a, name);
  if (f < 0 || i >= numel(a)) mexErrMsgIdAndTxt("stub:field", "mxSetField: no field %s", name);
  a->items[i * (mwSize)a->nfields + (mwSize)f] = v;
}
mxArray* mxGetCell(const mxArray* a, mwIndex i) { return (a && a->cls == mxCELL_CLASS && i < numel(a)) ? a->items[i] : NULL; }
void mxSetCell(mxArray* a, mwIndex i, mxArray* v) { a->items[i] = v; }
mwSize mxGetNumberOfElements(const mxArray* a) { return numel(a); }
mwSize mxGetM(const mxArray* a) { return a->dims[0]; }
mwSize mxGetN(const mxArray* a) { mwSize n = 1; for (mwSize i = 1; i < a->ndim; ++i) n *= a->dims[i]; return n; }
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->ndim; }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims; }
mxClassID mxGetClassID(const mxArray* a) { return a->cls; }
int mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
int mxIsInt8(const mxArray* a) { return a->cls == mxINT8_CLASS; }
int mxIsInt16(const mxArray* a) { return a->cls == mxINT16_CLASS; }
int mxIsInt32(const mxArray* a) { return a->cls == mxINT32_CLASS; }
int mxIsSingle(const mxArray* a) { return a->cls == mxSINGLE_CLASS; }
int mxIsEmpty(const mxArray* a) { return numel(a) == 0; }
int mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
int mxIsStruct(const mxArray* a) { return a->cls == mxSTRUCT_CLASS; }
int mxIsCell(const mxArray* a) { return a->cls == mxCELL_CLASS; }
double* mxGetDoubles(const mxArray* a) {
  if (a->cls != mxDOUBLE_CLASS) mexErrMsgIdAndTxt("stub:type", "mxGetDoubles of a non-double array (class %d)", (int)a->cls);
  return (double*)a->data;
}
double* mxGetPr(const mxArray* a) { return mxGetDoubles(a); }
void* mxGetData(const mxArray* a) { return a->data; }
void* mxMalloc(size_t n) { return malloc(n ? n : 1); }
void* mxCalloc(size_t n, size_t size) { return calloc(n ? n : 1, size ? size : 1); }
void mxFree(void* p) { free(p); }

static void (*g_atexit)(void);
static int g_locks;
int mexAtExit(void (*fn)(void)) { g_atexit = fn; return 0; }
void mexLock(void) { ++g_locks; }
void mexUnlock(void) { if (g_locks > 0) --g_locks; }
int mexPrintf(const char* fmt, ...) { va_list ap; va_start(ap, fmt); int n = vprintf(fmt, ap); va_end(ap); return n; }

/* ---- harness entry points (Python, tests/mexstub/harness.py) ---------------------------------------------------- */
int stub_call(int nlhs, mxArray** plhs, int nrhs, const mxArray** prhs) {
  g_err[0] = 0;
  for (int i = 0; i < (nlhs > 0 ? nlhs : 1); ++i) plhs[i] = NULL;
  if (setjmp(g_jmp)) {
    g_active = 0;
    return 1; /* mexErrMsgIdAndTxt was raised: message in stub_last_error() */
  }
  g_active = 1;
  mexFunction(nlhs, plhs, nrhs, prhs);
  g_active = 0;
  return 0;
}
const char* stub_last_error(void) { return g_err; }
int stub_lock_count(void) { return g_locks; }
void stub_run_atexit(void) { if (g_atexit) g_atexit(); }
int stub_field_count(const mxArray* a) { return a->nfields; }
const char* stub_field_name(const mxArray* a, int i) { return a->names[i]; }
