/*
 * mex.h — TEST-ONLY stand-in for MATLAB's MEX API, just large enough to compile matlab/gnsscorr_mex.c without MATLAB and
 * drive its mexFunction from Python (tests/mexstub/harness.py, tests/test_gpu_mex_gateway.py).  It is not a product file and
 * makes no claim to be MATLAB's mex.h: it implements the two dozen entry points the gateway uses with the documented
 * semantics (column-major numeric arrays, struct and cell arrays, char rows, mexErrMsgIdAndTxt unwinding the call).
 */
#ifndef GNSSCORR_TEST_MEX_H
#define GNSSCORR_TEST_MEX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS, mxSINGLE_CLASS,
               mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef struct mxArray_tag mxArray;

mxArray* mxCreateDoubleScalar(double v);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity c);
mxArray* mxCreateString(const char* s);
mxArray* mxCreateCellMatrix(mwSize m, mwSize n);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names);
void mxDestroyArray(mxArray* a);
double mxGetScalar(const mxArray* a);
int mxGetString(const mxArray* a, char* buf, mwSize buflen);
mxArray* mxGetField(const mxArray* a, mwIndex i, const char* name);
void mxSetField(mxArray* a, mwIndex i, const char* name, mxArray* v);
mxArray* mxGetCell(const mxArray* a, mwIndex i);
void mxSetCell(mxArray* a, mwIndex i, mxArray* v);
mwSize mxGetNumberOfElements(const mxArray* a);
mwSize mxGetM(const mxArray* a);
mwSize mxGetN(const mxArray* a);
mwSize mxGetNumberOfDimensions(const mxArray* a);
const mwSize* mxGetDimensions(const mxArray* a);
mxClassID mxGetClassID(const mxArray* a);
int mxIsDouble(const mxArray* a);
int mxIsInt8(const mxArray* a);
int mxIsInt16(const mxArray* a);
int mxIsInt32(const mxArray* a);
int mxIsSingle(const mxArray* a);
int mxIsEmpty(const mxArray* a);
int mxIsChar(const mxArray* a);
int mxIsStruct(const mxArray* a);
int mxIsCell(const mxArray* a);
double* mxGetDoubles(const mxArray* a);
double* mxGetPr(const mxArray* a);
void* mxGetData(const mxArray* a);
void* mxMalloc(size_t n);
void* mxCalloc(size_t n, size_t size);
void mxFree(void* p);
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
int mexAtExit(void (*fn)(void));
void mexLock(void);
void mexUnlock(void);
int mexPrintf(const char* fmt, ...);

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#ifdef __cplusplus
}
#endif
#endif
