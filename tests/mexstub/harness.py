"""Python side of the test-only MEX harness: compiles matlab/gnsscorr_mex.c against tests/mexstub/mex.h + mexstub.c into
tests/mexstub/_build/libgnsscorr_mexstub.so (gcc, links the product's libgnsscorr.so) and calls its mexFunction with mxArrays
built from Python values.  What a MATLAB host does with `mex gnsscorr_mex.c`, minus MATLAB."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build", "libgnsscorr_mexstub.so")
LIBDIR = os.path.join(ROOT, "cu-sdr-collection_amd", "lib")

CLS = {"cell": 1, "struct": 2, "logical": 3, "char": 4, "double": 6, "single": 7, "int8": 8, "uint8": 9, "int16": 10, "uint16": 11,
       "int32": 12, "uint32": 13, "int64": 14, "uint64": 15}
NP_OF = {6: np.float64, 7: np.float32, 8: np.int8, 9: np.uint8, 10: np.int16, 11: np.uint16, 12: np.int32, 13: np.uint32, 14: np.int64, 15: np.uint64,
         3: np.uint8}


def build(force: bool = False) -> str:
    srcs = [os.path.join(ROOT, "matlab", "gnsscorr_mex.c"), os.path.join(HERE, "mexstub.c")]
    deps = srcs + [os.path.join(HERE, "mex.h"), os.path.join(ROOT, "include", "gnsscorr.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-Wall", "-Wextra", "-Wno-unused-parameter", "-fPIC", "-shared", f"-I{HERE}", f"-I{os.path.join(ROOT, 'include')}",
           *srcs, f"-L{LIBDIR}", "-lgnsscorr", f"-Wl,-rpath,{LIBDIR}", "-o", OUT]
    subprocess.run(cmd, check=True)
    return OUT


class MexError(RuntimeError):
    pass


class Gateway:
    """mexFunction of the compiled gateway: call(cmd, *args, nargout=1) with Python values in, Python values out."""

    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        P = C.c_void_p
        for name, res, args in (
                ("mxCreateNumericArray", P, [C.c_size_t, C.POINTER(C.c_size_t), C.c_int, C.c_int]), ("mxCreateString", P, [C.c_char_p]),
                ("mxCreateCellMatrix", P, [C.c_size_t, C.c_size_t]), ("mxCreateStructMatrix", P, [C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_char_p)]),
                ("mxSetField", None, [P, C.c_size_t, C.c_char_p, P]), ("mxSetCell", None, [P, C.c_size_t, P]), ("mxGetField", P, [P, C.c_size_t, C.c_char_p]),
                ("mxGetCell", P, [P, C.c_size_t]), ("mxGetData", P, [P]), ("mxGetClassID", C.c_int, [P]), ("mxGetNumberOfDimensions", C.c_size_t, [P]),
                ("mxGetDimensions", C.POINTER(C.c_size_t), [P]), ("mxGetNumberOfElements", C.c_size_t, [P]), ("mxDestroyArray", None, [P]),
                ("mxGetString", C.c_int, [P, C.c_char_p, C.c_size_t]), ("stub_call", C.c_int, [C.c_int, C.POINTER(P), C.c_int, C.POINTER(P)]),
                ("stub_last_error", C.c_char_p, []), ("stub_lock_count", C.c_int, []), ("stub_run_atexit", None, []),
                ("stub_field_count", C.c_int, [P]), ("stub_field_name", C.c_char_p, [P, C.c_int])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args

    # ---- Python -> mxArray ------------------------------------------------------------------------------------
    def to_mx(self, v):
        L = self.lib
        if isinstance(v, str):
            return L.mxCreateString(v.encode("latin-1"))
        if isinstance(v, dict):
            names = (C.c_char_p * len(v))(*[k.encode() for k in v])
            s = L.mxCreateStructMatrix(1, 1, len(v), names)
            for k, x in v.items():
                L.mxSetField(s, 0, k.encode(), self.to_mx(x))
            return s
        if isinstance(v, (list, tuple)) and (not v or not all(isinstance(x, (int, float, bool, np.number)) for x in v)):
            c = L.mxCreateCellMatrix(1, len(v))
            for i, x in enumerate(v):
                L.mxSetCell(c, i, self.to_mx(x))
            return c
        a = np.asarray(v)
        if a.dtype == np.bool_ or a.dtype.kind in "iu" and a.dtype.itemsize == 8 and not isinstance(v, np.ndarray):
            a = a.astype(np.float64)
        if a.dtype not in (np.float64, np.float32, np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64):
            a = a.astype(np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(1, -1)
        cls = {np.dtype(t): k for k, t in NP_OF.items() if k != 3}[a.dtype]
        dims = (C.c_size_t * a.ndim)(*a.shape)
        m = L.mxCreateNumericArray(a.ndim, dims, cls, 0)
        if a.size:
            col_major = np.asfortranarray(a)          # kept alive until the copy is done
            C.memmove(L.mxGetData(m), col_major.ctypes.data, a.nbytes)
        return m

    # ---- mxArray -> Python ------------------------------------------------------------------------------------
    def from_mx(self, m):
        L = self.lib
        if not m:
            return None
        cls = L.mxGetClassID(m)
        nd = L.mxGetNumberOfDimensions(m)
        dims = tuple(int(L.mxGetDimensions(m)[i]) for i in range(nd))
        n = int(L.mxGetNumberOfElements(m))
        if cls == CLS["char"]:
            buf = C.create_string_buffer(n + 1)
            L.mxGetString(m, buf, n + 1)
            return buf.value.decode("latin-1")
        if cls == CLS["cell"]:
            return [self.from_mx(L.mxGetCell(m, i)) for i in range(n)]
        if cls == CLS["struct"]:
            return {L.stub_field_name(m, i).decode(): self.from_mx(L.mxGetField(m, 0, L.stub_field_name(m, i))) for i in range(L.stub_field_count(m))}
        dt = np.dtype(NP_OF[cls])
        raw = (C.c_char * (n * dt.itemsize)).from_address(L.mxGetData(m)) if n else b""
        return np.frombuffer(bytes(raw), dtype=dt).reshape(dims, order="F").copy()

    def call(self, cmd, *args, nargout=1):
        L = self.lib
        prhs_vals = [self.to_mx(cmd)] + [self.to_mx(a) for a in args]
        prhs = (C.c_void_p * len(prhs_vals))(*prhs_vals)
        plhs = (C.c_void_p * max(nargout, 1))()
        rc = L.stub_call(nargout, plhs, len(prhs_vals), prhs)
        try:
            if rc:
                raise MexError(L.stub_last_error().decode("latin-1"))
            outs = [self.from_mx(plhs[i]) for i in range(max(nargout, 1))]
        finally:
            for p in prhs_vals:
                L.mxDestroyArray(p)
            for i in range(max(nargout, 1)):
                if plhs[i]:
                    L.mxDestroyArray(plhs[i])
        if nargout <= 1:
            return outs[0]
        return tuple(outs[:nargout])
