"""ctypes access to oracle/_build/liboracle.so (TEST INFRASTRUCTURE ONLY — see gnss_oracle.c)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "liboracle.so")
FIELDS = ["absoluteSample", "codeFreq", "carrFreq", "I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L",
          "dllDiscr", "dllDiscrFilt", "pllDiscr", "pllDiscrFilt", "remCodePhase", "remCarrPhase"]
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "gnss_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B", "_build/liboracle.so"], check=True, capture_output=True)
    return SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.orc_generate_ca.argtypes = [C.c_int, C.c_void_p]
        _lib.orc_correlate_block.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int] + \
            [C.c_double] * 9 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_track_l1ca.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + \
            [C.c_double] * 9 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        _lib.orc_track_l1ca.restype = C.c_int
    return _lib


def generate_ca(prn: int) -> np.ndarray:
    out = np.empty(1023)
    lib().orc_generate_ca(prn, out.ctypes.data)
    return out


def correlate_block(iq: np.ndarray, first_sample: int, n: int, tables, rem, step, d, carr_freq, rem_carr,
                    fs, code_length, r=1.0, mult=1.0, swap_iq=False):
    """iq: int8 interleaved I,Q.  tables: list of equal-length padded float64 tables."""
    iq = np.ascontiguousarray(iq, dtype=np.int8)
    tabs = np.ascontiguousarray(np.stack([np.asarray(t, dtype=np.float64) for t in tables]))
    arms, tlen = tabs.shape
    sums = np.empty(arms * 6)
    rc = C.c_double()
    rp = C.c_double()
    lib().orc_correlate_block(iq.ctypes.data, first_sample, n, tabs.ctypes.data, arms, tlen, rem, step, d,
                              r, mult, carr_freq, rem_carr, fs, code_length, int(swap_iq), sums.ctypes.data,
                              C.byref(rc), C.byref(rp))
    return sums.reshape(arms, 6), rc.value, rp.value


def track_l1ca(iq: np.ndarray, channel, settings):
    """Closed loop, channels serial (tracking.m:133-368).  Returns (dict field -> [nch, n_epochs],
    epochs_done, aborted)."""
    iq = np.ascontiguousarray(iq, dtype=np.int8)
    nch = len(channel)
    prn = np.array([c.PRN for c in channel], dtype=np.int32)
    fa = np.array([c.acquiredFreq for c in channel], dtype=np.float64)
    cp = np.array([c.codePhase for c in channel], dtype=np.int64)
    n_ep = int(settings.msToProcess)
    out = np.zeros((nch, len(FIELDS), n_ep))
    done = np.zeros(nch, dtype=np.int32)
    rc = lib().orc_track_l1ca(iq.ctypes.data, iq.shape[0] // 2, nch, prn.ctypes.data, fa.ctypes.data,
                              cp.ctypes.data, settings.samplingFreq, settings.codeFreqBasis,
                              settings.codeLength, settings.dllCorrelatorSpacing, settings.intTime,
                              settings.dllNoiseBandwidth, settings.dllDampingRatio,
                              settings.pllNoiseBandwidth, settings.pllDampingRatio,
                              int(settings.skipNumberOfBytes), n_ep, out.ctypes.data, done.ctypes.data)
    return {f: out[:, i, :] for i, f in enumerate(FIELDS)}, done, bool(rc)
