"""Where a wavefront of the fused passes of the default GPS L1 C/A search spends its cycles (a tuning build:
scripts/variants.sh acq "CLK:-DGC_ACQ_STAGE_CLOCKS=1"; GC_LIB_PATH=.../libgnsscorr_CLK.so python scripts/acq_stage_clocks.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cu_sdr_collection_amd as P
from cu_sdr_collection_amd import _lib as L

S = P.initSettings()
sats = P.synth.scene(12, 5, S.samplingFreq)
iq = P.synth.generate_if(sats, int(0.1 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=3)
eng = P.Engine(0)
eng.load_if(iq, fs=S.samplingFreq)
lib = C.CDLL(L.LIB_PATH)
buf = (C.c_ulonglong * 128)()
for _ in range(10):
    P.acquisition(eng, S)
lib.gc_debug_acq_stage_clocks(buf, 1)
n = 20
for _ in range(n):
    P.acquisition(eng, S)
lib.gc_debug_acq_stage_clocks(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(2, 8, 8)
names = ["wait tile + first stage", "fetch issue + barrier", "middle stage", "barrier", "last stage", "loop top"]
for k, pas in enumerate(("columns pass", "rows pass")):
    print(pas)
    for w in range(8):
        hops = a[k, w, 7]
        if hops == 0:
            continue
        per = a[k, w, :6] / hops
        print("  wavefront %d: %6.0f cycles per hop = " % (w, per.sum()) + ", ".join("%s %5.0f" % (nm, v) for nm, v in zip(names, per)))
