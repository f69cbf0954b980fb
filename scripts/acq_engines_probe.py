"""Does the default GPS L1 C/A search take the same time in every engine of a process?  Six engines made one after the other, each timed
while the earlier ones are alive, with two PRN lanes and with one (GC_ACQ_LANE_STREAMS=own: a stream pair per context - the second engine
searched in 3.65 instead of 2.77 ms; default: the device's search streams)."""
import os as _os
_os.environ.setdefault("GC_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cu-sdr-collection_amd", "lib", "libgnsscorr_tuning.so"))  # the GC_* switches used below exist in the tuning build only (docs/KNOBS.md)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cu_sdr_collection_amd as P
from cu_sdr_collection_amd.receiver import acquisition
S = P.initSettings()
sats = P.synth.scene(12, 5, S.samplingFreq)
iq = P.synth.generate_if(sats, int(0.1 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=3)
def timeit(eng, n=40):
    for _ in range(12): acquisition(eng, S)
    ts = []
    for _ in range(n):
        t = time.perf_counter(); acquisition(eng, S); ts.append(time.perf_counter() - t)
    return np.median(ts) * 1e3
engs = []
for i in range(6):
    e = P.Engine(0); e.load_if(iq, fs=S.samplingFreq); engs.append(e)
    print("engine %d (all earlier ones alive): %.3f ms" % (i, timeit(e)), flush=True)
for i, e in enumerate(engs):
    print("engine %d again: %.3f ms, lanes=1: " % (i, timeit(e)), end="", flush=True)
    os.environ["GC_ACQ_LANES"] = "1"; print("%.3f ms" % timeit(e), flush=True); del os.environ["GC_ACQ_LANES"]
