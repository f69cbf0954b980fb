"""Does the intermediate's footprint matter?  The default GPS L1 C/A search with fewer frequency bins (the intermediate of one PRN is
bins x 20 hops x 288 KB; two PRN lanes hold two): microseconds per inverse transform, one and two lanes.  The last-level cache holds 256 MB."""
import os as _os
_os.environ.setdefault("GC_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cu-sdr-collection_amd", "lib", "libgnsscorr_tuning.so"))  # the GC_* switches used below exist in the tuning build only (docs/KNOBS.md)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cu_sdr_collection_amd as P

S0 = P.initSettings()
sats = P.synth.scene(12, 5, S0.samplingFreq)
iq = P.synth.generate_if(sats, int(0.1 * S0.samplingFreq), S0.samplingFreq, S0.IF, P.codes.generateCAcode, S0.codeFreqBasis, 1023, seed=3)
for lanes in ("2", "1"):
    os.environ["GC_ACQ_LANES"] = lanes
    for band in (7000, 5000, 3500, 2500, 1500, 500):
        S = P.initSettings()
        S.acqSearchBand = band
        bins = int(round(band * 2 / S.acqSearchStep)) + 1
        with P.Engine(0) as eng:
            eng.load_if(iq, fs=S.samplingFreq)
            for _ in range(12):
                P.acquisition(eng, S)
            ts = []
            for _ in range(30):
                t = time.perf_counter(); P.acquisition(eng, S); ts.append(time.perf_counter() - t)
        ms = float(np.median(ts)) * 1e3
        print("lanes %s, %2d bins: intermediate %3.0f MB per lane, %.3f ms per search, %.4f us per inverse transform (fine stage and spectra included)"
              % (lanes, bins, bins * 20 * 0.288, ms, ms * 1e3 / (32 * bins * 20)))
