"""Where the milliseconds of the default GPS L1 C/A acquisition go on the host's clock: the coarse call (one C call: spectra, 32 PRNs on
two lanes, peak keys), the Python between the calls, the fine call (one C call), the Python after it.  Sustained calls (clocks up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cu_sdr_collection_amd as P

S = P.initSettings()
sats = P.synth.scene(12, 5, S.samplingFreq)
iq = P.synth.generate_if(sats, int(0.1 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=3)
eng = P.Engine(0)
eng.load_if(iq, fs=S.samplingFreq)
stamps = []
for name in ("acquire_coarse", "acquire_fine_l1ca_batch"):
    f = getattr(eng, name)
    def wrap(*a, _f=f, _n=name, **k):
        t0 = time.perf_counter(); r = _f(*a, **k); stamps.append((_n, t0, time.perf_counter())); return r
    setattr(eng, name, wrap)
rows = []
for i in range(300):
    stamps.clear()
    t0 = time.perf_counter(); P.acquisition(eng, S); t1 = time.perf_counter()
    (_, c0, c1), (_, f0, f1) = stamps
    rows.append((c0 - t0, c1 - c0, f0 - c1, f1 - f0, t1 - f1, t1 - t0))
r = np.median(np.array(rows[100:]), axis=0) * 1e3
print("median of 200 sustained calls, ms: before coarse %.3f | coarse call %.3f | between %.3f | fine call %.3f | after %.3f | total %.3f" % tuple(r))
