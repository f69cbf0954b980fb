"""Times the default GPS L1 C/A acquisition (32 PRNs x 29 bins x 20 ms, fine stage included) on the bench's scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cu_sdr_collection_amd as P

S = P.initSettings()
sats = P.synth.scene(12, 5, S.samplingFreq)
iq = P.synth.generate_if(sats, int(0.1 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=3)
eng = P.Engine(0)
eng.load_if(iq, fs=S.samplingFreq)
r = P.acquisition(eng, S)
best = 1e9
for _ in range(5):
    t = time.perf_counter(); r = P.acquisition(eng, S); best = min(best, time.perf_counter() - t)
print("acquisition best of 5: %.2f ms, PRNs found: %s" % (best * 1e3, [int(i) + 1 for i in np.nonzero(r.carrFreq)[0]]))
