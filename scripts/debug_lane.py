import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cu_sdr_collection_amd as P
from oracle import gnss_oracle as O
fs = 18e6; n_if = 400000
rng = np.random.default_rng(11)
iq = np.clip(np.rint(20 * rng.standard_normal(2 * n_if)), -127, 127).astype(np.int8)
rng = np.random.default_rng(12)
eng = P.Engine(0); eng.load_if(iq, fs=fs)
L = {6: 10230.0, 7: 16382.0}; tabs = {}
for ch in (6, 7):
    tabs[ch] = [O.pad_code(rng.choice([-1.0, 1.0], size=int(L[ch]))) for _ in range(2)]
only6 = "--only6" in sys.argv
for ch in ((6,) if only6 else (6, 7)):
    eng.set_channel(ch, [t.astype(np.int8) for t in tabs[ch]])
step = 10.23e6 / fs
for (n, d, stp, f) in [(64, 0.3, 0.2, 0.0), (64, 0.5, 0.2, 0.0), (64, 0.3, step, 0.0), (256, 0.3, 0.2, 0.0), (300, 0.3, 0.2, 0.0), (1000, 0.3, 0.2, 0.0), (5000, 0.3, 0.2, 0.0), (5000, 0.5, 0.2, 0.0)]:
    rem = 0.1
    b = eng.make_blocks(1)
    b[0].channel = 6; b[0].blksize = n; b[0].first_sample = 1000; b[0].rem_code_phase = rem; b[0].code_phase_step = stp
    b[0].el_spacing = d; b[0].carr_freq = f; b[0].rem_carr_phase = 0.0
    got = eng.correlate(b)
    ref, _, _ = O.correlate_block(O.raw_from_if(iq, 1000, n), tabs[6], rem, stp, d, f, 0.0, fs, L[6])
    print("n", n, "d", d, "step", round(stp, 3), "err", np.round(np.abs(got[0, :2] - ref)[0], 2), "got", np.round(got[0, 0], 1), "ref", np.round(ref[0], 1))
