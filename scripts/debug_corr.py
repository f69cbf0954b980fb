import sys, numpy as np
sys.path.insert(0, '.')
import cu_sdr_collection_amd as P
from oracle import c_oracle as CO, gnss_oracle as O
S = P.initSettings(); fs, fc = S.samplingFreq, S.codeFreqBasis
n = int(0.02 * fs); A = 60.0
code = P.codes.generateCAcode(7).astype(np.float64)
t = np.arange(n) / fs
f0 = S.IF + 1234.567
chip = np.floor(np.arange(n) * (fc / fs)).astype(np.int64) % 1023
x = A * code[chip] * np.exp(1j * (2 * np.pi * f0 * t + 0.7))
iq = np.empty(2 * n, dtype=np.int8); iq[0::2] = np.rint(x.real); iq[1::2] = np.rint(x.imag)
tab = O.pad_code(O.generate_ca_code(7))
eng = P.Engine(0); eng.load_if(iq, fs=fs); eng.set_channel(0, [tab.astype(np.int8)])
step = fc / fs
for nrep in (1, 64, 600):
    b = eng.make_blocks(nrep)
    descs = []
    for k in range(nrep):
        s0 = (k * 18000) % (n - 36000) if nrep > 1 else 0
        s0 = int(round(s0 / (fs / fc)) ) # arbitrary
        e = k % 10
        s0 = 18000 * e
        rem = (s0 * step) % 1.0 if False else 0.0
        N = int(np.ceil((1023 - rem) / step))
        b[k].channel = 0; b[k].blksize = N; b[k].first_sample = s0
        b[k].rem_code_phase = (s0 * step) - np.floor(s0 * step) if s0 else 0.0
        b[k].code_phase_step = step; b[k].el_spacing = 0.5
        b[k].carr_freq = f0; b[k].rem_carr_phase = float(np.fmod(2 * np.pi * f0 * s0 / fs + 0.7, 2 * np.pi))
    got = eng.correlate(b)[:, 0]
    worst = 0
    for k in range(min(nrep, 12)):
        ref, _, _ = CO.correlate_block(iq, b[k].first_sample, b[k].blksize, [tab], b[k].rem_code_phase, step, 0.5, f0, b[k].rem_carr_phase, fs, 1023.0)
        err = got[k] - ref[0]
        worst = max(worst, np.abs(err).max())
        if k < 3: print(nrep, k, 'ref', ref[0].round(2), 'err', err.round(3))
    print('nrep', nrep, 'worst abs err', worst)
