"""Where does the GPU closed loop first differ from the C oracle on the bench record?  Prints the worst (channel, epoch), then
recomputes that block with gc_correlate, the C oracle and the NumPy oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import cu_sdr_collection_amd as P
from cu_sdr_collection_amd import _lib as L
from cu_sdr_collection_amd.receiver import track_params
from oracle import c_oracle as CO
from oracle import gnss_oracle as O
S = P.initSettings(); fs = S.samplingFreq; nch = 12
n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
S.msToProcess = n_ep; S.numberOfChannels = nch
eng = P.Engine(0)
sats = P.synth.scene(nch, 20241008 + 2, fs)
P.synth.generate_if_gpu(eng, sats, int((n_ep + 5) * 1e-3 * fs), fs, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=20241008 + 2)
eng.set_sampling_freq(fs)
inits = []
for i, s in enumerate(sats):
    eng.set_channel(i, [P.codes.padded_table(P.codes.generateCAcode(s.prn))])
    inits.append(L.gc_channel_init(channel=i, prn=s.prn, acquired_freq=S.IF + s.doppler + 3.0, code_freq=S.codeFreqBasis, code_phase=int(np.ceil(s.code_phase_samples)) + 1))
fields, done, st = eng.track(track_params(S), inits)
CO.build(force=False)
iq = eng.read_if(0, int((n_ep + 3) * 1e-3 * fs))
ch = [SimpleNamespace(PRN=i.prn, acquiredFreq=i.acquired_freq, codePhase=i.code_phase, status="T") for i in inits]
ref, cdone, ab = CO.track_l1ca(iq, ch, S)
same = ref["absoluteSample"] == fields["absoluteSample"]
same[:, :-1] &= same[:, 1:]
worst = (0, None)
for c in range(nch):
    n = int(np.argmin(same[c])) if not same[c].all() else n_ep
    for f in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L"):
        d = np.abs(ref[f][c, :n] - fields[f][c, :n])
        e = int(np.argmax(d))
        if d[e] > worst[0]: worst = (float(d[e]), (c, e, f, n))
print("worst", worst)
c, e, f, n = worst[1]
# first epoch of that channel with a deviation above 1e-2
for f2 in ("I_E", "I_P", "I_L"):
    d = np.abs(ref[f2][c, :n] - fields[f2][c, :n]); idx = np.nonzero(d > 1e-2)[0]
    print(f2, "first epochs with |dev| > 1e-2:", idx[:6], d[idx[:6]])
e = int(min(np.nonzero(np.abs(ref[ff][c, :n] - fields[ff][c, :n]) > 1e-2)[0][0] for ff in ("I_E", "I_P", "I_L") if (np.abs(ref[ff][c, :n] - fields[ff][c, :n]) > 1e-2).any()))
print("channel", c, "epoch", e, "state GPU vs C: rem", repr(fields["remCodePhase"][c, e]), repr(ref["remCodePhase"][c, e]), "codeFreq", repr(fields["codeFreq"][c, e]), repr(ref["codeFreq"][c, e]),
      "carr", repr(fields["carrFreq"][c, e]), repr(ref["carrFreq"][c, e]), "remCarr", repr(fields["remCarrPhase"][c, e]), repr(ref["remCarrPhase"][c, e]))
for src, F in (("gpu", fields), ("C  ", ref)):
    b = eng.make_blocks(1)
    step = F["codeFreq"][c, e] / fs
    b[0].channel = c; b[0].first_sample = int(F["absoluteSample"][c, e]); b[0].rem_code_phase = F["remCodePhase"][c, e]
    b[0].code_phase_step = step; b[0].blksize = int(np.ceil((S.codeLength - F["remCodePhase"][c, e]) / step)); b[0].el_spacing = 0.5
    b[0].carr_freq = F["carrFreq"][c, e]; b[0].rem_carr_phase = F["remCarrPhase"][c, e]
    g = eng.correlate(b)[0, 0]
    tab = O.pad_code(O.generate_ca_code(sats[c].prn))
    r, _, _ = CO.correlate_block(iq, b[0].first_sample, b[0].blksize, [tab], b[0].rem_code_phase, b[0].code_phase_step, 0.5, b[0].carr_freq, b[0].rem_carr_phase, fs, 1023.0)
    r2, _, _ = O.correlate_block(iq, b[0].first_sample, b[0].blksize, [tab], b[0].rem_code_phase, b[0].code_phase_step, 0.5, b[0].carr_freq, b[0].rem_carr_phase, fs, 1023.0)
    print(src, "state: gc_correlate", g, "\n     C oracle   ", r[0], "\n     numpy     ", np.asarray(r2[0]))
    print("     recorded gpu", [F2[c, e] for F2 in (fields["I_E"], fields["Q_E"], fields["I_P"], fields["Q_P"], fields["I_L"], fields["Q_L"])])
