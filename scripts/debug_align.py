import sys, numpy as np
sys.path.insert(0, '.')
import cu_sdr_collection_amd as P
from oracle import c_oracle as CO, gnss_oracle as O
S = P.initSettings()
sats = P.synth.scene(2, 7, S.samplingFreq)
iq = P.synth.generate_if(sats, int(0.03 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=3)
eng = P.Engine(0); eng.load_if(iq, fs=S.samplingFreq)
tab = O.pad_code(O.generate_ca_code(sats[1].prn)); eng.set_channel(0, [tab.astype(np.int8)])
step = 1.023e6/18e6
for rem in (0.0, 1e-9, step*0.5):
  for s0 in range(88, 104):
    b = eng.make_blocks(1)
    b[0].channel = 0; b[0].blksize = 18000; b[0].first_sample = s0; b[0].rem_code_phase = rem
    b[0].code_phase_step = step; b[0].el_spacing = 0.5; b[0].carr_freq = 21000.0; b[0].rem_carr_phase = 0.0
    g = eng.correlate(b)[0, 0]
    ref, _, _ = CO.correlate_block(iq, s0, 18000, [tab], rem, step, 0.5, 21000.0, 0.0, S.samplingFreq, 1023.0)
    print(rem, s0, s0 & 7, (g - ref[0]).round(3), iq[2*s0:2*s0+2])
