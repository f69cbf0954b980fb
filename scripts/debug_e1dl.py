import sys, numpy as np
sys.path.insert(0, ".")
from types import SimpleNamespace
import cu_sdr_collection_amd as P
from cu_sdr_collection_amd.settings import initSettings_GAL_E1C
rng = np.random.default_rng(9)
for _ in range(2):
    [rng.uniform(-3e3, 3e3), rng.uniform(0, 18000), rng.uniform(0, 6.28)]
S = initSettings_GAL_E1C()
fs = S.samplingFreq
S.msToProcess = 80
S.numberOfChannels = 2
sats = [P.synth.SatSpec(prn=p, doppler=float(rng.uniform(-3e3, 3e3)), code_phase_samples=float(rng.uniform(0, 72000)),
                        carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=48.0) for p in (4, 19)]
iq = P.synth.generate_if(sats, int(0.090 * fs), fs, S.IF, P.codes.generateE1Bcode, 2 * S.codeFreqBasis, 8184, seed=21,
                         bit_periods=1, pilot_fn=P.codes.generateE1Ccode)
ch = [SimpleNamespace(PRN=s.prn, acquiredFreq=S.IF + s.doppler + 2.0, status="T", codePhase=int(np.ceil(s.code_phase_samples)) + 1) for s in sats]
eng = P.Engine()
eng.load_if(iq, fs=fs)
host, _ = P.tracking(eng, ch, S, signal="GAL_E1C")
dev, _ = P.tracking(eng, ch, S, signal="GAL_E1C", device_loop=True)
np.set_printoptions(linewidth=220, precision=3, suppress=True)
for f in ("remCarrPhase", "I_E", "Q_P", "I_P", "Pilot_I_P", "dllDiscr", "pllDiscr", "pllDiscrFilt"):
    print(f, "dev ", getattr(dev[0], f)[:8])
    print(f, "host", getattr(host[0], f)[:8])
