"""Debug: acquisition -> tracking on the acq_scene record; prints per-epoch prompt sums and block geometry."""
import sys, numpy as np
sys.path.insert(0, ".")
import cu_sdr_collection_amd as P
S = P.initSettings()
S.acqNonCohTime = 4
S.acqSatelliteList = [3, 7, 11, 14, 19, 22, 28, 31]
rng = np.random.default_rng(5)
sats = [P.synth.SatSpec(prn=p, doppler=float(rng.uniform(-5e3, 5e3)), code_phase_samples=float(rng.uniform(0, 18000)),
                        carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=cn0)
        for p, cn0 in ((7, 50.0), (14, 47.0), (22, 44.0), (31, 52.0))]
n = 44 * 18000
iq = P.synth.generate_if(sats, n, S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=12)
eng = P.Engine()
eng.load_if(iq, fs=S.samplingFreq)
S.numberOfChannels = 6
S.msToProcess = 40
acq = P.acquisition(eng, S)
ch = P.preRun(acq, S)
print([(c.PRN, c.codePhase, c.acquiredFreq) for c in ch])
tr, _ = P.tracking(eng, ch, S)
np.set_printoptions(linewidth=200, precision=1, suppress=True)
for k in range(4):
    print(k, tr[k].status, "abs", tr[k].absoluteSample[:4], "rem", tr[k].remCodePhase[:4])
    for f in ("I_E", "Q_E", "I_P", "Q_P", "I_L", "Q_L"):
        print("   ", f, getattr(tr[k], f)[:5])
