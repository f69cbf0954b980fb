import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["GC_LIB_PATH"] = "/root/repo/cu-sdr-collection_amd/lib/libgnsscorr_tuning.so"
import numpy as np
import cu_sdr_collection_amd as P
from cu_sdr_collection_amd import _lib as L
from cu_sdr_collection_amd.receiver import track_params
S = P.initSettings(); fs = S.samplingFreq; seconds = 10.0; nch = 12
rng = np.random.default_rng(20241010)
prns = rng.choice(np.arange(1, 33), size=nch, replace=False)
sats = [P.synth.SatSpec(prn=int(p), doppler=float(rng.uniform(-5e3, 5e3)), code_phase_samples=float(rng.uniform(0, 18000)), carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=45.0) for p in prns]
eng = P.Engine(0)
P.synth.generate_if_gpu(eng, sats, int(seconds * fs), fs, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=7)
eng.set_sampling_freq(fs)
S.msToProcess = int(seconds * 1000) - 3
p = track_params(S)
inits = []
for i, s in enumerate(sats):
    eng.set_channel(i, [P.codes.padded_table(P.codes.generateCAcode(s.prn))])
    inits.append(L.gc_channel_init(channel=i, prn=s.prn, acquired_freq=S.IF + s.doppler + 3.0, code_freq=S.codeFreqBasis, code_phase=int(np.ceil(s.code_phase_samples)) + 1))
eng.track(p, inits, device_loop=True)
for scope in ("0", "2"):
    os.environ["GC_DEVLOOP_SCOPE"] = scope
    for rep in range(2):
        t0 = time.time(); eng.track(p, inits, device_loop=True); td = time.time() - t0
    print("scope", scope, "us/epoch", round(td / p.n_epochs * 1e6, 2), flush=True)
os.environ["GC_DEVLOOP_SCOPE"] = "0"
os.environ["GC_DEVLOOP_TIMING"] = "1"
eng.track(p, inits, device_loop=True)
