#!/usr/bin/env python3
"""The device-closed loop of 12 GPS L1 C/A channels (and 16 GPS L5 channels at 50 Msps) with and without the next epoch's first chunk
fetched during the closure (corr_fast.hip pf_w; GC_DEVLOOP_NO_PREFETCH=1 on the tuning build): microseconds per epoch, the closer's
phase clocks (GC_DEVLOOP_TIMING=1 - they cost ~1 us per epoch themselves), and that both runs return the same records."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GC_LIB_PATH", os.path.join(ROOT, "cu-sdr-collection_amd", "lib", "libgnsscorr_tuning.so"))
import numpy as np  # noqa: E402
import cu_sdr_collection_amd as P  # noqa: E402
from cu_sdr_collection_amd import _lib as L  # noqa: E402
from cu_sdr_collection_amd.receiver import track_params  # noqa: E402

S = P.initSettings()
fs, seconds = S.samplingFreq, float(os.environ.get("AB_SECONDS", "10"))
for nch in [int(v) for v in os.environ.get("AB_CHANNELS", "12,48").split(",")]:
    rng = np.random.default_rng(20241010)
    prns = rng.choice(np.arange(1, 33), size=min(nch, 32), replace=False)
    sats = [P.synth.SatSpec(prn=int(prns[i % len(prns)]), doppler=float(rng.uniform(-5e3, 5e3)), code_phase_samples=float(rng.uniform(0, 18000)),
                            carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=45.0) for i in range(min(nch, 32))]
    eng = P.Engine(0)
    P.synth.generate_if_gpu(eng, sats, int(seconds * fs), fs, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=7)
    eng.set_sampling_freq(fs)
    S.msToProcess = int(seconds * 1000) - 3
    p = track_params(S)
    inits = []
    for i in range(nch):
        s = sats[i % len(sats)]
        eng.set_channel(i, [P.codes.padded_table(P.codes.generateCAcode(s.prn))])
        inits.append(L.gc_channel_init(channel=i, prn=s.prn, acquired_freq=S.IF + s.doppler + 3.0, code_freq=S.codeFreqBasis, code_phase=int(np.ceil(s.code_phase_samples)) + 1))
    eng.track(p, inits, device_loop=True)
    ref = None
    for label, env in (("prefetch", None), ("no prefetch", "1"), ("prefetch", None), ("no prefetch", "1")):
        if env:
            os.environ["GC_DEVLOOP_NO_PREFETCH"] = env
        else:
            os.environ.pop("GC_DEVLOOP_NO_PREFETCH", None)
        best = 1e9
        for rep in range(3):
            t0 = time.time()
            fields, done, st = eng.track(p, inits, device_loop=True)
            best = min(best, time.time() - t0)
        key = np.concatenate([np.asarray(fields[f]).ravel() for f in ("I_P", "Q_P", "carrFreq", "codeFreq", "absoluteSample")])
        same = True if ref is None else bool(np.array_equal(key, ref))
        ref = key if ref is None else ref
        print(f"{nch} ch  {label:12s} {best / p.n_epochs * 1e6:6.2f} us/epoch = {1e-3 / (best / p.n_epochs):6.1f} x real time   records identical to the first run: {same}   mode {eng.last_track_mode()}", flush=True)
    eng.close()
