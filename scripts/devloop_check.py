"""Device-side vs host-side loop closure over a long record: timing per split count, first divergence of the block geometry."""
import os as _os
_os.environ.setdefault("GC_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cu-sdr-collection_amd", "lib", "libgnsscorr_tuning.so"))  # the GC_* switches used below exist in the tuning build only (docs/KNOBS.md)
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cu_sdr_collection_amd as P
from cu_sdr_collection_amd import _lib as L
from cu_sdr_collection_amd.receiver import track_params
S = P.initSettings()
fs = S.samplingFreq
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
nch = 12
rng = np.random.default_rng(20241010)
prns = rng.choice(np.arange(1, 33), size=nch, replace=False)
sats = [P.synth.SatSpec(prn=int(p), doppler=float(rng.uniform(-5e3, 5e3)), code_phase_samples=float(rng.uniform(0, 18000)),
                        carrier_phase=float(rng.uniform(0, 6.28)), cn0_dbhz=45.0) for p in prns]
eng = P.Engine(0)
P.synth.generate_if_gpu(eng, sats, int(seconds * fs), fs, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=7)
eng.set_sampling_freq(fs)
S.msToProcess = int(seconds * 1000) - 3
p = track_params(S)
inits = []
for i, s in enumerate(sats):
    eng.set_channel(i, [P.codes.padded_table(P.codes.generateCAcode(s.prn))])
    inits.append(L.gc_channel_init(channel=i, prn=s.prn, acquired_freq=S.IF + s.doppler + 3.0, code_freq=S.codeFreqBasis,
                                   code_phase=int(np.ceil(s.code_phase_samples)) + 1))
t0 = time.time(); hf, hd, hs = eng.track(p, inits); th = time.time() - t0
print("host loop us/epoch", round(th / p.n_epochs * 1e6, 2))
for sp in (None, 2, 4, 8, 12, 17, 24, 32, 48, 64):
    if sp is None:
        os.environ.pop("GC_DEVLOOP_MEMBERS", None)
    else:
        os.environ["GC_DEVLOOP_MEMBERS"] = str(sp)
    t0 = time.time(); df, dd, ds = eng.track(p, inits, device_loop=True); td = time.time() - t0
    same = np.array_equal(df["absoluteSample"], hf["absoluteSample"])
    print("device loop splits", sp, "us/epoch", round(td / p.n_epochs * 1e6, 2), "status", ds, "same geometry", same,
          "max carrFreq dev", float(np.max(np.abs(df["carrFreq"] - hf["carrFreq"]))))
    if not same and sp is None:
        d = np.argwhere(df["absoluteSample"] != hf["absoluteSample"])
        c, e = d[0]
        print("  first difference: channel", c, "epoch", e, "abs", df["absoluteSample"][c, e - 1:e + 2], hf["absoluteSample"][c, e - 1:e + 2])
        print("  remCodePhase", df["remCodePhase"][c, e - 2:e + 1], hf["remCodePhase"][c, e - 2:e + 1])
        print("  codeFreq", df["codeFreq"][c, e - 2:e + 1] - 1.023e6, hf["codeFreq"][c, e - 2:e + 1] - 1.023e6)
        n_d = (1023 - df["remCodePhase"][c, e - 1]) / (df["codeFreq"][c, e - 1] / fs)
        n_h = (1023 - hf["remCodePhase"][c, e - 1]) / (hf["codeFreq"][c, e - 1] / fs)
        print("  (L - rem)/step before ceil:", repr(n_d), repr(n_h))

# which loop follows the float64 oracle?  (channel 0, first 8000 epochs)
if "--oracle" in sys.argv:
    from types import SimpleNamespace
    from oracle import c_oracle as CO
    os.environ.pop("GC_DEVLOOP_MEMBERS", None)
    nE = 8000
    S.msToProcess = nE
    p2 = track_params(S)
    hf, _, _ = eng.track(p2, inits[:1])
    df, _, _ = eng.track(p2, inits[:1], device_loop=True)
    iq = eng.read_if(0, int((nE + 5) * 1e-3 * fs))
    ch = [SimpleNamespace(PRN=sats[0].prn, acquiredFreq=inits[0].acquired_freq, codePhase=inits[0].code_phase, status="T")]
    S.numberOfChannels = 1
    ref, done, ab = CO.track_l1ca(iq, ch, S)
    for name, f in (("host", hf), ("device", df)):
        for fld in ("codeFreq", "carrFreq", "remCodePhase"):
            d = np.abs(f[fld][0] - ref[fld][0])
            print(name, fld, "max dev vs oracle over epochs [0,2000) [2000,6000) [6000,8000):", d[:2000].max(), d[2000:6000].max(), d[6000:].max())
        print(name, "geometry equal to oracle:", np.array_equal(f["absoluteSample"][0], ref["absoluteSample"][0]))
