"""Where a wavefront of the fused passes spends its cycles (a tuning build: scripts/variants.sh acq "CLK:-DGC_ACQ_STAGE_CLOCKS=1";
GC_LIB_PATH=.../libgnsscorr_CLK.so python scripts/acq_stage_clocks.py [package ...]).  Without arguments: the default GPS L1 C/A search
on the bench's scene; with package names (GPS_L5C, GAL_E1C, BDS_B1C, GPS_L2C ...): that package's default search on its fixture's record.
Cycles per hop (per tile where a bin has one hop) and wavefront, shader clock: [wait for the prefetched tile + first stage | issue of the
next fetch + barrier | middle stage | barrier | the rest: second middle stage of four-stage plans, last stage with |.| or the stores]."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cu_sdr_collection_amd as P
from cu_sdr_collection_amd import _lib as L

lib = C.CDLL(L.LIB_PATH)
if not hasattr(lib, "gc_debug_acq_stage_clocks"):
    sys.exit("this library was not built with -DGC_ACQ_STAGE_CLOCKS=1")
buf = (C.c_ulonglong * 128)()
names = ["wait tile + first stage", "fetch issue + barrier", "middle stage", "barrier", "rest / last stage", "loop top"]


def report(tag):
    a = np.array(list(buf), dtype=np.float64).reshape(2, 8, 8)
    print("==", tag)
    for k, pas in enumerate(("columns pass", "rows pass")):
        for w in range(8):
            hops = a[k, w, 7]
            if hops == 0:
                continue
            per = a[k, w, :6] / hops
            print("  %s, wavefront %d: %6.0f cycles per hop = " % (pas, w, per.sum()) + ", ".join("%s %5.0f" % (nm, v) for nm, v in zip(names, per)))


if len(sys.argv) == 1:
    S = P.initSettings()
    sats = P.synth.scene(12, 5, S.samplingFreq)
    iq = P.synth.generate_if(sats, int(0.1 * S.samplingFreq), S.samplingFreq, S.IF, P.codes.generateCAcode, S.codeFreqBasis, 1023, seed=3)
    eng = P.Engine(0)
    eng.load_if(iq, fs=S.samplingFreq)
    for _ in range(10):
        P.acquisition(eng, S)
    lib.gc_debug_acq_stage_clocks(buf, 1)
    for _ in range(20):
        P.acquisition(eng, S)
    lib.gc_debug_acq_stage_clocks(buf, 0)
    report("GPS L1 C/A, default search, bench scene (20 searches)")
else:
    import ref_scenes as RS
    for want in sys.argv[1:]:
        sc = next(s for s in RS.DEFAULT_ACQ_SCENES if s.name == want + "_default")
        S, rec = RS.acq_inputs(P, sc)
        with P.Engine(0) as eng:
            eng.load_if(rec, fs=S.samplingFreq)
            sc.product(P, eng, S)
            sc.product(P, eng, S)
            lib.gc_debug_acq_stage_clocks(buf, 1)
            sc.product(P, eng, S)
            lib.gc_debug_acq_stage_clocks(buf, 0)
        report(want + ", default search on its fixture's record")
