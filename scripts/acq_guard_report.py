#!/usr/bin/env python3
"""Per acquisition scene (reduced, guard and default-size): peakMetric deviation from the reference fixture, the float64 guard's
statistics (ties resolved, largest float32-vs-float64 peak deviation, eps) and the search time.  Run on the GPU box:
    python scripts/acq_guard_report.py > gpurun_out/<tag>/acq_guard.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cu_sdr_collection_amd as P  # noqa: E402
import ref_scenes as RS  # noqa: E402

out = {}
for sc in RS.ACQ_SCENES + RS.GUARD_ACQ_SCENES + RS.DEFAULT_ACQ_SCENES:
    path = os.path.join(ROOT, "tests", "golden", f"ref_acq_{sc.name}.npz")
    if not os.path.exists(path):
        continue
    z = np.load(path)
    S, rec = RS.acq_inputs(P, sc)
    with P.Engine(0) as eng:
        eng.load_if(rec, fs=S.samplingFreq)
        sc.product(P, eng, S)
        t0 = time.perf_counter()
        got = sc.product(P, eng, S)
        ms = (time.perf_counter() - t0) * 1e3
        try:
            st = eng.acq_guard_stats()
        except Exception as e:  # noqa: BLE001
            st = {"error": repr(e)}
    want = z["f_peakMetric"]
    have = np.asarray(got.peakMetric, dtype=np.float64)
    out[sc.name] = {"ms": round(ms, 3), "peak_metric_max_rel_dev": float(np.max(np.abs(have - want)) / np.max(np.abs(want))),
                    "positions_equal": bool(np.array_equal(np.asarray(got.codePhase, dtype=np.float64), z["f_codePhase"])
                                            and np.array_equal(np.asarray(got.carrFreq, dtype=np.float64), z["f_carrFreq"])), "guard": st}
    print(sc.name, json.dumps(out[sc.name]), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
