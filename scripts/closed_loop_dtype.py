"""Closed-loop time of config 4's two concurrent jobs (8 x GPS L5 + 8 x BDS B2a on one 50-Msps record, gc_track_multi) for an int8
and an int16 record, host-closed and device-closed:  python scripts/closed_loop_dtype.py [seconds] [order: 8,16 | 16,8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench_workloads as W
import cu_sdr_collection_amd as P

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
order = [np.int8, np.int16] if (len(sys.argv) < 3 or sys.argv[2].startswith("8")) else [np.int16, np.int8]
for dtype in order:
    parts = [("GPS_L5C", 8), ("BDS_B2a", 8)]
    engines = [P.Engine(0) for _ in parts]
    made = W.make_band(P, engines[0], parts, seconds, 50e6, 20e3, 4004, dtype=dtype)
    engines[1].share_if(engines[0])
    jobs = []
    for (pkg, S, sats), eng in zip(made, engines):
        n_ep = int((seconds - 3 * S.intTime) / S.intTime) - 1
        jobs.append(W.prepare_job(P, W.Job(pkg.signal, pkg, S, sats, eng), n_ep))
    for dl in (False, True):
        W.run_closed_loops(P, jobs, device_loop=dl)
        t, recs = W.run_closed_loops(P, jobs, device_loop=dl)
        t1, _ = W.run_closed_loops(P, jobs[:1], device_loop=dl)
        print(np.dtype(dtype).name, "device" if dl else "host", "both jobs %.3f s = %.2f us/epoch; one job alone %.2f us/epoch" % (t, t / n_ep * 1e6, t1 / n_ep * 1e6), flush=True)
    for e in engines:
        e.close()
