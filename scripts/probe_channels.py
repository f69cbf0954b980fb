import sys, os, copy
sys.path.insert(0, '/root/repo')
import numpy as np
import bench_workloads as W
import cu_sdr_collection_amd as P
from cu_sdr_collection_amd import _lib as L
owner = P.Engine(0)
(pkg, S, scene), = W.make_band(P, owner, [("GPS_L1CA", 32)], 3.0, 18e6, 20e3, 7007)
n_ep = int((3.0 - 3 * S.intTime) / S.intTime) - 1
for n in [int(x) for x in os.environ.get("PROBE_N", "24,32,40,48,96").split(",")]:
    eng = P.Engine(0); eng.share_if(owner); eng.set_sampling_freq(18e6)
    sats = [scene[i % 32] for i in range(n)]
    job = W.prepare_job(P, W.Job("x", pkg, copy.copy(S), sats, eng), n_ep)
    for dl in (False, True):
        try:
            W.run_closed_loops(P, [job], device_loop=dl)
            t, _ = W.run_closed_loops(P, [job], device_loop=dl)
            print(n, dl, eng.last_track_mode(), round(t / n_ep * 1e6, 2), flush=True)
        except Exception as e:
            print(n, dl, "ERR", str(e)[:200], flush=True)
    lib = L.load()
    print("  last error:", lib.gc_last_error().decode()[:200] if hasattr(lib, 'gc_last_error') else None, flush=True)
    eng.close()
