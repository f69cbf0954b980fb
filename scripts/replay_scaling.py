"""Replay time of a periodic block list against its length: packages x channels x record seconds -> ms per launch, fraction of the
HBM figure, kernel chosen (gc_debug_last_kernel).  Run on the GPU box: python scripts/replay_scaling.py"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bench_workloads as W
import cu_sdr_collection_amd as P
import test_gpu_full_size as T
for name, nch, fs in (("GPS_L1CA", 3, 18e6), ("GPS_L1CA", 12, 18e6), ("GAL_E1C", 3, 18e6), ("GAL_E1C", 8, 18e6)):
    for secs in (2.0, 5.0, 10.0, 20.0):
        engines, jobs = T._band(P, W, [(name, nch)], secs, fs, 20241008 + 3)
        _, recs = W.run_closed_loops(P, jobs, device_loop=True)
        j = jobs[0]
        W.keep_records(j, recs[0])
        ms, dev, kern = W.time_replay(j, 20, 3, warm=W.warm_engine(P, j))
        cs = float(j.blks.sum())
        print(f"{name} x{nch} {secs:5.1f} s: {ms:.4f} ms, {2 * cs / ms / 1e6 / 8000:.3f} of 8 TB/s, kernel {kern}, blocks {j.blks.size}", flush=True)
        for e in engines: e.close()
