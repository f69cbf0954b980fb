#!/usr/bin/env python3
"""One BASELINE shape's replay kernel, a few launches, for rocprofv3 (kernel trace / PMC passes):
    python scripts/prof_shapes.py l5|b2a|cboc|b1c|e1|l1ca3 [seconds] [launches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as W  # noqa: E402
import cu_sdr_collection_amd as P  # noqa: E402

SHAPES = {"l5": ([("GPS_L5C", 8)], 50e6), "b2a": ([("BDS_B2a", 8)], 50e6), "l5_18": ([("GPS_L5C", 8)], 18e6), "cboc": ([("GAL_E1C_CBOC", 8)], 18e6),
          "b1c": ([("BDS_B1C_NB", 2)], 18e6), "e1": ([("GAL_E1C", 3)], 18e6), "l1ca3": ([("GPS_L1CA", 3)], 18e6), "e1x8": ([("GAL_E1C", 8)], 18e6),
          "b1i": ([("BDS_B1I", 8)], 18e6), "b3i": ([("BDS_B3I", 8)], 18e6), "l2c": ([("GPS_L2C", 4)], 8e6), "glo": ([("GLO_GL1", 8)], 12e6)}


def main():
    shape = sys.argv[1]
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    parts, fs = SHAPES[shape]
    eng = P.Engine(0)
    (pkg, S, sats), = W.make_band(P, eng, parts, seconds, fs, 20e3, 4004)
    n_ep = int((seconds - 3 * S.intTime) / S.intTime) - 1
    job = W.prepare_job(P, W.Job(shape, pkg, S, sats, eng), n_ep)
    _, recs = W.run_closed_loops(P, [job], device_loop=False)
    W.keep_records(job, recs[0])
    ms, dev, kern = W.time_replay(job, launches, 2, warm=W.warm_engine(P, job))
    cs = float(job.blks.sum())
    print({"shape": shape, "kernel": W.KERNEL_NAMES.get(kern), "ms": round(ms, 4), "GBps": round(2 * cs / ms / 1e6, 1), "frac": round(2 * cs / ms / 1e6 / 8000, 4),
           "channel_samples_per_launch": cs, "dev": dev, "locked": W.locked(job)})


if __name__ == "__main__":
    main()
