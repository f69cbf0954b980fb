"""The twelve default-size searches (bench.py's acquisition.packages leg) on their own: timings, and with --profile the host
side's cProfile top entries per package (where a call's milliseconds go outside the kernels).
    python scripts/acq_packages.py [--profile] [--only GPS_L5C,GAL_E5a]
    rocprofv3 --kernel-trace --stats -d gpurun_out/acqpkg -- python scripts/acq_packages.py --only GAL_E5b"""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import cu_sdr_collection_amd as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--profile", action="store_true")
ap.add_argument("--only", default=None)
a = ap.parse_args()
only = a.only.split(",") if a.only else None
if a.profile:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_scenes as RS
    for sc in RS.DEFAULT_ACQ_SCENES:
        name = sc.name[:-len("_default")]
        if only and name not in only:
            continue
        S, rec = RS.acq_inputs(P, sc)
        with P.Engine(0) as eng:
            eng.load_if(rec, fs=S.samplingFreq)
            sc.product(P, eng, S)
            pr = cProfile.Profile()
            pr.enable()
            sc.product(P, eng, S)
            pr.disable()
        out = io.StringIO()
        pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(14)
        print("=" * 30, name)
        print("\n".join(l for l in out.getvalue().splitlines()[4:] if l.strip()))
else:
    print(json.dumps(bench.run_acquisition_packages(P, 0, only), indent=1))
