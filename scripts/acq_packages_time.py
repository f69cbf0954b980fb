#!/usr/bin/env python3
"""bench.py's acquisition.packages leg alone (the twelve default-size searches, warm-up calls, median of three, array_equal to the
fixtures) + the float64 guard's statistics per package.  python scripts/acq_packages_time.py [PKG ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import cu_sdr_collection_amd as P  # noqa: E402

out = bench.run_acquisition_packages(P, 0, only=set(sys.argv[1:]) or None)
print(json.dumps({k: {"ms": v.get("ms"), "event_ms": v.get("event_ms"), "equal": v.get("equal_to_the_references_acquisition_m"), "guard": v.get("float64_guard"),
                      "metric_dev": v.get("peak_metric_max_rel_dev")} for k, v in out.items()}, indent=1))
