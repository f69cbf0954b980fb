"""Condenses rocprofv3 CSV output (kernel trace / stats / counter collection) into small text
summaries that can be committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
lines = []
for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
    lines.append(f"# {os.path.basename(f)}")
    lines += [l.rstrip() for l in open(f)]
for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
    agg = defaultdict(list)
    meta = {}
    rd = csv.DictReader(open(f))
    lines.append(f"# columns: {rd.fieldnames}")
    for r in rd:
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        k = f'{r["Kernel_Name"][:70]} grid={grid}'
        agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = {c: r.get(c) for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                         "Workgroup_Size_X", "Grid_Size_X") if c in r}
    lines.append(f"# {os.path.basename(f)} (durations in ns, grouped by kernel and grid size)")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k[:90]}: calls={len(v)} total_ns={sum(v)} avg_ns={sum(v)/len(v):.0f} min_ns={min(v)} max_ns={max(v)} {meta[k]}")
for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(list))
    rd = csv.DictReader(open(f))
    lines.append(f"# columns: {rd.fieldnames}")
    for r in rd:
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        agg[f'{r["Kernel_Name"][:70]} grid={grid}'][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append(f"# {os.path.basename(f)} (per-kernel mean counter value per dispatch)")
    for k, cs in agg.items():
        for c, v in cs.items():
            lines.append(f"{k[:90]}: {c} mean={sum(v)/len(v):.6g} n={len(v)}")
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
