#!/bin/bash
# rocprofv3 kernel statistics of the default-size acquisition of single packages (scripts/acq_packages.py --only <pkg>):
#   usage (GPU box): scripts/prof_acq_packages.sh <tag> PKG [PKG ...]   -> gpurun_out/<tag>/acq_<PKG>_stats.txt
TAG=$1; shift
OUT=/root/repo/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for p in "$@"; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/acq_${p}_stats" -- python /root/repo/scripts/acq_packages.py --only $p > "$OUT/acq_${p}.json" 2>/dev/null
  (python /root/repo/scripts/prof_summarize.py "$OUT/acq_${p}_stats" "$OUT/acq_${p}_stats.txt" > /dev/null; rm -rf "$OUT/acq_${p}_stats")
done
