OUT=/root/repo/gpurun_out/r3s
mkdir -p $OUT
cd /root/repo; python -m pytest tests -x -q -m gpu -k "acq or acquisition" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/acq_hg" -- python /root/repo/scripts/acq_time.py > "$OUT/acq_hg.txt" 2>&1
f=$(find $OUT/acq_hg -name '*kernel_stats.csv' | head -1)
grep "best of" $OUT/acq_hg.txt; head -5 "$f" | cut -c1-60,130-220
