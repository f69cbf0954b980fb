OUT=/root/repo/gpurun_out/r3s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in blocked natural; do
  if [ $v = natural ]; then export GC_ACQ_NATURAL_ORDER=1; else unset GC_ACQ_NATURAL_ORDER; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/acq_$v" -- python /root/repo/scripts/acq_time.py > "$OUT/acq_$v.txt" 2>&1
  f=$(find $OUT/acq_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; grep "best of" $OUT/acq_$v.txt; head -4 "$f" | cut -c1-60,130-220
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/acq_pmc_$v" -- python /root/repo/scripts/acq_time.py > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
for f in glob.glob('$OUT/acq_pmc_$v/**/*counter_collection.csv',recursive=True):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'][:75]].append(float(r['Counter_Value']))
    for k,v in d.items():
        if 'fft_pass' in k and len(v)>100 or 'combine' in k: print(k, len(v), sum(v)/len(v))
PY
done
unset GC_ACQ_NATURAL_ORDER; cd /root/repo; python -m pytest tests -x -q -m gpu -k "acq or acquisition" 2>&1 | tail -3
