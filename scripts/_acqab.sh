OUT=gpurun_out/r3t
mkdir -p $OUT
python -m pytest tests -x -q -m gpu -k "closed_loop or tracking or track or persistent or mix" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for t in 1 2 3 4; do
  GC_TRACK_THREADS=$t python bench.py --no-cpu > $OUT/b_$t.json 2> $OUT/b_$t.err
  python - <<PY
import json
d=json.loads(open('$OUT/b_$t.json').read().strip().splitlines()[-1])
c=d['configs']
print('threads $t: l1ca', d['closed_loop']['us_per_epoch'], 'dev', d['closed_loop_device']['us_per_epoch'],
      {k:(v.get('closed_loop_host') or {}).get('us_per_epoch') for k,v in c.items() if isinstance(v,dict)})
PY
done
