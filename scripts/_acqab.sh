OUT=gpurun_out/r3v
mkdir -p $OUT
for rep in 1 2; do
for v in OLD NEW; do
  for shape in cboc l5; do
    GC_LIB_PATH=$PWD/cu-sdr-collection_amd/lib/libgnsscorr_$v.so python scripts/prof_shapes.py $shape 10 8 2>/dev/null | tail -1 | cut -c1-200
  done
done
done
GC_LIB_PATH=$PWD/cu-sdr-collection_amd/lib/libgnsscorr_NEW.so python -m pytest tests -x -q -m gpu -k "not bench_ranks and not smoke and not build" 2>&1 | tail -4
