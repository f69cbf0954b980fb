for rep in 1 2; do
for v in "" _NOPN; do
    GC_LIB_PATH=$PWD/cu-sdr-collection_amd/lib/libgnsscorr$v.so python scripts/prof_shapes.py cboc 10 8 2>/dev/null | tail -1 | cut -c1-120
    GC_LIB_PATH=$PWD/cu-sdr-collection_amd/lib/libgnsscorr$v.so python scripts/prof_shapes.py cboc 60 6 2>/dev/null | tail -1 | cut -c1-120
done
done
GC_LIB_PATH=$PWD/cu-sdr-collection_amd/lib/libgnsscorr_NOPN.so python -m pytest tests -x -q -m gpu -k "derived or cboc or CBOC or b1c or B1C or variants" 2>&1 | tail -3
