#!/bin/bash
# The hybrid CBOC kernel (corr_cboc.hip) on config 3's shape in the tuning build: time at several wave counts against the lane
# kernel (GC_NO_CBOC=1), then SQ / LDS counter passes of one wave count.   usage (GPU box): scripts/prof_cboc.sh <tag> [waves for the counters] [seconds]
set -u
TAG=$1; W=${2:-16}; SEC=${3:-20}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p "$OUT"
export GC_LIB_PATH=/root/repo/cu-sdr-collection_amd/lib/libgnsscorr_tuning.so
cd /root/repo
{ GC_NO_CBOC=1 python scripts/prof_shapes.py cboc $SEC 10; for w in 16 12 8; do GC_CBOC_WAVES=$w python scripts/prof_shapes.py cboc $SEC 10; done; } > "$OUT/cboc_ab.txt" 2>&1
cat "$OUT/cboc_ab.txt"
cd /tmp; export TMPDIR=/tmp
export GC_CBOC_WAVES=$W
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/cboc_hybrid_sq" -- python /root/repo/scripts/prof_shapes.py cboc $SEC 4 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/cboc_hybrid_lds" -- python /root/repo/scripts/prof_shapes.py cboc $SEC 4 > /dev/null 2>&1
cd /root/repo
for n in sq lds; do python scripts/prof_summarize.py "$OUT/cboc_hybrid_$n" "$OUT/cboc_hybrid_pmc_$n.txt" > /dev/null; rm -rf "$OUT/cboc_hybrid_$n"; done
grep -i "cboc" "$OUT"/cboc_hybrid_pmc_*.txt | head -20
