#!/bin/bash
# SQ counter passes of one BASELINE shape's replay kernel (scripts/prof_shapes.py), e.g. to compare kernels under GC_NO_MULTI:
#   usage (on the GPU box): scripts/prof_kernel.sh <tag> <shape> [seconds]     -> gpurun_out/<tag>/<shape>_{stats,sq,lds}.txt
set -u
TAG=$1; SHAPE=$2; SEC=${3:-20}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${SHAPE}_stats" -- python /root/repo/scripts/prof_shapes.py $SHAPE $SEC 6 > "$OUT/${SHAPE}.txt" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/${SHAPE}_sq" -- python /root/repo/scripts/prof_shapes.py $SHAPE $SEC 4 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/${SHAPE}_lds" -- python /root/repo/scripts/prof_shapes.py $SHAPE $SEC 4 > /dev/null 2>&1
cd /root/repo
for n in stats sq lds; do python scripts/prof_summarize.py "$OUT/${SHAPE}_$n" "$OUT/${SHAPE}_$n.txt" > /dev/null; rm -rf "$OUT/${SHAPE}_$n"; done
