#!/bin/bash
# kernel statistics + SQ / LDS counters of the default GPS L1 C/A search (scripts/acq_packages.py --only GPS_L1CA): gpurun_out/<tag>/acq_l1ca_{stats,sq,lds}.txt
TAG=$1; PKG=${2:-GPS_L1CA}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/acq_${PKG}_stats" -- python /root/repo/scripts/acq_packages.py --only $PKG > "$OUT/acq_${PKG}.json" 2>/dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/acq_${PKG}_sq" -- python /root/repo/scripts/acq_packages.py --only $PKG > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/acq_${PKG}_lds" -- python /root/repo/scripts/acq_packages.py --only $PKG > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/acq_${PKG}_$c" -- python /root/repo/scripts/acq_packages.py --only $PKG > /dev/null 2>&1
done
for n in stats sq lds FETCH_SIZE WRITE_SIZE; do python /root/repo/scripts/prof_summarize.py "$OUT/acq_${PKG}_$n" "$OUT/acq_${PKG}_$n.txt" > /dev/null; rm -rf "$OUT/acq_${PKG}_$n"; done
