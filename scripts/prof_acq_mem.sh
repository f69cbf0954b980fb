#!/bin/bash
# Memory-path counters of the default GPS L1 C/A search (scripts/acq_time.py), two counters per pass (a TCP or TCC block takes no more in
# one pass on gfx950: "Request exceeds the capabilities of the hardware to collect"): gpurun_out/<tag>/acq_mem_<group>.txt
TAG=$1
OUT=/root/repo/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/acq_mem_$n" -- python /root/repo/scripts/acq_time.py > /dev/null 2>&1
        python /root/repo/scripts/prof_summarize.py "$OUT/acq_mem_$n" "$OUT/acq_mem_$n.txt" > /dev/null 2>&1; rm -rf "$OUT/acq_mem_$n"; }
run read_latency TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run write_latency TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
run tcp_pending TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run l2_hit TCC_HIT_sum TCC_MISS_sum
run l2_write_stall TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
run l2_read_stall TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
run l2_requests TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
