#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command and the PMC
# passes the roofline / traffic figures of DESIGN.md come from.  Output under gpurun_out/<tag>/; condensed into
# profiles/<round>/ afterwards with scripts/prof_summarize.py (see profiles/r02/README.md).
#   usage: scripts/collect_profiles.sh <tag> [quick]
set -u
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p "$OUT"
cd /root/repo
export TMPDIR=/tmp
sha256sum cu-sdr-collection_amd/lib/libgnsscorr.so | cut -d" " -f1 > "$OUT/lib_sha256.txt"   # ties every counter pass to the build it ran on (bench.py drops traffic of another build)
python bench.py --detail "$OUT/bench.json" > "$OUT/bench_line.json" 2> "$OUT/bench.err"   # the printed line (what the driver parses) and the long form
cd /tmp
# per-kernel durations of the same command (no CPU leg: it adds nothing on the device)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python /root/repo/bench.py --no-cpu --no-side-by-side --no-sweep --detail "$OUT/bench_under_profiler.json" > /dev/null 2>&1
# HBM traffic of the main replay kernel: separate passes (TCC slots), main line only
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/l1ca_pmc_$c" -- python /root/repo/bench.py --config l1ca --no-cpu --steps 4 --warmup 1 > /dev/null 2>&1
done
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/l1ca_pmc_sq" -- python /root/repo/bench.py --config l1ca --no-cpu --steps 4 --warmup 1 > /dev/null 2>&1
if [ "${2:-}" != "quick" ]; then
  # the other BASELINE shapes (scripts/prof_shapes.py: one replay kernel per shape)
  for shape in l5 b2a cboc e1x8 e1 b1c b1i l1ca3; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${shape}_stats" -- python /root/repo/scripts/prof_shapes.py $shape 5 6 > "$OUT/${shape}.txt" 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${shape}_pmc_FETCH_SIZE" -- python /root/repo/scripts/prof_shapes.py $shape 5 4 > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/${shape}_pmc_sq" -- python /root/repo/scripts/prof_shapes.py $shape 5 4 > /dev/null 2>&1
  done
  # acquisition
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/acq_stats" -- python /root/repo/scripts/acq_time.py > "$OUT/acq.txt" 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/acq_pmc_sq" -- python /root/repo/scripts/acq_time.py > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/acq_pmc_FETCH_SIZE" -- python /root/repo/scripts/acq_time.py > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/acq_pmc_WRITE_SIZE" -- python /root/repo/scripts/acq_time.py > /dev/null 2>&1
  # the twelve default-size searches (bench.py acquisition.packages) under the kernel trace
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/acqpkg_stats" -- python /root/repo/scripts/acq_packages.py > "$OUT/acq_packages.json" 2>/dev/null
  # counter passes (FETCH / WRITE / SQ / LDS) of the default L1 C/A search and of the two plans whose searches take 60 ms (GPS L2C 320 x 1000,
  # BDS B1C 600 x 600)
  for pkg in GPS_L1CA GPS_L2C BDS_B1C; do bash /root/repo/scripts/prof_acq_pmc.sh "$TAG" $pkg; done
fi
cd /root/repo
for d in "$OUT"/*/; do
  n=$(basename "$d")
  python scripts/prof_summarize.py "$d" "$OUT/summary/$n.txt"
  rm -rf "$d"   # the raw CSV traces are tens of MB; the summaries are what gets merged back
done
mv "$OUT/summary"/* "$OUT"/ 2>/dev/null; rmdir "$OUT/summary" 2>/dev/null
ls -la "$OUT"
