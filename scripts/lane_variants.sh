#!/bin/bash
# Builds tuning variants of libgnsscorr.so that differ only in corr_lane.hip macros:  scripts/lane_variants.sh "GRP8:-DGC_LANE_GRP=8" ...
set -e
cd "$(dirname "$0")/../cu-sdr-collection_amd"
python -m cu_sdr_collection_amd.build >/dev/null 2>&1 || (cd .. && python -m cu_sdr_collection_amd.build >/dev/null)
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -ffp-contract=off $flags -c csrc/corr_lane.hip -o build/corr_lane_$name.o &
done
wait
for spec in "$@"; do
  name="${spec%%:*}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared build/gnsscorr.o build/corr_kernel.o build/corr_fast.o build/track.o build/acq.o build/corr_lane_$name.o -o lib/libgnsscorr_$name.so
  echo built lib/libgnsscorr_$name.so
done
