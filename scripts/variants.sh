#!/bin/bash
# Tuning variants of libgnsscorr.so that differ in ONE translation unit's macros:
#   scripts/variants.sh corr_fast "PFX:-DGC_FAST_PREFIX=1" ...   -> cu-sdr-collection_amd/lib/libgnsscorr_PFX.so
set -e
cd "$(dirname "$0")/../cu-sdr-collection_amd"
unit="$1"; shift
CONTRACT="-ffp-contract=off"; case "$unit" in acq_*) CONTRACT="";; esac  # build.py NO_CONTRACT: the acq_*.hip units are compiled with the default contraction
(cd .. && python -m cu_sdr_collection_amd.build >/dev/null)
objs=""
# (the objects of the tuning build: a variant library reads the GC_* switches too)
for o in gnsscorr corr_kernel corr_fast corr_multi corr_cboc corr_lane_GC_LANE_PART_0 corr_lane_GC_LANE_PART_1 corr_lane_GC_LANE_PART_2 corr_lane_GC_LANE_PART_3 track multi stream acq_fft acq_coarse acq_shift acq_fine acq_cond acq_guard navsync; do case "$o" in ${unit}|${unit}_GC_*) ;; *) objs="$objs build/${o}_GC_TUNING_1.o";; esac; done
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -DGC_TUNING=1 $CONTRACT $flags -c csrc/$unit.hip -o build/${unit}_$name.o &
done
wait
for spec in "$@"; do
  name="${spec%%:*}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $objs build/${unit}_$name.o -o lib/libgnsscorr_$name.so
  echo built lib/libgnsscorr_$name.so
done
