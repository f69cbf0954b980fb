#!/bin/bash
# closed-loop (host loop) time per epoch against the number of workgroups per block; GC_TRACK_PERSIST=0 for launch-per-epoch
for sp in ${SPLITS:-8 16 24 32}; do
  GC_TRACK_TIMING=1 GC_TRACK_SPLITS=$sp timeout 200 python bench.py --no-cpu --steps 2 --warmup 1 --seconds 20 2>&1 | grep "^gc_track:"
done
