#!/bin/bash
# Sweep of tools/rmw_wall2 on the GPU box; writes gpurun_out/rmw_wall2.txt (+ PMC csv for a few points).
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
B=$R/tools/rmw_wall2
OUT=$R/gpurun_out/rmw_wall2.txt
: > $OUT
# (a) entry size x write shape, DRAM-resident (16 GB) and Infinity-Cache-sized (128 MB), shared table
for tab in 16384 128; do
  for entry in 32 64 128; do
    for mode in 0 1 2 3; do
      timeout -k 5 60 $B $entry $mode $tab 8 0 0 >> $OUT || exit 1
    done
  done
done
# (b) private slices per wavefront: how small must the live state be before the caches carry it?  with and
#     without the row stream beside it (20 B per edge in the real kernel)
for wpc in 1 2 4 8; do
  for tab in 64 128 192 256 384 512 1024 4096; do
    for st in 0 16; do
      timeout -k 5 60 $B 32 2 $tab $wpc 1 $st >> $OUT || exit 1
    done
  done
done
# (c) occupancy: the DRAM-resident RMW rate vs wavefronts per CU (latency-bound below how many?)
for wpc in 1 2 3 4 6 8 12 16; do
  timeout -k 5 60 $B 32 2 16384 $wpc 0 0 >> $OUT || exit 1
done
cat $OUT
