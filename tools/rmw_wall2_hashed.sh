#!/bin/bash
# What would a compact (hashed) cold state buy?  One private table per wavefront, 4 wavefronts per CU, the row stream
# beside it; the dense state of today = a 32 MB slice per wavefront (32 GB in all), a hashed one 0.5-4 MB.
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
B=$R/tools/rmw_wall2
OUT=$R/gpurun_out/rmw_wall2_hashed.txt
: > $OUT
for tab in 32768 8192 4096 2048 1024 512 256; do
  for spec in "32 2" "64 2" "128 2"; do
    timeout -k 5 60 $B $spec $tab 4 1 16 >> $OUT || exit 1
  done
done
for tab in 32768 2048 1024 512; do
  timeout -k 5 60 $B 32 2 $tab 8 1 16 >> $OUT || exit 1
  timeout -k 5 60 $B 64 2 $tab 8 1 16 >> $OUT || exit 1
done
cat $OUT
