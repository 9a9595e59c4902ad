#!/bin/bash
# Is the random-access wall a wall of READS, of WRITES or of their sum?  Private slices (the kernel's layout), the row
# stream beside them, DRAM-resident and near-cache footprints.  Writes gpurun_out/rmw_wall3.txt.
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
B=$R/tools/rmw_wall2
OUT=$R/gpurun_out/rmw_wall3.txt
: > $OUT
for wpc in 6 8; do
  for tab in 32768 2048; do
    for st in 16 0; do
      for em in "32 0" "32 1" "32 2" "32 3" "16 0" "16 1" "16 3" "64 0" "64 1" "64 2" "64 3"; do
        timeout -k 5 60 $B $em $tab $wpc 1 $st >> $OUT || exit 1
      done
    done
  done
done
cat $OUT
