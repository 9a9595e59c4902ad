#!/bin/bash
# round 4: re-touches of a state line as memory-side float64 atomics against load + store (tools/atomic_wall.hip)
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
OUT=$R/gpurun_out/r04_atomic_wall.txt
: > $OUT
for rep in 1 2; do
  for blind in 0 65; do
    for mode in 0 1 2; do
      timeout -k 5 60 $R/tools/atomic_wall $mode $blind 4 12 >> $OUT || exit 1
    done
  done
done
for mode in 0 1 2; do timeout -k 5 60 $R/tools/atomic_wall $mode 65 4 6 >> $OUT || exit 1; done
for mode in 0 1; do timeout -k 5 60 $R/tools/atomic_wall $mode 65 0.0625 12 >> $OUT || exit 1; done
cat $OUT
