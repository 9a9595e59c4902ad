#!/bin/bash
# Round-4 records, part F: the default bench and the process spread on the final source id.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1     # graph cache
python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench default $?"
for i in 1 2 3 4 5; do python bench.py --steps 2 --cpu-seconds 0 2>/dev/null > $O/bench_process_$i.json; done; echo "process spread done"
