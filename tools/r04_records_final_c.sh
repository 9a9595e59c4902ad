#!/bin/bash
# Round-4 records, part C (after the CSR assembly's buffers went into the cache): first and later calls, end to end, the default
# bench under the profiler and alone.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
FIRST_CALL_CALLS=4 ARCTE_HIP_VERBOSE=1 python tools/first_call_time.py 1000000 50000000 > $O/first_call_1m.txt 2>&1; echo "first call $?"
ARCTE_HIP_VERBOSE=1 python tools/e2e_time.py 1000000 50000000 > $O/e2e_arcte_1m.txt 2>&1; echo "e2e $?"
python tools/e2e_time.py 100000 2000000 > $O/e2e_arcte_config1.txt 2>&1
python tools/multi_worker_time.py 1000000 50000000 3 > $O/multi_worker_1m.txt 2>&1; echo "multi worker 1M $?"
python tools/measure_centrality_weighting.py 1000000 50000000 > $O/centrality_weighting_1m.txt 2>&1; echo "f rows $?"
bash tools/rocprof_passes.sh > $O/rocprof_passes.txt 2>&1; echo "rocprof $?"
