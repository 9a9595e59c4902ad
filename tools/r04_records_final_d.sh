#!/bin/bash
# Round-4 records, part D: the default bench (with the PMC traffic stamped for this source), the process spread, and the first calls
# with and without the placement draw.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1     # graph cache
python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench default $?"
for i in 1 2 3 4 5; do python bench.py --steps 2 --cpu-seconds 0 2>/dev/null > $O/bench_process_$i.json; done; echo "process spread done"
FIRST_CALL_CALLS=3 ARCTE_HIP_SPREAD_TRIES=1 ARCTE_HIP_VERBOSE=1 python tools/first_call_time.py 1000000 50000000 > $O/first_call_1m_no_draw.txt 2>&1; echo "first call, no draw $?"
ARCTE_HIP_SPREAD_TRIES=1 python tools/e2e_time.py 1000000 50000000 > $O/e2e_arcte_1m_no_draw.txt 2>&1; echo "e2e, no draw $?"
python bench.py --gpus 1 --shards 8 --sub-launches 4 --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_one_rank_of_8_four_sub_launches.json 2>/dev/null; echo "rank of 8 $?"
python bench.py --gpus 1 --shards 8 --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_one_rank_of_8_one_launch.json 2>/dev/null; echo "rank of 8, one launch $?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --cpu-seconds 0 --verify > $O/bench_two_ranks_gloo_one_gpu.json 2> $O/bench_two_ranks_gloo_one_gpu.log; echo "two ranks $?"
