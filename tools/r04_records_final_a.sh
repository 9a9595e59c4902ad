#!/bin/bash
# Round-4 records, part A (run on the GPU box from the repo root): rocprofv3 passes over the default bench, the default
# bench itself, the spread over processes, the two-rank rehearsal (gloo transport: one GPU), the multi-worker arcte().
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1     # graph cache
python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench default $?"
for i in 1 2 3 4 5; do python bench.py --steps 2 --cpu-seconds 0 2>/dev/null > $O/bench_process_$i.json; done; echo "process spread done"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --cpu-seconds 0 --verify > $O/bench_two_ranks_gloo_one_gpu.json 2> $O/bench_two_ranks_gloo_one_gpu.log; echo "two ranks $?"
python tools/multi_worker_time.py 100000 2000000 3 > $O/multi_worker_config1.txt 2>&1; echo "multi worker $?"
python tools/multi_worker_time.py 1000000 50000000 3 > $O/multi_worker_1m.txt 2>&1; echo "multi worker 1M $?"
bash tools/rocprof_passes.sh > $O/rocprof_passes.txt 2>&1; echo "rocprof $?"
