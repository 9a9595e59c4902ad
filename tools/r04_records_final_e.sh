#!/bin/bash
# Round-4 records, part E: push flavours and larger graphs on the final source, the two-rank rehearsal on one GPU.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --cpu-seconds 0 --verify > $O/bench_two_ranks_gloo_one_gpu.json 2> $O/bench_two_ranks_gloo_one_gpu.log; echo "two ranks $?"
python bench.py --variant pagerank --steps 3 --cpu-seconds 0 > $O/bench_variant_pagerank.json 2>/dev/null; echo "pagerank $?"
python bench.py --variant lazy --steps 3 --cpu-seconds 0 > $O/bench_variant_lazy_pagerank.json 2>/dev/null; echo "lazy $?"
python bench.py --nodes 4000000 --edges 100000000 --steps 2 --cpu-seconds 0 > $O/bench_n4M_m100M.json 2>/dev/null; echo "4M $?"
python bench.py --nodes 8000000 --edges 100000000 --steps 2 --cpu-seconds 0 > $O/bench_n8M_m100M.json 2>/dev/null; echo "8M $?"
python bench.py --nodes 16000000 --edges 200000000 --shards 16 --steps 2 --cpu-seconds 0 > $O/bench_n16M_m200M_every_16th_seed.json 2>/dev/null; echo "16M $?"
