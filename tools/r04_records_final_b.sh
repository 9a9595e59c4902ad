#!/bin/bash
# Round-4 records, part B: push flavours, larger graphs, end-to-end and console-script timings, f-rows at size, phases.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.log; echo "bench default $?"
python bench.py --variant pagerank --steps 3 --cpu-seconds 0 > $O/bench_variant_pagerank.json 2>/dev/null; echo "pagerank $?"
python bench.py --variant lazy --steps 3 --cpu-seconds 0 > $O/bench_variant_lazy_pagerank.json 2>/dev/null; echo "lazy $?"
python bench.py --nodes 4000000 --edges 100000000 --steps 2 --cpu-seconds 0 > $O/bench_n4M_m100M.json 2>/dev/null; echo "4M $?"
python bench.py --nodes 8000000 --edges 100000000 --steps 2 --cpu-seconds 0 > $O/bench_n8M_m100M.json 2>/dev/null; echo "8M $?"
python bench.py --nodes 16000000 --edges 200000000 --shards 16 --steps 2 --cpu-seconds 0 > $O/bench_n16M_m200M_every_16th_seed.json 2>/dev/null; echo "16M $?"
python tools/e2e_time.py 1000000 50000000 > $O/e2e_arcte_1m.txt 2>&1; echo "e2e $?"
python tools/e2e_time.py 100000 2000000 > $O/e2e_arcte_config1.txt 2>&1
python tools/cli_time.py 100000 2000000 3 > $O/cli_time_config1.txt 2>&1; echo "cli $?"
python tools/measure_centrality_weighting.py 1000000 50000000 > $O/centrality_weighting_1m.txt 2>&1; echo "f rows $?"
ARCTE_HIP_VERBOSE=1 python tools/first_call_time.py 1000000 50000000 > $O/first_call_1m.txt 2>&1; echo "first call $?"
python bench.py --gpus 1 --shards 8 --sub-launches 4 --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_one_rank_of_8_four_sub_launches.json 2>/dev/null; echo "rank of 8 $?"
