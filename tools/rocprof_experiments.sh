#!/bin/bash
# Per-experiment evidence for the push kernel (run on the GPU box): for each launch shape, the kernel's duration
# (rocprofv3 --kernel-trace --stats) and, in separate --pmc passes, the memory-side request counters.
# Workload: bench.py --shards 8 (seed shard 0 of 8 of the 1M/50M graph, 81 434 seeds per launch).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_exp
rm -rf $OUT && mkdir -p $OUT
ARGS="$R/bench.py --shards 8 --steps 2 --warmup 1 --cpu-seconds 0 --placement-tries 1"
python3 $R/bench.py --shards 8 --steps 1 --warmup 0 --cpu-seconds 0 --placement-tries 1 > /dev/null 2>&1   # graph cache
arm() {  # name, env assignments...
  name=$1; shift
  for kv in "$@"; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- python3 $ARGS > $OUT/$name.bench.json 2> $OUT/$name.trace.err
  for pass in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_LDS"; do
    p=$(echo $pass | tr ' ' '_' | cut -c1-24)
    rocprofv3 --pmc $pass --output-format csv -d $OUT/$name/pmc_$p -- python3 $ARGS > /dev/null 2> $OUT/$name.pmc_$p.err
  done
  for kv in "$@"; do unset "${kv%%=*}"; done
  echo "arm $name done"
}
if [ "$ROUND" = "r02" ]; then
arm a_r1_shape_tables_off ARCTE_HIP_STATE=dense ARCTE_HIP_HOT=0 ARCTE_HIP_WAVES_PER_CU=8 ARCTE_HIP_NARROW=0
arm b_lds_table_w8 ARCTE_HIP_STATE=dense ARCTE_HIP_WAVES_PER_CU=8 ARCTE_HIP_WARM=0 ARCTE_HIP_NARROW=0
arm c_lds_table_w4 ARCTE_HIP_STATE=dense ARCTE_HIP_WAVES_PER_CU=4 ARCTE_HIP_WARM=0 ARCTE_HIP_NARROW=0
arm d_lds_table_w4_narrow_rows ARCTE_HIP_STATE=dense ARCTE_HIP_WAVES_PER_CU=4 ARCTE_HIP_WARM=0
arm e_lds_warm_narrow_w4 ARCTE_HIP_STATE=dense ARCTE_HIP_WAVES_PER_CU=4
arm f_default_lds_warm_narrow_w6 ARCTE_HIP_STATE=dense
arm g_lds_warm_narrow_w8 ARCTE_HIP_STATE=dense ARCTE_HIP_WAVES_PER_CU=8
else
# round 3: the dense state of round 2 -> the line state, step by step (each arm = the previous + one thing)
arm a_dense_state_r02_default ARCTE_HIP_STATE=dense
arm b_lines_two_tiles_w8_bitmap_16k ARCTE_HIP_TILES=2 ARCTE_HIP_WAVES_PER_CU=8 ARCTE_HIP_LINES_LDS=131072
arm c_lines_two_tiles_w8_bitmap_4k ARCTE_HIP_TILES=2 ARCTE_HIP_WAVES_PER_CU=8
arm d_lines_one_tile_w8 ARCTE_HIP_WAVES_PER_CU=8
arm e_lines_one_tile_w10 ARCTE_HIP_WAVES_PER_CU=10
arm f_default_lines_one_tile_w12
arm g_lines_one_tile_w16_spilling ARCTE_HIP_WAVES_PER_CU=16
fi
python3 $R/tools/summarise_experiments.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
