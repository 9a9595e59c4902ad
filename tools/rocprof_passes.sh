#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box): kernel trace + separate PMC passes (never combined).
# BENCH_ARGS narrows the workload (default: bench.py's own default = all seeds of the 1M/50M graph).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
ARGS="$R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --placement-tries 1 $BENCH_ARGS"
python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --placement-tries 1 $BENCH_ARGS > /dev/null 2>&1   # graph cache
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.json 2> $OUT/trace.err
echo "trace done $?"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name done $?"
done
find $OUT -name "*.csv" | head -50
