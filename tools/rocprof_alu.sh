#!/bin/bash
# round 3: is the propagation kernel bound by instruction issue or by memory?  PMC passes (never combined with tracing)
# over bench.py --shards 8 for the line-state kernel and the dense-state kernel.  Run on the GPU box.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_alu
rm -rf $OUT && mkdir -p $OUT
ARGS="$R/bench.py --shards 8 --steps 2 --warmup 1 --cpu-seconds 0"
python3 $R/bench.py --shards 8 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1   # graph cache
arm() {
  name=$1; shift
  for kv in "$@"; do export "$kv"; done
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
    p=$(echo $pass | tr ' ' '_' | cut -c1-28)
    rocprofv3 --pmc $pass --output-format csv -d $OUT/$name/pmc_$p -- python3 $ARGS > $OUT/$name.$p.json 2> $OUT/$name.pmc_$p.err
  done
  for kv in "$@"; do unset "${kv%%=*}"; done
  echo "arm $name done"
}
arm lines_w8 ARCTE_HIP_WAVES_PER_CU=8
arm dense_w6 ARCTE_HIP_STATE=dense
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_alu"
for arm in sorted(os.listdir(out)):
    if not os.path.isdir(os.path.join(out, arm)): continue
    tot = collections.defaultdict(float); calls = collections.defaultdict(int)
    for f in glob.glob(os.path.join(out, arm, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_arcte_lines" not in k and "k_arcte_seeds" not in k: continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); calls[row["Counter_Name"]] += 1
    print(arm)
    for c in sorted(tot): print("   %-28s %.4g per launch (%d dispatch rows)" % (c, tot[c] / 3.0, calls[c]))
PY
