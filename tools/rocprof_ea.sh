#!/bin/bash
# round 3: memory-side request sizes of the line-state kernel (bench.py --shards 8)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ea
rm -rf $OUT && mkdir -p $OUT
ARGS="$R/bench.py --shards 8 --steps 2 --warmup 1 --cpu-seconds 0"
python3 $R/bench.py --shards 8 --steps 1 --warmup 0 --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['per_seed'], d['roofline']['kernel_ms_per_launch'])"
for pass in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_WRITEBACK_sum"; do
  p=$(echo $pass | tr ' ' '_' | cut -c1-28)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$p -- python3 $ARGS > /dev/null 2> $OUT/pmc_$p.err
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_ea"
tot = collections.defaultdict(float)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_arcte_lines" in row["Kernel_Name"]: tot[row["Counter_Name"]] += float(row["Counter_Value"])
edges = 2.864e9
for c in sorted(tot): print("   %-36s %.4g per launch = %.3f per edge" % (c, tot[c] / 3.0, tot[c] / 3.0 / edges))
PY
