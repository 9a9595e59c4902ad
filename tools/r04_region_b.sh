#!/bin/bash
# round 4: region B of the line state on large graphs -- dense lines against indirect lines whose claim hands out the pool line.
#   tools/r04_region_b.sh NODES EDGES SHARDS   (SHARDS > 1: every SHARDS-th seed per launch)
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
N=$1; M=$2; S=${3:-1}
mkdir -p $R/gpurun_out
for arm in dense indirect; do
  if [ $arm = indirect ]; then export ARCTE_HIP_B_INDIRECT=1; else export ARCTE_HIP_B_INDIRECT=0; fi
  timeout -k 10 900 python $R/bench.py --nodes $N --edges $M --shards $S --steps 2 --warmup 1 --cpu-seconds 0 \
      > $R/gpurun_out/r04_regionb_n${N}_${arm}.json 2> $R/gpurun_out/r04_regionb_n${N}_${arm}.err || { tail -5 $R/gpurun_out/r04_regionb_n${N}_${arm}.err; exit 1; }
done
python - <<PY
import json
for arm in ("dense", "indirect"):
    d = json.load(open("$R/gpurun_out/r04_regionb_n${N}_%s.json" % arm))
    c, r = d["config"], d["roofline"]
    print("n=$N %-8s slots %d  indirect %d  seeds/s %.0f  kernel ms %.1f  frac %.4f  slot GB %.1f  in use GB %.1f  draw %s  updates/edge %s" % (
        arm, c["slots_per_gpu"], c["state"].get("region_b_indirect", -1), d["value"], r["kernel_ms_per_launch"], r["frac"], c["state"]["slot_bytes"] / 1e9,
        c["device_memory"]["in_use_bytes_after_create"] / 1e9, c["slot_memory_probe_gups"], {k: round(v, 3) for k, v in r["updates_per_edge"].items()}))
PY
