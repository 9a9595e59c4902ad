#!/bin/bash
# round 4: the split of a wavefront's LDS (touched-line bitmap against on-chip values) and the wavefronts per CU on a LARGE graph, where
# region B's claims are memory-side atomics: does a bitmap that covers more ranks pay for the wavefronts it costs?
#   tools/r04_large_graph_sweep.sh NODES EDGES SHARDS
R=${GRAFT_REPO_ROOT:-.}
N=$1; M=$2; S=${3:-1}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py --nodes $N --edges $M --shards 64 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1
for cfg in "12 65536" "12 32768" "9 131072" "8 131072" "12 65536"; do
  set -- $cfg
  ARCTE_HIP_WAVES_PER_CU=$1 ARCTE_HIP_LINES_LDS=$2 timeout -k 10 600 python $R/bench.py --nodes $N --edges $M --shards $S --steps 2 --warmup 1 --cpu-seconds 0 \
      > $O/r04_large_${N}_$1_$2.json 2>/dev/null || { echo "failed $cfg"; continue; }
  python - <<PY
import json
d = json.loads([l for l in open("$O/r04_large_${N}_$1_$2.json") if l.startswith("{")][-1])
c, r = d["config"], d["roofline"]
print("n=$N waves/CU $1 lines in LDS $2: K %d slots %d indirect %d  kernel ms %.1f  frac %.4f  draw %s  updates/edge %s" % (
    c["hot_values_per_wave"], c["slots_per_gpu"], c["state"]["region_b_indirect"], r["kernel_ms_per_launch"], r["frac"], c["slot_memory_probe_gups"],
    {k: round(v, 3) for k, v in r["updates_per_edge"].items()}))
PY
done
