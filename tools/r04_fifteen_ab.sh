#!/bin/bash
# round 4: fourteen against fifteen wavefronts per CU on the 1M/50M graph (both leave 256 on-chip values beside 8 KB of touched-bits),
# alternating processes, whole launches.
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1
for i in 1 2 3; do for w in 14 15; do
  ARCTE_HIP_WAVES_PER_CU=$w timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $O/r04_fifteen_${w}_$i.json 2>/dev/null || { echo "failed $w"; continue; }
  python - "$O/r04_fifteen_${w}_$i.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c, r = d["config"], d["roofline"]
best = max(c["slot_memory_probe_gups"])
print("waves/CU %d K %d slots %d  kernel ms %.1f  frac %.4f  draw %s  ms x level %.0f" % (c["waves_per_cu"], c["hot_values_per_wave"], c["slots_per_gpu"],
      r["kernel_ms_per_launch"], r["frac"], c["slot_memory_probe_gups"], r["kernel_ms_per_launch"] * c["slot_memory_probe_gups"][c["slot_memory_kept"]]))
PY
done; done
