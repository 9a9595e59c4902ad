#!/bin/bash
# round 4: interleaved A/B at the bench workload (all seeds of the 1M/50M graph): twelve wavefronts per CU (156 VGPRs, 12 KB of LDS each)
# against fourteen on the 128-VGPR build (120 bytes of scratch per lane, 10 KB of LDS each: 8 KB of touched-bits + 256 on-chip values)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1
for rep in 1 2 3 4; do
  for w in 12 14; do
    ARCTE_HIP_WAVES_PER_CU=$w timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $O/r04_occab_${w}_$rep.json 2>/dev/null || exit 1
  done
done
python - <<PY
import json, glob
for w in (12, 14):
    for f in sorted(glob.glob("$O/r04_occab_%d_*.json" % w)):
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        c, r = d["config"], d["roofline"]
        kept = c["slot_memory_probe_gups"][c["slot_memory_kept"]]
        print("waves/CU %d: K %d  kernel ms %.1f  frac %.4f  kept level %.2f  ms x level %.0f  draw %s" % (w, c["hot_values_per_wave"], r["kernel_ms_per_launch"], r["frac"], kept, r["kernel_ms_per_launch"] * kept, c["slot_memory_probe_gups"]))
PY
