#!/bin/bash
# round 4: the stride of the spread slot layout (ARCTE_HIP_SLOT_SPREAD_MB: one slot every so many MB of ONE allocation) against the
# probe level and the kernel's duration, fresh processes interleaved on one box, one candidate each (no draw).
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1
for rep in 1 2 3 4; do
  for mb in 16 24 32 48; do
    ARCTE_HIP_SPREAD_TRIES=1 ARCTE_HIP_SLOT_SPREAD_MB=$mb timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $O/r04_spread_${mb}_$rep.json 2>/dev/null || exit 1
  done
done
python - <<PY
import json, glob, os
for mb in (16, 24, 32, 48):
    rows = []
    for f in sorted(glob.glob("$O/r04_spread_%d_*.json" % mb)):
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        rows.append((d["config"]["slot_memory_probe_gups"][0], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"], d["config"]["state"]["slot_bytes"] / 1e9))
    print("spread %2d MB (%.1f GB): " % (mb, rows[0][3]) + "  ".join("probe %.2f -> %.1f ms (%.4f)" % r[:3] for r in rows))
PY
