#!/bin/bash
# round 4: interleaved A/B of the packed (4 bytes per traversed edge) against the narrow (8 bytes) row stream, whole bench launches
# on one box.  Writes gpurun_out/r04_ab_pack_{packed,narrow}_N.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
for i in 1 2; do
  ARCTE_HIP_PACK=1 timeout -k 10 300 python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/r04_ab_pack_packed_$i.json 2> $R/gpurun_out/r04_ab_pack_packed_$i.err || exit 1
  ARCTE_HIP_PACK=0 timeout -k 10 300 python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/r04_ab_pack_narrow_$i.json 2> $R/gpurun_out/r04_ab_pack_narrow_$i.err || exit 1
done
python - <<'PY'
import json, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
for f in sorted(glob.glob(R + "/gpurun_out/r04_ab_pack_*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), "rows", d["config"]["narrow_rows"], "kernel ms", round(d["roofline"]["kernel_ms_per_launch"], 1), "frac", round(d["roofline"]["frac"], 4),
          "draw", d["config"]["slot_memory_probe_gups"], "kept", d["config"]["slot_memory_kept"], "in use GB", round(d["config"]["device_memory"]["in_use_bytes_after_create"] / 1e9, 1))
PY
