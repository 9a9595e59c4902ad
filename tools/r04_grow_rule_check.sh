#!/bin/bash
# round 4: after grow_lines learnt to give up slots half a wavefront per CU at a time (and pools to double from 8 MB on): the 16M graph
# at 12 / 14 / 16 wavefronts per CU (launches of a sixteenth of the seeds) and the 8M graph's WHOLE launch on the defaults.
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
mkdir -p $O
show() {
python - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c, r = d["config"], d["roofline"]
print("%s: waves/CU %d K %d lines %d slots %d indirect %d pool %d  kernel ms %.1f  frac %.4f  draw %s  in use after create GB %.1f" % (
    sys.argv[2], c["waves_per_cu"], c["hot_values_per_wave"], c["state"]["bitmap_lds_bytes"] * 8, c["slots_per_gpu"],
    c["state"]["region_b_indirect"], c["state"].get("region_b_pool_lines", -1), r["kernel_ms_per_launch"], r["frac"], c["slot_memory_probe_gups"],
    c["device_memory"]["in_use_bytes_after_create"] / 1e9))
PY
}
python $R/bench.py --nodes 16000000 --edges 200000000 --shards 64 --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1
for cfg in "16 32768" "12 65536" "14 65536"; do
  set -- $cfg
  ARCTE_HIP_WAVES_PER_CU=$1 ARCTE_HIP_LINES_LDS=$2 timeout -k 10 500 python $R/bench.py --nodes 16000000 --edges 200000000 --shards 16 --steps 2 --warmup 1 --cpu-seconds 0 \
      > $O/r04_grow_16m_$1.json 2>$O/r04_grow_16m_$1.err || { echo "failed $cfg"; continue; }
  show $O/r04_grow_16m_$1.json "16M/200M sixteenth ($cfg)"
done
timeout -k 10 500 python $R/bench.py --nodes 8000000 --edges 100000000 --steps 2 --warmup 1 --cpu-seconds 0 > $O/r04_grow_8m_whole.json 2>$O/r04_grow_8m_whole.err && show $O/r04_grow_8m_whole.json "8M/100M whole launch, defaults"
