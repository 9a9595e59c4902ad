#!/bin/bash
# round 4, records on the final source (one GPU-box call): the N = 8 rank's step without transport, arcte() end to end, the first
# call of a process, and the placement draw's two policies over six fresh processes each.
set -o pipefail
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1      # (graph cache)
timeout -k 10 300 python $R/bench.py --gpus 1 --shards 8 --sub-launches 4 --steps 5 --warmup 1 --cpu-seconds 0 > $O/r04_shard_of_8_sub4.json 2> $O/r04_shard_of_8_sub4.err || exit 1
timeout -k 10 300 python $R/bench.py --gpus 1 --shards 8 --sub-launches 1 --steps 5 --warmup 1 --cpu-seconds 0 > $O/r04_shard_of_8_sub1.json 2> $O/r04_shard_of_8_sub1.err || exit 1
timeout -k 10 400 python $R/tools/e2e_time.py 1000000 50000000 > $O/r04_e2e_arcte_1m.txt 2>&1 || exit 1
ARCTE_HIP_VERBOSE=1 timeout -k 10 300 python $R/tools/first_call_time.py 1000000 50000000 > $O/r04_first_call_1m.txt 2>&1 || exit 1
for i in 1 2 3 4 5 6; do
  timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $O/r04_lottery_new_$i.json 2> $O/r04_lottery_new_$i.err || exit 1
  ARCTE_HIP_PARK_MAX=8 ARCTE_HIP_DRAW_ALLOC_MS=100000 timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $O/r04_lottery_r03policy_$i.json 2> $O/r04_lottery_r03policy_$i.err || exit 1
done
python - <<PY
import json, glob, os
O = "$O"
for f in ("r04_shard_of_8_sub4.json", "r04_shard_of_8_sub1.json"):
    d = json.loads([l for l in open(os.path.join(O, f)) if l.startswith("{")][-1])
    k = d["roofline"]["kernel_ms_per_launch"]
    print(f, "ms_per_step %.2f  kernel ms per step %.2f  host share %.3f  seeds/step %d  sub-launches %d" % (
        d["ms_per_step"], k, 1 - k / d["ms_per_step"], d["config"]["seeds_per_step"], d["config"]["sub_launches"]))
for pol in ("new", "r03policy"):
    for f in sorted(glob.glob(os.path.join(O, "r04_lottery_%s_*.json" % pol))):
        d = json.load(open(f))
        print(os.path.basename(f), "frac %.4f  kernel ms %.1f  draw %s kept %d  in use GB %.1f" % (
            d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["config"]["slot_memory_probe_gups"], d["config"]["slot_memory_kept"],
            d["config"]["device_memory"]["in_use_bytes_after_create"] / 1e9))
PY
cat $O/r04_e2e_arcte_1m.txt $O/r04_first_call_1m.txt
