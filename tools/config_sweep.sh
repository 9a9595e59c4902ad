#!/bin/bash
# A/B of launch configurations of the propagation kernel on the bench graph, one eighth of the seeds per launch, interleaved
# repetitions (the spread between processes is that of the placement draw: see the probe rates).
#   tools/config_sweep.sh REPS "tiles waves lines_lds reserve_kb" ...        (extra bench.py arguments in $BENCH_ARGS)
reps=$1; shift
cfgs=("$@")
for rep in $(seq 1 $reps); do
    for cfg in "${cfgs[@]}"; do
        read -r tiles waves lines reserve <<< "$cfg"
        ARCTE_HIP_TILES=$tiles ARCTE_HIP_WAVES_PER_CU=$waves ARCTE_HIP_LINES_LDS=$lines ARCTE_HIP_LDS_RESERVE_KB=$reserve python bench.py --shards 8 --steps 3 --cpu-seconds 0 $BENCH_ARGS 2>/dev/null |
            python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tiles/waves/lines_lds/reserve $cfg:', d['config']['hot_values_per_wave'], d['config']['slots_per_gpu'], round(d['roofline']['kernel_ms_per_launch'],1), round(d['roofline']['frac'],4), d['config']['slot_memory_probe_gups'])"
    done
done
