#!/bin/bash
# round 4: do the probe levels of the slot-memory candidates follow the ADDRESS the allocation got (its alignment), or the order
# in which the candidates were allocated?  Four candidates forced per process, six processes.
R=${GRAFT_REPO_ROOT:-.}
python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > /dev/null 2>&1
for i in 1 2 3 4 5 6; do
  ARCTE_HIP_VERBOSE=1 ARCTE_HIP_SPREAD_TRIES=4 ARCTE_HIP_DRAW_GOOD_X10=999 ARCTE_HIP_DRAW_OK_X10=999 ARCTE_HIP_DRAW_ALLOC_MS=100000 \
    timeout -k 10 300 python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 2>&1 >/dev/null | grep "slot memory" 
  echo "--"
done
