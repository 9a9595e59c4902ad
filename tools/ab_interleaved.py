"""Interleaved A/B of launch shapes of the push kernel: the configurations are run round-robin ROUNDS times (a new
context each time), so that slow drifts of a box (clock, neighbours) hit every configuration alike; prints mean, min
and spread per configuration.

usage: python tools/ab_interleaved.py NODES EDGES STRIDE ROUNDS CONFIG [CONFIG ...]
  CONFIG = name=VAR:val,VAR:val,...   (environment variables without the ARCTE_HIP_ prefix), e.g.
           w4warm=WAVES_PER_CU:4  w4cold=WAVES_PER_CU:4,WARM:0
"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hot_sweep import load_graph, result_hash
from reveal_graph_embedding_amd import _native

KNOBS = ("WAVES_PER_CU", "HOT", "WARM", "NARROW", "TILES", "LDS_RESERVE_KB", "WAVES_PER_BLOCK", "COOP", "COOP_MIN")


def main():
    n, m, stride, rounds = (int(x) for x in sys.argv[1:5])
    configs = []
    for spec in sys.argv[5:]:
        name, _, body = spec.partition("=")
        configs.append((name, dict(kv.split(":") for kv in body.split(",") if kv)))
    A = load_graph(n, m)
    with _native.Context.from_adjacency(A.indptr, A.indices, A.data, n_slots=256) as ctx:
        seeds = ctx.seed_list()[::stride]
    print("graph n=%d nnz=%d seeds=%d, %d rounds" % (n, A.nnz, seeds.size, rounds), flush=True)
    times = {name: [] for name, _ in configs}
    ref = None
    for rnd in range(rounds):
        for name, env in configs:
            for k in KNOBS:
                os.environ.pop("ARCTE_HIP_" + k, None)
            for k, v in env.items():
                os.environ["ARCTE_HIP_" + k] = v
            with _native.Context.from_adjacency(A.indptr, A.indices, A.data) as ctx:
                best = 1e30
                for _ in range(3):
                    ctx.run_seeds(seeds, 0.1, 1e-5)
                    best = min(best, ctx.timing()["push_ms"])
                if rnd == 0:
                    h = result_hash(*ctx.fetch())
                    ref = ref or h
                    assert h == ref, "results differ between configurations"
            times[name].append(best)
        print("round %d: %s" % (rnd, "  ".join("%s %.1f" % (name, times[name][-1]) for name, _ in configs)), flush=True)
    print()
    for name, _ in configs:
        t = np.array(times[name])
        print("%-28s mean %.1f  median %.1f  min %.1f  max %.1f ms  (%d runs)" % (name, t.mean(), np.median(t), t.min(), t.max(), t.size))


if __name__ == "__main__":
    main()
