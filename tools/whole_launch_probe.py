"""(round 4) Why is a WHOLE launch of a large graph slower than its parts?  One context, the seed list run whole and in interleaved
parts, with the library's own report of slots, reruns and timings after every run.

usage: python tools/whole_launch_probe.py NODES EDGES
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reveal_graph_embedding_amd import _native
from hot_sweep import load_graph


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    adj = load_graph(n, m)
    ctx = _native.Context.from_adjacency(adj.indptr, adj.indices, adj.data)
    seeds = ctx.seed_list()
    print("seeds", seeds.size, "info", ctx.info(), "placement", ctx.placement_info(), flush=True)
    plan = (("whole", [seeds]), ("halves", [seeds[0::2], seeds[1::2]]), ("quarters", [seeds[k::4] for k in range(4)]), ("whole", [seeds]))
    if os.environ.get("PROBE_SHORT"):
        plan = (("whole", [seeds]), ("whole", [seeds]), ("quarters", [seeds[k::4] for k in range(4)]))
    for what, parts in plan:
        total = 0.0
        for p in parts:
            t = time.time()
            ctx.run_seeds(p, 0.1, 1e-5)
            tm = ctx.timing()
            st = ctx.stats()
            total += tm["push_ms"]
            print("  %-8s part of %8d seeds: push %.1f ms (call %.1f), launches %d reruns %d, slots now %d, state %s" % (
                what, p.size, tm["push_ms"], (time.time() - t) * 1e3, st["launches"], st["reruns"], ctx.info()["slots"],
                {k: ctx.state_info()[k] for k in ("pushed_capacity", "candidate_capacity", "slot_bytes", "region_b_pool_lines")}), flush=True)
        print("%-8s: %.1f ms of push kernel in all" % (what, total), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
