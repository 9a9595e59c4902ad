"""Is the speed of a context decided per process or per allocation?  One process, several contexts in a row on the same
graph: with the library's buffer cache kept (same slot buffers every time) and with trim() in between (fresh allocations).

usage: python tools/context_lottery.py NODES EDGES STRIDE ROUNDS
"""
import os
import sys

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hot_sweep import load_graph
from reveal_graph_embedding_amd import _native


def one(A, seeds):
    with _native.Context.from_adjacency(A.indptr, A.indices, A.data) as ctx:
        if seeds is None:
            seeds = ctx.seed_list()
        best = 1e30
        for _ in range(3):
            ctx.run_seeds(seeds, 0.1, 1e-5)
            best = min(best, ctx.timing()["push_ms"])
    return best, seeds


def main():
    n, m, stride, rounds = (int(x) for x in sys.argv[1:5])
    A = load_graph(n, m)
    _, seeds = one(A, None)
    seeds = seeds[::stride]
    for mode in ("cache kept", "trim() before every context", "cache kept", "trim() before every context"):
        out = []
        for _ in range(rounds):
            if mode.startswith("trim"):
                _native.trim()
            out.append(one(A, seeds)[0])
        print("%-30s %s" % (mode, "  ".join("%.1f" % x for x in out)), flush=True)


if __name__ == "__main__":
    main()
