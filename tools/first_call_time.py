"""Where does the FIRST propagation of a process spend its time?  (The second one reuses the slot memory, the module, the pages
of the host arrays.)  Context creation, run, device assembly + fetch, twice, with the library's own report of its allocations
(ARCTE_HIP_VERBOSE=1).

usage: python tools/first_call_time.py NODES EDGES
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("ARCTE_HIP_VERBOSE", "1")
from reveal_graph_embedding_amd import _native
from hot_sweep import load_graph


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    adj = load_graph(n, m)
    for it in range(int(os.environ.get("FIRST_CALL_CALLS", "2"))):
        t0 = time.time()
        ctx = _native.Context.from_adjacency(adj.indptr, adj.indices, adj.data)
        t1 = time.time()
        seeds = np.sort(ctx.seed_list())
        ctx.run_seeds(seeds, 0.1, 1e-5)
        t2 = time.time()
        indptr, indices = ctx.fetch_csr(True)
        t3 = time.time()
        ctx.close()
        print("call %d: context %.3f s, run %.3f s (push kernel %.3f), assembly + fetch of %d ids %.3f s, close %.3f s" % (
            it, t1 - t0, t2 - t1, ctx_push(ctx), indices.size, t3 - t2, time.time() - t3), flush=True)


def ctx_push(ctx):
    try:
        return ctx.timing()["push_ms"] / 1e3
    except Exception:
        return float("nan")


if __name__ == "__main__":
    main()
