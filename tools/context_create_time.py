import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from hot_sweep import load_graph
from reveal_graph_embedding_amd import _native
A = load_graph(1000000, 50000000)
for k in range(3):
    t = time.time()
    ctx = _native.Context.from_adjacency(A.indptr, A.indices, A.data)
    t1 = time.time()
    s = ctx.seed_list()[:2000]
    ctx.run_seeds(s, 0.1, 1e-5)
    t2 = time.time()
    print("context %d: create %.3f s, first small run %.3f s, probe %s" % (k, t1 - t, t2 - t1, ctx.placement_info()), flush=True)
    ctx.close()
    print("  close %.3f s" % (time.time() - t2), flush=True)
