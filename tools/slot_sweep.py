import sys
sys.path.insert(0, ".")
from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.synthetic import rmat_graph
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes
n, m, stride = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A = rmat_graph(n, m, 0)
w, od, idg = get_natural_random_walk_matrix(A)
seeds = seed_nodes(A)[::stride]
print("graph", n, A.nnz, "seeds", seeds.size, flush=True)
for slots in [int(x) for x in sys.argv[4:]]:
    ctx = _native.Context(w.indptr, w.indices, w.data, od, idg, n_slots=slots)
    best = None
    for it in range(3):
        ctx.run_seeds(seeds, 0.1, 1e-5)
        tm = ctx.timing(); st = ctx.stats()
        if best is None or tm['push_ms'] < best: best = tm['push_ms']
    byt = 52*st['edges']+36*st['pushes']+4*st['enqueues']+36*st['support']
    print("slots %5d  push_ms %.1f  seeds/s %.0f  Gedges/s %.2f  alg GB/s %.0f  dev GB %.1f" % (
        ctx.info()['slots'], best, seeds.size/best*1e3, st['edges']/best/1e6, byt/best/1e6, ctx.info()['device_bytes']/1e9), flush=True)
    ctx.close()
