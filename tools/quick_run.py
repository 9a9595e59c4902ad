import sys, time
sys.path.insert(0, ".")
from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.synthetic import rmat_graph
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes
n, m = int(sys.argv[1]), int(sys.argv[2])
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t=time.time(); A = rmat_graph(n, m, 0); print("graph", time.time()-t, A.nnz, flush=True)
w, od, idg = get_natural_random_walk_matrix(A)
seeds = seed_nodes(A)
if len(sys.argv) > 4: seeds = seeds[::int(sys.argv[4])]
t=time.time(); ctx = _native.Context(w.indptr, w.indices, w.data, od, idg, n_slots=slots); print("ctx", time.time()-t, ctx.info(), flush=True)
for it in range(3):
    t=time.time(); ctx.run_seeds(seeds, 0.1, 1e-5); dt=time.time()-t
    st=ctx.stats(); tm=ctx.timing(); ns, tot = ctx.result_sizes()
    byt = 52*st['edges']+36*st['pushes']+4*st['enqueues']+36*st['support']
    print("run", it, "wall %.3fs"%dt, "seeds/s %.0f"%(ns/dt), tm, st, "rows", tot, "alg GB/s (push kernel) %.1f"%(byt/tm['push_ms']/1e6), flush=True)
