import sys, time, numpy as np
sys.path.insert(0, ".")
from reveal_graph_embedding_amd.synthetic import rmat_graph
from reveal_graph_embedding_amd.embedding.arcte import arcte as A
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd import _native
n, m = int(sys.argv[1]), int(sys.argv[2])
adj = rmat_graph(n, m, 0)
for it in range(2):
    t0 = time.time(); f = A.arcte(adj, 0.1, 1e-5, 1); t1 = time.time()
    print("arcte() end-to-end %.3f s, nnz %d" % (t1 - t0, f.nnz), flush=True)
if n > 200000:
    sys.exit(0)
t = time.time(); w, od, idg = get_natural_random_walk_matrix(adj); print(" a1 transition %.3f" % (time.time() - t))
t = time.time(); seeds = A.seed_nodes(adj); print(" seed list %.3f" % (time.time() - t))
t = time.time(); ctx = _native.Context(w.indptr, w.indices, w.data, od, idg); print(" context (upload+slots) %.3f" % (time.time() - t))
t = time.time(); ctx.run_seeds(seeds, 0.1, 1e-5); print(" run_seeds %.3f" % (time.time() - t))
t = time.time(); colptr, rows = ctx.fetch(); print(" fetch D2H %.3f (%d rows)" % (time.time() - t, rows.size))
t = time.time(); ctx.close(); print(" close %.3f" % (time.time() - t))
t = time.time(); loc = A._seed_matrix(n, seeds, colptr, rows); print(" seed matrix (coo->csr) %.3f" % (time.time() - t))
import scipy.sparse as sparse
t = time.time()
identity = sparse.csr_matrix(sparse.eye(n, n, dtype=np.float64)); ones = adj.copy(); ones.data = np.ones_like(ones.data); base = identity + ones
print(" base block %.3f" % (time.time() - t))
t = time.time(); f = sparse.hstack([base, loc]).tocsr(); print(" hstack %.3f" % (time.time() - t))
