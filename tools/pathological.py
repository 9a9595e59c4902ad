"""Pathological shapes at scale: one giant hub, a long path, a dense bipartite block.  HIP vs oracle on all
(or sampled) seeds; prints timings so that cliffs show."""
import sys, time
import numpy as np, scipy.sparse as sparse
sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes


def star(n):
    i = np.zeros(n - 1, dtype=np.int64); j = np.arange(1, n)
    a = sparse.coo_matrix((np.ones(n - 1), (i, j)), shape=(n, n)).tocsr()
    return sparse.csr_matrix(a + a.T)


def star_with_ring(n):
    a = star(n).tolil()
    a = sparse.csr_matrix(a)
    j = np.arange(1, n); k = np.roll(j, 1)
    ring = sparse.coo_matrix((np.ones(n - 1), (j, k)), shape=(n, n)).tocsr()
    return sparse.csr_matrix(((a + ring + ring.T) > 0).astype(np.float64))


def path(n):
    i = np.arange(n - 1)
    a = sparse.coo_matrix((np.ones(n - 1), (i, i + 1)), shape=(n, n)).tocsr()
    return sparse.csr_matrix(a + a.T)


def bipartite(m):
    b = sparse.csr_matrix(np.ones((m, m)))
    return sparse.bmat([[None, b], [b.T, None]], format="csr")


for name, a in (("star+ring 200k", star_with_ring(200000)), ("path 200k", path(200000)), ("K(1500,1500)", bipartite(1500))):
    a = sparse.csr_matrix(a, dtype=np.float64)
    w, od, idg = get_natural_random_walk_matrix(a)
    seeds = seed_nodes(a)
    rng = np.random.default_rng(0)
    if seeds.size > 20000:
        seeds = np.concatenate([seeds[:5], rng.choice(seeds[5:], size=20000, replace=False)])
    seeds = np.sort(seeds)
    t = time.time()
    with _native.Context(w.indptr, w.indices, w.data, od, idg) as ctx:
        t1 = time.time()
        ctx.run_seeds(seeds, 0.1, 1e-5)
        t2 = time.time()
        colptr, rows, nop = ctx.fetch(want_nop=True)
        st = ctx.stats(); tm = ctx.timing()
    t = time.time()
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(w, od, idg, seeds, 0.1, 1e-5, threads=oracle.lib().oracle_max_threads(), want_stats=True)
    to = time.time() - t
    ok = np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
    seg = np.repeat(np.arange(seeds.size), np.diff(colptr))
    ok = ok and np.array_equal(rows[np.lexsort((rows, seg))], o_rows)
    print("%-16s n %7d nnz %9d seeds %6d  hip run %.3f s (eps %.1f ms, push %.1f ms)  oracle %.2f s  pushes %d edges %d reruns %d  identical %s"
          % (name, a.shape[0], a.nnz, seeds.size, t2 - t1, tm["eps_ms"], tm["push_ms"], to, st["pushes"], st["edges"], st["reruns"], ok), flush=True)
