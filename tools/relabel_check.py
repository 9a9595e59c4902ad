"""Experiment: does an internal relabelling of the nodes (degree-descending) raise the random-write rate?
Row order (= FIFO enqueue order) is kept: only the integer names change; results are mapped back."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.synthetic import rmat_graph
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes

n, m, stride = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A = rmat_graph(n, m, 0)
w, od, idg = get_natural_random_walk_matrix(A)
seeds = np.sort(seed_nodes(A)[::stride])
deg = np.diff(w.indptr)


def relabel(order):
    """order[new] = old.  Returns CSR arrays in the new naming with the ORIGINAL within-row order."""
    perm = np.empty(n, dtype=np.int64); perm[order] = np.arange(n)          # old -> new
    lens = deg[order]
    indptr = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=indptr[1:])
    src = np.repeat(w.indptr[:-1][order] - indptr[:-1], lens) + np.arange(indptr[-1])
    return indptr, perm[w.indices[src]].astype(np.int32), w.data[src], od[order], idg[order], perm


def run(tag, indptr, indices, data, odx, idx_, sd):
    ctx = _native.Context(indptr, indices, data, odx, idx_)
    best = 1e9
    for it in range(3):
        ctx.run_seeds(sd, 0.1, 1e-5)
        best = min(best, ctx.timing()["push_ms"])
    st = ctx.stats(); colptr, rows = ctx.fetch()
    print("%-18s push_ms %.1f  Gedges/s %.2f  rows %d" % (tag, best, st["edges"] / best / 1e6, rows.size), flush=True)
    ctx.close()
    return colptr, rows

c0, r0 = run("original ids", w.indptr, w.indices, w.data, od, idg, seeds)
for name, order in (("degree-descending", np.argsort(-deg, kind="stable")),
                    ("random", np.random.default_rng(0).permutation(n))):
    indptr, indices, data, odx, idx_, perm = relabel(order)
    sd_new = perm[seeds]
    srt = np.argsort(sd_new)
    c1, r1 = run(name, indptr, indices, data, odx, idx_, sd_new[srt])
    back = order[r1]                       # new -> old
    ok = True
    for j in range(0, seeds.size, max(1, seeds.size // 200)):
        k = srt[j]                         # seed k of the original list is entry j of the sorted new list
        a = np.sort(r0[c0[k]:c0[k + 1]]); b = np.sort(back[c1[j]:c1[j + 1]])
        ok &= np.array_equal(a, b)
    print("   results identical after mapping back:", ok, flush=True)
