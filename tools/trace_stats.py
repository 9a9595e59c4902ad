import sys, numpy as np, collections
sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd.synthetic import rmat_graph
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes
n, m = int(sys.argv[1]), int(sys.argv[2])
A = rmat_graph(n, m, 0)
w, od, idg = get_natural_random_walk_matrix(A)
deg = np.diff(w.indptr).astype(np.int64)
seeds = seed_nodes(A)
rng = np.random.default_rng(1)
sample = rng.choice(seeds, size=200, replace=False)
cnt = collections.Counter(); edges_by_node = collections.Counter(); tot_edges = 0; tot_push = 0
per_seed_edges = []
for sd in sample:
    t = oracle.push_trace(w, od, idg, sd, 0.1, 1e-5)
    e = deg[t].sum(); per_seed_edges.append(e)
    tot_edges += e; tot_push += len(t)
    for u in set(t.tolist()): cnt[u] += 1
    for u in t.tolist(): edges_by_node[u] += deg[u]
print("seeds", len(sample), "pushes/seed", tot_push/len(sample), "edges/seed", tot_edges/len(sample))
top = sorted(edges_by_node.items(), key=lambda kv: -kv[1])[:15]
for u, e in top:
    print("node %7d deg %6d pushed by %3d/200 seeds, share of all edges %.3f" % (u, deg[u], cnt[u], e / tot_edges))
# edge share by degree class
classes = [(0, 64), (64, 512), (512, 4096), (4096, 10**9)]
for lo, hi in classes:
    e = sum(v for u, v in edges_by_node.items() if lo <= deg[u] < hi)
    print("deg in [%d,%d): edge share %.3f" % (lo, hi, e / tot_edges))
pe = np.array(per_seed_edges); print("per-seed edges: p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(pe, [10, 50, 90, 100])))
