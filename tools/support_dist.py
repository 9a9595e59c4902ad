import sys, numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd.synthetic import rmat_graph
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes
n, m = int(sys.argv[1]), int(sys.argv[2])
A = rmat_graph(n, m, 0)
w, od, idg = get_natural_random_walk_matrix(A)
seeds = seed_nodes(A)
rng = np.random.default_rng(3)
sample = rng.choice(seeds, size=400, replace=False)
sup, edg = [], []
for sd in sample:
    _, _, _, _, st = oracle.worker(w, od, idg, np.array([sd]), 0.1, 1e-5, want_stats=True)
    sup.append(st[3]); edg.append(st[1])
sup = np.array(sup); edg = np.array(edg)
print("support percentiles p10/p25/p50/p75/p90/p99:", np.percentile(sup, [10, 25, 50, 75, 90, 99]).astype(int))
for cap in (4000, 6500, 10000, 13000, 20000):
    fit = sup <= cap
    print("support <= %5d: %.1f%% of seeds, %.1f%% of edge work" % (cap, 100 * fit.mean(), 100 * edg[fit].sum() / edg.sum()))
