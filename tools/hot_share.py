"""(CPU, oracle) Which share of the traversed edges lands on the K highest-degree nodes?

Sizing study for the LDS-resident hot state of the push kernel: the entries of the K hottest nodes live in
LDS (zeroed per seed), everything else stays in the dense per-slot state in HBM.  Prints, per K, the share
of edge visits that would be served from LDS, over a sample of seeds of the R-MAT graph.

usage: python tools/hot_share.py NODES EDGES [SAMPLE]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.synthetic import rmat_graph


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    nsample = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    A = rmat_graph(n, m, 0)
    w, od, idg = get_natural_random_walk_matrix(A)
    deg = np.diff(w.indptr).astype(np.int64)
    seeds = seed_nodes(A)
    rng = np.random.default_rng(1)
    sample = rng.choice(seeds, size=min(nsample, len(seeds)), replace=False)
    order = np.argsort(-deg, kind="stable")
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n)
    edge_rank = rank[w.indices]                       # rank of every edge's target
    Ks = [512, 1024, 1280, 2048, 2560, 4096, 5120, 8192, 16384, 65536]
    # per node: number of neighbours among the K hottest, by a cumulative count over sorted ranks
    hot_in_row = {}
    for K in Ks:
        flag = (edge_rank < K).astype(np.int64)
        csum = np.concatenate([[0], np.cumsum(flag)])
        hot_in_row[K] = csum[w.indptr[1:]] - csum[w.indptr[:-1]]
    tot = 0
    hot = {K: 0 for K in Ks}
    pushes = 0
    pushed_hot = {K: 0 for K in Ks}
    for sd in sample:
        t = oracle.push_trace(w, od, idg, int(sd), 0.1, 1e-5)
        tot += int(deg[t].sum())
        pushes += len(t)
        for K in Ks:
            hot[K] += int(hot_in_row[K][t].sum())
            pushed_hot[K] += int((rank[t] < K).sum())
    print("graph n=%d nnz=%d seeds sampled %d: %.0f edges/seed, %.1f pushes/seed" % (n, w.nnz, len(sample), tot / len(sample), pushes / len(sample)))
    print("degree mass of the K hottest nodes vs share of traversed edges that land on them vs share of pushes of them")
    dsum = deg.sum()
    for K in Ks:
        print("K=%6d  endpoint mass %.3f  edge-visit share %.3f  pushed-node share %.3f  (LDS %.0f KB per slot at 16 B)" %
              (K, deg[order[:K]].sum() / dsum, hot[K] / tot, pushed_hot[K] / pushes, K * 16 / 1024))


if __name__ == "__main__":
    main()
