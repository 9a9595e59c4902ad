"""(CPU, oracle) Would ranking the hot-table candidates by MEASURED visit frequency beat ranking by degree?

Push traces of one sample of seeds give push counts per node; a node's visit score is the sum of the push counts of
its neighbours (every push of u visits all of N(u)).  The share of traversed edges that land on the K best nodes is
then evaluated on a DIFFERENT sample of seeds, for the degree ranking and for the visit-score ranking.

usage: python tools/hot_rank_study.py NODES EDGES [CALIBRATION_SEEDS] [EVALUATION_SEEDS]
"""
import sys

import numpy as np
import scipy.sparse as sparse

sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes
from reveal_graph_embedding_amd.synthetic import rmat_graph


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    ncal = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    nev = int(sys.argv[4]) if len(sys.argv) > 4 else 300
    A = rmat_graph(n, m, 0)
    w, od, idg = oracle.get_natural_random_walk_matrix(A)
    deg = np.diff(w.indptr).astype(np.int64)
    seeds = seed_nodes(A)
    rng = np.random.default_rng(1)
    pick = rng.choice(seeds, size=ncal + nev, replace=False)
    cal, ev = pick[:ncal], pick[ncal:]
    pushes = np.zeros(n, dtype=np.float64)
    for sd in cal:
        t = oracle.push_trace(w, od, idg, int(sd), 0.1, 1e-5)
        np.add.at(pushes, t, 1.0)
    pattern = sparse.csr_matrix((np.ones(w.nnz), w.indices, w.indptr), shape=w.shape)
    score = pattern.T @ pushes                      # visits of v = pushes of its in-neighbours
    rankings = {"degree": np.argsort(-deg, kind="stable"), "visit score": np.argsort(-score, kind="stable")}
    ev_push = np.zeros(n, dtype=np.float64)
    tot = 0
    for sd in ev:
        t = oracle.push_trace(w, od, idg, int(sd), 0.1, 1e-5)
        np.add.at(ev_push, t, 1.0)
        tot += int(deg[t].sum())
    ev_visits = pattern.T @ ev_push                 # exact visit counts of the evaluation sample
    print("graph n=%d nnz=%d; calibration %d seeds, evaluation %d seeds (%.0f edges/seed)" % (n, w.nnz, ncal, nev, tot / nev))
    for K in (1280, 2560, 5120, 10240, 20480):
        line = "K=%6d" % K
        for name, order in rankings.items():
            line += "   %s: %.3f" % (name, ev_visits[order[:K]].sum() / tot)
        best = np.sort(ev_visits)[::-1][:K].sum() / tot
        print(line + "   (hindsight optimum %.3f)" % best)


if __name__ == "__main__":
    main()
