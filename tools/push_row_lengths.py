"""How long are the rows the push loop walks?  Edge-weighted distribution of the degree of the pushed nodes over a
sample of seeds of the R-MAT graph (CPU oracle's push trace): what a workgroup-cooperative push could split.

usage: python tools/push_row_lengths.py NODES EDGES [SAMPLE]
"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hot_sweep import load_graph
from oracle import oracle


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    sample = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    a = load_graph(n, m)
    w, od, idg = oracle.get_natural_random_walk_matrix(a)
    seeds = oracle.seed_list(a)
    deg = np.diff(w.indptr)
    rng = np.random.default_rng(1)
    lens = []
    for sd in rng.choice(seeds, size=min(sample, seeds.size), replace=False):
        lens.append(deg[oracle.push_trace(w, od, idg, int(sd), 0.1, 1e-5)])
    lens = np.concatenate(lens).astype(np.int64)
    total = lens.sum()
    print("graph n=%d nnz=%d; %d seeds, %d pushes, %d edges (%.0f per push)" % (n, a.nnz, sample, lens.size, total, total / lens.size))
    print("%-22s %10s %10s %14s" % ("row length", "pushes", "edges", "steps of 128"))
    steps = (lens + 127) // 128
    for lo, hi in ((1, 64), (65, 128), (129, 256), (257, 512), (513, 2048), (2049, 8192), (8193, 1 << 40)):
        sel = (lens >= lo) & (lens <= hi)
        print("%-22s %9.1f%% %9.1f%% %13.1f%%" % ("%d .. %s" % (lo, hi if hi < 1 << 40 else "max %d" % lens.max()), 100 * sel.mean(),
                                                  100 * lens[sel].sum() / total, 100 * steps[sel].sum() / steps.sum()))
    for wv in (2, 4):
        # a workgroup of wv wavefronts walks a row in ceil(steps / wv) rounds
        rounds = (steps + wv - 1) // wv
        print("%d wavefronts per seed: %.2fx fewer rounds per seed than steps now" % (wv, steps.sum() / rounds.sum()))


if __name__ == "__main__":
    main()
