"""(CPU, oracle) How many of a seed's state updates are FIRST touches of a 64-byte line?

Sizing study for the round-3 state layout: one float64 per node in RANK order (descending pattern in-count), eight
ranks per 64-byte line, a touched-line bitmap in LDS.  A first touch of a line is a blind whole-line write (no read);
every other update is a read-modify-write.  Prints, per LDS-table size K and ranks-per-line, the share of the
traversed edges that are on chip, blind line writes and read-modify-writes, plus the number of distinct pushed
nodes and the largest number of enqueues a single push makes.

usage: python tools/line_study.py NODES EDGES [SAMPLE]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd.synthetic import rmat_graph


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    nsample = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    A = rmat_graph(n, m, 0)
    w, od, idg = oracle.get_natural_random_walk_matrix(A)
    deg = np.diff(w.indptr).astype(np.int64)
    seeds = np.asarray(oracle.seed_list(A))
    rng = np.random.default_rng(1)
    sample = rng.choice(seeds, size=min(nsample, len(seeds)), replace=False)
    order = np.argsort(-deg, kind="stable")
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n)
    indptr = w.indptr.astype(np.int64)
    Ks = [1024, 1280, 2048, 2816, 3200]
    per_line = [4, 8, 16]
    tot = 0
    onchip = {K: 0 for K in Ks}
    blind = {(K, L): 0 for K in Ks for L in per_line}
    # strided layout: line = rank mod M, slot = rank div M -- a line's eight nodes are ranks M apart, so a line
    # of a frequently touched (high) rank shares with seven rarely touched ones
    Ms = [32768, 65536, 131072]
    strided = {(K, M): [0, 0] for K in Ks for M in Ms}      # [blind, covered visits]
    node_first = {K: 0 for K in Ks}
    pushed_distinct = []
    support = []
    cover = {R: 0 for R in (65536, 131072, 262144, 524288)}
    for sd in sample:
        t = oracle.push_trace(w, od, idg, int(sd), 0.1, 1e-5, cap=1 << 20)
        pushed_distinct.append(len(np.unique(t)))
        # visit sequence: the rows of the pushed nodes, in push order
        lens = deg[t]
        starts = indptr[t]
        total = int(lens.sum())
        idx = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(total)
        vr = rank[w.indices[idx]]
        tot += total
        support.append(len(np.unique(vr)))
        for R in cover:
            cover[R] += int((vr < R).sum())
        for K in Ks:
            off = vr[vr >= K]
            onchip[K] += total - off.size
            node_first[K] += len(np.unique(off))
            for L in per_line:
                blind[(K, L)] += len(np.unique(off // L))
            for M in Ms:
                cov = off[off < 8 * M]
                strided[(K, M)][0] += len(np.unique(cov % M))
                strided[(K, M)][1] += cov.size
    ns = len(sample)
    print("graph n=%d nnz=%d, %d seeds: %.0f edges/seed, distinct nodes/seed %.0f (p50 %d, p90 %d, max %d)" % (
        n, w.nnz, ns, tot / ns, np.mean(support), np.percentile(support, 50), np.percentile(support, 90), np.max(support)))
    print("distinct pushed nodes per seed: mean %.1f p50 %d p90 %d p99 %d max %d" % (
        np.mean(pushed_distinct), np.percentile(pushed_distinct, 50), np.percentile(pushed_distinct, 90),
        np.percentile(pushed_distinct, 99), np.max(pushed_distinct)))
    for R, c in cover.items():
        print("edge visits to ranks < %7d: %.3f" % (R, c / tot))
    for K in Ks:
        print("K=%5d: on chip %.3f | off chip: node first touches %.3f of the off-chip visits" % (
            K, onchip[K] / tot, node_first[K] / (tot - onchip[K])))
        for L in per_line:
            b = blind[(K, L)]
            off = tot - onchip[K]
            print("         %2d ranks per line: blind line writes %.3f of the off-chip visits (%.3f per edge), read-modify-writes %.3f per edge"
                  " -> memory requests per edge %.3f (today %.3f)" % (L, b / off, b / tot, (off - b) / tot,
                                                                    (b + 2 * (off - b)) / tot, 2 * off / tot))
        for M in Ms:
            b, cov = strided[(K, M)]
            off = tot - onchip[K]
            print("         strided, %6d lines (ranks < %7d, %5.1f KB of bitmap): covers %.3f of the off-chip visits, blind %.3f of them"
                  " -> requests per edge %.3f (the uncovered tail as read-modify-writes)" % (
                      M, 8 * M, M / 8 / 1024, cov / off, b / cov, (b + 2 * (off - b)) / tot))


if __name__ == "__main__":
    main()
