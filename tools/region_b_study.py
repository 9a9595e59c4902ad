"""(CPU, oracle) What does region B of the line state see?  Sizing study for its indirect lines (arcte_lines.hpp, IND).

Region A = the 8 M highest ranks (M lines with touched-bits in LDS), region B = the rest, MB = next power of two of
(n - 8 M) / 8 lines, eight ranks MB apart per line.  Per sampled seed: the share of the traversed edges that lands in region
B, how many of those are the first touch of their LINE (blind pool-line writes with indirect lines) and how many re-touches
(the updates that pay the entry read), and how many distinct lines a seed touches (= pool lines it needs).

usage: python tools/region_b_study.py NODES EDGES [SAMPLE] [M]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd.synthetic import rmat_graph


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    nsample = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    M = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
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
    RA = 8 * M
    MB = 1
    while MB < (n - RA + 7) // 8:
        MB *= 2
    tot = in_b = first_line = first_node = 0
    lines_per_seed, same_step = [], 0
    for sd in sample:
        t = oracle.push_trace(w, od, idg, int(sd), 0.1, 1e-5, cap=1 << 20)
        lens = deg[t]
        starts = indptr[t]
        total = int(lens.sum())
        idx = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(total)
        vr = rank[w.indices[idx]]
        tot += total
        b = vr[vr >= RA] - RA
        in_b += b.size
        ln = b % MB
        _, first_pos = np.unique(ln, return_index=True)
        first_line += first_pos.size
        first_node += np.unique(b).size
        lines_per_seed.append(first_pos.size)
        # re-touches of a line inside the 64-edge step of its first touch (the entry read ahead is stale: the slow path)
        step = np.arange(total)[vr >= RA] // 64          # (an upper bound: steps restart with every row)
        first_step = np.full(MB, -1, dtype=np.int64)
        first_step[ln[first_pos]] = step[first_pos]
        again = np.ones(b.size, dtype=bool)
        again[first_pos] = False
        same_step += int((first_step[ln[again]] == step[again]).sum())
    ns = len(sample)
    lp = np.asarray(lines_per_seed)
    print("graph n=%d nnz=%d, %d seeds, M=%d (region A: ranks < %d), MB=%d lines" % (n, w.nnz, ns, M, RA, MB))
    print("traversed edges per seed %.0f; in region B %.3f of them" % (tot / ns, in_b / tot))
    print("of region B's updates: first touch of the line %.3f (of the node %.3f), re-touches %.3f; re-touches inside the step of the"
          " first touch <= %.5f" % (first_line / max(in_b, 1), first_node / max(in_b, 1), 1 - first_line / max(in_b, 1),
                                    same_step / max(in_b, 1)))
    print("distinct region-B lines per seed (= pool lines): mean %.0f p50 %d p90 %d p99 %d max %d" % (
        lp.mean(), np.percentile(lp, 50), np.percentile(lp, 90), np.percentile(lp, 99), lp.max()))
    print("requests per traversed edge spent on region B: dense %.3f (claim hits the L2; blind line write or read + write),"
          " indirect %.3f (+ the entry read of every update, + the entry write of every first touch)" % (
              (first_line + 2 * (in_b - first_line)) / tot, (first_line * 3 + 3 * (in_b - first_line)) / tot))


if __name__ == "__main__":
    main()
