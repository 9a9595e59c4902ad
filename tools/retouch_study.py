"""(round 4; CPU, oracle) Where do the read-modify-writes of the line state come from, and what could take them away?

The push kernel sits at the memory system's random-request wall (profiles/r03): a blind first touch of a line costs one
64-byte request, a re-touch two (read + write-back).  This study replays the visit sequence of sampled seeds (ranks of the
targets of every traversed edge, in order) against

  * the shipped layout (K on-chip values, M strided lines with a touched-bit in LDS),
  * other ways to map ranks to lines with the same number of touched-bits (hot ranks share their line with colder ones),
  * a DYNAMIC on-chip cache of C tagged values beside / instead of the static table (direct mapped and LRU),
  * the reuse distance of node re-touches (how far back, in off-chip visits, was the node touched last),

and prints memory requests per traversed edge for each.  Also: candidates per seed as the kernel's lower bound admits them.

usage: python tools/retouch_study.py NODES EDGES [SAMPLE]
"""
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd.synthetic import rmat_graph


def visits_of(w, deg, indptr, rank, t):
    lens = deg[t]
    starts = indptr[t]
    total = int(lens.sum())
    idx = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(total)
    return rank[w.indices[idx]]


def requests_static(vr, K, line_of):
    """on-chip ranks < K; every other visit: first touch of its line blind (1 request), else read-modify-write (2)."""
    off = vr[vr >= K]
    ln = line_of(off)
    nblind = len(np.unique(ln))
    return vr.size - off.size, nblind, off.size - nblind


def sim_dynamic(vr, K, M, C, lru):
    """static table of K ranks + a tagged cache of C values.  A miss evicts: the evicted value goes to its home line
    (blind whole-line write when the line is untouched, else a read-modify-write: the line holds other nodes); the
    incoming node's value is 0 when its line is untouched, else one read."""
    touched = set()
    req = 0
    hits = 0
    if lru:
        cache = OrderedDict()
    else:
        cache = {}
    for r in vr.tolist():
        if r < K:
            continue
        if lru:
            if r in cache:
                cache.move_to_end(r)
                hits += 1
                continue
            if len(cache) >= C:
                ev, _ = cache.popitem(last=False)
                ln = ev % M
                if ln in touched:
                    req += 2
                else:
                    touched.add(ln)
                    req += 1
            cache[r] = 1
        else:
            s = r % C
            ev = cache.get(s)
            if ev == r:
                hits += 1
                continue
            if ev is not None:
                ln = ev % M
                if ln in touched:
                    req += 2
                else:
                    touched.add(ln)
                    req += 1
            cache[s] = r
        # the incoming node: was it written to memory before (evicted earlier)?  Its LINE's bit says "maybe": one read
        if (r % M) in touched:
            req += 1
    # what is still cached at the end is never written: extraction reads it on chip
    return req, hits


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    nsample = int(sys.argv[3]) if len(sys.argv) > 3 else 100
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
    K, M = 512, 65536
    tot = 0
    acc = {}

    def add(name, onchip, blind, rmw):
        a = acc.setdefault(name, [0, 0, 0])
        a[0] += onchip; a[1] += blind; a[2] += rmw

    dyn = {}
    dist_hist = np.zeros(24, dtype=np.int64)
    retouch_rank_hist = np.zeros(8, dtype=np.int64)
    rank_edges = [512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 62]
    for sd in sample:
        t = oracle.push_trace(w, od, idg, int(sd), 0.1, 1e-5, cap=1 << 20)
        vr = visits_of(w, deg, indptr, rank, t)
        tot += vr.size
        add("shipped: K=512, strided 65536 lines", *requests_static(vr, K, lambda r: r % M))
        add("K=1024, strided 32768 lines (4 KB bitmap)", *requests_static(vr, 1024, lambda r: r % 32768))
        add("K=0, strided 65536 lines", *requests_static(vr, 0, lambda r: r % M))
        add("K=512, one node per line (upper bound of any line mapping)", *requests_static(vr, K, lambda r: r))
        # hot ranks alone with far-colder ones: ranks < 65536 keep line = rank and share it with ranks >= 262144 only (7 of
        # them, strided); ranks 65536..262143 packed eight consecutive ranks per line behind them (24 576 lines more)
        def two_tier(r):
            out = np.empty_like(r)
            hotm = r < 65536
            mid = (r >= 65536) & (r < 262144)
            cold = r >= 262144
            out[hotm] = r[hotm]
            out[mid] = 65536 + (r[mid] - 65536) // 8
            out[cold] = (r[cold] - 262144) % 65536
            return out
        add("K=512, two tiers (90 112 lines, 11 KB bitmap)", *requests_static(vr, K, two_tier))
        # strided with the eight members of a line chosen so that a hot rank's companions are the COLDEST: line l holds
        # rank l and ranks 8M-1-l-k*M ... (reverse pairing)
        def reverse_pair(r):
            out = np.where(r < M, r, (8 * M - 1 - r) % M)
            return out
        add("K=512, 65536 lines, companions reversed", *requests_static(vr, K, reverse_pair))
        # reuse distance of node re-touches among off-chip visits
        off = vr[vr >= K]
        last = {}
        for i, r in enumerate(off.tolist()):
            j = last.get(r)
            if j is not None:
                d = i - j
                dist_hist[min(23, int(np.log2(d)) if d > 0 else 0)] += 1
                for b, e in enumerate(rank_edges):
                    if r < e:
                        retouch_rank_hist[b] += 1
                        break
            last[r] = i
        for C, lru, Kd in ((512, True, 0), (512, False, 0), (341, False, 0), (256, False, 256), (2048, True, 0), (8192, True, 0)):
            rq, hits = sim_dynamic(vr, Kd, M, C, lru)
            d = dyn.setdefault((C, lru, Kd), [0, 0, 0])
            d[0] += rq; d[1] += hits; d[2] += int((vr < Kd).sum())
    ns = len(sample)
    print("graph n=%d nnz=%d, %d seeds, %.0f traversed edges per seed" % (n, w.nnz, ns, tot / ns))
    for name, (oc, b, r) in acc.items():
        print("%-62s on chip %.3f  blind %.3f  rmw %.3f  -> state requests per edge %.3f" % (name, oc / tot, b / tot, r / tot, (b + 2 * r) / tot))
    for (C, lru, Kd), (rq, hits, st) in dyn.items():
        print("dynamic cache of %5d values (%s) + static %4d: on chip %.3f (cache hits %.3f)  -> state requests per edge %.3f" % (
            C, "LRU" if lru else "direct mapped", Kd, (hits + st) / tot, hits / tot, rq / tot))
    tot_rt = dist_hist.sum()
    print("node re-touches among off-chip visits: %.3f per edge; reuse distance (off-chip visits), cumulative:" % (tot_rt / tot))
    cum = 0
    for b in range(24):
        cum += dist_hist[b]
        if dist_hist[b]:
            print("   < %8d: %.3f" % (2 << b, cum / tot_rt))
    print("re-touched node's rank: " + "  ".join("<%s: %.3f" % ("inf" if e > 1 << 40 else e, c / max(1, tot_rt)) for e, c in zip(rank_edges, retouch_rank_hist)))


if __name__ == "__main__":
    main()
