"""SURVEY.md section 8 rows f3 / f4 at the bench graph's size: arcte_and_centrality on the R-MAT graph (EVERY node a
seed, raw epsilon) and the feature weighting of its n x (n + communities) matrix, timed on the GPU, with parity at
size: a node block against the CPU oracle bit for bit, the block sums against the full run, and the weighting
against vectorised numpy restatements of the reference's formulas on the fetched matrix.

usage: python tools/measure_centrality_weighting.py NODES EDGES [ORACLE_BLOCK_NODES]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sparse

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hot_sweep import load_graph
from oracle import oracle
from reveal_graph_embedding_amd import _native

RHO, EPS = 0.1, 1e-5


def timed(label, fn, nbytes=None):
    t = time.perf_counter()
    out = fn()
    dt = time.perf_counter() - t
    extra = "  %.0f GB/s of %.1f GB" % (nbytes / dt / 1e9, nbytes / 1e9) if nbytes else ""
    print(" %-58s %8.1f ms%s" % (label, dt * 1e3, extra), flush=True)
    return out


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    block = int(sys.argv[3]) if len(sys.argv) > 3 else 1500          # 0: time only (for runs under rocprofv3)
    check = block > 0
    adj = load_graph(n, m)
    print("graph n=%d nnz=%d" % (n, adj.nnz), flush=True)
    read_gbps, _ = _native.stream_bandwidth(0)
    print("streaming read rate of this box %.0f GB/s" % read_gbps, flush=True)
    with _native.Context.from_adjacency(adj.indptr, adj.indices, adj.data) as ctx:
        # ---- f3: the propagation of every node + the centrality fold
        for it in range(2):
            t = time.perf_counter()
            ctx.run_centrality(RHO, EPS)
            wall = time.perf_counter() - t
            tm, st = ctx.timing(), ctx.stats()
            print("run_centrality, all %d nodes: %.3f s wall, push kernel %.1f ms, %.0f seeds/s; per seed: %.0f pushes %.0f edges "
                  "%.0f support; reruns %d" % (n, wall, tm["push_ms"], n / wall, st["pushes"] / n, st["edges"] / n,
                                               st["support"] / n, st["reruns"]), flush=True)
        full = ctx.centrality()
        ns, total = ctx.result_sizes()
        print(" emitted communities' members %d, centrality sum %.6f" % (total, full.sum()), flush=True)
        alg = 52 * st["edges"] + 36 * st["pushes"] + 4 * st["enqueues"] + 36 * st["support"]
        print(" algorithmic bytes %.0f GB -> %.0f GB/s = %.3f of 8 TB/s (same model as the bench kernel; the contribution"
              " stream of this mode is extra)" % (alg / 1e9, alg / tm["push_ms"] / 1e6, alg / tm["push_ms"] / 1e6 / 8000), flush=True)

        # ---- the feature matrix of that run, weighted on the device
        t = time.perf_counter()
        f = _native.Features.from_result(ctx, with_base_block=True)
        rows, cols, nnz = f.sizes()
        print(" device assembly of the %d x %d matrix, %d stored entries  %.1f ms" % (rows, cols, nnz, (time.perf_counter() - t) * 1e3),
              flush=True)
        fetch = f.to_scipy if check else (lambda: None)
        before = timed("fetch (D2H of indptr, indices, data)", fetch, nnz * 12)
        # bytes per stored entry: column counts read the index (4); the division reads index + value and writes the value
        timed("normalize_columns (k_feat_column_counts, _idf, _divide_columns)", f.normalize_columns, nnz * (4 + 4 + 8 + 8))
        nc = timed("fetch", fetch, nnz * 12)
        timed("normalize_rows (k_feat_normalize_rows: read, read, write)", f.normalize_rows, nnz * (8 + 8 + 8))
        nr = timed("fetch", fetch, nnz * 12)
        rng = np.random.default_rng(3)
        labels = rng.integers(0, 5, size=rows)
        y = sparse.csr_matrix((np.ones(rows), (np.arange(rows), labels)), shape=(rows, 5))
        weights = timed("chi2 contingency + peak-SNR weights (5 classes, every row)",
                        lambda: f.chi2_psnr_weights(y.indptr, y.indices, 5), nnz * 4)
        timed("community_weighting (scale, drop zeros, row-normalise)", lambda: f.community_weighting(weights), nnz * (4 + 8 + 8 + 24))
        cw = timed("fetch", fetch, f.sizes()[2] * 12)
        f.close()
        if not check:
            return

        # ---- parity at size
        # (1) a node block against the oracle: members and partial centrality bit for bit
        ctx.run_centrality(RHO, EPS, 0, block)
        colptr, members = ctx.fetch()
        part = ctx.centrality()
        t = time.perf_counter()
        o_colptr, o_members, o_part = oracle.centrality_block(adj, RHO, EPS, 0, block)
        print(" oracle on nodes [0, %d): %.1f s" % (block, time.perf_counter() - t), flush=True)
        has_row = np.diff(adj.indptr)[:block] > 0              # the HIP result holds the seeds only: nodes with out-edges
        o_sizes = np.diff(o_colptr)
        assert not o_sizes[~has_row].any()
        assert np.array_equal(np.diff(colptr), o_sizes[has_row]), "community sizes differ"
        seg = np.repeat(np.arange(colptr.size - 1), np.diff(colptr))
        assert np.array_equal(members[np.lexsort((members, seg))], o_members[np.lexsort((o_members, seg))]), "members differ"
        assert np.array_equal(part, o_part), "partial centrality differs"
        print(" nodes [0, %d): communities and partial centrality IDENTICAL to the oracle (%d members)" % (block, members.size), flush=True)
        # (2) two half blocks add up to the full run (another summation order: rounding only)
        ctx.run_centrality(RHO, EPS, 0, n // 2)
        lo = ctx.centrality()
        ctx.run_centrality(RHO, EPS, n // 2, n)
        hi = ctx.centrality()
        err = np.max(np.abs(lo + hi - full) / np.maximum(full, 1e-300))
        assert err < 1e-12, err
        print(" half-block sums vs full run: max relative difference %.2e" % err, flush=True)

    # (3) the weighting against numpy restatements of common.py:49-67 / :29-46 and community_weighting.py:87-125
    assert np.array_equal(before.indptr, nc.indptr) and np.array_equal(before.indices, nc.indices)
    df = np.bincount(before.indices, minlength=cols)
    scale = np.ones(cols)
    scale[df > 1] = np.sqrt(np.log(df[df > 1]))
    want = before.data / scale[before.indices]
    err = np.max(np.abs(nc.data - want) / want)
    assert err < 1e-14, err
    print(" normalize_columns vs numpy: max relative difference %.2e over %d entries" % (err, nnz), flush=True)
    starts = nc.indptr[:-1][np.diff(nc.indptr) > 0]
    norms = np.ones(rows)
    norms[np.diff(nc.indptr) > 0] = np.sqrt(np.add.reduceat(nc.data * nc.data, starts))
    want = nc.data / np.repeat(norms, np.diff(nc.indptr))
    err = np.max(np.abs(nr.data - want) / want)
    assert err < 1e-13, err
    print(" normalize_rows vs numpy: max relative difference %.2e; row norms within %.2e of 1" % (
        err, np.max(np.abs(np.sqrt(np.add.reduceat(nr.data * nr.data, starts)) - 1.0))), flush=True)
    factor = np.ones(cols)
    big = df > 1
    factor[big] = np.where(weights[big] == 0.0, 0.0, np.log(1.0 + weights[big]))
    x = nr.data * factor[nr.indices]
    keep = x != 0.0
    kept_rows = np.repeat(np.arange(rows), np.diff(nr.indptr))[keep]
    x = x[keep]
    assert cw.nnz == x.size and np.array_equal(cw.indices, nr.indices[keep]), "surviving pattern differs"
    cnt = np.bincount(kept_rows, minlength=rows)
    starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])[cnt > 0]
    norms = np.ones(rows)
    norms[cnt > 0] = np.sqrt(np.add.reduceat(x * x, starts))
    want = x / norms[kept_rows]
    err = np.max(np.abs(cw.data - want) / want)
    assert err < 1e-12, err
    print(" community_weighting vs numpy: pattern identical (%d of %d entries survive), max relative difference %.2e" % (
        x.size, nnz, err), flush=True)
    print("ALL CHECKS PASSED", flush=True)


if __name__ == "__main__":
    main()
