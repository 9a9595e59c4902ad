"""The two-wavefront launch shape (ARCTE_HIP_COOP=1: a helper wavefront walks the second half of long rows) must be
invisible in the results: communities, push counts and work counters equal the oracle's whether rows are split or not,
with small tables (LDS, warm and dense targets inside one split row), small rings (the staged enqueues overflow and the
seed is re-run) and a small output arena."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle

from reveal_graph_embedding_amd import _native as _n

# the helper-wavefront shape lost its A/B in round 2 (profiles/r02/ab_interleaved_4_helper_wavefront.txt) and is compiled by
# `make AB=1` only
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _n.has_ab_builds(), reason="library built without the A/B launch shapes (make AB=1)")]


def hip(monkeypatch, w, out_degree, in_degree, seeds, rho, epsilon, coop, coop_min=256, hot=None, warm=None, **kw):
    from reveal_graph_embedding_amd import _native
    monkeypatch.setenv("ARCTE_HIP_COOP", "1" if coop else "0")
    monkeypatch.setenv("ARCTE_HIP_COOP_MIN", str(coop_min))
    for name, val in (("ARCTE_HIP_HOT", hot), ("ARCTE_HIP_WARM", warm)):
        if val is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(val))
    with _native.Context(w.indptr, w.indices, w.data, out_degree, in_degree, **kw) as ctx:
        ctx.run_seeds(seeds, rho, epsilon)
        colptr, rows, nop = ctx.fetch(want_nop=True)
        st = ctx.stats()
        info = ctx.info()
    info["split_rows"] = st["split_rows"]
    return colptr, rows, nop, [st["pushes"], st["edges"], st["enqueues"], st["support"]], st["reruns"], info


def sorted_rows(colptr, rows):
    seg = np.repeat(np.arange(colptr.size - 1), np.diff(colptr))
    return rows[np.lexsort((rows, seg))]


def hub_graph(n=6000, hubs=6, seed=5):
    """A few hubs with rows of 1 500-4 000 edges over a sparse random background: most pushes of a hub are split."""
    import scipy.sparse as sparse
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for h in range(hubs):
        nb = rng.choice(np.arange(hubs, n), size=int(rng.integers(1500, 4000)), replace=False)
        rows.append(np.full(nb.size, h)); cols.append(nb)
    r = rng.integers(0, n, size=4 * n); c = rng.integers(0, n, size=4 * n)
    keep = r != c
    rows.append(r[keep]); cols.append(c[keep])
    r = np.concatenate(rows); c = np.concatenate(cols)
    a = sparse.coo_matrix((np.ones(r.size), (r, c)), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.float64).tocsr()
    a.sort_indices()
    return a


@pytest.mark.parametrize("weighted", [False, True])
def test_split_rows_match_the_oracle(monkeypatch, weighted):
    a = hub_graph()
    if weighted:                                   # wide row streams (per-edge weights) instead of the narrow ones
        a = a.copy()
        a.data = np.random.default_rng(2).uniform(0.5, 2.0, size=a.nnz)
        a = ((a + a.T) * 0.5).tocsr()
        a.sort_indices()
    w, od, idg = oracle.get_natural_random_walk_matrix(a)
    seeds = oracle.seed_list(a)[:1500]
    rho, eps = 0.1, 1e-5
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(w, od, idg, seeds, rho, eps, threads=8, want_stats=True)
    long_rows = int((np.diff(w.indptr) >= 256).sum())
    assert long_rows >= 6
    for hot, warm in ((None, None), (8, 40), (8, 0), (0, 0)):
        colptr, rows, nop, stats, reruns, info = hip(monkeypatch, w, od, idg, seeds, rho, eps, coop=True, hot=hot, warm=warm)
        tag = "hot %s warm %s" % (hot, warm)
        assert np.array_equal(colptr, o_colptr), tag
        assert np.array_equal(nop, o_nop), tag
        assert np.array_equal(sorted_rows(colptr, rows), o_rows), tag
        assert stats == list(o_stats), tag
        assert (info["split_rows"] > 1000) == (hot != 0), tag           # (no LDS table, no helper)
    assert info["waves_per_cu"] % 2 == 0


def test_staged_enqueues_overflow_and_small_arena(monkeypatch):
    a = hub_graph(n=5000, hubs=4, seed=9)
    w, od, idg = oracle.get_natural_random_walk_matrix(a)
    seeds = oracle.seed_list(a)[:600]
    rho, eps = 0.1, 1e-6
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(w, od, idg, seeds, rho, eps, threads=8, want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_ARENA_ROWS", "20000")
    colptr, rows, nop, stats, reruns, info = hip(monkeypatch, w, od, idg, seeds, rho, eps, coop=True, queue_capacity=64, n_slots=16)
    assert reruns > 0
    assert info["split_rows"] > 0
    assert np.array_equal(colptr, o_colptr)
    assert np.array_equal(nop, o_nop)
    assert np.array_equal(sorted_rows(colptr, rows), o_rows)
    assert stats == list(o_stats)


@pytest.mark.parametrize("name", ["ba300", "corner", "selfloop", "directed", "rmat2000"])
def test_fixtures_with_helpers(monkeypatch, name):
    g = load_golden(name)
    w = g["w"]
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"],
                                                        want_stats=True)
    colptr, rows, nop, stats, _, _ = hip(monkeypatch, w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"], coop=True)
    assert np.array_equal(colptr, o_colptr)
    assert np.array_equal(nop, o_nop)
    assert np.array_equal(sorted_rows(colptr, rows), o_rows)
    assert stats == list(o_stats)
