"""The PageRank-flavoured siblings of the ARCTE path: PageRank and lazy-PageRank pushes,
their propagation drivers and workers.  CPU part: oracle vs reference fixtures.  GPU part: HIP vs both."""
import os

import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import GOLDEN, assert_same_sparse, load_golden
from oracle import oracle

PAGERANK_GRAPHS = ["ba300", "grid25", "corner", "weighted", "selfloop", "directed", "rmat2000"]
FLAVOURS = [("pr", oracle.PAGERANK), ("lazy", oracle.LAZY_PAGERANK)]


def load(name):
    g = load_golden(name)
    z = np.load(os.path.join(GOLDEN, name + "_pagerank.npz"))
    p = {k: z[k] for k in z.files}
    n = g["n"]
    for tag, _ in FLAVOURS:
        p[tag + "_worker"] = sparse.csr_matrix((np.ones(p[tag + "_worker_indices"].size), p[tag + "_worker_indices"],
                                                p[tag + "_worker_indptr"]), shape=(n, n))
        p[tag + "_feat"] = sparse.csr_matrix((p[tag + "_feat_data"], p[tag + "_feat_indices"], p[tag + "_feat_indptr"]),
                                             shape=(n, 2 * n))
    return g, p


def rho_of(tag, g, p):
    return float(p["lazy_rho"]) if tag == "lazy" else g["rho"]


def check_slices(g, p, tag, run):
    n = g["n"]
    for k, seed in enumerate(g["seeds"]):
        s, r = np.zeros(n), np.zeros(n)
        nop = run(seed, rho_of(tag, g, p), g["eps_eff"][k], s, r)
        assert nop == p[tag + "_nop"][k]
        for vec, v in ((s, "s"), (r, "r")):
            lo, hi = p[tag + "_" + v + "_ptr"][k], p[tag + "_" + v + "_ptr"][k + 1]
            nz = np.nonzero(vec)[0]
            assert np.array_equal(nz, p[tag + "_" + v + "_idx"][lo:hi])
            assert np.array_equal(vec[nz], p[tag + "_" + v + "_val"][lo:hi])


@pytest.mark.parametrize("name", PAGERANK_GRAPHS)
@pytest.mark.parametrize("tag,variant", FLAVOURS)
def test_oracle_matches_reference_fixtures(name, tag, variant):
    g, p = load(name)
    w = g["w"]
    u = int(g["seeds"][0])
    s, r = p[tag + "_push_s_in"].copy(), p[tag + "_push_r_in"].copy()
    oracle.push_variant(variant, s, r, w.data[w.indptr[u]:w.indptr[u + 1]], w.indices[w.indptr[u]:w.indptr[u + 1]],
                        u, g["rho"], 0.5)
    assert np.array_equal(s, p[tag + "_push_s_out"]) and np.array_equal(r, p[tag + "_push_r_out"])
    check_slices(g, p, tag, lambda seed, rho, eps, s, r: oracle.similarity_variant(
        variant, w, g["in_degree"], seed, rho, eps, s, r, 0.5))
    got = oracle.worker_matrix(w, g["out_degree"], g["in_degree"], g["seeds"], g["rho"], g["epsilon"], variant=variant)
    assert_same_sparse(got, p[tag + "_worker"], values=False)
    assert_same_sparse(oracle.arcte(g["adjacency"], g["rho"], g["epsilon"], 2, variant=variant), p[tag + "_feat"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", PAGERANK_GRAPHS)
@pytest.mark.parametrize("tag,variant", FLAVOURS)
def test_hip_matches_reference_fixtures_and_oracle(name, tag, variant):
    from reveal_graph_embedding_amd import _native
    from reveal_graph_embedding_amd.embedding.arcte import arcte as A
    from reveal_graph_embedding_amd.eps_randomwalk import push as P, similarity as S
    g, p = load(name)
    w = g["w"]
    n = g["n"]
    a_i = np.ndarray(n, dtype=np.ndarray)
    w_i = np.ndarray(n, dtype=np.ndarray)
    for i in range(n):
        a_i[i] = w.indices[w.indptr[i]:w.indptr[i + 1]]
        w_i[i] = w.data[w.indptr[i]:w.indptr[i + 1]]
    # one push (push.py:4-38)
    u = int(g["seeds"][0])
    s, r = p[tag + "_push_s_in"].copy(), p[tag + "_push_r_in"].copy()
    if tag == "pr":
        P.pagerank_limit_push(s, r, w_i[u], a_i[u], u, g["rho"])
    else:
        P.pagerank_lazy_push(s, r, w_i[u], a_i[u], u, g["rho"], 0.5)
    assert np.array_equal(s, p[tag + "_push_s_out"]) and np.array_equal(r, p[tag + "_push_r_out"])
    # propagation slices with the reference's own call shape (similarity.py:11-146)
    if tag == "pr":
        run = lambda seed, rho, eps, s, r: S.fast_approximate_personalized_pagerank(
            s, r, w_i, a_i, g["out_degree"], g["in_degree"], seed, rho, eps)
    else:
        run = lambda seed, rho, eps, s, r: S.lazy_approximate_personalized_pagerank(
            s, r, w_i, a_i, g["out_degree"], g["in_degree"], seed, rho, eps, 0.5)
    check_slices(g, p, tag, run)
    # workers and drivers (arcte.py:53-276, 391-588)
    worker = A.arcte_with_pagerank_worker if tag == "pr" else A.arcte_with_lazy_pagerank_worker
    got = worker(g["seeds"], w.indices, w.indptr, w.data, g["out_degree"], g["in_degree"], g["rho"], g["epsilon"])
    assert_same_sparse(got, p[tag + "_worker"], values=False)
    driver = A.arcte_with_pagerank if tag == "pr" else A.arcte_with_lazy_pagerank
    assert_same_sparse(driver(g["adjacency"], g["rho"], g["epsilon"], 1), p[tag + "_feat"])
    # counters against the oracle on every seed
    with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
        rho = g["rho"]
        ctx.run_seeds(g["all_seeds"], (rho * 0.5) / (1 - 0.5 * rho) if tag == "lazy" else rho, g["epsilon"],
                      variant=variant, laziness_factor=0.5)
        colptr, rows, nop = ctx.fetch(want_nop=True)
        st = ctx.stats()
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"],
                                                        g["epsilon"], want_stats=True, variant=variant)
    assert np.array_equal(nop, o_nop) and np.array_equal(colptr, o_colptr)
    assert [st["pushes"], st["edges"], st["enqueues"], st["support"]] == list(o_stats)


def _config1_check(f, z, tag):
    import hashlib
    f.sum_duplicates()
    f.sort_indices()
    assert f.nnz == int(z[tag + "_nnz"])
    local = sparse.csc_matrix(f[:, 100000:])
    assert np.array_equal(np.diff(local.indptr), z[tag + "_local_col_counts"])
    h = hashlib.sha256()
    h.update(f.indptr.astype(np.int64).tobytes())
    h.update(f.indices.astype(np.int64).tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), z[tag + "_sha256"])


@pytest.mark.parametrize("tag,variant", FLAVOURS)
def test_oracle_config1_full_size_hash(tag, variant):
    """All 63 070 seeds of the config-1 R-MAT graph against the reference's own 8-process run."""
    from reveal_graph_embedding_amd.synthetic import rmat_graph
    z = np.load(os.path.join(GOLDEN, "rmat100k_pagerank_summary.npz"))
    adjacency = rmat_graph(100000, 2000000, seed=0)
    f = oracle.arcte(adjacency, float(z["rho"]), float(z["epsilon"]), oracle.lib().oracle_max_threads(), variant=variant)
    _config1_check(f, z, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,variant", FLAVOURS)
def test_hip_config1_full_size_hash(tag, variant):
    from reveal_graph_embedding_amd.embedding.arcte import arcte as A
    from reveal_graph_embedding_amd.synthetic import rmat_graph
    z = np.load(os.path.join(GOLDEN, "rmat100k_pagerank_summary.npz"))
    adjacency = rmat_graph(100000, 2000000, seed=0)
    driver = A.arcte_with_pagerank if tag == "pr" else A.arcte_with_lazy_pagerank
    _config1_check(driver(adjacency, float(z["rho"]), float(z["epsilon"]), 1), z, tag)
