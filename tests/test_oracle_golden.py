"""The CPU oracle against outputs of the reference itself (tests/golden/*.npz).

Everything here is bit-exact: float64 values are compared with array_equal.
"""
import hashlib
import os

import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import GOLDEN, assert_same_sparse, load_golden
from oracle import oracle


def test_np_pairwise_sum_matches_numpy():
    rng = np.random.default_rng(0)
    for n in list(range(0, 40)) + [127, 128, 129, 130, 136, 137, 255, 256, 257, 1000, 4097, 20001]:
        a = rng.uniform(0.0, 1000.0, size=n) * rng.choice([1e-6, 1.0, 1e6], size=n)
        assert oracle.np_sum(a) == (np.sum(a) if n else 0.0), n
        if n:
            assert oracle.np_sum(a) / n == a.mean()


def test_transition_matrix(golden):
    w, out_degree, in_degree = oracle.get_natural_random_walk_matrix(golden["adjacency"])
    assert_same_sparse(w, golden["w"])
    assert np.array_equal(out_degree, golden["out_degree"])
    assert np.array_equal(in_degree, golden["in_degree"])


def test_epsilon_effective_all_seeds(golden):
    w, od = golden["w"], golden["out_degree"]
    got = np.array([oracle.calculate_epsilon_effective(golden["rho"], golden["epsilon"], od[s],
                                                       od[w.indices[w.indptr[s]:w.indptr[s + 1]]])
                    for s in golden["all_seeds"]])
    assert np.array_equal(got, golden["all_eps_eff"])


def test_single_push(golden):
    w = golden["w"]
    u = int(golden["push_node"])
    s, r = golden["push_s_in"].copy(), golden["push_r_in"].copy()
    oracle.cumulative_pagerank_difference_limit_push(
        s, r, w.data[w.indptr[u]:w.indptr[u + 1]], w.indices[w.indptr[u]:w.indptr[u + 1]], u, golden["rho"])
    assert np.array_equal(s, golden["push_s_out"])
    assert np.array_equal(r, golden["push_r_out"])


@pytest.mark.parametrize("flavour", ["", "raw_"])
def test_similarity_slices(golden, flavour):
    n = golden["n"]
    for k, seed in enumerate(golden["seeds"]):
        eps = golden["eps_eff"][k] if flavour == "" else golden["epsilon"]
        s = np.zeros(n)
        r = np.zeros(n)
        nop = oracle.similarity(golden["w"], golden["in_degree"], seed, golden["rho"], eps, s, r)
        assert nop == golden[flavour + "nop"][k]
        for vec, tag in ((s, "s"), (r, "r")):
            lo, hi = golden[flavour + tag + "_ptr"][k], golden[flavour + tag + "_ptr"][k + 1]
            nz = np.nonzero(vec)[0]
            assert np.array_equal(nz, golden[flavour + tag + "_idx"][lo:hi])
            assert np.array_equal(vec[nz], golden[flavour + tag + "_val"][lo:hi])


def test_worker(golden):
    got = oracle.worker_matrix(golden["w"], golden["out_degree"], golden["in_degree"], golden["seeds"],
                               golden["rho"], golden["epsilon"])
    assert_same_sparse(got, golden["worker"])
    _, _, eps_eff, nop, _ = oracle.worker(golden["w"], golden["out_degree"], golden["in_degree"],
                                          golden["seeds"], golden["rho"], golden["epsilon"], want_stats=True)
    assert np.array_equal(eps_eff, golden["eps_eff"])
    assert np.array_equal(nop, golden["nop"])


@pytest.mark.parametrize("threads", [1, 3])
def test_arcte_full(golden, threads):
    got = oracle.arcte(golden["adjacency"], golden["rho"], golden["epsilon"], threads)
    assert_same_sparse(got, golden["feat%d" % threads])


def test_config1_rmat_full_size():
    """All 63 070 seeds of the config-1 graph against the reference's own run (8 processes)."""
    from reveal_graph_embedding_amd.synthetic import rmat_graph
    z = np.load(os.path.join(GOLDEN, "rmat100k_summary.npz"))
    adjacency = rmat_graph(100000, 2000000, seed=0)
    f = oracle.arcte(adjacency, float(z["rho"]), float(z["epsilon"]), oracle.lib().oracle_max_threads())
    f.sum_duplicates()
    f.sort_indices()
    assert f.nnz == int(z["nnz"])
    h = hashlib.sha256()
    h.update(f.indptr.astype(np.int64).tobytes())
    h.update(f.indices.astype(np.int64).tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), z["sha256"])
    local = sparse.csc_matrix(f[:, 100000:])
    assert np.array_equal(np.diff(local.indptr), z["local_col_counts"])


def _summary_hash(f):
    import hashlib
    f = sparse.csr_matrix(f)
    f.sum_duplicates()
    f.sort_indices()
    h = hashlib.sha256()
    h.update(f.indptr.astype(np.int64).tobytes())
    h.update(f.indices.astype(np.int64).tobytes())
    return f, np.frombuffer(h.digest(), dtype=np.uint8)


def load_ba20000():
    """BASELINE.json configs[0], stand-in no. 2: BA(20 000, 10) at rho = 1e-3, eps = 1e-5 as the
    reference's own arcte() answered it (tests/golden/make_golden_ba20000.py)."""
    z = np.load(os.path.join(GOLDEN, "ba20000_rho1e-3_summary.npz"))
    n = z["adj_indptr"].size - 1
    a = sparse.csr_matrix((np.ones(z["adj_indices"].size), z["adj_indices"], z["adj_indptr"]), shape=(n, n))
    return z, a


def test_oracle_config0_ba20000_rho1e3_matches_reference_hash():
    z, a = load_ba20000()
    f, digest = _summary_hash(oracle.arcte(a, float(z["rho"]), float(z["epsilon"]), oracle.lib().oracle_max_threads()))
    assert f.nnz == int(z["nnz"])
    assert np.array_equal(np.diff(sparse.csc_matrix(f[:, a.shape[0]:]).indptr), z["local_col_counts"])
    assert np.array_equal(digest, z["sha256"])
