"""Oracle vs the reference's OTHER driver and the feature weighting behind it (CPU).

arcte_and_centrality (embedding/arcte/cython_opt/arcte.pyx:125-241), normalize_columns / normalize_rows
(embedding/common.py:29-67) and community_weighting.py:11-125, pinned by tests/golden/centrality_*.npz and
weighting_*.npz -- outputs of the reference itself (tests/golden/make_golden_centrality.py compiles its .pyx files
in a temporary directory and runs them).  No fixture graph has a seed whose community depends on the tie order of
numpy's unstable argsort (`ambiguous` is all zero), so the patterns are pinned exactly."""
import os

import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import GOLDEN, load_golden
from oracle import oracle

CENTRALITY_GRAPHS = ["ba300", "weighted", "selfloop", "grid25", "ws1000", "rmat2000", "directed"]
WEIGHTING_GRAPHS = ["ba300", "weighted", "rmat2000", "selfloop"]


def load_centrality(name):
    z = np.load(os.path.join(GOLDEN, "centrality_%s.npz" % name))
    g = {k: z[k] for k in z.files}
    n = int(g["n"])
    g["adjacency"] = sparse.csr_matrix((g["adj_data"], g["adj_indices"], g["adj_indptr"]), shape=(n, n))
    g["features"] = sparse.csr_matrix((g["feat_data"], g["feat_indices"], g["feat_indptr"]), shape=tuple(g["feat_shape"]))
    return g


def load_weighting(name):
    z = np.load(os.path.join(GOLDEN, "weighting_%s.npz" % name))
    g = {k: z[k] for k in z.files}
    n = int(g["n"])
    for tag, rows in (("nc", n), ("nr", n), ("xt", g["train"].size), ("xs", g["test"].size)):
        g[tag] = sparse.csr_matrix((g[tag + "_data"], g[tag + "_indices"], g[tag + "_indptr"]), shape=(rows, 2 * n))
    return g


def assert_close_sparse(a, b, rtol):
    a, b = sparse.csr_matrix(a), sparse.csr_matrix(b)
    a.sort_indices()
    b.sort_indices()
    assert a.shape == b.shape
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    np.testing.assert_allclose(a.data, b.data, rtol=rtol, atol=0)


@pytest.mark.parametrize("name", CENTRALITY_GRAPHS)
def test_oracle_arcte_and_centrality_matches_reference(name):
    g = load_centrality(name)
    assert int(g["ambiguous"].sum()) == 0
    f, c = oracle.arcte_and_centrality(g["adjacency"], float(g["rho"]), float(g["epsilon"]))
    assert f.shape[1] - f.shape[0] == int(g["ncols_local"])
    assert_close_sparse(f, g["features"], rtol=1e-13)
    # per node a left fold over the seeds in index order (arcte.pyx:190-191): bit for bit
    assert np.array_equal(c, g["centrality"])


@pytest.mark.parametrize("name", WEIGHTING_GRAPHS)
def test_oracle_feature_weighting_matches_reference(name):
    w = load_weighting(name)
    x = load_golden(name)["feat1"]
    nc = oracle.normalize_columns(x)
    assert_close_sparse(nc, w["nc"], rtol=1e-15)
    assert_close_sparse(oracle.normalize_rows(nc), w["nr"], rtol=1e-14)
    x_train, x_test = sparse.csr_matrix(w["nc"][w["train"]]), sparse.csr_matrix(w["nc"][w["test"]])
    cm = oracle.chi2_contingency_matrix(x_train, w["labels"][w["train"]])
    np.testing.assert_allclose(cm, w["contingency"], rtol=1e-13, atol=0)
    wts = oracle.peak_snr_weight_aggregation(cm)
    np.testing.assert_allclose(wts, w["weights"], rtol=1e-12, atol=0)
    xt, xs = oracle.community_weighting(x_train, x_test, w["weights"])
    assert_close_sparse(xt, w["xt"], rtol=1e-13)
    assert_close_sparse(xs, w["xs"], rtol=1e-13)
