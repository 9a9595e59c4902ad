import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

GOLDEN_GRAPHS = ["ba300", "ba300_rho1e-3", "ba1500", "ws1000", "grid25", "corner",
                 "weighted", "selfloop", "directed", "rmat2000"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import scipy.sparse as sparse
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    n = int(g["n"])
    g["adjacency"] = sparse.csr_matrix((g["adj_data"], g["adj_indices"], g["adj_indptr"]), shape=(n, n))
    g["w"] = sparse.csr_matrix((g["w_data"], g["w_indices"], g["w_indptr"]), shape=(n, n))
    for t in (1, 3):
        g["feat%d" % t] = sparse.csr_matrix(
            (g["feat%d_data" % t], g["feat%d_indices" % t], g["feat%d_indptr" % t]), shape=(n, 2 * n))
    g["worker"] = sparse.csr_matrix((g["worker_data"], g["worker_indices"], g["worker_indptr"]), shape=(n, n))
    g["rho"] = float(g["rho"])
    g["epsilon"] = float(g["epsilon"])
    g["n"] = n
    return g


def canon(m):
    import scipy.sparse as sparse
    m = sparse.csr_matrix(m).copy()
    m.sum_duplicates()
    m.sort_indices()
    return m


def assert_same_sparse(a, b, values=True):
    a, b = canon(a), canon(b)
    assert a.shape == b.shape
    assert np.array_equal(a.indptr, b.indptr)
    assert np.array_equal(a.indices, b.indices)
    if values:
        assert np.array_equal(a.data, b.data)


@pytest.fixture(scope="session")
def rmat_1m():
    """BASELINE.json configs[2]'s graph (R-MAT 1M nodes / 50M sampled edges), generated once per session."""
    from reveal_graph_embedding_amd.synthetic import rmat_graph
    adjacency = rmat_graph(1000000, 50000000, seed=0)
    assert adjacency.nnz == 88123742
    return adjacency


@pytest.fixture(params=GOLDEN_GRAPHS)
def golden(request):
    return load_golden(request.param)
