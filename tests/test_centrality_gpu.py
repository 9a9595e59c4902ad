"""arcte_and_centrality (embedding/arcte/cython_opt/arcte.pyx:125-241) on the GPU against the reference's own outputs
(tests/golden/centrality_*.npz) and the oracle: community pattern exactly, centrality BIT FOR BIT (per node the
contributions are folded in seed order, as the reference adds them), normalised feature values within 1e-12."""
import numpy as np
import pytest

from conftest import GOLDEN  # noqa: F401  (conftest puts the repo root on sys.path)
from oracle import oracle
from test_centrality_weighting_cpu import CENTRALITY_GRAPHS, assert_close_sparse, load_centrality

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CENTRALITY_GRAPHS)
def test_arcte_and_centrality_matches_reference(name):
    from reveal_graph_embedding_amd.embedding.arcte.cython_opt.arcte import arcte, arcte_and_centrality
    g = load_centrality(name)
    f, c = arcte_and_centrality(g["adjacency"], float(g["rho"]), float(g["epsilon"]))
    assert f.shape == tuple(g["feat_shape"])
    assert_close_sparse(f, g["features"], 1e-12)
    assert np.array_equal(c, g["centrality"])
    if name == "ba300":
        assert_close_sparse(arcte(g["adjacency"], float(g["rho"]), float(g["epsilon"])), g["features"], 1e-12)


@pytest.mark.parametrize("name", ["rmat2000", "weighted"])
def test_small_batches_and_blocks(name, monkeypatch):
    """A contribution arena far smaller than one pass forces many batches (and batch halving); node blocks give the
    partial sums of a sharded run.  Communities, push counts and -- for the whole range -- the bits of the centrality
    must not depend on the batching."""
    from reveal_graph_embedding_amd import _native
    g = load_centrality(name)
    a = g["adjacency"]
    n = a.shape[0]
    rho, eps = float(g["rho"]), float(g["epsilon"])
    o_colptr, o_rows, o_cent = oracle.centrality_block(a, rho, eps)
    monkeypatch.setenv("ARCTE_HIP_CONTRIB_ENTRIES", str(3 * n))
    with _native.Context.from_adjacency(a.indptr, a.indices, a.data) as ctx:
        ctx.run_centrality(rho, eps)
        colptr, rows = ctx.fetch()
        cent = ctx.centrality()
        st = ctx.stats()
    assert st["launches"] > 3
    assert np.array_equal(colptr, o_colptr)
    for k in range(colptr.size - 1):
        assert np.array_equal(np.sort(rows[colptr[k]:colptr[k + 1]]), o_rows[o_colptr[k]:o_colptr[k + 1]])
    assert np.array_equal(cent, o_cent) and np.array_equal(cent, g["centrality"])
    monkeypatch.delenv("ARCTE_HIP_CONTRIB_ENTRIES")
    total = np.zeros(n)
    with _native.Context.from_adjacency(a.indptr, a.indices, a.data) as ctx:
        for lo, hi in ((0, n // 3), (n // 3, n // 3), (n // 3, n)):
            ctx.run_centrality(rho, eps, lo, hi)
            bc, br, bcent = oracle.centrality_block(a, rho, eps, lo, hi)
            c2, r2 = ctx.fetch()
            assert np.array_equal(c2, bc)
            assert np.array_equal(ctx.centrality(), bcent)
            total += ctx.centrality()
    np.testing.assert_allclose(total, g["centrality"], rtol=1e-13, atol=0)


def test_nodes_without_out_edges():
    """arcte.pyx:210: centrality 1.0 for nodes that were no seeds (the reference itself raises there; see the oracle)."""
    import scipy.sparse as sparse
    from reveal_graph_embedding_amd.embedding.arcte.cython_opt.arcte import arcte_and_centrality
    g = load_centrality("ba300")
    a = sparse.lil_matrix(g["adjacency"])
    a[5, :] = 0
    a[17, :] = 0
    a = sparse.csr_matrix(a)
    a.eliminate_zeros()
    f, c = arcte_and_centrality(a, 0.1, 1e-4)
    fo, co = oracle.arcte_and_centrality(a, 0.1, 1e-4)
    assert c[5] == 1.0 and c[17] == 1.0
    assert np.array_equal(c, co)
    assert_close_sparse(f, fo, 1e-12)
