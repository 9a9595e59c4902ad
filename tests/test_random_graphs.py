"""Randomised parity sweep on the GPU: many small graphs of different shape (density, weights, direction,
self-loops, isolated nodes, hubs wider than two 64-edge tiles), all three push flavours, random rho / epsilon,
HIP against the oracle: community sets, push counts, work counters, and bit-exact similarity slices."""
import numpy as np
import pytest
import scipy.sparse as sparse

from oracle import oracle
from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix

pytestmark = pytest.mark.gpu


def random_graph(rng, case):
    n = int(rng.integers(3, 420))
    kind = case % 6
    density = float(rng.choice([0.01, 0.03, 0.1, 0.4]))
    m = max(2, int(density * n * n / 2))
    i = rng.integers(0, n, size=m)
    j = rng.integers(0, n, size=m)
    w = rng.uniform(0.1, 3.0, size=m) if kind in (1, 4) else np.ones(m)
    a = sparse.coo_matrix((w, (i, j)), shape=(n, n)).tocsr()
    a.sum_duplicates()
    if kind != 5:
        a = a.tolil()
        a.setdiag(0)
        a = sparse.csr_matrix(a)
        a.eliminate_zeros()
    if kind in (0, 1, 2, 5):
        a = sparse.csr_matrix(a + a.T)                     # symmetric (kind 5 keeps self-loops)
    if kind == 2 and n > 200:                              # a hub wider than two tiles
        row = np.zeros(n)
        row[rng.choice(n, size=min(n - 1, 190), replace=False)] = 1.0
        row[0] = 0.0
        a = sparse.lil_matrix(a)
        a[0, :] = row
        a[:, 0] = row.reshape(-1, 1)
        a = sparse.csr_matrix(a)
    if kind in (3, 4):                                     # directed: every node keeps at least one out-edge
        a = sparse.lil_matrix(a)
        for r in range(n):
            if a[r, :].nnz == 0:
                a[r, (r + 1) % n] = 1.0
        a = sparse.csr_matrix(a)
    return sparse.csr_matrix(a, dtype=np.float64)


@pytest.mark.parametrize("case", range(36))
def test_random_graph_parity(case):
    rng = np.random.default_rng(1000 + case)
    a = random_graph(rng, case)
    n = a.shape[0]
    w, od, idg = get_natural_random_walk_matrix(a)
    rho = float(rng.choice([0.05, 0.1, 0.2, 0.5]))
    eps = float(rng.choice([1e-3, 1e-4, 1e-5, 3e-6]))
    variant = case % 3
    run_rho = (rho * 0.5) / (1 - 0.5 * rho) if variant == 2 else rho
    out_len = np.diff(w.indptr)
    seeds = np.flatnonzero(out_len > 0)                    # every node the reference could process
    if seeds.size == 0:
        pytest.skip("no seed with out-neighbours")
    rng.shuffle(seeds)
    with _native.Context(w.indptr, w.indices, w.data, od, idg, n_slots=64,
                         queue_capacity=int(rng.choice([64, 4096]))) as ctx:
        try:
            ctx.run_seeds(seeds, run_rho, eps, variant=variant)
            hip_err = None
        except _native.ArcteHipError as e:
            hip_err = e
        try:
            o = oracle.worker(w, od, idg, seeds, rho, eps, want_stats=True, variant=variant)
            ora_err = None
        except RuntimeError as e:
            ora_err = e
        assert (hip_err is None) == (ora_err is None), (hip_err, ora_err)      # both reject the same inputs
        if hip_err is None:
            colptr, rows, eps_used, nop = ctx.fetch(want_eps=True, want_nop=True)
            st = ctx.stats()
            o_colptr, o_rows, o_eps, o_nop, o_stats = o
            np.testing.assert_allclose(eps_used, o_eps, rtol=4e-15, atol=0)
            assert np.array_equal(nop, o_nop)
            assert np.array_equal(colptr, o_colptr)
            for k in range(seeds.size):
                assert np.array_equal(np.sort(rows[colptr[k]:colptr[k + 1]]), o_rows[o_colptr[k]:o_colptr[k + 1]])
            assert [st["pushes"], st["edges"], st["enqueues"], st["support"]] == list(o_stats)
        # similarity slices with a raw epsilon, bit for bit
        for seed in seeds[:3]:
            s_h, r_h, s_o, r_o = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
            nop_h = ctx.similarity_slice(seed, run_rho, eps * 10, s_h, r_h, variant=variant)
            nop_o = oracle.similarity_variant(variant, w, idg, seed, run_rho, eps * 10, s_o, r_o, 0.5)
            assert nop_h == nop_o
            assert np.array_equal(s_h, s_o) and np.array_equal(r_h, r_o)
