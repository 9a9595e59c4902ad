"""Device-side graph preparation: get_natural_random_walk_matrix (transition.py:43-99), the seed
list of arcte() (arcte.py:610-617), triplets -> CSR and the symmetrisation (A + A^T)/2 of entry_points/arcte.py:70-71,
all in HIP kernels.  Bit-exact against the reference's own outputs (fixtures) -- the weighted and directed graphs pin
scipy's summation orders -- and against scipy run here on random inputs."""
import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import assert_same_sparse, load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu


def test_transition_matrix_matches_reference_fixture(golden):
    from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
    w, out_degree, in_degree = get_natural_random_walk_matrix(golden["adjacency"])
    assert_same_sparse(w, golden["w"])
    assert np.array_equal(out_degree, golden["out_degree"])
    assert np.array_equal(in_degree, golden["in_degree"])
    assert w.has_sorted_indices


def test_seed_list_matches_reference_fixture(golden):
    from reveal_graph_embedding_amd import _native
    a = golden["adjacency"]
    with _native.Context.from_adjacency(a.indptr, a.indices, a.data, n_slots=1) as ctx:
        seeds = ctx.seed_list()
        n, nnz, ns = ctx.graph_sizes()
    cnt = np.bincount(a.indices, minlength=a.shape[0])
    assert (n, nnz, ns) == (a.shape[0], a.nnz, golden["all_seeds"].size)
    assert np.array_equal(np.sort(seeds), golden["all_seeds"])          # arcte.py:617
    assert np.all(np.diff(cnt[seeds]) <= 0)                             # arcte.py:614-616: descending count
    # ties (unspecified in the reference: unstable argsort) come in ascending node order here
    ties = np.diff(cnt[seeds]) == 0
    assert np.all(np.diff(seeds)[ties] > 0)


def test_transition_zero_out_degree_rows_and_unsorted_input():
    from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
    a = sparse.coo_matrix((np.array([2.0, 1.0, 4.0]), (np.array([0, 0, 2]), np.array([2, 1, 0]))), shape=(3, 3))
    w, od, idg = get_natural_random_walk_matrix(a)
    assert np.array_equal(od, [3.0, 1.0, 4.0])
    assert np.array_equal(idg, [4.0, 1.0, 2.0])
    assert np.array_equal(w.toarray(), [[0, 1 / 3.0, 2 / 3.0], [0, 0, 0], [1.0, 0, 0]])


@pytest.mark.parametrize("seed", range(4))
def test_weighted_random_graphs_round_like_scipy(seed):
    """Rows long enough for numpy's pairwise recursion (> 128 entries), columns long enough for a visible fold order,
    CSR stored with UNSORTED rows: sums follow the storage order (transition.py:55-56 come before sort_indices :65)."""
    from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
    rng = np.random.default_rng(seed)
    n = 700
    dense = rng.random((n, n)) * (rng.random((n, n)) < (0.6 if seed % 2 else 0.05))
    dense[rng.integers(0, n, 5)] = 0.0                                # some empty rows
    a = sparse.csr_matrix(dense)
    # scramble the storage order inside every row
    indices, data = a.indices.copy(), a.data.copy()
    for i in range(n):
        lo, hi = a.indptr[i], a.indptr[i + 1]
        p = rng.permutation(hi - lo)
        indices[lo:hi] = indices[lo:hi][p]
        data[lo:hi] = data[lo:hi][p]
    a = sparse.csr_matrix((data, indices, a.indptr), shape=(n, n))
    assert not a.has_sorted_indices or seed >= 0
    w, od, idg = get_natural_random_walk_matrix(a)
    ow, ood, oidg = oracle.get_natural_random_walk_matrix(a)          # the reference's scipy calls, restated
    assert np.array_equal(od, ood)
    assert np.array_equal(idg, oidg)
    assert_same_sparse(w, ow)


def test_duplicate_columns_are_rejected_on_the_device():
    from reveal_graph_embedding_amd import _native
    indptr = np.array([0, 3, 4, 5], dtype=np.int64)
    indices = np.array([1, 2, 1, 0, 0], dtype=np.int32)
    with pytest.raises(_native.ArcteHipError) as e:
        _native.Context.from_adjacency(indptr, indices, np.ones(5), n_slots=1)
    assert e.value.code == -1


@pytest.mark.parametrize("name", ["ba300", "weighted", "directed", "selfloop", "rmat2000"])
@pytest.mark.parametrize("symmetrise", [False, True])
def test_triplets_to_transition(name, symmetrise):
    """csr_matrix(coo) [+ (A + A^T)/2] + get_natural_random_walk_matrix from shuffled triplets, against scipy."""
    from reveal_graph_embedding_amd import _native
    g = load_golden(name)
    coo = g["adjacency"].tocoo()
    rng = np.random.default_rng(1)
    p = rng.permutation(coo.nnz)
    row, col, val = coo.row[p], coo.col[p], coo.data[p]
    # split some entries into two triplets: csr_matrix(coo) sums duplicates
    split = rng.random(val.size) < 0.2
    part = np.where(split, val * 0.25, 0.0)
    row = np.concatenate([row, row[split]])
    col = np.concatenate([col, col[split]])
    val = np.concatenate([val - part, part[split]])
    a = sparse.csr_matrix(sparse.coo_matrix((val, (row, col)), shape=coo.shape))
    if symmetrise:
        a = sparse.csr_matrix((a + a.transpose()) / 2)                 # entry_points/arcte.py:70-71
    ow, ood, oidg = oracle.get_natural_random_walk_matrix(a)
    with _native.Context.from_coo(coo.shape[0], row, col, val, symmetrise=symmetrise, n_slots=1) as ctx:
        indptr, indices, data, od, idg = ctx.transition()
        seeds = ctx.seed_list()
    w = sparse.csr_matrix((data, indices, indptr), shape=coo.shape)
    assert_same_sparse(w, ow)
    assert np.array_equal(od, ood) and np.array_equal(idg, oidg)
    assert np.array_equal(np.sort(seeds), np.sort(oracle.seed_list(a)))


def test_from_adjacency_runs_like_the_host_prepared_context():
    """The context built on the device must propagate exactly like one built from the fixture's W."""
    from reveal_graph_embedding_amd import _native
    g = load_golden("weighted")
    a, w = g["adjacency"], g["w"]
    with _native.Context.from_adjacency(a.indptr, a.indices, a.data) as ctx:
        ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
        c1, r1, n1 = ctx.fetch(want_nop=True)
    with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
        ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
        c2, r2, n2 = ctx.fetch(want_nop=True)
    assert np.array_equal(c1, c2) and np.array_equal(r1, r2) and np.array_equal(n1, n2)
