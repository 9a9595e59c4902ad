"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's fixtures.

Bit-exact bar: sparsity patterns, push counts, and -- because the kernels keep the reference's
operation order with FMA contraction off -- the float64 values of s and r as well.  The one
tolerance is the effective epsilon, whose two logarithms come from the device libm (rtol below).
"""
import hashlib
import os

import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import GOLDEN, assert_same_sparse, load_golden
from oracle import oracle

from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.embedding.arcte.arcte import arcte, arcte_worker, calculate_epsilon_effective
from reveal_graph_embedding_amd.eps_randomwalk.push import cumulative_pagerank_difference_limit_push
from reveal_graph_embedding_amd.eps_randomwalk.similarity import (
    fast_approximate_cumulative_pagerank_difference, similarity_slice_cython)
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.synthetic import rmat_graph

pytestmark = pytest.mark.gpu

EPS_RTOL = 4e-15   # a few ulp: eps*log(1+d)/log(1+mean) with device log (<= 1 ulp each)


def ctx_of(g, **kw):
    w = g["w"]
    return _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"], **kw)


def arrays_of_arrays(w):
    n = w.shape[0]
    a_i = np.ndarray(n, dtype=np.ndarray)
    w_i = np.ndarray(n, dtype=np.ndarray)
    for i in range(n):
        a_i[i] = w.indices[w.indptr[i]:w.indptr[i + 1]]
        w_i[i] = w.data[w.indptr[i]:w.indptr[i + 1]]
    return w_i, a_i


def test_device_visible():
    assert _native.device_count() >= 1


def test_epsilon_effective_all_seeds(golden):
    with ctx_of(golden) as ctx:
        got = ctx.epsilon_effective(golden["all_seeds"], golden["epsilon"])
    want = golden["all_eps_eff"]
    np.testing.assert_allclose(got, want, rtol=EPS_RTOL, atol=0)
    print("eps_eff bit-identical: %d / %d" % (int((got == want).sum()), want.size))


def test_epsilon_effective_scalar_api(golden):
    w, od = golden["w"], golden["out_degree"]
    for k, s in enumerate(golden["seeds"][:4]):
        e = calculate_epsilon_effective(golden["rho"], golden["epsilon"], od[s],
                                        od[w.indices[w.indptr[s]:w.indptr[s + 1]]], od.mean())
        np.testing.assert_allclose(e, golden["eps_eff"][k], rtol=EPS_RTOL, atol=0)


def test_single_push_bit_exact(golden):
    w = golden["w"]
    u = int(golden["push_node"])
    s, r = golden["push_s_in"].copy(), golden["push_r_in"].copy()
    cumulative_pagerank_difference_limit_push(s, r, w.data[w.indptr[u]:w.indptr[u + 1]],
                                              w.indices[w.indptr[u]:w.indptr[u + 1]], u, golden["rho"])
    assert np.array_equal(s, golden["push_s_out"])
    assert np.array_equal(r, golden["push_r_out"])


@pytest.mark.parametrize("flavour", ["", "raw_"])
def test_similarity_slices_bit_exact(golden, flavour):
    """fast_approximate_cumulative_pagerank_difference with the reference's own call shape."""
    n = golden["n"]
    w_i, a_i = arrays_of_arrays(golden["w"])
    assert similarity_slice_cython is fast_approximate_cumulative_pagerank_difference
    for k, seed in enumerate(golden["seeds"]):
        eps = golden["eps_eff"][k] if flavour == "" else golden["epsilon"]
        s = np.zeros(n)
        r = np.zeros(n)
        nop = fast_approximate_cumulative_pagerank_difference(s, r, w_i, a_i, golden["out_degree"],
                                                              golden["in_degree"], seed, golden["rho"], eps)
        assert nop == golden[flavour + "nop"][k]
        for vec, tag in ((s, "s"), (r, "r")):
            lo, hi = golden[flavour + tag + "_ptr"][k], golden[flavour + tag + "_ptr"][k + 1]
            nz = np.nonzero(vec)[0]
            assert np.array_equal(nz, golden[flavour + tag + "_idx"][lo:hi])
            assert np.array_equal(vec[nz], golden[flavour + tag + "_val"][lo:hi])


def test_similarity_slice_continues_from_caller_state(golden):
    """s and r are the caller's: a non-zero start state must be carried like the reference does."""
    n = golden["n"]
    rng = np.random.default_rng(5)
    s0 = rng.random(n) * 1e-3
    r0 = rng.random(n) * 1e-4
    seed = int(golden["seeds"][0])
    s_o, r_o = s0.copy(), r0.copy()
    nop_o = oracle.similarity(golden["w"], golden["in_degree"], seed, golden["rho"], 1e-3, s_o, r_o)
    with ctx_of(golden, n_slots=4) as ctx:
        s_h, r_h = s0.copy(), r0.copy()
        nop_h = ctx.similarity_slice(seed, golden["rho"], 1e-3, s_h, r_h)
    assert nop_h == nop_o
    assert np.array_equal(s_h, s_o) and np.array_equal(r_h, r_o)


def test_worker_matches_fixture_and_oracle(golden):
    w = golden["w"]
    got = arcte_worker(golden["seeds"], w.indices, w.indptr, w.data, golden["out_degree"], golden["in_degree"],
                       golden["rho"], golden["epsilon"])
    assert_same_sparse(got, golden["worker"])
    with ctx_of(golden) as ctx:
        ctx.run_seeds(golden["seeds"], golden["rho"], golden["epsilon"])
        colptr, rows, eps, nop = ctx.fetch(want_eps=True, want_nop=True)
        st = ctx.stats()
    assert np.array_equal(nop, golden["nop"])
    np.testing.assert_allclose(eps, golden["eps_eff"], rtol=EPS_RTOL, atol=0)
    o_colptr, o_rows, _, _, o_stats = oracle.worker(w, golden["out_degree"], golden["in_degree"], golden["seeds"],
                                                    golden["rho"], golden["epsilon"], want_stats=True)
    assert np.array_equal(colptr, o_colptr)
    for k in range(golden["seeds"].size):
        assert np.array_equal(np.sort(rows[colptr[k]:colptr[k + 1]]), o_rows[o_colptr[k]:o_colptr[k + 1]])
    assert [st["pushes"], st["edges"], st["enqueues"], st["support"]] == list(o_stats)


def test_arcte_full_matches_reference_fixture(golden):
    got = arcte(golden["adjacency"], golden["rho"], golden["epsilon"], 1)
    assert got.shape == (golden["n"], 2 * golden["n"])
    assert_same_sparse(got, golden["feat1"])
    assert_same_sparse(got, golden["feat3"])


def test_raw_epsilon_mode_against_oracle():
    g = load_golden("ba1500")
    seeds = g["all_seeds"][::7]
    with ctx_of(g) as ctx:
        ctx.run_seeds(seeds, 0.15, 3e-5, use_effective_epsilon=False)
        colptr, rows, eps, nop = ctx.fetch(want_eps=True, want_nop=True)
    assert np.all(eps == 3e-5)
    n = g["n"]
    for k, seed in enumerate(seeds[:40]):
        s, r = np.zeros(n), np.zeros(n)
        assert oracle.similarity(g["w"], g["in_degree"], seed, 0.15, 3e-5, s, r) == nop[k]


@pytest.mark.parametrize("qcap", [64, 256])
def test_queue_overflow_is_detected_and_recovered(qcap):
    g = load_golden("rmat2000")
    seeds = g["all_seeds"]
    with ctx_of(g, n_slots=64, queue_capacity=qcap) as ctx:
        assert ctx.info()["queue_capacity"] == qcap
        ctx.run_seeds(seeds, g["rho"], g["epsilon"])
        colptr, rows = ctx.fetch()
        st = ctx.stats()
        assert ctx.info()["queue_capacity"] > qcap
    assert st["reruns"] > 0 and st["launches"] > 1
    o_colptr, o_rows = oracle.worker(g["w"], g["out_degree"], g["in_degree"], seeds, g["rho"], g["epsilon"])
    assert np.array_equal(colptr, o_colptr)
    for k in range(seeds.size):
        assert np.array_equal(np.sort(rows[colptr[k]:colptr[k + 1]]), o_rows[o_colptr[k]:o_colptr[k + 1]])


def test_output_arena_overflow_is_detected_and_recovered(monkeypatch):
    g = load_golden("rmat2000")
    seeds = g["all_seeds"]
    monkeypatch.setenv("ARCTE_HIP_ARENA_ROWS", "20000")
    with ctx_of(g, n_slots=64) as ctx:
        ctx.run_seeds(seeds, g["rho"], g["epsilon"])
        colptr, rows = ctx.fetch()
        st = ctx.stats()
    assert st["launches"] > 1
    o_colptr, o_rows = oracle.worker(g["w"], g["out_degree"], g["in_degree"], seeds, g["rho"], g["epsilon"])
    assert np.array_equal(colptr, o_colptr)
    for k in range(seeds.size):
        assert np.array_equal(np.sort(rows[colptr[k]:colptr[k + 1]]), o_rows[o_colptr[k]:o_colptr[k + 1]])


def test_empty_seed_list_and_reuse_of_context():
    g = load_golden("ba300")
    with ctx_of(g) as ctx:
        ctx.run_seeds(np.zeros(0, dtype=np.int64), 0.1, 1e-5)
        colptr, rows = ctx.fetch()
        assert colptr.tolist() == [0] and rows.size == 0
        for _ in range(2):     # slots must come back all-zero after every run
            ctx.run_seeds(g["seeds"], g["rho"], g["epsilon"])
            colptr, rows, nop = ctx.fetch(want_nop=True)
            assert np.array_equal(nop, g["nop"])


def test_c_abi_error_codes():
    g = load_golden("ba300")
    with ctx_of(g) as ctx:
        with pytest.raises(_native.ArcteHipError) as e:
            ctx.fetch()
        assert e.value.code == -4
        with pytest.raises(_native.ArcteHipError) as e:
            ctx.run_seeds(np.array([10 ** 9]), 0.1, 1e-5)
        assert e.value.code == -1
    with pytest.raises(_native.ArcteHipError) as e:
        _native.Context(g["w"].indptr, g["w"].indices, g["w"].data, g["out_degree"], g["in_degree"], device=99)
    assert e.value.code == -2


def test_zero_weight_edge_is_reported_not_miscomputed():
    """Star whose hub->leaf-1 transition weight is an explicit zero: leaf 1 can never receive mass, so
    the hub's closed neighbourhood is not inside its support and the reference mis-indexes (arcte.py:359)."""
    a = sparse.csr_matrix(np.array([[0, 1, 1, 1], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]], dtype=np.float64))
    w, od, idg = get_natural_random_walk_matrix(a)
    w.data[0] = 0.0
    with _native.Context(w.indptr, w.indices, w.data, od, idg) as ctx:
        with pytest.raises(_native.ArcteHipError) as e:
            ctx.run_seeds(np.array([0]), 0.1, 1e-5)
        assert e.value.code == -5
        ctx.run_seeds(np.array([1, 2]), 0.1, 1e-5)     # the context stays usable
    with pytest.raises(RuntimeError):
        oracle.worker(w, od, idg, np.array([0]), 0.1, 1e-5)


@pytest.mark.parametrize("state", ["default", "region B dense", "region B indirect", "packed words escape"])
def test_config1_rmat_all_seeds_matches_reference_hash(state, monkeypatch):
    """All 63 070 seeds of the config-1 graph: SHA-256 of the canonical CSR against the reference's own run -- with the
    defaults (everything in region A), and with the LDS bitmap cut to 1 024 lines, so that 92 % of the nodes live in region B
    (dense lines, then indirect ones: hub rows of 15 000 edges put several lanes of one 64-edge step into one line of region
    B, the case the claims must order), and with packed row words that leave 3 bits for the in_degree."""
    if state.startswith("region B"):
        monkeypatch.setenv("ARCTE_HIP_LINES_LDS", "1024")
        monkeypatch.setenv("ARCTE_HIP_HOT", "64")
        monkeypatch.setenv("ARCTE_HIP_B_INDIRECT", "1" if state.endswith("indirect") else "0")
    if state.startswith("packed"):
        monkeypatch.setenv("ARCTE_HIP_PACK_RANK_BITS", "29")
    z = np.load(os.path.join(GOLDEN, "rmat100k_summary.npz"))
    adjacency = rmat_graph(100000, 2000000, seed=0)
    f = arcte(adjacency, float(z["rho"]), float(z["epsilon"]), 1)
    f.sum_duplicates()
    f.sort_indices()
    assert f.nnz == int(z["nnz"])
    local = sparse.csc_matrix(f[:, 100000:])
    assert np.array_equal(np.diff(local.indptr), z["local_col_counts"])
    h = hashlib.sha256()
    h.update(f.indptr.astype(np.int64).tobytes())
    h.update(f.indices.astype(np.int64).tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), z["sha256"])


def test_full_size_1m_graph_properties_and_sampled_oracle_parity(rmat_1m):
    """BASELINE.json configs[2] at full size (R-MAT 1M nodes / 50M sampled edges): size-independent
    properties on a seed shard plus exact agreement with the oracle on a random sample of it."""
    adjacency = rmat_1m
    w, od, idg = get_natural_random_walk_matrix(adjacency)
    from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes
    seeds = seed_nodes(adjacency)
    assert seeds.size == 651465
    shard = seeds[5::64]                      # ~10k seeds spread over the whole degree range
    deg = np.diff(w.indptr)
    results = []
    for slots in (0, 512):                    # result must not depend on how many seeds are in flight
        with _native.Context(w.indptr, w.indices, w.data, od, idg, n_slots=slots) as ctx:
            ctx.run_seeds(shard, 0.1, 1e-5)
            results.append(ctx.fetch(want_nop=True) + (ctx.stats(),))
    (colptr, rows, nop, st), (colptr2, rows2, nop2, st2) = results
    assert np.array_equal(colptr, colptr2) and np.array_equal(nop, nop2)
    assert [st[k] for k in ("pushes", "edges", "enqueues", "support")] == \
           [st2[k] for k in ("pushes", "edges", "enqueues", "support")]
    sizes = np.diff(colptr)
    emitted = sizes > 0
    assert np.all(sizes[emitted] > deg[shard][emitted] + 1)          # arcte.py:370
    assert np.all(nop >= 1)
    for k in np.flatnonzero(emitted)[::97]:
        members = np.sort(rows[colptr[k]:colptr[k + 1]])
        assert np.array_equal(members, np.sort(rows2[colptr2[k]:colptr2[k + 1]]))
        assert np.unique(members).size == members.size
        base = np.append(w.indices[w.indptr[shard[k]]:w.indptr[shard[k] + 1]], shard[k])
        assert np.all(np.isin(base, members))                         # community contains N[seed]
    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(shard.size, size=1500, replace=False))
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(w, od, idg, shard[pick], 0.1, 1e-5,
                                                  threads=oracle.lib().oracle_max_threads(), want_stats=True)
    assert np.array_equal(o_nop, nop[pick])
    assert np.array_equal(np.diff(o_colptr), sizes[pick])
    for j, k in enumerate(pick):
        assert np.array_equal(np.sort(rows[colptr[k]:colptr[k + 1]]), o_rows[o_colptr[j]:o_colptr[j + 1]])


def test_seed_without_out_neighbours_is_rejected_like_the_reference():
    """calculate_epsilon_effective on an empty neighbourhood raises in the reference (np.max of an empty
    array, arcte.py:39); directed input is the only way to get there (the pipeline symmetrises)."""
    a = sparse.csr_matrix(np.array([[0, 1, 1], [1, 0, 1], [0, 0, 0]], dtype=np.float64))   # node 2: in-count 2, no out-edges
    w, od, idg = get_natural_random_walk_matrix(a)
    with _native.Context(w.indptr, w.indices, w.data, od, idg) as ctx:
        with pytest.raises(_native.ArcteHipError) as e:
            ctx.run_seeds(np.array([2]), 0.1, 1e-5)
        assert e.value.code == -5
        ctx.run_seeds(np.array([2]), 0.1, 1e-5, use_effective_epsilon=False)    # raw epsilon needs no neighbours
        colptr, rows, nop = ctx.fetch(want_nop=True)
        assert nop.tolist() == [1] and rows.size == 0


def test_degenerate_graphs_and_seed_lists():
    """Ragged / tiny inputs: single node, all-isolated graph, duplicate and unsorted seeds, a seed list that
    is not in degree order, more slots than seeds."""
    one = sparse.csr_matrix((1, 1), dtype=np.float64)
    w, od, idg = get_natural_random_walk_matrix(one)
    assert arcte(one, 0.1, 1e-5, 1).toarray().tolist() == [[1.0, 0.0]]
    iso = sparse.csr_matrix((5, 5), dtype=np.float64)
    f = arcte(iso, 0.1, 1e-5, 1)
    assert f.shape == (5, 10) and np.array_equal(f.toarray()[:, :5], np.eye(5)) and f[:, 5:].nnz == 0
    g = load_golden("ba300")
    seeds = np.array([7, 3, 250, 3, 7, 120, 0], dtype=np.int64)      # duplicates, unsorted
    got = arcte_worker(seeds, g["w"].indices, g["w"].indptr, g["w"].data, g["out_degree"], g["in_degree"],
                       g["rho"], g["epsilon"])
    with ctx_of(g) as ctx:
        ctx.run_seeds(seeds, g["rho"], g["epsilon"])
        colptr, rows = ctx.fetch()
    o_colptr, o_rows = oracle.worker(g["w"], g["out_degree"], g["in_degree"], seeds, g["rho"], g["epsilon"])
    assert np.array_equal(colptr, o_colptr)
    for k in range(seeds.size):
        assert np.array_equal(np.sort(rows[colptr[k]:colptr[k + 1]]), o_rows[o_colptr[k]:o_colptr[k + 1]])
    # duplicate seeds write the same column twice: COO -> CSR sums them, exactly like the reference's coo_matrix
    want = oracle.worker_matrix(g["w"], g["out_degree"], g["in_degree"], seeds, g["rho"], g["epsilon"])
    assert_same_sparse(got, want)


def test_in_process_multi_worker_path(monkeypatch):
    """arcte() with several workers in one process (one host thread + one context per worker, seeds dealt
    round-robin, results summed: arcte.py:650-673).  Both workers are placed on GPU 0 here."""
    monkeypatch.setenv("ARCTE_HIP_DEVICES", "0,0,0")
    for name in ("rmat2000", "selfloop", "corner"):
        g = load_golden(name)
        got = arcte(g["adjacency"], g["rho"], g["epsilon"], 3)
        assert_same_sparse(got, g["feat3"])
    monkeypatch.setenv("ARCTE_HIP_DEVICES", "0")
    g = load_golden("ba300")
    assert_same_sparse(arcte(g["adjacency"], g["rho"], g["epsilon"], None), g["feat1"])


def test_host_assembly_fallback_when_the_device_sort_is_too_small(monkeypatch):
    monkeypatch.setenv("ARCTE_HIP_MAX_SORT_KEYS", "1000")
    for name in ("rmat2000", "selfloop"):
        g = load_golden(name)
        assert_same_sparse(arcte(g["adjacency"], g["rho"], g["epsilon"], 1), g["feat1"])
        got = arcte_worker(g["seeds"], g["w"].indices, g["w"].indptr, g["w"].data, g["out_degree"], g["in_degree"],
                           g["rho"], g["epsilon"])
        assert_same_sparse(got, g["worker"])


def test_epsilon_effective_on_big_weighted_rows():
    """Rows beyond 4096 neighbours take the workgroup-per-row kernel (16 subtrees of numpy's pairwise
    recursion); with random weights the summation ORDER shows in the last bits, so bit-identity with the
    oracle on (nearly) every seed proves the order."""
    a = rmat_graph(100000, 2000000, seed=0)
    rng = np.random.default_rng(5)
    u = sparse.triu(a, k=1).tocoo()
    wts = rng.uniform(0.05, 7.0, size=u.nnz)
    wa = sparse.coo_matrix((wts, (u.row, u.col)), shape=a.shape)
    wa = sparse.csr_matrix(wa + wa.T)
    w, od, idg = get_natural_random_walk_matrix(wa)
    deg = np.diff(w.indptr)
    big = np.flatnonzero(deg >= 4096)
    assert big.size >= 5
    seeds = np.concatenate([big, rng.choice(np.flatnonzero((deg > 1) & (deg < 4096)), size=300, replace=False)])
    with _native.Context(w.indptr, w.indices, w.data, od, idg, n_slots=64) as ctx:
        got = ctx.epsilon_effective(seeds, 1e-5)
    want = np.array([oracle.calculate_epsilon_effective(0.1, 1e-5, od[s], od[w.indices[w.indptr[s]:w.indptr[s + 1]]])
                     for s in seeds])
    np.testing.assert_allclose(got, want, rtol=EPS_RTOL, atol=0)
    # Every value that is not bit-identical must explain itself: the neighbour mean is reproduced exactly (numpy's
    # pairwise order) and the two clipping bounds are correctly rounded IEEE operations, so a difference can only
    # come from the two logarithms (device libm vs glibc, <= 1 ulp each) -- i.e. it must sit on the unclipped /
    # lower-clipped branch, where the result is eps*log(1+d)/log(1+mean) [or its average with the lower bound], and be
    # a couple of ulp at most.  (Pattern identity then rests on no r/in_degree landing inside that band.)
    ulp = np.abs(got.view(np.int64) - want.view(np.int64))
    for k in np.flatnonzero(ulp):
        s_ = seeds[k]
        nd = od[w.indices[w.indptr[s_]:w.indptr[s_ + 1]]]
        upper = np.max(1.0 / (od[s_] * nd))
        assert want[k] != upper, "an exactly-computed clipping bound differs: not a logarithm effect"
        print("seed %d (row %d): %d ulp, value %.17g on the logarithm branch (upper bound %.3g)" % (s_, nd.size, ulp[k], want[k], upper))
    assert ulp.max() <= 2
    print("big rows: %d, bit-identical eps: %d; all seeds bit-identical: %d / %d, max %d ulp"
          % (big.size, int((ulp[:big.size] == 0).sum()), int((ulp == 0).sum()), seeds.size, int(ulp.max())))


def test_bench_two_ranks_gather_equals_reference_hash(tmp_path):
    """bench.py's own N > 1 step (shard, run, variable-length gather to rank 0, merge) with two ranks sharing the GPU
    over gloo -- the one thing it cannot exercise on a one-GPU box is the RCCL transport itself.  Every seed of the
    config-1 graph: what rank 0 gathered equals the sum of what the ranks emitted, and the merged n x 2n matrix is
    the reference's own (SHA-256 of the canonical CSR from its 8-process run)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    z = np.load(os.path.join(GOLDEN, "rmat100k_summary.npz"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--nodes", "100000",
           "--edges", "2000000", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--verify",
           "--rho", str(float(z["rho"])), "--epsilon", str(float(z["epsilon"]))]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and cfg["seeds_per_step"] == 63070
    assert cfg["gathered_rows_rank0"] == cfg["emitted_rows_all_ranks"] == int(z["nnz"]) - 100000 - 3554220
    assert cfg["merged_sha256"] == bytes(z["sha256"]).hex()


def test_config0_ba20000_rho1e3_matches_reference_hash():
    """BASELINE.json configs[0], stand-in no. 2: BA(20 000, 10), rho = 1e-3, eps = 1e-5 -- every seed, against the
    SHA-256 of the reference's own 8-process run."""
    from test_oracle_golden import _summary_hash, load_ba20000
    z, a = load_ba20000()
    f, digest = _summary_hash(arcte(a, float(z["rho"]), float(z["epsilon"]), 1))
    assert f.nnz == int(z["nnz"])
    assert np.array_equal(np.diff(sparse.csc_matrix(f[:, a.shape[0]:]).indptr), z["local_col_counts"])
    assert np.array_equal(digest, z["sha256"])


def test_slot_buffers_are_reused_across_contexts_and_trimmed():
    """A destroyed context leaves its big slot buffers in the library's cache (freeing and re-allocating tens of GB
    costs seconds); the next context of the same shape takes them back CLEARED: same results, and trim() empties it."""
    adjacency = rmat_graph(100000, 2000000, seed=0)
    results = []
    for _ in range(2):
        with _native.Context.from_adjacency(adjacency.indptr, adjacency.indices, adjacency.data) as ctx:
            seeds = np.sort(ctx.seed_list())[::7]
            ctx.run_seeds(seeds, 0.1, 1e-5)
            colptr, rows, nop = ctx.fetch(want_nop=True)
            results.append((colptr, rows, nop))
    assert all(np.array_equal(a, b) for a, b in zip(results[0], results[1]))
    _native.trim()
    with _native.Context.from_adjacency(adjacency.indptr, adjacency.indices, adjacency.data) as ctx:
        ctx.run_seeds(seeds, 0.1, 1e-5)
        colptr, rows, nop = ctx.fetch(want_nop=True)
    assert all(np.array_equal(a, b) for a, b in zip(results[0], (colptr, rows, nop)))
    _native.trim()


@pytest.mark.gpu
def test_fastest_context_returns_a_working_context():
    """Several contexts alive at once, one kept: the kept one computes what any context computes."""
    adjacency = rmat_graph(20000, 200000, seed=1)

    def make():
        return _native.Context.from_adjacency(adjacency.indptr, adjacency.indices, adjacency.data, n_slots=64)

    def calibrate(ctx):
        ctx.run_seeds(ctx.seed_list()[::4], 0.1, 1e-5)
        return ctx.timing()["push_ms"]

    ctx, results = _native.fastest_context(make, calibrate, tries=3)
    with ctx:
        assert len(results) == 3 and all(r > 0 for r in results)
        seeds = np.sort(ctx.seed_list())
        ctx.run_seeds(seeds, 0.1, 1e-5)
        got = ctx.fetch(want_nop=True)
    with make() as ref:
        ref.run_seeds(seeds, 0.1, 1e-5)
        want = ref.fetch(want_nop=True)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_appended_runs_equal_one_run(golden):
    """arcte_hip_run_seeds_append: a seed list run in parts on one context (the reference sums the chunk matrices of a worker,
    arcte.py:384-386) gives every seed the column, push count and counters of a single run."""
    seeds = golden["all_seeds"]
    if seeds.size < 6:
        pytest.skip("too few seeds to split")
    with ctx_of(golden) as ctx:
        ctx.run_seeds(seeds, golden["rho"], golden["epsilon"])
        colptr, rows, nop = ctx.fetch(want_nop=False) + (None,)
        st = ctx.stats()
        ref = {int(s): np.sort(rows[colptr[k]:colptr[k + 1]]) for k, s in enumerate(seeds)}
        ref_csr = ctx.fetch_csr(True)
        parts = [seeds[0::3], seeds[1::3], seeds[2::3]]
        ctx.run_seeds(parts[0], golden["rho"], golden["epsilon"])
        ctx.run_seeds(parts[1], golden["rho"], golden["epsilon"], append=True)
        ctx.run_seeds(parts[2], golden["rho"], golden["epsilon"], append=True)
        colptr2, rows2 = ctx.fetch()
        order = np.concatenate([parts[2], parts[1], parts[0]])          # an appended run lists its own seeds first
        assert colptr2.size == seeds.size + 1
        for k, s in enumerate(order):
            assert np.array_equal(np.sort(rows2[colptr2[k]:colptr2[k + 1]]), ref[int(s)]), int(s)
        st2 = ctx.stats()
        for key in ("pushes", "edges", "enqueues", "support"):
            assert st2[key] == st[key], key
        got_csr = ctx.fetch_csr(True)
        assert np.array_equal(got_csr[0], ref_csr[0]) and np.array_equal(got_csr[1], ref_csr[1])
    with ctx_of(golden) as ctx:
        with pytest.raises(_native.ArcteHipError) as e:
            ctx.run_seeds(seeds, golden["rho"], golden["epsilon"], append=True)         # nothing to append to
        assert e.value.code == -4


def test_arcte_in_two_launches_with_background_host_arrays(golden, monkeypatch):
    """Large runs of arcte() / arcte_worker() go in two launches (a sizing part, then the rest while the host arrays are
    faulted in): forced onto the fixtures, the matrices must be the reference's."""
    from reveal_graph_embedding_amd.embedding.arcte import arcte as A
    monkeypatch.setattr(A, "_SPLIT_MIN_SEEDS", 4)
    got = A.arcte(golden["adjacency"], golden["rho"], golden["epsilon"], 1)
    assert_same_sparse(got, golden["feat1"])
    w = golden["w"]
    got = A.arcte_worker(golden["seeds"], w.indices, w.indptr, w.data, golden["out_degree"], golden["in_degree"],
                         golden["rho"], golden["epsilon"])
    assert_same_sparse(got, golden["worker"])
    # the host-assembly fallback (more entries than the device assembly takes) with the parts' seed order
    monkeypatch.setenv("ARCTE_HIP_MAX_SORT_KEYS", "10")
    got = A.arcte(golden["adjacency"], golden["rho"], golden["epsilon"], 1)
    assert_same_sparse(got, golden["feat1"])
