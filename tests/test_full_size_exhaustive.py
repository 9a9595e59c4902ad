"""BASELINE.json configs[2] EXHAUSTIVELY: every one of the 651 465 seeds of the 1M-node / 50M-edge R-MAT graph, the
production path (graph preparation on the device, effective epsilons, k_arcte_lines, extraction) against the CPU oracle
(OpenMP over all host cores): community sizes, push counts, community members and the four work counters identical.
Round 3 ran this as a tool (tools/full_parity_1m.py); as a test it is part of what the driver records."""
import hashlib
import os
import time

import numpy as np
import pytest
import scipy.sparse as sparse

from oracle import oracle

pytestmark = pytest.mark.gpu


def test_every_seed_of_the_1m_graph_equals_the_oracle(rmat_1m):
    from reveal_graph_embedding_amd import _native
    a = rmat_1m
    t = time.time()
    # the graph is prepared ON THE DEVICE (transition matrix, degrees, seed list) and handed to the oracle from there: the
    # oracle's own preparation is the scipy path the device version is held to in tests/test_device_prepare.py
    with _native.Context.from_adjacency(a.indptr, a.indices, a.data) as ctx:
        seeds = np.sort(ctx.seed_list())
        assert seeds.size == 651465
        indptr, indices, data, od, idg = ctx.transition()
        w = sparse.csr_matrix((data, indices, indptr), shape=a.shape)
        ctx.run_seeds(seeds, 0.1, 1e-5)
        colptr, rows, nop = ctx.fetch(want_nop=True)
        st = ctx.stats()
        info = ctx.state_info()
    t_hip = time.time() - t
    assert info["line_state"] == 1
    t = time.time()
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(w, od, idg, seeds, 0.1, 1e-5, threads=oracle.lib().oracle_max_threads(),
                                                        want_stats=True)
    t_oracle = time.time() - t
    assert np.array_equal(colptr, o_colptr), "community sizes differ"
    assert np.array_equal(nop, o_nop), "push counts differ"
    assert [st[k] for k in ("pushes", "edges", "enqueues", "support")] == list(o_stats)
    # members: the oracle's segments are sorted; sort the device's by (seed, member)
    seg = np.repeat(np.arange(seeds.size, dtype=np.int64), np.diff(colptr))
    order = np.lexsort((rows, seg))
    assert np.array_equal(rows[order], o_rows), "community members differ"
    print("IDENTICAL: %d seeds, %d emitted rows, %d pushes, %d traversed edges; device %.1f s, oracle %.1f s on %d threads; sha256(rows) %s" % (
        seeds.size, rows.size, st["pushes"], st["edges"], t_hip, t_oracle, oracle.lib().oracle_max_threads(),
        hashlib.sha256(o_rows.tobytes()).hexdigest()[:16]))


def test_heaviest_and_sampled_seeds_of_the_8m_graph_equal_the_oracle():
    """The configuration the library picks for LARGE graphs (round 4), on the graph it was measured on (R-MAT 8M nodes / 100M sampled
    edges, SURVEY.md 8(d)'s next size): packed rows, region B's lines indirect behind one claiming atomic, sixteen wavefronts per CU on
    the 128-VGPR build, 4 KB of touched-bits -- and pools that GROW under the heaviest seeds (every claim takes a pool line), with the
    seeds that overflowed run again.  The 128 heaviest seeds (smallest effective epsilon; the oracle spends seconds on each) + 6 000
    drawn at random: community sizes, push
    counts, members and work counters identical to the CPU oracle."""
    from reveal_graph_embedding_amd import _native
    from reveal_graph_embedding_amd.synthetic import rmat_graph
    a = rmat_graph(8000000, 100000000, seed=0)
    t = time.time()
    with _native.Context.from_adjacency(a.indptr, a.indices, a.data) as ctx:
        info, state = ctx.info(), ctx.state_info()
        assert state["line_state"] == 1 and state["lines_region_b"] > 0
        if "ARCTE_HIP_WAVES_PER_CU" not in os.environ and "ARCTE_HIP_B_INDIRECT" not in os.environ:
            assert state["region_b_indirect"] == 1 and info["narrow_rows"] == 2 and info["waves_per_cu"] > 12, (info, state)
        all_seeds = ctx.seed_list()
        eps = ctx.epsilon_effective(all_seeds, 1e-5)
        heaviest = all_seeds[np.argsort(eps, kind="stable")[:128]]
        drawn = np.random.default_rng(8).choice(all_seeds, size=6000, replace=False)
        seeds = np.unique(np.concatenate([heaviest, drawn]))
        indptr, indices, data, od, idg = ctx.transition()
        w = sparse.csr_matrix((data, indices, indptr), shape=a.shape)
        ctx.run_seeds(seeds, 0.1, 1e-5)
        colptr, rows, nop = ctx.fetch(want_nop=True)
        st = ctx.stats()
        grown = ctx.state_info()
    t_hip = time.time() - t
    if state["region_b_indirect"] == 1 and "ARCTE_HIP_B_POOL" not in os.environ:
        assert st["reruns"] > 0 and grown["region_b_pool_lines"] > state["region_b_pool_lines"], (st, grown)
    t = time.time()
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(w, od, idg, seeds, 0.1, 1e-5, threads=oracle.lib().oracle_max_threads(),
                                                        want_stats=True)
    t_oracle = time.time() - t
    assert np.array_equal(colptr, o_colptr), "community sizes differ"
    assert np.array_equal(nop, o_nop), "push counts differ"
    assert [st[k] for k in ("pushes", "edges", "enqueues", "support")] == list(o_stats)
    seg = np.repeat(np.arange(seeds.size, dtype=np.int64), np.diff(colptr))
    order = np.lexsort((rows, seg))
    assert np.array_equal(rows[order], o_rows), "community members differ"
    print("IDENTICAL: %d seeds of the 8M graph (%d re-run after their pool grew to %d lines), %d emitted rows, %d traversed edges; "
          "device %.1f s, oracle %.1f s" % (seeds.size, st["reruns"], grown["region_b_pool_lines"], rows.size, st["edges"], t_hip, t_oracle))
