"""BASELINE.json configs[2] EXHAUSTIVELY: every one of the 651 465 seeds of the 1M-node / 50M-edge R-MAT graph, the
production path (graph preparation on the device, effective epsilons, k_arcte_lines, extraction) against the CPU oracle
(OpenMP over all host cores): community sizes, push counts, community members and the four work counters identical.
Round 3 ran this as a tool (tools/full_parity_1m.py); as a test it is part of what the driver records."""
import hashlib
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
