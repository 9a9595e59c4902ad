"""The line state of the propagation kernel (csrc/arcte_lines.hpp) must be invisible in the results.

Nodes are named by rank inside the kernel; ranks below K keep their value in LDS, ranks below 8 M in strided 64-byte
lines whose touched-bits are an LDS bitmap (first touch = blind whole-line write), the ranks beyond in lines whose
touched-bits live in global memory (region B); pushed nodes move to a compact {r, s} array.  Whatever the split --
everything on chip, nothing on chip, region B forced onto a 300-node graph -- communities, push counts and work
counters equal the oracle's, for the three push flavours and for arcte_and_centrality."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle
from test_hot_table import sorted_rows

pytestmark = pytest.mark.gpu

GRAPHS = ["ba300", "grid25", "corner", "weighted", "selfloop", "directed", "rmat2000", "ws1000", "ba1500"]
FLAVOURS = [oracle.ARCTE, oracle.PAGERANK, oracle.LAZY_PAGERANK]
# (values of the LDS level, lines with touched-bits in LDS): nothing on chip and every rank >= 512 in region B; a handful
# of on-chip values inside the tiles; a mid-size split; the defaults (a graph this small lives on chip entirely)
SPLITS = [(0, 64), (4, 64), (24, 128), (-1, None)]


def run(g, hot, lines_lds, variant, monkeypatch, **kw):
    from reveal_graph_embedding_amd import _native
    monkeypatch.setenv("ARCTE_HIP_HOT", str(hot))
    monkeypatch.delenv("ARCTE_HIP_STATE", raising=False)
    if lines_lds is None:
        monkeypatch.delenv("ARCTE_HIP_LINES_LDS", raising=False)
    else:
        monkeypatch.setenv("ARCTE_HIP_LINES_LDS", str(lines_lds))
    w = g["w"]
    rho = g["rho"]
    with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"], **kw) as ctx:
        ctx.run_seeds(g["all_seeds"], (rho * 0.5) / (1 - 0.5 * rho) if variant == oracle.LAZY_PAGERANK else rho, g["epsilon"],
                      variant=variant, laziness_factor=0.5)
        colptr, rows, nop = ctx.fetch(want_nop=True)
        st = ctx.stats()
        info = ctx.state_info()
    return colptr, rows, nop, st, info


@pytest.mark.parametrize("name", GRAPHS)
@pytest.mark.parametrize("variant", FLAVOURS)
def test_every_split_matches_the_oracle(name, variant, monkeypatch):
    g = load_golden(name)
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(g["w"], g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"],
                                                        g["epsilon"], want_stats=True, variant=variant)
    for hot, lines_lds in SPLITS:
        colptr, rows, nop, st, info = run(g, hot, lines_lds, variant, monkeypatch)
        tag = "%d on-chip values, %s lines in LDS" % (hot, lines_lds)
        assert info["line_state"] == 1, tag
        if lines_lds is not None and g["n"] > 8 * lines_lds:
            assert info["lines_region_b"] > 0, tag
        assert np.array_equal(colptr, o_colptr), tag
        assert np.array_equal(nop, o_nop), tag
        assert np.array_equal(sorted_rows(colptr, rows), o_rows), tag
        assert [st["pushes"], st["edges"], st["enqueues"], st["support"]] == list(o_stats), tag
        # every traversed edge is exactly one update of one of the four kinds
        assert info["lds_updates"] + info["blind_line_writes"] + info["line_read_modify_writes"] + info["pushed_node_updates"] == st["edges"], tag
        if hot == 0:
            assert info["lds_updates"] == 0 and info["blind_line_writes"] > 0, tag


def test_small_capacities_grow_and_rerun(monkeypatch):
    """A pushed-state array, a candidate list and a ring that are too small flag the seed; the host grows them by four
    and runs it again: nothing is dropped."""
    g = load_golden("rmat2000")
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(g["w"], g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"],
                                                  want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_PUSHED", "64")
    monkeypatch.setenv("ARCTE_HIP_CANDIDATES", "64")
    colptr, rows, nop, st, info = run(g, 4, 64, oracle.ARCTE, monkeypatch, queue_capacity=64)
    assert st["reruns"] > 0 and st["launches"] > 1
    assert info["pushed_capacity"] > 64 or info["candidate_capacity"] > 64
    assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
    assert np.array_equal(sorted_rows(colptr, rows), o_rows)


def test_centrality_with_region_b(monkeypatch):
    """arcte_and_centrality's seed loop (every node a seed, the whole support a candidate) through every level."""
    from reveal_graph_embedding_amd import _native
    from test_centrality_weighting_cpu import load_centrality
    g = load_centrality("ba300")
    a = g["adjacency"]
    out = []
    for hot, lines_lds in ((-1, None), (0, 64), (8, 64)):
        monkeypatch.setenv("ARCTE_HIP_HOT", str(hot))
        if lines_lds is None:
            monkeypatch.delenv("ARCTE_HIP_LINES_LDS", raising=False)
        else:
            monkeypatch.setenv("ARCTE_HIP_LINES_LDS", str(lines_lds))
        with _native.Context.from_adjacency(a.indptr, a.indices, a.data) as ctx:
            ctx.run_centrality(float(g["rho"]), float(g["epsilon"]))
            colptr, rows = ctx.fetch()
            out.append((colptr, sorted_rows(colptr, rows), ctx.centrality()))
    np.testing.assert_array_equal(out[0][2], g["centrality"])            # the reference's own vector, bit for bit
    for colptr, rows, cent in out[1:]:
        assert np.array_equal(colptr, out[0][0]) and np.array_equal(rows, out[0][1])
        assert np.array_equal(cent, out[0][2])


def test_consecutive_contexts_get_the_same_slots():
    """Buffers parked in the process-wide cache count as free when the slot count is chosen (round-2 advisor finding)."""
    from reveal_graph_embedding_amd import _native
    g = load_golden("rmat2000")
    w = g["w"]
    slots = []
    for _ in range(3):
        with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
            ctx.run_seeds(g["all_seeds"][:50], g["rho"], g["epsilon"])
            slots.append(ctx.info()["slots"])
    assert slots[0] == slots[1] == slots[2]


@pytest.mark.parametrize("poison", [255, 165])
def test_slot_memory_content_is_irrelevant(poison, monkeypatch):
    """Nothing is cleared between seeds or contexts: a value is only read after a bitmap said its line was written by
    this seed.  So the slot memory may start as any garbage (ARCTE_HIP_POISON fills it with a byte: 0xFF = NaNs)."""
    g = load_golden("rmat2000")
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(g["w"], g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"],
                                                  want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_POISON", str(poison))
    for hot, lines_lds in ((0, 64), (16, 64), (-1, None)):
        colptr, rows, nop, st, info = run(g, hot, lines_lds, oracle.ARCTE, monkeypatch)
        assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
        assert np.array_equal(sorted_rows(colptr, rows), o_rows)


def test_rows_staged_through_lds_give_the_same_results(monkeypatch):
    """ARCTE_HIP_STAGE_ROWS=1 (A/B of the north_star's "rows staged through LDS"): global_load_lds into a ring of five
    stages instead of VGPR stages; long rows only.  Same results."""
    g = load_golden("rmat2000")
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(g["w"], g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"],
                                                  want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_STAGE_ROWS", "1")
    for hot, lines_lds in ((0, 64), (16, 64), (-1, None)):
        colptr, rows, nop, st, info = run(g, hot, lines_lds, oracle.ARCTE, monkeypatch)
        assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
        assert np.array_equal(sorted_rows(colptr, rows), o_rows)


def test_placement_draw_is_invisible_in_the_results(monkeypatch):
    """A context that is large enough probes candidate allocations of its slot memory and keeps the fastest (the losers are
    parked until arcte_hip_trim()); forced onto a small graph here: same results, the draws are reported."""
    from reveal_graph_embedding_amd import _native
    g = load_golden("rmat2000")
    w = g["w"]
    with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
        assert ctx.placement_info() == (-1, [])                      # too small to care by default
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"], want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_MIN_NODES", "1")
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_MIN_MB", "1")
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_TRIES", "4")
    monkeypatch.setenv("ARCTE_HIP_SLOT_SPREAD_MB", "0")              # the packed layout (hot blocks and region B apart)
    try:
        with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
            kept, rates = ctx.placement_info()
            assert 1 <= len(rates) <= 4 and 0 <= kept < len(rates) and all(r > 0 for r in rates)
            ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
            colptr, rows, nop = ctx.fetch(want_nop=True)
        assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
        assert np.array_equal(sorted_rows(colptr, rows), o_rows)
    finally:
        _native.trim()


@pytest.mark.parametrize("lines_lds", [0, 64])
def test_slots_spread_over_one_allocation(lines_lds, monkeypatch):
    """Large contexts lay their slots out with a fixed stride inside ONE allocation, region B's values behind the slot's hot
    block and unused bytes up to the next slot (csrc/arcte_hip.hip, setup_lines); forced onto a small graph here, with and
    without a region B, under the placement draw: same results."""
    from reveal_graph_embedding_amd import _native
    g = load_golden("rmat2000")
    w = g["w"]
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"], want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_MIN_NODES", "1")
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_MIN_MB", "1")
    monkeypatch.setenv("ARCTE_HIP_SPREAD_TRIES", "2")
    monkeypatch.setenv("ARCTE_HIP_SLOT_SPREAD_MB", "4")
    if lines_lds:
        monkeypatch.setenv("ARCTE_HIP_LINES_LDS", str(lines_lds))
        monkeypatch.setenv("ARCTE_HIP_HOT", "0")
    try:
        with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
            state = ctx.state_info()
            assert state["slot_bytes"] == ctx.info()["slots"] * (4 << 20)          # one slot every 4 MB
            assert (state["lines_region_b"] > 0) == bool(lines_lds)
            kept, rates = ctx.placement_info()
            assert 1 <= len(rates) <= 2 and 0 <= kept < len(rates)
            ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
            colptr, rows, nop = ctx.fetch(want_nop=True)
        assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
        assert np.array_equal(sorted_rows(colptr, rows), o_rows)
    finally:
        _native.trim()


def test_spread_slots_fall_back_to_packed_ones(monkeypatch):
    """When the one large allocation of the spread layout cannot be had (test hook), the context packs its slots: same results."""
    from reveal_graph_embedding_amd import _native
    g = load_golden("rmat2000")
    w = g["w"]
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"], want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_MIN_NODES", "1")
    monkeypatch.setenv("ARCTE_HIP_SLOT_SPREAD_MB", "4")
    monkeypatch.setenv("ARCTE_HIP_TEST_SPREAD_FAILS", "1")
    try:
        with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
            assert ctx.state_info()["slot_bytes"] < ctx.info()["slots"] * (4 << 20)          # packed
            ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
            colptr, rows, nop = ctx.fetch(want_nop=True)
        assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
        assert np.array_equal(sorted_rows(colptr, rows), o_rows)
    finally:
        _native.trim()


@pytest.mark.parametrize("name", ["ba300", "weighted", "selfloop", "directed", "rmat2000", "ws1000", "ba1500"])
@pytest.mark.parametrize("variant", FLAVOURS)
def test_indirect_region_b_matches_the_oracle(name, variant, monkeypatch):
    """Region B's lines INDIRECT (arcte_lines.hpp, IND: an 8-byte entry per line + a pool of lines instead of eight float64 per
    line; the default from 16 MB of region B per slot, forced here): same communities, push counts and work counters."""
    g = load_golden(name)
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(g["w"], g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"],
                                                        g["epsilon"], want_stats=True, variant=variant)
    monkeypatch.setenv("ARCTE_HIP_B_INDIRECT", "1")
    for hot, lines_lds, poison in ((0, 64, None), (4, 64, 255), (24, 128, 165)):
        if poison is None:
            monkeypatch.delenv("ARCTE_HIP_POISON", raising=False)
        else:
            monkeypatch.setenv("ARCTE_HIP_POISON", str(poison))
        colptr, rows, nop, st, info = run(g, hot, lines_lds, variant, monkeypatch)
        tag = "%d on-chip values, %s lines in LDS, poison %s" % (hot, lines_lds, poison)
        if g["n"] > 8 * lines_lds:
            assert info["lines_region_b"] > 0, tag
        assert np.array_equal(colptr, o_colptr), tag
        assert np.array_equal(nop, o_nop), tag
        assert np.array_equal(sorted_rows(colptr, rows), o_rows), tag
        assert [st["pushes"], st["edges"], st["enqueues"], st["support"]] == list(o_stats), tag


def test_indirect_region_b_pool_grows(monkeypatch):
    """A pool that is too small flags the seed; the host makes it four times as large and runs the seed again."""
    g = load_golden("rmat2000")
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(g["w"], g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"],
                                                  want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_B_INDIRECT", "1")
    monkeypatch.setenv("ARCTE_HIP_B_POOL", "64")
    colptr, rows, nop, st, info = run(g, 0, 64, oracle.ARCTE, monkeypatch)
    assert st["reruns"] > 0 and st["launches"] > 1
    assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
    assert np.array_equal(sorted_rows(colptr, rows), o_rows)


def test_growing_slot_memory_gives_up_slots_half_a_wavefront_per_cu_at_a_time(monkeypatch):
    """Round 4: when the grown slot memory does not fit, the context keeps as many slots as do -- fewer by half a wavefront per
    CU at a time, not half of them (the first whole launch of the 8M-node graph lost 2 048 of 4 096 slots that way) -- the
    wavefronts per CU (and with them the LDS share and the kernel build) follow, and the results stay the oracle's."""
    from reveal_graph_embedding_amd import _native
    g = load_golden("rmat2000")
    w = g["w"]
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"], want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_B_INDIRECT", "1")
    monkeypatch.setenv("ARCTE_HIP_B_POOL", "64")
    monkeypatch.setenv("ARCTE_HIP_HOT", "0")
    monkeypatch.setenv("ARCTE_HIP_LINES_LDS", "64")
    # what the slots take once every pool has grown to what the graph's heaviest seed claims, with room for all of them
    with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
        ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
        assert ctx.stats()["reruns"] > 0
        all_slots = ctx.info()["slots"]
        grown_bytes = ctx.state_info()["slot_bytes"]
    # ... and on a device (as the hook shows it) that has 80 % of that: three quarters of it may go to slots
    room_mb = max(1, int(grown_bytes * 0.8) >> 20)
    monkeypatch.setenv("ARCTE_HIP_TEST_GROW_FREE_MB", str(room_mb))
    with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
        before = ctx.info()
        cus = before["compute_units"]
        assert before["slots"] == all_slots
        ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
        assert ctx.stats()["reruns"] > 0
        after = ctx.info()
        slot_bytes = ctx.state_info()["slot_bytes"]
        budget = (room_mb << 20) // 4 * 3
        assert slot_bytes <= budget
        assert all_slots // 2 < after["slots"] < all_slots, (all_slots, after["slots"])
        assert (all_slots - after["slots"]) % max(1, cus // 2) == 0
        # one more step of half a wavefront per CU would not have fitted
        assert (after["slots"] + cus // 2) * (slot_bytes // after["slots"]) > budget
        assert after["waves_per_cu"] == -(-after["slots"] // cus)
        colptr, rows, nop = ctx.fetch(want_nop=True)
        assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
        assert np.array_equal(sorted_rows(colptr, rows), o_rows)
        # ... and a second run on the smaller context
        ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
        assert ctx.stats()["reruns"] == 0
        colptr, rows, nop = ctx.fetch(want_nop=True)
        assert np.array_equal(colptr, o_colptr) and np.array_equal(sorted_rows(colptr, rows), o_rows)


@pytest.mark.parametrize("name", ["rmat2000", "ws1000"])
def test_indirect_region_b_centrality(name, monkeypatch):
    """arcte_and_centrality's seed loop through indirect lines: the reference's own vector, bit for bit."""
    from reveal_graph_embedding_amd import _native
    from test_centrality_weighting_cpu import load_centrality
    g = load_centrality(name)
    a = g["adjacency"]
    monkeypatch.setenv("ARCTE_HIP_B_INDIRECT", "1")
    monkeypatch.setenv("ARCTE_HIP_HOT", "0")
    monkeypatch.setenv("ARCTE_HIP_LINES_LDS", "64")
    with _native.Context.from_adjacency(a.indptr, a.indices, a.data) as ctx:
        assert ctx.state_info()["lines_region_b"] > 0
        ctx.run_centrality(float(g["rho"]), float(g["epsilon"]))
        np.testing.assert_array_equal(ctx.centrality(), g["centrality"])


def test_at_most_one_draw_loser_stays_allocated(monkeypatch):
    """Round-3 review / advisor: the placement draw kept EVERY loser allocated (155 GB for a 1M-node graph) and counted them
    as free when it sized slot memory.  Now at most ARCTE_HIP_PARK_MAX (1) loser per device stays, losers of another shape
    are returned before a draw and when a context's slot memory grows, and the library reports what it holds."""
    from reveal_graph_embedding_amd import _native
    g = load_golden("rmat2000")
    w = g["w"]
    o_colptr, o_rows, _, o_nop, _ = oracle.worker(w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"], g["epsilon"], want_stats=True)
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_MIN_NODES", "1")
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_MIN_MB", "1")
    monkeypatch.setenv("ARCTE_HIP_PLACEMENT_TRIES", "4")
    monkeypatch.setenv("ARCTE_HIP_SLOT_SPREAD_MB", "0")
    _native.trim()
    try:
        assert _native.memory_info()["parked_bytes"] == 0
        with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
            kept, rates = ctx.placement_info()
            held = _native.memory_info()
            slot_bytes = ctx.state_info()["slot_bytes"]
            assert held["parked_bytes"] <= slot_bytes                                   # one candidate at most
            assert (held["parked_bytes"] > 0) == (len(rates) > 1)
            # the slot memory grows (a pushed-state array of 64 entries overflows): the old shape's loser goes back
            monkeypatch.setenv("ARCTE_HIP_PUSHED", "64")
        with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
            before = ctx.state_info()["slot_bytes"]
            ctx.run_seeds(g["all_seeds"], g["rho"], g["epsilon"])
            assert ctx.stats()["reruns"] > 0
            after = ctx.state_info()["slot_bytes"]
            assert after >= before
            held = _native.memory_info()
            assert held["parked_bytes"] <= after                                        # of the NEW shape, one at most
            colptr, rows, nop = ctx.fetch(want_nop=True)
        assert np.array_equal(colptr, o_colptr) and np.array_equal(nop, o_nop)
        assert np.array_equal(sorted_rows(colptr, rows), o_rows)
        monkeypatch.delenv("ARCTE_HIP_PUSHED")
        monkeypatch.setenv("ARCTE_HIP_PARK_MAX", "0")
        _native.trim()
        with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
            assert _native.memory_info()["parked_bytes"] == 0
    finally:
        _native.trim()
    assert _native.memory_info()["parked_bytes"] == 0 and _native.memory_info()["cached_bytes"] == 0
