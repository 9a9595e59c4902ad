"""The LDS-resident hot table of the push kernels must be invisible in the results.

The state of the highest-degree nodes lives in LDS (one value per node while r == s, moved to the pushed-state array --
line state -- or to the dense HBM entry -- dense state -- when the node is pushed).  Whatever the table size -- off, a
handful of nodes (hot and cold targets mixed inside one tile), every node on chip -- communities, push counts and work
counters must equal the oracle's, for the three push flavours, both state layouts (ARCTE_HIP_STATE) and both arithmetic
types (float32 against its own table-off run).  tests/test_line_state.py holds the line state's own levels."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu

GRAPHS = ["ba300", "grid25", "corner", "weighted", "selfloop", "directed", "rmat2000", "ws1000"]
FLAVOURS = [oracle.ARCTE, oracle.PAGERANK, oracle.LAZY_PAGERANK]


def run(g, cap, variant, monkeypatch, float32=False, warm=None, state=None, **kw):
    from reveal_graph_embedding_amd import _native
    monkeypatch.setenv("ARCTE_HIP_HOT", str(cap))
    if state is None:
        monkeypatch.delenv("ARCTE_HIP_STATE", raising=False)
    else:
        monkeypatch.setenv("ARCTE_HIP_STATE", state)
    if warm is None:
        monkeypatch.delenv("ARCTE_HIP_WARM", raising=False)
    else:
        monkeypatch.setenv("ARCTE_HIP_WARM", str(warm))
    w = g["w"]
    rho = g["rho"]
    with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"], **kw) as ctx:
        if float32:
            ctx.set_float32(True)
        ctx.run_seeds(g["all_seeds"], (rho * 0.5) / (1 - 0.5 * rho) if variant == oracle.LAZY_PAGERANK else rho, g["epsilon"],
                      variant=variant, laziness_factor=0.5)
        colptr, rows, nop = ctx.fetch(want_nop=True)
        st = ctx.stats()
        if state is not None and not float32:
            assert ctx.state_info()["line_state"] == (1 if state == "lines" else 0)
    return colptr, rows, nop, [st["pushes"], st["edges"], st["enqueues"], st["support"]]


def sorted_rows(colptr, rows):
    seg = np.repeat(np.arange(colptr.size - 1), np.diff(colptr))
    return rows[np.lexsort((rows, seg))]


@pytest.mark.parametrize("state", ["lines", "dense"])
@pytest.mark.parametrize("name", GRAPHS)
@pytest.mark.parametrize("variant", FLAVOURS)
def test_every_table_size_matches_the_oracle(name, variant, state, monkeypatch):
    g = load_golden(name)
    w = g["w"]
    o_colptr, o_rows, _, o_nop, o_stats = oracle.worker(w, g["out_degree"], g["in_degree"], g["all_seeds"], g["rho"],
                                                        g["epsilon"], want_stats=True, variant=variant)
    # (LDS table size, warm-table end rank): off; LDS only; LDS + warm + dense mixed inside one tile; no LDS share but
    # warm; everything in LDS / warm (the defaults on a graph this small)
    for cap, warm in ((0, 0), (4, 0), (4, 24), (32, 100), (4, None), (-1, None)):
        colptr, rows, nop, stats = run(g, cap, variant, monkeypatch, warm=warm, state=state)
        tag = "%s state, hot cap %d, warm end %s" % (state, cap, warm)
        assert np.array_equal(colptr, o_colptr), tag
        assert np.array_equal(nop, o_nop), tag
        assert np.array_equal(sorted_rows(colptr, rows), o_rows), tag
        assert stats == list(o_stats), tag


@pytest.mark.parametrize("shape", [dict(n_slots=4), dict(n_slots=512), dict()])
def test_launch_shapes(shape, monkeypatch):
    """One wavefront per CU with the whole LDS, many, and the default; two wavefronts per workgroup."""
    g = load_golden("rmat2000")
    ref = run(g, 0, oracle.ARCTE, monkeypatch, **shape)
    for wpb, state in (("1", "lines"), ("1", "dense"), ("2", "dense"), ("4", "dense")):
        monkeypatch.setenv("ARCTE_HIP_WAVES_PER_BLOCK", wpb)
        for tiles in ("1", "2", "4"):
            monkeypatch.setenv("ARCTE_HIP_TILES", tiles)
            got = run(g, -1, oracle.ARCTE, monkeypatch, state=state, **shape)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2]) and got[3] == ref[3]
            assert np.array_equal(sorted_rows(got[0], got[1]), sorted_rows(ref[0], ref[1]))


@pytest.mark.parametrize("variant", FLAVOURS)
def test_float32_table_is_invisible_too(variant, monkeypatch):
    g = load_golden("rmat2000")
    ref = run(g, 0, variant, monkeypatch, float32=True)
    for cap, warm in ((16, 0), (16, 200), (-1, None)):
        got = run(g, cap, variant, monkeypatch, float32=True, warm=warm)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2]) and got[3] == ref[3]
        assert np.array_equal(sorted_rows(got[0], got[1]), sorted_rows(ref[0], ref[1]))


def test_duplicate_columns_are_rejected():
    """A column stored twice in a row would make two lanes of one push race: EINVAL at create."""
    from reveal_graph_embedding_amd import _native
    indptr = np.array([0, 3, 4, 5], dtype=np.int64)
    indices = np.array([1, 2, 1, 0, 0], dtype=np.int32)
    with pytest.raises(_native.ArcteHipError) as e:
        _native.Context(indptr, indices, np.ones(5), np.ones(3), np.ones(3))
    assert e.value.code == -1 and "twice" in str(e.value)


@pytest.mark.parametrize("variant", FLAVOURS)
def test_narrow_and_packed_rows_are_invisible(variant, monkeypatch):
    """Unweighted graphs stream 8 bytes per edge (one weight per row, float32 in_degrees that widen back exactly) or, packed
    (the default of the line state), ONE 32-bit word: rank + integer in_degree, the in_degrees that do not fit looked up in
    a float32 table by rank; weighted graphs can do neither.  Same results every way -- also when the word is left so few
    bits for the in_degree (test hook) that most lanes take the table."""
    from reveal_graph_embedding_amd import _native
    monkeypatch.delenv("ARCTE_HIP_PACK_RANK_BITS", raising=False)
    for name, expect in (("ba300", 2), ("rmat2000", 2), ("selfloop", 2), ("weighted", 0)):
        g = load_golden(name)
        w = g["w"]
        monkeypatch.delenv("ARCTE_HIP_PACK", raising=False)
        monkeypatch.delenv("ARCTE_HIP_NARROW", raising=False)
        with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
            assert ctx.info()["narrow_rows"] == expect, name
        monkeypatch.setenv("ARCTE_HIP_NARROW", "0")
        ref = run(g, -1, variant, monkeypatch)
        monkeypatch.setenv("ARCTE_HIP_NARROW", "1")
        for pack, bits in (("0", None), ("1", None), ("1", "29"), ("1", "30")):
            monkeypatch.setenv("ARCTE_HIP_PACK", pack)
            if bits is None:
                monkeypatch.delenv("ARCTE_HIP_PACK_RANK_BITS", raising=False)
            else:
                monkeypatch.setenv("ARCTE_HIP_PACK_RANK_BITS", bits)
            if expect:
                with _native.Context(w.indptr, w.indices, w.data, g["out_degree"], g["in_degree"]) as ctx:
                    assert ctx.info()["narrow_rows"] == (2 if pack == "1" else 1), (name, pack, bits)
            got = run(g, -1, variant, monkeypatch)
            tag = (name, pack, bits)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2]) and got[3] == ref[3], tag
            assert np.array_equal(sorted_rows(got[0], got[1]), sorted_rows(ref[0], ref[1])), tag
