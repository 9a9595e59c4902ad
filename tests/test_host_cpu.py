"""CPU-side checks (no GPU): host logic and the C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import ROOT

from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.embedding.arcte.arcte import parallel_chunks, roundrobin_chunks, seed_nodes
from reveal_graph_embedding_amd.synthetic import rmat_graph


def header_symbols():
    text = open(os.path.join(ROOT, "include", "arcte_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arcte_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_native.SIGNATURES) == names
    assert _native.lib().arcte_hip_abi_version() == 9


def test_no_gpu_means_loud_failure_not_fallback():
    if _native.device_count() > 0:
        pytest.skip("a GPU is visible")
    from reveal_graph_embedding_amd.embedding.arcte.arcte import arcte
    with pytest.raises(_native.ArcteHipError):
        arcte(sparse.eye(8, format="csr"), 0.1, 1e-5)
    with pytest.raises(_native.ArcteHipError):
        _native.Context(np.array([0, 1, 2]), np.array([1, 0]), np.ones(2), np.ones(2), np.ones(2))


def test_argument_validation_happens_before_any_device_work():
    bad = np.array([0, 3, 2], dtype=np.int64)
    with pytest.raises(_native.ArcteHipError) as e:
        _native.Context(bad, np.array([1, 0, 1]), np.ones(3), np.ones(2), np.ones(2))
    assert e.value.code == -1
    with pytest.raises(_native.ArcteHipError) as e:
        _native.Context(np.array([0, 1, 2]), np.array([1, 7]), np.ones(2), np.ones(2), np.ones(2))
    assert e.value.code == -1


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "reveal-graph-embedding_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(base, f)).read()
                for needle in ("import oracle", "from oracle", "liboracle", "oracle/", "oracle."):
                    assert needle not in text, (needle, os.path.join(base, f))


def test_chunkers_match_reference_semantics():
    l = list(range(10))
    assert list(parallel_chunks(l, 3)) == [[0, 3, 6, 9], [1, 4, 7], [2, 5, 8]]
    assert roundrobin_chunks([1, 2], 4, 3) is None


def test_seed_list_is_count_descending_and_excludes_degree_le_1(golden):
    a = golden["adjacency"]
    seeds = seed_nodes(a)
    cnt = np.bincount(a.indices, minlength=a.shape[0])
    assert np.array_equal(np.sort(seeds), golden["all_seeds"])
    assert np.all(np.diff(cnt[seeds]) <= 0)
    assert np.all(cnt[seeds] > 1)


def test_rmat_generator_reproduces_survey_counts():
    a = rmat_graph(100000, 2000000, seed=0)
    deg = np.diff(a.indptr)
    assert a.nnz == 3554220
    assert int((deg == 0).sum()) == 25004
    assert int((deg > 1).sum()) == 63070
    assert int(deg.max()) == 14891
    assert (a != a.T).nnz == 0


def test_background_ones_cover_every_entry():
    from reveal_graph_embedding_amd.embedding.arcte.arcte import _OnesInBackground
    for size in (0, 1, 5, (1 << 22) * 3 + 17):
        out = _OnesInBackground(size, threads=4).result()
        assert out.dtype == np.float64 and out.shape == (size,)
        assert np.all(out == 1.0)


def test_fastest_context_keeps_the_best_draw_and_closes_the_rest():
    from reveal_graph_embedding_amd import _native

    class Fake:
        def __init__(self, speed):
            self.speed, self.closed = speed, False

        def close(self):
            self.closed = True

    speeds = iter([3.0, 1.0, 2.0])
    made = []

    def make():
        made.append(Fake(next(speeds)))
        return made[-1]

    best, results = _native.fastest_context(make, lambda c: c.speed, tries=3)
    assert results == [3.0, 1.0, 2.0] and best is made[1]
    assert [c.closed for c in made] == [True, False, True]
    # a draw that fails for lack of memory ends the drawing; the first failure is the caller's
    def make_then_fail():
        if made_b:
            raise _native.ArcteHipError(-3, "out of memory")
        made_b.append(Fake(5.0))
        return made_b[-1]
    made_b = []
    best, results = _native.fastest_context(make_then_fail, lambda c: c.speed, tries=3)
    assert best is made_b[0] and results == [5.0] and not best.closed


def test_sizes_beyond_int32_are_refused_by_name_before_the_cast():
    """The reference's shared path carries int64 indices (transition.py:82-87); here node ids are int32.  An id that does not
    fit must raise a ValueError naming the limit instead of being wrapped by the dtype cast."""
    big = np.array([1, 2 ** 31 + 5], dtype=np.int64)
    with pytest.raises(ValueError, match="int32"):
        _native.Context.from_adjacency(np.array([0, 1, 2], dtype=np.int64), big, np.ones(2))
    with pytest.raises(ValueError, match="int32"):
        _native.Context.from_coo(2, big, np.array([0, 1]), np.ones(2))
    with pytest.raises(ValueError, match="int32"):
        _native.Context(np.array([0, 1, 2]), big, np.ones(2), np.ones(2), np.ones(2))
    with pytest.raises(ValueError, match="2\\^31"):
        _native._ids32(np.zeros(3, dtype=np.int64), 2 ** 31, 2 ** 31, "indices")
    # what fits goes through to the library (which has no device here, or builds the context)
    assert _native._ids32(np.array([0, 5], dtype=np.int64), 6, 2 ** 31, "indices").dtype == np.int32
