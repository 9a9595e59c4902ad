"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end of oracle/arcte_oracle.c plus numpy/scipy restatements of the two
driver-level reference functions that are scipy calls rather than arithmetic:

  eps_randomwalk/transition.py:43-68   -> get_natural_random_walk_matrix
  embedding/arcte/arcte.py:591-688     -> arcte  (seed list :610-617, base block :676-683)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Parity status: pinned by tests/golden/*.npz (outputs of the reference run in
the build container, tests/golden/make_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sparse

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # ORACLE_LIB: sanitizer build
_lib = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "arcte_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        l = C.CDLL(_LIB_PATH)
        l.oracle_np_sum.restype = C.c_double
        l.oracle_np_sum.argtypes = [_f64p, C.c_int64]
        l.oracle_epsilon_effective.restype = C.c_double
        l.oracle_epsilon_effective.argtypes = [C.c_double, C.c_double, _f64p, C.c_int64]
        l.oracle_push.restype = None
        l.oracle_push.argtypes = [_f64p, _f64p, _f64p, _i32p, C.c_int64, C.c_int64, C.c_double]
        l.oracle_similarity.restype = C.c_int64
        l.oracle_similarity.argtypes = [C.c_int64, _i64p, _i32p, _f64p, _f64p, C.c_int64, C.c_double, C.c_double,
                                        _f64p, _f64p]
        l.oracle_worker.restype = C.c_int
        l.oracle_worker.argtypes = [C.c_int64, _i64p, _i32p, _f64p, _f64p, _f64p, _i64p, C.c_int64,
                                    C.c_double, C.c_double, C.c_int, _i64p, C.POINTER(C.POINTER(C.c_int32)),
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        l.oracle_push_trace.restype = C.c_int64
        l.oracle_push_trace.argtypes = [C.c_int64, _i64p, _i32p, _f64p, _f64p, _f64p, C.c_int64, C.c_double, C.c_double,
                                        _i32p, C.c_int64]
        l.oracle_worker_variant.restype = C.c_int
        l.oracle_worker_variant.argtypes = [C.c_int] + l.oracle_worker.argtypes
        l.oracle_similarity_variant.restype = C.c_int64
        l.oracle_similarity_variant.argtypes = [C.c_int, C.c_double] + l.oracle_similarity.argtypes
        l.oracle_push_variant.restype = None
        l.oracle_push_variant.argtypes = [C.c_int, C.c_double] + l.oracle_push.argtypes
        l.oracle_arcte_and_centrality.restype = C.c_int
        l.oracle_arcte_and_centrality.argtypes = [C.c_int64, _i64p, _i32p, _f64p, _f64p, C.c_double, C.c_double, C.c_int64,
                                                  C.c_int64, _i64p, C.POINTER(C.POINTER(C.c_int32)), _f64p]
        l.oracle_free.restype = None
        l.oracle_free.argtypes = [C.c_void_p]
        l.oracle_max_threads.restype = C.c_int
        _lib = l
    return _lib


def _csr_arrays(w):
    return (np.ascontiguousarray(w.indptr, dtype=np.int64),
            np.ascontiguousarray(w.indices, dtype=np.int32),
            np.ascontiguousarray(w.data, dtype=np.float64))


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().oracle_np_sum(a, a.size)


def calculate_epsilon_effective(rho, epsilon, seed_degree, neighbor_degrees, mean_degree=None):
    """arcte.py:26-50 (rho and mean_degree unused there as well)."""
    nd = np.ascontiguousarray(neighbor_degrees, dtype=np.float64)
    return lib().oracle_epsilon_effective(float(epsilon), float(seed_degree), nd, nd.size)


def cumulative_pagerank_difference_limit_push(s, r, w_i, a_i, push_node, rho):
    """push.py:41-64, in place on s, r."""
    w_i = np.ascontiguousarray(w_i, dtype=np.float64)
    a_i = np.ascontiguousarray(a_i, dtype=np.int32)
    lib().oracle_push(s, r, w_i, a_i, a_i.size, int(push_node), float(rho))


def similarity(w, in_degree, seed, rho, epsilon, s, r):
    """similarity.py:149-222 on the CSR `w`; s, r dense float64, mutated in place; returns nop."""
    indptr, indices, data = _csr_arrays(w)
    return lib().oracle_similarity(w.shape[0], indptr, indices, data,
                                   np.ascontiguousarray(in_degree, dtype=np.float64),
                                   int(seed), float(rho), float(epsilon), s, r)


ARCTE, PAGERANK, LAZY_PAGERANK = 0, 1, 2


def push_variant(variant, s, r, w_i, a_i, push_node, rho, lazy=0.5):
    """push.py:4-17 (PAGERANK), :20-38 (LAZY_PAGERANK), :41-64 (ARCTE), in place on s, r."""
    w_i = np.ascontiguousarray(w_i, dtype=np.float64)
    a_i = np.ascontiguousarray(a_i, dtype=np.int32)
    lib().oracle_push_variant(int(variant), float(lazy), s, r, w_i, a_i, a_i.size, int(push_node), float(rho))


def similarity_variant(variant, w, in_degree, seed, rho, epsilon, s, r, lazy=0.5):
    """similarity.py:11-63 (PAGERANK), :66-146 (LAZY_PAGERANK), :149-222 (ARCTE); returns nop."""
    indptr, indices, data = _csr_arrays(w)
    return lib().oracle_similarity_variant(int(variant), float(lazy), w.shape[0], indptr, indices, data,
                                           np.ascontiguousarray(in_degree, dtype=np.float64),
                                           int(seed), float(rho), float(epsilon), s, r)


def worker(w, out_degree, in_degree, seeds, rho, epsilon, threads=1, want_stats=False, variant=0):
    """arcte.py:279-388.  Returns (colptr[int64, nseeds+1], rows[int32]) and, with
    want_stats, also (eps_eff, nop, stats4 = [pushes, edges, enqueues, support])."""
    indptr, indices, data = _csr_arrays(w)
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    n = w.shape[0]
    colptr = np.zeros(seeds.size + 1, dtype=np.int64)
    rows_p = C.POINTER(C.c_int32)()
    eps_eff = np.zeros(seeds.size, dtype=np.float64)
    nop = np.zeros(seeds.size, dtype=np.int64)
    stats = np.zeros(4, dtype=np.int64)
    rc = lib().oracle_worker_variant(int(variant), n, indptr, indices, data,
                             np.ascontiguousarray(out_degree, dtype=np.float64),
                             np.ascontiguousarray(in_degree, dtype=np.float64),
                             seeds, seeds.size, float(rho), float(epsilon), int(threads),
                             colptr, C.byref(rows_p),
                             eps_eff.ctypes.data, nop.ctypes.data, stats.ctypes.data)
    total = int(colptr[-1])
    rows = np.ctypeslib.as_array(rows_p, shape=(max(total, 1),))[:total].copy()
    lib().oracle_free(rows_p)
    if rc != 0:
        raise RuntimeError("oracle_worker failed with status %d" % rc)
    if want_stats:
        return colptr, rows, eps_eff, nop, stats
    return colptr, rows


def push_trace(w, out_degree, in_degree, seed, rho, epsilon, cap=1 << 16):
    """Node ids pushed for `seed` (effective epsilon), in push order."""
    indptr, indices, data = _csr_arrays(w)
    buf = np.zeros(cap, dtype=np.int32)
    n = lib().oracle_push_trace(w.shape[0], indptr, indices, data,
                                np.ascontiguousarray(out_degree, dtype=np.float64),
                                np.ascontiguousarray(in_degree, dtype=np.float64), int(seed), float(rho), float(epsilon),
                                buf, cap)
    return buf[:min(n, cap)].copy()


def worker_matrix(w, out_degree, in_degree, seeds, rho, epsilon, threads=1, variant=0):
    """arcte_worker's return value: n x n CSR of ones, column = seed id (arcte.py:379-388)."""
    n = w.shape[0]
    seeds = np.asarray(seeds, dtype=np.int64)
    colptr, rows = worker(w, out_degree, in_degree, seeds, rho, epsilon, threads, variant=variant)
    cols = np.repeat(seeds, np.diff(colptr))
    m = sparse.coo_matrix((np.ones(rows.size, dtype=np.float64), (rows.astype(np.int64), cols)), shape=(n, n))
    return sparse.csr_matrix(m)


def get_natural_random_walk_matrix(adjacency_matrix):
    """transition.py:43-68 (make_shared only changes where the arrays live)."""
    rw = sparse.csr_matrix(adjacency_matrix, dtype=np.float64, copy=True)          # :52
    out_degree = rw.sum(axis=1)                                                     # :55
    in_degree = rw.sum(axis=0)                                                      # :56
    out_degree[out_degree == 0.0] = 1.0                                             # :58
    od = np.asarray(out_degree).reshape(-1)
    for i in range(rw.shape[0]):                                                    # :61-63
        rw.data[rw.indptr[i]: rw.indptr[i + 1]] = rw.data[rw.indptr[i]: rw.indptr[i + 1]] / od[i]
    rw.sort_indices()                                                               # :65
    out_degree = np.array(out_degree).astype(np.float64).reshape(out_degree.size)   # :67
    in_degree = np.array(in_degree).astype(np.float64).reshape(in_degree.size)      # :68
    return rw, out_degree, in_degree


def seed_list(adjacency_matrix):
    """arcte.py:610-617: pattern in-count > 1, descending count (tie order is
    unspecified in the reference -- unstable argsort -- and does not affect the output)."""
    a = sparse.csr_matrix(adjacency_matrix).copy()
    a.data = np.ones_like(a.data)
    cnt = np.squeeze(np.asarray(a.sum(axis=0), dtype=np.int64)).reshape(-1)
    nodes = np.where(cnt != 0)[0]
    nodes = nodes[np.argsort(cnt[nodes], kind="stable")][::-1]
    return nodes[cnt[nodes] > 1]


def arcte(adjacency_matrix, rho, epsilon, number_of_threads=1, variant=0):
    """arcte.py:591-688.  Chunking over processes (:650-673) only partitions the
    seed list and sums disjoint columns, so the thread count cannot change the result."""
    adjacency_matrix = sparse.csr_matrix(adjacency_matrix)
    n = adjacency_matrix.shape[0]
    w, out_degree, in_degree = get_natural_random_walk_matrix(adjacency_matrix)
    seeds = seed_list(adjacency_matrix)
    local = worker_matrix(w, out_degree, in_degree, seeds, rho, epsilon, threads=number_of_threads, variant=variant)
    identity = sparse.csr_matrix(sparse.eye(n, n, dtype=np.float64))               # :676
    ones = adjacency_matrix.copy()
    ones.data = np.ones_like(ones.data)                                             # :677-678
    base = identity + ones                                                          # :679
    return sparse.hstack([base, local]).tocsr()                                     # :683


# ---------------------------------------------------------------------------------------------------------------------
# embedding/arcte/cython_opt/arcte.pyx:125-241 and the feature weighting behind it (embedding/common.py,
# embedding/community_weighting.py)
# ---------------------------------------------------------------------------------------------------------------------

def normalize_columns(features):
    """common.py:49-67: every column with more than one stored entry is divided by sqrt(log(stored entries))."""
    f = sparse.csc_matrix(features, dtype=np.float64, copy=True)
    df = np.diff(f.indptr)
    scale = np.ones(f.shape[1])
    big = df > 1
    scale[big] = np.sqrt(np.log(df[big]))
    f.data = f.data / np.repeat(scale, df)
    return f.tocsr()


def normalize_rows(features):
    """common.py:29-46: sklearn normalize(norm="l2") = every row divided by sqrt(sum of squares) (zero rows stay),
    the squares summed in storage order (sklearn/utils/sparsefuncs_fast.pyx inplace_csr_row_normalize_l2)."""
    f = sparse.csr_matrix(features, dtype=np.float64, copy=True)
    for i in range(f.shape[0]):
        lo, hi = f.indptr[i], f.indptr[i + 1]
        acc = 0.0
        for x in f.data[lo:hi]:
            acc += x * x
        if acc != 0.0:
            f.data[lo:hi] /= np.sqrt(acc)
    return f


def normalize_community_features(features):
    """common.py:8-26"""
    return normalize_rows(normalize_columns(features))


def centrality_block(adjacency_matrix, rho, epsilon, node_begin=0, node_end=None):
    """The seed loop of arcte.pyx:165-217 for the nodes in [node_begin, node_end): (colptr, rows, partial centrality)."""
    a = sparse.csr_matrix(adjacency_matrix, dtype=np.float64)
    n = a.shape[0]
    node_end = n if node_end is None else int(node_end)
    w, out_degree, in_degree = get_natural_random_walk_matrix(a)
    indptr, indices, data = _csr_arrays(w)
    colptr = np.zeros(node_end - node_begin + 1, dtype=np.int64)
    rows_p = C.POINTER(C.c_int32)()
    centrality = np.zeros(n, dtype=np.float64)
    rc = lib().oracle_arcte_and_centrality(n, indptr, indices, data, np.ascontiguousarray(in_degree, dtype=np.float64),
                                           float(rho), float(epsilon), int(node_begin), node_end, colptr, C.byref(rows_p),
                                           centrality)
    if rc != 0:
        raise RuntimeError("oracle_arcte_and_centrality failed with status %d" % rc)
    total = int(colptr[-1])
    rows = np.ctypeslib.as_array(rows_p, shape=(max(total, 1),))[:total].copy()
    lib().oracle_free(rows_p)
    return colptr, rows, centrality


def arcte_and_centrality(adjacency_matrix, rho, epsilon):
    """arcte.pyx:125-241.  Returns (features n x (n + emitted communities) CSR, centrality[n])."""
    a = sparse.csr_matrix(adjacency_matrix, dtype=np.float64)
    n = a.shape[0]
    w, out_degree, in_degree = get_natural_random_walk_matrix(a)
    colptr, rows, centrality = centrality_block(a, rho, epsilon)
    sizes = np.diff(colptr)
    emitted = np.flatnonzero(sizes)
    cols = np.repeat(np.arange(emitted.size), sizes[emitted])                      # arcte.pyx:213-215: running counter
    local = sparse.coo_matrix((np.ones(rows.size), (rows.astype(np.int64), cols)), shape=(n, emitted.size))
    # arcte.pyx:227-228 says identity + adjacency_matrix, but by then the reference's adjacency_matrix IS the
    # transition matrix: cython_opt/transition.pyx:19 wraps its float64 CSR argument without copy=True and
    # normalises the rows in place, so the base block carries the weights of W (row sums 1), not of A
    base = sparse.csr_matrix(sparse.eye(n, n, dtype=np.float64)) + w
    features = sparse.hstack([base, local]).tocoo() if emitted.size else base      # :231-234
    return normalize_community_features(features), centrality


def chi2_contingency_matrix(x_train, y_train):
    """community_weighting.py:11-45 (LabelBinarizer = one column per sorted class; two classes get both columns)."""
    x = sparse.csr_matrix(x_train, dtype=np.float64, copy=True)
    x.data = np.ones_like(x.data)
    classes = np.unique(y_train)
    y = (np.asarray(y_train).reshape(-1, 1) == classes.reshape(1, -1)).astype(np.int64)
    if y.shape[1] == 1:                     # a single class: LabelBinarizer gives one all-zero column
        y = np.zeros((y.shape[0], 1), dtype=np.int64)
        y = np.append(1 - y, y, axis=1)
    elif y.shape[1] == 2:                   # binary: LabelBinarizer gives the second class only, then [1 - y, y]
        y = y[:, 1:2]
        y = np.append(1 - y, y, axis=1)
    observed = np.asarray((x.T @ y).T, dtype=np.float64)
    feature_count = np.asarray(x.sum(axis=0)).reshape(1, -1)
    class_prob = y.mean(axis=0).reshape(1, -1)
    expected = np.dot(class_prob.T, feature_count)
    m = observed
    m -= expected
    m **= 2
    expected[expected == 0.0] = 1.0
    m /= expected
    return m


def peak_snr_weight_aggregation(contingency_matrix):
    """community_weighting.py:48-84"""
    c = np.array(contingency_matrix, dtype=np.float64)
    c[np.isnan(c)] = 0.0
    variance = np.sqrt(np.mean([np.var(c[k, :]) for k in range(c.shape[0])]))
    weights = np.zeros(c.shape[1])
    for f in range(c.shape[1]):
        d = c[:, f]
        d = d[d > 0.0]
        if d.size > 1:
            weights[f] = (np.max(d) - np.min(d)) / variance
        elif d.size == 1:
            weights[f] = np.max(d) / variance
    return weights


def community_weighting(x_train, x_test, community_weights):
    """community_weighting.py:87-125"""
    out = []
    for x in (x_train, x_test):
        f = sparse.csc_matrix(x, dtype=np.float64, copy=True)
        df = np.diff(f.indptr)
        reinforcement = np.where(np.asarray(community_weights) == 0.0, 0.0, np.log(1.0 + np.asarray(community_weights, dtype=np.float64)))
        scale = np.where(df > 1, reinforcement, 1.0)
        f.data = f.data * np.repeat(scale, df)
        f = f.tocsr()
        f.eliminate_zeros()
        out.append(normalize_rows(f))
    return out[0], out[1]
