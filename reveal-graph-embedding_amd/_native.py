"""ctypes binding of csrc/libarcte_hip.so (C ABI: include/arcte_hip.h).

There is deliberately no fallback: if the HIP library is missing or no GPU is
visible, every compute call raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libarcte_hip.so")

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

# every symbol include/arcte_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "arcte_hip_abi_version": (C.c_int, []),
    "arcte_hip_last_error": (C.c_char_p, []),
    "arcte_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "arcte_hip_create": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _i64p, _i32p, _f64p, _f64p, _f64p,
                                   C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]),
    "arcte_hip_destroy": (C.c_int, [C.c_void_p]),
    "arcte_hip_trim": (C.c_int, []),
    "arcte_hip_create_from_adjacency": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _i64p, _i32p, _f64p, C.c_int64, C.c_int64,
                                                  C.POINTER(C.c_void_p)]),
    "arcte_hip_create_from_coo": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _i32p, _i32p, _f64p, C.c_int, C.c_int64, C.c_int64,
                                            C.POINTER(C.c_void_p)]),
    "arcte_hip_graph_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "arcte_hip_fetch_transition": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "arcte_hip_fetch_seed_list": (C.c_int, [C.c_void_p, C.c_void_p]),
    "arcte_hip_epsilon_effective": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_double, _f64p]),
    "arcte_hip_epsilon_effective_scalar": (C.c_int, [C.c_int, C.c_double, C.c_double, _f64p, C.c_int64, C.POINTER(C.c_double)]),
    "arcte_hip_run_seeds": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_double, C.c_double, C.c_int]),
    "arcte_hip_run_seeds_variant": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int,
                                              C.c_double]),
    "arcte_hip_run_seeds_append": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int,
                                             C.c_double]),
    "arcte_hip_run_centrality": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_double]),
    "arcte_hip_fetch_centrality": (C.c_int, [C.c_void_p, _f64p]),
    "arcte_hip_result_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "arcte_hip_fetch_result": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "arcte_hip_result_csr_size": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "arcte_hip_fetch_result_csr": (C.c_int, [C.c_void_p, C.c_int, _i64p, C.c_void_p, C.POINTER(C.c_int64)]),
    "arcte_hip_result_device_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "arcte_hip_copy_result_rows_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "arcte_hip_run_stats": (C.c_int, [C.c_void_p, _i64p]),
    "arcte_hip_run_counters": (C.c_int, [C.c_void_p, _i64p, C.c_int]),
    "arcte_hip_run_timing": (C.c_int, [C.c_void_p, _f64p]),
    "arcte_hip_similarity_slice": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_double, _f64p, _f64p,
                                             C.POINTER(C.c_int64)]),
    "arcte_hip_similarity_slice_variant": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_double,
                                                     _f64p, _f64p, C.POINTER(C.c_int64)]),
    "arcte_hip_seed_state": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, _f64p, _f64p,
                                       C.POINTER(C.c_int64)]),
    "arcte_hip_push_variant": (C.c_int, [C.c_int, C.c_int64, _f64p, _f64p, _f64p, _i32p, C.c_int64, C.c_int64, C.c_double,
                                         C.c_int, C.c_double]),
    "arcte_hip_push": (C.c_int, [C.c_int, C.c_int64, _f64p, _f64p, _f64p, _i32p, C.c_int64, C.c_int64, C.c_double]),
    "arcte_hip_set_float32": (C.c_int, [C.c_void_p, C.c_int]),
    "arcte_hip_stream_bandwidth": (C.c_int, [C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "arcte_hip_append_result": (C.c_int, [C.c_void_p, _i64p, _i64p, C.c_int64, C.c_void_p, C.c_int64]),
    "arcte_hip_edge_list_read": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "arcte_hip_edge_list_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "arcte_hip_edge_list_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "arcte_hip_edge_list_destroy": (C.c_int, [C.c_void_p]),
    "arcte_hip_write_feature_triplets": (C.c_int, [C.c_char_p, C.c_int64, _i64p, C.c_void_p, _i64p, _i64p, C.c_int64, C.c_char_p]),
    "arcte_hip_info": (C.c_int, [C.c_void_p, _i64p]),
    "arcte_hip_state_info": (C.c_int, [C.c_void_p, _i64p]),
    "arcte_hip_placement_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), _f64p, C.c_int, C.POINTER(C.c_int)]),
    "arcte_hip_memory_info": (C.c_int, [C.c_int, _i64p]),
    "arcte_hip_has_ab_builds": (C.c_int, []),
    "arcte_hip_launch_occupancy": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "arcte_hip_features_from_result": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "arcte_hip_features_upload": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, _i64p, _i32p, _f64p, C.POINTER(C.c_void_p)]),
    "arcte_hip_features_destroy": (C.c_int, [C.c_void_p]),
    "arcte_hip_features_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "arcte_hip_features_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "arcte_hip_features_select_rows": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.POINTER(C.c_void_p)]),
    "arcte_hip_features_normalize_columns": (C.c_int, [C.c_void_p]),
    "arcte_hip_features_normalize_rows": (C.c_int, [C.c_void_p]),
    "arcte_hip_features_chi2_psnr_weights": (C.c_int, [C.c_void_p, _i64p, _i32p, C.c_int64, C.c_void_p, _f64p]),
    "arcte_hip_features_community_weighting": (C.c_int, [C.c_void_p, _f64p]),
    "arcte_hip_peak_snr_weights": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _f64p, _f64p]),
}

# push flavours (include/arcte_hip.h: `variant`)
ARCTE, PAGERANK, LAZY_PAGERANK = 0, 1, 2

_lib = None
_lock = threading.Lock()


class ArcteHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("arcte_hip error %d: %s" % (code, message))
        self.code = code


def lib():
    """Load the library once; raise ImportError loudly when it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    "HIP extension %s is missing; build it with `make -C %s` or "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)"
                    % (LIB_PATH, os.path.dirname(LIB_PATH)))
            l = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(l, name)
                fn.restype = res
                fn.argtypes = args
            _lib = l
    return _lib


def _check(rc):
    if rc != 0:
        raise ArcteHipError(rc, lib().arcte_hip_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    _check(lib().arcte_hip_device_count(C.byref(n)))
    return n.value


def trim():
    """Return the slot buffers that destroyed contexts left in the library's cache to the driver."""
    _check(lib().arcte_hip_trim())


def has_ab_builds():
    """True when the library was built with `make AB=1` (the launch shapes that lost their A/B exist)."""
    return bool(lib().arcte_hip_has_ab_builds())


def memory_info(device=0):
    """Device memory the library holds outside any context + the device's fill (arcte_hip_memory_info)."""
    i = np.zeros(4, dtype=np.int64)
    _check(lib().arcte_hip_memory_info(int(device), i))
    return dict(parked_bytes=int(i[0]), cached_bytes=int(i[1]), free_bytes=int(i[2]), total_bytes=int(i[3]))


def epsilon_effective_scalar(epsilon, seed_degree, neighbor_degrees, device=0):
    """calculate_epsilon_effective (arcte.py:26-50) for one seed, by the kernel the bulk path uses."""
    nd = np.ascontiguousarray(neighbor_degrees, dtype=np.float64).reshape(-1)
    out = C.c_double(0)
    _check(lib().arcte_hip_epsilon_effective_scalar(int(device), float(epsilon), float(seed_degree), nd, nd.size, C.byref(out)))
    return out.value


def stream_bandwidth(device=0, nbytes=4 << 30):
    """On-box streaming rates in GB/s: (coalesced read sweep, copy counted as read + write bytes)."""
    rd, cp = C.c_double(0), C.c_double(0)
    _check(lib().arcte_hip_stream_bandwidth(int(device), int(nbytes), C.byref(rd), C.byref(cp)))
    return rd.value, cp.value


def read_edge_list(file_path, separator, undirected):
    """The edge list as flat arrays (arcte_hip_edge_list_read: datautil/datarw.py:54-120 natively):
    (number_of_nodes, row int32, col int32, data float64, node_ids int64 with node_ids[new id] = original id)."""
    h = C.c_void_p()
    _check(lib().arcte_hip_edge_list_read(str(file_path).encode(), str(separator).encode(), 1 if undirected else 0, C.byref(h)))
    try:
        n, m = C.c_int64(0), C.c_int64(0)
        _check(lib().arcte_hip_edge_list_sizes(h, C.byref(n), C.byref(m)))
        row = np.empty(m.value, dtype=np.int32)
        col = np.empty(m.value, dtype=np.int32)
        val = np.empty(m.value, dtype=np.float64)
        ids = np.empty(n.value, dtype=np.int64)
        _check(lib().arcte_hip_edge_list_fetch(h, row.ctypes.data, col.ctypes.data, val.ctypes.data, ids.ctypes.data))
    finally:
        lib().arcte_hip_edge_list_destroy(h)
    return int(n.value), row, col, val, ids


def write_feature_triplets(file_path, indptr, indices, node_ids, doubled_diagonal_nodes, separator):
    """arcte()'s matrix as `<original id><sep><column><sep><value>` lines (arcte_hip_write_feature_triplets)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    node_ids = np.ascontiguousarray(node_ids, dtype=np.int64)
    doubled = np.ascontiguousarray(doubled_diagonal_nodes, dtype=np.int64).reshape(-1)
    _check(lib().arcte_hip_write_feature_triplets(str(file_path).encode(), int(indptr.size - 1), indptr, indices.ctypes.data, node_ids,
                                                  doubled, int(doubled.size), str(separator).encode()))


def fastest_context(make, calibrate, tries=3):
    """Draw `tries` contexts and keep the fastest.

    The duration of the propagation kernel belongs to the ALLOCATION a context's slot buffers happen to get: contexts on
    the same buffers repeat to 0.5 %, fresh allocations differ by up to 20 % (DESIGN.md section 5,
    profiles/r02/context_lottery_1m.txt).  A long-running service pays for a good draw once: `make()` builds a context
    (they are alive at the same time, so they cannot be handed the same memory), `calibrate(ctx)` returns a duration
    for it (e.g. the kernel time of a sample of seeds); the slower contexts are closed and their memory returned.
    Returns (context, [calibration results in draw order])."""
    drawn, results = [], []
    try:
        for _ in range(max(1, int(tries))):
            try:
                ctx = make()
            except ArcteHipError:
                if drawn:                     # no room for another one: choose among those we have
                    break
                raise
            drawn.append(ctx)
            results.append(calibrate(ctx))
        best = min(range(len(drawn)), key=lambda k: results[k])
        keep = drawn[best]
        drawn[best] = None
        return keep, results
    finally:
        for ctx in drawn:
            if ctx is not None:
                ctx.close()
        if len(results) > 1:
            trim()


def _ids32(a, n, nnz_limit, what):
    """Node ids as the C ABI takes them (int32): the reference's shared path carries int64 indices (transition.py:82-87);
    sizes beyond this library's limits are refused HERE, by name, instead of being wrapped by the cast."""
    if n >= 2 ** 31:
        raise ValueError("%d nodes: node ids are int32 in this library (n < 2^31)" % n)
    a = np.asarray(a)
    if a.size >= nnz_limit:
        raise ValueError("%d stored entries in %s: this entry point takes fewer than 2^%d" % (a.size, what, int(np.log2(nnz_limit))))
    if a.size and a.dtype.kind in "iu" and a.dtype.itemsize > 4 and (int(a.max()) >= 2 ** 31 or int(a.min()) < 0):
        raise ValueError("%s holds ids outside [0, 2^31): node ids are int32 in this library" % what)
    return np.ascontiguousarray(a, dtype=np.int32)


class Context:
    """Device-resident transition matrix + propagation slots on one GPU."""

    def __init__(self, indptr, indices, data, out_degree, in_degree, device=0, n_slots=0, queue_capacity=0):
        self._h = None
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = _ids32(indices, int(np.asarray(out_degree).size), 2 ** 31, "indices")
        data = np.ascontiguousarray(data, dtype=np.float64)
        out_degree = np.ascontiguousarray(out_degree, dtype=np.float64)
        in_degree = np.ascontiguousarray(in_degree, dtype=np.float64)
        self.n = int(out_degree.size)
        if indptr.size != self.n + 1 or in_degree.size != self.n or indices.size != data.size:
            raise ValueError("inconsistent CSR / degree array sizes")
        h = C.c_void_p()
        _check(lib().arcte_hip_create(int(device), self.n, int(indices.size), indptr, indices, data,
                                      out_degree, in_degree, int(n_slots), int(queue_capacity), C.byref(h)))
        self._h = h
        self.device = int(device)

    @classmethod
    def _adopt(cls, handle, device):
        self = cls.__new__(cls)
        self._h = handle
        self.device = int(device)
        self.n = self.graph_sizes()[0]
        return self

    @classmethod
    def from_adjacency(cls, indptr, indices, data, device=0, n_slots=0, queue_capacity=0):
        """Context from the ADJACENCY matrix (CSR): W, the degree vectors and the seed list are made on the device."""
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = _ids32(indices, int(indptr.size - 1), 2 ** 31, "indices")
        data = np.ascontiguousarray(data, dtype=np.float64)
        if indices.size != data.size or indptr.size < 2:
            raise ValueError("inconsistent CSR array sizes")
        h = C.c_void_p()
        _check(lib().arcte_hip_create_from_adjacency(int(device), int(indptr.size - 1), int(indices.size), indptr, indices, data,
                                                     int(n_slots), int(queue_capacity), C.byref(h)))
        return cls._adopt(h, device)

    @classmethod
    def from_coo(cls, n, row, col, val, symmetrise=False, device=0, n_slots=0, queue_capacity=0):
        """Context from edge-list triplets; symmetrise=True makes (A + A^T)/2 first (entry_points/arcte.py:70-71)."""
        row = _ids32(row, int(n), 2 ** 30, "row")
        col = _ids32(col, int(n), 2 ** 30, "col")
        val = np.ascontiguousarray(val, dtype=np.float64)
        if not (row.size == col.size == val.size):
            raise ValueError("row, col and val must have one entry per triplet")
        h = C.c_void_p()
        _check(lib().arcte_hip_create_from_coo(int(device), int(n), int(row.size), row, col, val, 1 if symmetrise else 0,
                                               int(n_slots), int(queue_capacity), C.byref(h)))
        return cls._adopt(h, device)

    def graph_sizes(self):
        """(nodes, stored transitions, length of arcte()'s seed list)."""
        n, nnz, ns = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(lib().arcte_hip_graph_sizes(self._h, C.byref(n), C.byref(nnz), C.byref(ns)))
        return n.value, nnz.value, ns.value

    def transition(self):
        """(indptr, indices, data, out_degree, in_degree) of W as held on the device."""
        n, nnz, _ = self.graph_sizes()
        indptr = np.zeros(n + 1, dtype=np.int64)
        indices = np.zeros(nnz, dtype=np.int32)
        data = np.zeros(nnz, dtype=np.float64)
        od = np.zeros(n, dtype=np.float64)
        idg = np.zeros(n, dtype=np.float64)
        _check(lib().arcte_hip_fetch_transition(self._h, indptr.ctypes.data, indices.ctypes.data if nnz else None,
                                                data.ctypes.data if nnz else None, od.ctypes.data, idg.ctypes.data))
        return indptr, indices, data, od, idg

    def seed_list(self):
        """arcte()'s seed list (arcte.py:610-617), made on the device."""
        ns = self.graph_sizes()[2]
        seeds = np.zeros(ns, dtype=np.int64)
        _check(lib().arcte_hip_fetch_seed_list(self._h, seeds.ctypes.data if ns else None))
        return seeds

    def close(self):
        if self._h is not None:
            lib().arcte_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_float32(self, enable=True):
        """Switch the propagation arithmetic to float32 (tolerance sweep) or back to float64 (default)."""
        _check(lib().arcte_hip_set_float32(self._h, 1 if enable else 0))

    def epsilon_effective(self, seeds, epsilon):
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        out = np.zeros(seeds.size, dtype=np.float64)
        _check(lib().arcte_hip_epsilon_effective(self._h, seeds, seeds.size, float(epsilon), out))
        return out

    def run_seeds(self, seeds, rho, epsilon, use_effective_epsilon=True, variant=ARCTE, laziness_factor=0.5, append=False):
        """append=True: one more part of a seed list; the completed run's result stays and this run's columns join it."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        fn = lib().arcte_hip_run_seeds_append if append else lib().arcte_hip_run_seeds_variant
        _check(fn(self._h, seeds, seeds.size, float(rho), float(epsilon), 1 if use_effective_epsilon else 0, int(variant),
                  float(laziness_factor)))

    def run_centrality(self, rho, epsilon, node_begin=0, node_end=None):
        """The loop of arcte_and_centrality (arcte.pyx:165-217) for the nodes in [node_begin, node_end)."""
        _check(lib().arcte_hip_run_centrality(self._h, int(node_begin), int(self.n if node_end is None else node_end),
                                              float(rho), float(epsilon)))

    def centrality(self):
        out = np.zeros(self.n, dtype=np.float64)
        _check(lib().arcte_hip_fetch_centrality(self._h, out))
        return out

    def result_sizes(self):
        ns, tot = C.c_int64(0), C.c_int64(0)
        _check(lib().arcte_hip_result_sizes(self._h, C.byref(ns), C.byref(tot)))
        return ns.value, tot.value

    def fetch(self, want_eps=False, want_nop=False, out_rows=None):
        """(colptr, rows[, eps][, nop]) of the last run.  out_rows: a C-contiguous int32 array of at least result_sizes()[1]
        entries to receive the rows (a buffer a caller reuses has its pages faulted in: the copy then runs at the PCIe rate
        instead of a fifth of it)."""
        ns, tot = self.result_sizes()
        colptr = np.zeros(ns + 1, dtype=np.int64)
        if out_rows is not None:
            if out_rows.dtype != np.int32 or not out_rows.flags.c_contiguous or out_rows.size < tot:
                raise ValueError("out_rows must be a C-contiguous int32 array of at least %d entries" % tot)
            rows = out_rows[:tot]
        else:
            rows = np.zeros(tot, dtype=np.int32)
        eps = np.zeros(ns, dtype=np.float64) if want_eps else None
        nop = np.zeros(ns, dtype=np.int64) if want_nop else None
        _check(lib().arcte_hip_fetch_result(
            self._h, colptr.ctypes.data, rows.ctypes.data if tot else None,
            eps.ctypes.data if want_eps and ns else None, nop.ctypes.data if want_nop and ns else None))
        out = [colptr, rows]
        if want_eps:
            out.append(eps)
        if want_nop:
            out.append(nop)
        return tuple(out)

    def result_csr_size(self, with_base_block=False):
        """Stored entries of the matrix fetch_csr() will return."""
        nnz = C.c_int64(0)
        _check(lib().arcte_hip_result_csr_size(self._h, 1 if with_base_block else 0, C.byref(nnz)))
        return int(nnz.value)

    def fetch_csr(self, with_base_block=False, out_indices=None):
        """The last run as CSR (indptr int64[n+1], indices int32), assembled on the device.  With the base block
        the columns are those of arcte()'s n x 2n matrix.  Seeds must have been unique.  out_indices: a C-contiguous int32
        array of at least result_csr_size() entries to receive the column ids (e.g. one whose pages are faulted in already)."""
        nnz = self.result_csr_size(with_base_block)
        indptr = np.zeros(self.n + 1, dtype=np.int64)
        if out_indices is not None:
            if out_indices.dtype != np.int32 or not out_indices.flags.c_contiguous or out_indices.size < max(nnz, 1):
                raise ValueError("out_indices must be a C-contiguous int32 array of at least %d entries" % max(nnz, 1))
            indices = out_indices
        else:
            indices = np.empty(max(nnz, 1), dtype=np.int32)
        got = C.c_int64(0)
        _check(lib().arcte_hip_fetch_result_csr(self._h, 1 if with_base_block else 0, indptr, indices.ctypes.data,
                                                C.byref(got)))
        return indptr, indices[:got.value]

    def result_device_rows(self):
        p = C.c_void_p()
        _check(lib().arcte_hip_result_device_rows(self._h, C.byref(p)))
        return p.value or 0

    def copy_rows_to_device(self, device_ptr, capacity_rows):
        """Device-to-device copy of the last run's rows into e.g. a torch tensor (tensor.data_ptr())."""
        _check(lib().arcte_hip_copy_result_rows_to_device(self._h, C.c_void_p(int(device_ptr)), int(capacity_rows)))

    def colptr(self):
        ns, _ = self.result_sizes()
        colptr = np.zeros(ns + 1, dtype=np.int64)
        _check(lib().arcte_hip_fetch_result(self._h, colptr.ctypes.data, None, None, None))
        return colptr

    def stats(self):
        s = np.zeros(8, dtype=np.int64)
        _check(lib().arcte_hip_run_counters(self._h, s, s.size))
        return dict(pushes=int(s[0]), edges=int(s[1]), enqueues=int(s[2]), support=int(s[3]),
                    reruns=int(s[4]), launches=int(s[5]), candidates=int(s[6]), split_rows=int(s[7]))

    def timing(self):
        t = np.zeros(4, dtype=np.float64)
        _check(lib().arcte_hip_run_timing(self._h, t))
        return dict(eps_ms=float(t[0]), push_ms=float(t[1]), compact_ms=float(t[2]), call_ms=float(t[3]))

    def info(self):
        i = np.zeros(10, dtype=np.int64)
        _check(lib().arcte_hip_info(self._h, i))
        return dict(slots=int(i[0]), queue_capacity=int(i[1]), device_bytes=int(i[2]), compute_units=int(i[3]),
                    waves_per_workgroup=int(i[4]), hot_values_per_wave=int(i[5]), tiles=int(i[6]),
                    waves_per_cu=int(i[7]), narrow_rows=int(i[8]), warm_end_rank=int(i[9]))

    def append_result(self, seeds, counts, rows, nrows=None):
        """Append another worker's part to this context's completed run (arcte_hip_append_result): `rows` is an int32
        numpy array, or a raw pointer (host or any GPU of this process) together with `nrows`."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        if isinstance(rows, np.ndarray):
            rows = np.ascontiguousarray(rows, dtype=np.int32)
            ptr, nrows = rows.ctypes.data, rows.size
        else:
            ptr = int(rows)
        _check(lib().arcte_hip_append_result(self._h, seeds, counts, int(seeds.size), C.c_void_p(ptr), int(nrows)))

    def state_info(self):
        """Where the per-seed state lives and, of the last run, how its updates were served (arcte_hip_state_info)."""
        i = np.zeros(14, dtype=np.int64)
        _check(lib().arcte_hip_state_info(self._h, i))
        return dict(line_state=int(i[0]), lines_per_slot=int(i[1]), pushed_capacity=int(i[2]), candidate_capacity=int(i[3]),
                    slot_bytes=int(i[4]), bitmap_lds_bytes=int(i[5]), lds_bytes_per_wave=int(i[6]), lines_region_b=int(i[7]),
                    lds_updates=int(i[8]), blind_line_writes=int(i[9]), line_read_modify_writes=int(i[10]),
                    pushed_node_updates=int(i[11]), region_b_indirect=int(i[12]), region_b_pool_lines=int(i[13]))

    def placement_info(self):
        """The candidate allocations of the slot memory probed at creation: (index kept, [G updates/s per candidate])."""
        kept, drawn = C.c_int(-1), C.c_int(0)
        rates = np.zeros(8, dtype=np.float64)
        _check(lib().arcte_hip_placement_info(self._h, C.byref(kept), rates, rates.size, C.byref(drawn)))
        return kept.value, [float(x) for x in rates[:drawn.value]]

    def launch_occupancy(self):
        """Workgroups of the propagation kernel per CU according to the runtime's occupancy query (diagnostic)."""
        k = C.c_int(0)
        _check(lib().arcte_hip_launch_occupancy(self._h, C.byref(k)))
        return k.value

    def similarity_slice(self, seed, rho, epsilon, s, r, variant=ARCTE, laziness_factor=0.5):
        if s.dtype != np.float64 or r.dtype != np.float64 or not s.flags.c_contiguous or not r.flags.c_contiguous:
            raise TypeError("s and r must be C-contiguous float64 arrays (they are updated in place)")
        if s.size != self.n or r.size != self.n:
            raise ValueError("s and r must have one entry per node")
        nop = C.c_int64(0)
        _check(lib().arcte_hip_similarity_slice_variant(self._h, int(seed), float(rho), float(epsilon), int(variant),
                                                        float(laziness_factor), s, r, C.byref(nop)))
        return nop.value

    def seed_state(self, seed, rho, epsilon, effective=False, variant=ARCTE, laziness_factor=0.5):
        """ONE seed through the production kernel (k_arcte_lines); returns (s, r, nop): the dense float64 vectors of
        similarity.py:149-222 gathered from every level of the kernel's state, starting from zeros."""
        s = np.zeros(self.n, dtype=np.float64)
        r = np.zeros(self.n, dtype=np.float64)
        nop = C.c_int64(0)
        _check(lib().arcte_hip_seed_state(self._h, int(seed), float(rho), float(epsilon), 1 if effective else 0, int(variant),
                                          float(laziness_factor), s, r, C.byref(nop)))
        return s, r, nop.value


class Features:
    """A CSR feature matrix resident on one GPU (include/arcte_hip.h: arcte_hip_features)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_result(cls, ctx, with_base_block=True):
        h = C.c_void_p()
        _check(lib().arcte_hip_features_from_result(ctx._h, 1 if with_base_block else 0, C.byref(h)))
        return cls(h)

    @classmethod
    def upload(cls, matrix, device=0):
        """From a scipy sparse matrix (converted to CSR float64)."""
        import scipy.sparse as sparse
        m = sparse.csr_matrix(matrix, dtype=np.float64)
        indptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(m.indices, dtype=np.int32)
        data = np.ascontiguousarray(m.data, dtype=np.float64)
        h = C.c_void_p()
        _check(lib().arcte_hip_features_upload(int(device), m.shape[0], m.shape[1], int(indices.size), indptr, indices, data,
                                               C.byref(h)))
        return cls(h)

    def close(self):
        if self._h is not None:
            lib().arcte_hip_features_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sizes(self):
        r, c, z = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(lib().arcte_hip_features_sizes(self._h, C.byref(r), C.byref(c), C.byref(z)))
        return r.value, c.value, z.value

    def to_scipy(self):
        import scipy.sparse as sparse
        rows, cols, nnz = self.sizes()
        indptr = np.zeros(rows + 1, dtype=np.int64)
        indices = np.zeros(nnz, dtype=np.int32)
        data = np.zeros(nnz, dtype=np.float64)
        _check(lib().arcte_hip_features_fetch(self._h, indptr.ctypes.data, indices.ctypes.data if nnz else None,
                                              data.ctypes.data if nnz else None))
        index_dtype = np.int32 if max(cols, nnz) < 2 ** 31 else np.int64
        return sparse.csr_matrix((data, indices.astype(index_dtype, copy=False), indptr.astype(index_dtype)), shape=(rows, cols))

    def select_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        h = C.c_void_p()
        _check(lib().arcte_hip_features_select_rows(self._h, rows, rows.size, C.byref(h)))
        return Features(h)

    def normalize_columns(self):
        _check(lib().arcte_hip_features_normalize_columns(self._h))
        return self

    def normalize_rows(self):
        _check(lib().arcte_hip_features_normalize_rows(self._h))
        return self

    def chi2_psnr_weights(self, y_indptr, y_indices, n_classes, want_contingency=False):
        y_indptr = np.ascontiguousarray(y_indptr, dtype=np.int64)
        y_indices = np.ascontiguousarray(y_indices, dtype=np.int32)
        _, cols, _ = self.sizes()
        weights = np.zeros(cols, dtype=np.float64)
        cont = np.zeros((int(n_classes), cols), dtype=np.float64) if want_contingency else None
        _check(lib().arcte_hip_features_chi2_psnr_weights(self._h, y_indptr, y_indices, int(n_classes),
                                                          cont.ctypes.data if want_contingency else None, weights))
        return (cont, weights) if want_contingency else weights

    def community_weighting(self, community_weights):
        w = np.ascontiguousarray(community_weights, dtype=np.float64)
        if w.size != self.sizes()[1]:
            raise ValueError("one weight per column")
        _check(lib().arcte_hip_features_community_weighting(self._h, w))
        return self


def peak_snr_weights(contingency_matrix, device=0):
    """peak_snr_weight_aggregation (community_weighting.py:48-84) of a host classes x communities matrix."""
    m = np.ascontiguousarray(contingency_matrix, dtype=np.float64)
    if m.ndim != 2:
        raise ValueError("the contingency matrix is classes x communities")
    out = np.zeros(m.shape[1], dtype=np.float64)
    _check(lib().arcte_hip_peak_snr_weights(int(device), m.shape[0], m.shape[1], m, out))
    return out


def single_push(s, r, w_i, a_i, push_node, rho, device=0, variant=ARCTE, laziness_factor=0.5):
    if s.dtype != np.float64 or r.dtype != np.float64 or not s.flags.c_contiguous or not r.flags.c_contiguous:
        raise TypeError("s and r must be C-contiguous float64 arrays (they are updated in place)")
    w_i = np.ascontiguousarray(w_i, dtype=np.float64)
    a_i = np.ascontiguousarray(a_i, dtype=np.int32)
    _check(lib().arcte_hip_push_variant(int(device), s.size, s, r, w_i, a_i, a_i.size, int(push_node), float(rho),
                                        int(variant), float(laziness_factor)))
