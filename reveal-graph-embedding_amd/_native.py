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
    "arcte_hip_epsilon_effective": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_double, _f64p]),
    "arcte_hip_run_seeds": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_double, C.c_double, C.c_int]),
    "arcte_hip_run_seeds_variant": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int,
                                              C.c_double]),
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
    "arcte_hip_push_variant": (C.c_int, [C.c_int, C.c_int64, _f64p, _f64p, _f64p, _i32p, C.c_int64, C.c_int64, C.c_double,
                                         C.c_int, C.c_double]),
    "arcte_hip_push": (C.c_int, [C.c_int, C.c_int64, _f64p, _f64p, _f64p, _i32p, C.c_int64, C.c_int64, C.c_double]),
    "arcte_hip_set_float32": (C.c_int, [C.c_void_p, C.c_int]),
    "arcte_hip_stream_bandwidth": (C.c_int, [C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "arcte_hip_info": (C.c_int, [C.c_void_p, _i64p]),
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


def stream_bandwidth(device=0, nbytes=4 << 30):
    """On-box streaming rates in GB/s: (coalesced read sweep, copy counted as read + write bytes)."""
    rd, cp = C.c_double(0), C.c_double(0)
    _check(lib().arcte_hip_stream_bandwidth(int(device), int(nbytes), C.byref(rd), C.byref(cp)))
    return rd.value, cp.value


class Context:
    """Device-resident transition matrix + propagation slots on one GPU."""

    def __init__(self, indptr, indices, data, out_degree, in_degree, device=0, n_slots=0, queue_capacity=0):
        self._h = None
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
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

    def run_seeds(self, seeds, rho, epsilon, use_effective_epsilon=True, variant=ARCTE, laziness_factor=0.5):
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        _check(lib().arcte_hip_run_seeds_variant(self._h, seeds, seeds.size, float(rho), float(epsilon),
                                                 1 if use_effective_epsilon else 0, int(variant),
                                                 float(laziness_factor)))

    def result_sizes(self):
        ns, tot = C.c_int64(0), C.c_int64(0)
        _check(lib().arcte_hip_result_sizes(self._h, C.byref(ns), C.byref(tot)))
        return ns.value, tot.value

    def fetch(self, want_eps=False, want_nop=False):
        ns, tot = self.result_sizes()
        colptr = np.zeros(ns + 1, dtype=np.int64)
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

    def fetch_csr(self, with_base_block=False):
        """The last run as CSR (indptr int64[n+1], indices int32), assembled on the device.  With the base block
        the columns are those of arcte()'s n x 2n matrix.  Seeds must have been unique."""
        nnz = C.c_int64(0)
        _check(lib().arcte_hip_result_csr_size(self._h, 1 if with_base_block else 0, C.byref(nnz)))
        indptr = np.zeros(self.n + 1, dtype=np.int64)
        indices = np.zeros(max(nnz.value, 1), dtype=np.int32)
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
        s = np.zeros(7, dtype=np.int64)
        _check(lib().arcte_hip_run_counters(self._h, s, s.size))
        return dict(pushes=int(s[0]), edges=int(s[1]), enqueues=int(s[2]), support=int(s[3]),
                    reruns=int(s[4]), launches=int(s[5]), candidates=int(s[6]))

    def timing(self):
        t = np.zeros(4, dtype=np.float64)
        _check(lib().arcte_hip_run_timing(self._h, t))
        return dict(eps_ms=float(t[0]), push_ms=float(t[1]), compact_ms=float(t[2]), call_ms=float(t[3]))

    def info(self):
        i = np.zeros(8, dtype=np.int64)
        _check(lib().arcte_hip_info(self._h, i))
        return dict(slots=int(i[0]), queue_capacity=int(i[1]), device_bytes=int(i[2]), compute_units=int(i[3]),
                    waves_per_workgroup=int(i[4]), hot_values_per_wave=int(i[5]), tiles=int(i[6]),
                    waves_per_cu=int(i[7]))

    def similarity_slice(self, seed, rho, epsilon, s, r, variant=ARCTE, laziness_factor=0.5):
        if s.dtype != np.float64 or r.dtype != np.float64 or not s.flags.c_contiguous or not r.flags.c_contiguous:
            raise TypeError("s and r must be C-contiguous float64 arrays (they are updated in place)")
        if s.size != self.n or r.size != self.n:
            raise ValueError("s and r must have one entry per node")
        nop = C.c_int64(0)
        _check(lib().arcte_hip_similarity_slice_variant(self._h, int(seed), float(rho), float(epsilon), int(variant),
                                                        float(laziness_factor), s, r, C.byref(nop)))
        return nop.value


def single_push(s, r, w_i, a_i, push_node, rho, device=0, variant=ARCTE, laziness_factor=0.5):
    if s.dtype != np.float64 or r.dtype != np.float64 or not s.flags.c_contiguous or not r.flags.c_contiguous:
        raise TypeError("s and r must be C-contiguous float64 arrays (they are updated in place)")
    w_i = np.ascontiguousarray(w_i, dtype=np.float64)
    a_i = np.ascontiguousarray(a_i, dtype=np.int32)
    _check(lib().arcte_hip_push_variant(int(device), s.size, s, r, w_i, a_i, a_i.size, int(push_node), float(rho),
                                        int(variant), float(laziness_factor)))
