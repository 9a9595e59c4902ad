"""(GPU) Time run_seeds over all seeds of an R-MAT graph through a given build of the library (bisecting a regression):
python tools/abi_time.py LIB.so NODES EDGES"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hot_sweep import load_graph

lib = C.CDLL(os.path.abspath(sys.argv[1]))
n, m = int(sys.argv[2]), int(sys.argv[3])
A = load_graph(n, m)
i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
lib.arcte_hip_last_error.restype = C.c_char_p
lib.arcte_hip_create_from_adjacency.argtypes = [C.c_int, C.c_int64, C.c_int64, i64, i32, f64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]
lib.arcte_hip_graph_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
lib.arcte_hip_fetch_seed_list.argtypes = [C.c_void_p, i64]
lib.arcte_hip_run_seeds.argtypes = [C.c_void_p, i64, C.c_int64, C.c_double, C.c_double, C.c_int]
lib.arcte_hip_run_timing.argtypes = [C.c_void_p, f64]
lib.arcte_hip_info.argtypes = [C.c_void_p, i64]
lib.arcte_hip_destroy.argtypes = [C.c_void_p]
ctx = C.c_void_p()
def chk(rc):
    if rc: raise RuntimeError(lib.arcte_hip_last_error().decode())
chk(lib.arcte_hip_create_from_adjacency(0, n, A.nnz, A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data, 0, 0, C.byref(ctx)))
ns = C.c_int64()
chk(lib.arcte_hip_graph_sizes(ctx, None, None, C.byref(ns)))
seeds = np.zeros(ns.value, dtype=np.int64)
chk(lib.arcte_hip_fetch_seed_list(ctx, seeds))
info = np.zeros(10, dtype=np.int64)
chk(lib.arcte_hip_info(ctx, info))
t = np.zeros(4)
for rep in range(2):
    chk(lib.arcte_hip_run_seeds(ctx, seeds, seeds.size, 0.1, 1e-5, 1))
    chk(lib.arcte_hip_run_timing(ctx, t))
    print(sys.argv[1], "slots", info[0], "qcap", info[1], "hot", info[5], "waves/CU", info[7], "push ms %.1f" % t[1], "->", round(seeds.size / (t[3] / 1e3)), "seeds/s", flush=True)
lib.arcte_hip_destroy(ctx)
