#!/usr/bin/env python3
"""BASELINE.json configs[0] stand-in no. 2 (the SNOW graph file is missing from the reference):
BA(20 000, 10, seed 1) [nnz 399 800] through the reference's arcte() with rho = 1e-3, eps = 1e-5, 8 processes.
Stores the SHA-256 of the canonical CSR, nnz and the local-community sizes (a summary: the matrix itself is large).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_golden_ba20000.py
"""
import hashlib
import os
import time

import networkx as nx
import numpy as np
import scipy.sparse as sparse

from reveal_graph_embedding.embedding.arcte.arcte import arcte

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = nx.barabasi_albert_graph(20000, 10, seed=1)
    a = sparse.csr_matrix(nx.to_scipy_sparse_array(g, dtype=np.float64, format="csr"))
    assert a.nnz == 399800
    t = time.time()
    f = sparse.csr_matrix(arcte(a.copy(), 1e-3, 1e-5, 8))
    dt = time.time() - t
    f.sum_duplicates()
    f.sort_indices()
    h = hashlib.sha256()
    h.update(f.indptr.astype(np.int64).tobytes())
    h.update(f.indices.astype(np.int64).tobytes())
    local = sparse.csc_matrix(f[:, 20000:])
    np.savez_compressed(os.path.join(HERE, "ba20000_rho1e-3_summary.npz"),
                        adj_indptr=a.indptr.astype(np.int64), adj_indices=a.indices.astype(np.int32),
                        sha256=np.frombuffer(h.digest(), dtype=np.uint8), nnz=np.int64(f.nnz),
                        local_col_counts=np.diff(local.indptr).astype(np.int32), rho=np.float64(1e-3), epsilon=np.float64(1e-5),
                        reference_seconds_8_processes=np.float64(dt))
    print("BA(20000,10) rho=1e-3: reference arcte() %.1f s with 8 processes, nnz %d, sha256 %s" % (dt, f.nnz, h.hexdigest()))


if __name__ == "__main__":
    main()
