"""Synthetic power-law graphs used by bench.py and the parity tests.

The R-MAT recipe is the one SURVEY.md section 8(d) fixes for the benchmark
configs (it is not part of the reference; the reference ships no generator).
"""
import numpy as np
import scipy.sparse as sparse


def rmat_edges(n, m, seed=0, a=0.57, b=0.19, c=0.19):
    """Directed R-MAT endpoint arrays (before de-duplication), SURVEY.md 8(d)."""
    bits = int(np.ceil(np.log2(n)))
    rng = np.random.default_rng(seed)
    src = np.zeros(m, dtype=np.int64)
    dst = np.zeros(m, dtype=np.int64)
    for _ in range(bits):
        u = rng.random(m)
        src_bit = u >= a + b
        dst_bit = ((u >= a) & (u < a + b)) | (u >= a + b + c)
        src = (src << 1) | src_bit
        dst = (dst << 1) | dst_bit
    perm = rng.permutation(2 ** bits)
    src = perm[src] % n
    dst = perm[dst] % n
    return src, dst


def rmat_graph(n, m, seed=0, a=0.57, b=0.19, c=0.19):
    """Symmetric unit-weight CSR adjacency: self-loops dropped, duplicates
    collapsed, pattern symmetrised (A or A^T), float64 ones."""
    src, dst = rmat_edges(n, m, seed, a, b, c)
    keep = src != dst
    src = src[keep]
    dst = dst[keep]
    key = np.concatenate([src * n + dst, dst * n + src])
    del src, dst
    key = np.unique(key)
    row = (key // n).astype(np.int32)
    col = (key % n).astype(np.int32)
    del key
    indptr = np.zeros(n + 1, dtype=np.int64)
    counts = np.bincount(row, minlength=n)
    indptr[1:] = np.cumsum(counts)
    data = np.ones(col.size, dtype=np.float64)
    idx_dtype = np.int32 if col.size < 2 ** 31 else np.int64
    adjacency = sparse.csr_matrix((data, col.astype(idx_dtype), indptr.astype(idx_dtype)), shape=(n, n))
    return adjacency
