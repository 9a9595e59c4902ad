"""Mirror of reveal_graph_embedding/eps_randomwalk/transition.py (reference lines 43-99)."""
import numpy as np
import scipy.sparse as sparse

from reveal_graph_embedding_amd import _native


def get_natural_random_walk_matrix(adjacency_matrix, make_shared=False, device=0):
    """
    Returns the natural random walk transition probability matrix given the adjacency matrix.

    Same contract as the reference (transition.py:43-99): returns (W, out_degree, in_degree) with
    W = D_out^-1 A as float64 CSR with sorted column indices, the weighted out-degree (zero rows get
    divisor 1, transition.py:58) and the weighted in-degree.  The sums, the row scaling and the column sort run on
    the GPU (arcte_hip_create_from_adjacency) with scipy's own rounding order; `make_shared` is accepted for call
    compatibility (there are no worker processes to share with here).  arcte() does not call this: it keeps W on
    the device and never copies it back.
    """
    a = sparse.csr_matrix(adjacency_matrix, dtype=np.float64)
    n = a.shape[0]
    if a.shape[0] != a.shape[1]:
        raise ValueError("the adjacency matrix must be square")
    with _native.Context.from_adjacency(a.indptr, a.indices, a.data, device=device, n_slots=1) as ctx:
        indptr, indices, data, out_degree, in_degree = ctx.transition()
    index_dtype = np.int32 if max(n, indices.size) < 2 ** 31 else np.int64
    rw_transition = sparse.csr_matrix((data, indices.astype(index_dtype, copy=False), indptr.astype(index_dtype)), shape=(n, n))
    rw_transition.has_sorted_indices = True
    return rw_transition, out_degree, in_degree
