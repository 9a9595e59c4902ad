"""Mirror of reveal_graph_embedding/eps_randomwalk/transition.py (reference lines 43-99)."""
import numpy as np
import scipy.sparse as sparse


def get_natural_random_walk_matrix(adjacency_matrix, make_shared=False):
    """
    Returns the natural random walk transition probability matrix given the adjacency matrix.

    Same contract as the reference (transition.py:43-99): returns (W, out_degree, in_degree) with
    W = D_out^-1 A as float64 CSR with sorted column indices, the weighted out-degree (zero rows get
    divisor 1, transition.py:58) and the weighted in-degree.  `make_shared` is accepted for call
    compatibility; there are no worker processes to share with here, the arrays go to the GPU instead.
    """
    rw_transition = sparse.csr_matrix(adjacency_matrix, dtype=np.float64, copy=True)

    # Same scipy reductions as the reference so that weighted degrees round identically (:55-56).
    out_degree = np.asarray(rw_transition.sum(axis=1), dtype=np.float64).reshape(-1)
    in_degree = np.asarray(rw_transition.sum(axis=0), dtype=np.float64).reshape(-1)
    out_degree[out_degree == 0.0] = 1.0

    # Row scaling (:61-63) as one vectorised division: element k of row i is divided by out_degree[i].
    row_of = np.repeat(np.arange(rw_transition.shape[0]), np.diff(rw_transition.indptr))
    rw_transition.data = rw_transition.data / out_degree[row_of]
    rw_transition.sort_indices()
    return rw_transition, out_degree, in_degree
