"""Mirror of reveal_graph_embedding/eps_randomwalk/push.py (reference lines 4-64): the three push flavours."""
from reveal_graph_embedding_amd import _native


def cumulative_pagerank_difference_limit_push(s, r, w_i, a_i, push_node, rho):
    """
    Performs a random step without a self-loop, in place on the dense float64 vectors s and r:
    c = (1-rho)*r[push_node]; r[push_node] = 0; s[a_i] += c*w_i; r[a_i] += c*w_i.

    The scatter runs on the GPU (arcte_hip_push); inside arcte()/arcte_worker() pushes never cross
    the host boundary, this entry exists for call compatibility and for the parity tests.
    """
    _native.single_push(s, r, w_i, a_i, push_node, rho)


def pagerank_limit_push(s, r, w_i, a_i, push_node, rho):
    """
    Performs a random step without a self-loop (reference push.py:4-17), in place:
    s[push_node] += rho*r[push_node]; r[push_node] = 0; r[a_i] += (1-rho)*r_old*w_i.
    """
    _native.single_push(s, r, w_i, a_i, push_node, rho, variant=_native.PAGERANK)


def pagerank_lazy_push(s, r, w_i, a_i, push_node, rho, lazy):
    """
    Performs a random step with a self-loop (reference push.py:20-38), in place:
    s[push_node] += rho*r; r[push_node] = (1-rho)*lazy*r; r[a_i] += (1-rho)*(1-lazy)*r*w_i.
    """
    _native.single_push(s, r, w_i, a_i, push_node, rho, variant=_native.LAZY_PAGERANK, laziness_factor=lazy)
