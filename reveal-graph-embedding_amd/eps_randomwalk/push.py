"""Mirror of reveal_graph_embedding/eps_randomwalk/push.py (ARCTE variant, reference lines 41-64)."""
from reveal_graph_embedding_amd import _native


def cumulative_pagerank_difference_limit_push(s, r, w_i, a_i, push_node, rho):
    """
    Performs a random step without a self-loop, in place on the dense float64 vectors s and r:
    c = (1-rho)*r[push_node]; r[push_node] = 0; s[a_i] += c*w_i; r[a_i] += c*w_i.

    The scatter runs on the GPU (arcte_hip_push); inside arcte()/arcte_worker() pushes never cross
    the host boundary, this entry exists for call compatibility and for the parity tests.
    """
    _native.single_push(s, r, w_i, a_i, push_node, rho)
