"""MI355X-native ARCTE hot path (eps-truncated absorbing-random-walk propagation).

Mirrors the call surface of MKLab-ITI/reveal-graph-embedding for that one path:

    from reveal_graph_embedding_amd.embedding.arcte.arcte import arcte, arcte_worker
    from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
    from reveal_graph_embedding_amd.eps_randomwalk.similarity import fast_approximate_cumulative_pagerank_difference

All arithmetic runs in hand-written HIP kernels (csrc/arcte_hip.hip) behind the C ABI
declared in include/arcte_hip.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
