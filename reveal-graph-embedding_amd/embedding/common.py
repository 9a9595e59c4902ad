"""Mirror of reveal_graph_embedding/embedding/common.py (reference lines 8-67): the tf-idf-like column scaling and
the row normalisation applied to community features.  Both run as streaming HIP kernels over the CSR
(reveal_graph_embedding_amd._native.Features); a Features object is accepted and returned as it is, so a pipeline
can keep arcte()'s matrix on the GPU from extraction to weighting."""
from reveal_graph_embedding_amd import _native


def _on_device(features, op, device=0):
    if isinstance(features, _native.Features):
        op(features)
        return features
    with _native.Features.upload(features, device=device) as f:
        op(f)
        return f.to_scipy()


def normalize_columns(features, device=0):
    """
    This performs column normalization of community embedding features (reference common.py:49-67): every column
    with more than one stored entry is divided by sqrt(log(number of stored entries)).  Returns CSR.
    `device` picks the GPU a scipy matrix is sent to (a Features object stays where it is).
    """
    return _on_device(features, lambda f: f.normalize_columns(), device)


def normalize_rows(features, device=0):
    """
    This performs row normalization to 1 of community embedding features (reference common.py:29-46).
    """
    return _on_device(features, lambda f: f.normalize_rows(), device)


def normalize_community_features(features, device=0):
    """
    This performs TF-IDF-like normalization of community embedding features (reference common.py:8-26).
    """
    return _on_device(features, lambda f: f.normalize_columns().normalize_rows(), device)
