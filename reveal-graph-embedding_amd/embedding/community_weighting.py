"""Mirror of reveal_graph_embedding/embedding/community_weighting.py (reference lines 11-125): chi-squared
contingency of community features against labels, peak-SNR aggregation into one weight per community, and the
weighting itself.  The matrices stay on the GPU (reveal_graph_embedding_amd._native.Features) between the steps."""
import numpy as np
import scipy.sparse as sparse

from reveal_graph_embedding_amd import _native


def _binarized_labels(y_train):
    """What LabelBinarizer().fit_transform leaves (reference :19-21) as CSR class lists: a 1-D label vector becomes
    one class per row (sorted label order; two labels give the pair [1 - Y, Y], one label an all-zero column and its
    complement), a 2-D indicator matrix is taken as it is."""
    if sparse.issparse(y_train) or np.ndim(y_train) == 2:
        y = sparse.csr_matrix(y_train)
        y.eliminate_zeros()
        if y.shape[1] == 1:
            col = np.asarray(y.todense()).reshape(-1) != 0
            y = sparse.csr_matrix(np.stack([~col, col], axis=1).astype(np.int8))
        return y.indptr.astype(np.int64), y.indices.astype(np.int32), y.shape[1]
    y = np.asarray(y_train).reshape(-1)
    classes, idx = np.unique(y, return_inverse=True)
    if classes.size == 1:
        idx = np.zeros(y.size, dtype=np.int64)          # column 0 = 1 - Y = all ones, column 1 = Y = all zeros
        return np.arange(y.size + 1, dtype=np.int64), idx.astype(np.int32), 2
    return np.arange(y.size + 1, dtype=np.int64), idx.astype(np.int32), max(int(classes.size), 2)


def _as_features(x):
    return (x, False) if isinstance(x, _native.Features) else (_native.Features.upload(x), True)


def chi2_contingency_matrix(X_train, y_train):
    """Reference :11-45: (observed - expected)^2 / expected per (class, community); classes x communities array."""
    f, own = _as_features(X_train)
    try:
        yp, yi, k = _binarized_labels(y_train)
        cont, _ = f.chi2_psnr_weights(yp, yi, k, want_contingency=True)
    finally:
        if own:
            f.close()
    return cont


def peak_snr_weight_aggregation(contingency_matrix):
    """Reference :48-84: one weight per community from the classes x communities statistic."""
    return _native.peak_snr_weights(contingency_matrix)


def community_weighting(X_train, X_test, community_weights):
    """Reference :87-125: scale the communities by log(1 + weight), drop zeros, l2-normalise the rows."""
    out = []
    for x in (X_train, X_test):
        f, own = _as_features(x)
        try:
            f.community_weighting(community_weights)
            out.append(f.to_scipy() if own else f)
        finally:
            if own:
                f.close()
    return out[0], out[1]


def chi2_psnr_community_weighting(X_train, X_test, y_train):
    """Reference :128-136: weights from the training part, applied to both parts."""
    ft, own = _as_features(X_train)
    try:
        yp, yi, k = _binarized_labels(y_train)
        weights = ft.chi2_psnr_weights(yp, yi, k)
    finally:
        if own:
            ft.close()
    return community_weighting(X_train, X_test, weights)
