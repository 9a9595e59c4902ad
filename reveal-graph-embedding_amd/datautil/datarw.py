"""Mirror of the two I/O functions of reveal_graph_embedding/datautil/datarw.py the ARCTE entry point
uses: the edge-list reader (reference datarw.py:54-120) and the triplet feature writer (:123-143)."""
import numpy as np
import scipy.sparse as spsp



def read_edge_triplets(file_path, separator, undirected):
    """The edge list as flat arrays: (number_of_nodes, row int32, col int32, data float64, node_ids) with
    node_ids[new id] = original id (the reference's node_to_id as an array).  Same parsing as read_adjacency_matrix
    (reference :54-120) -- `line.strip().split(separator)`, '#' lines skipped, first-seen renumbering, reciprocal edges --
    done by the library (arcte_hip_edge_list_read: several threads parse, one ordered pass renumbers)."""
    from reveal_graph_embedding_amd import _native
    return _native.read_edge_list(file_path, separator, undirected)


def read_adjacency_matrix(file_path, separator, undirected):
    """
    Reads an edge list (`source<sep>target<sep>weight` per line, lines starting with '#' skipped) and
    returns (adjacency_matrix as scipy COO float64, node_to_id).  Node ids are renumbered in first-seen
    order, source before target (reference :87-92); node_to_id maps the new numbers back.  With
    `undirected`, every non-loop edge also gets its reciprocal (:105-109).  Duplicate edges stay
    duplicate COO entries (they are summed when the matrix is converted, as in the reference).
    """
    number_of_nodes, row, col, data, node_ids = read_edge_triplets(file_path, separator, undirected)
    adjacency_matrix = spsp.coo_matrix((data, (row.astype(np.int64), col.astype(np.int64))),
                                       shape=(number_of_nodes, number_of_nodes))
    return adjacency_matrix, dict(enumerate(node_ids.tolist()))


def write_features(file_path, features, separator, node_to_id):
    """One line per stored entry, in COO-of-CSR order: `<original node id><sep><column><sep><int(value)>`."""
    features = spsp.coo_matrix(features)
    ids = np.array([node_to_id[i] for i in range(features.shape[0])], dtype=object) if features.shape[0] else []
    with open(file_path, "w") as f:
        f.writelines(str(ids[r]) + separator + str(c) + separator + str(int(v)) + "\n"
                     for r, c, v in zip(features.row.tolist(), features.col.tolist(), features.data.tolist()))


def write_feature_triplets(file_path, indptr, indices, doubled_diagonal_nodes, separator, node_to_id):
    """write_features (reference :123-143) for arcte()'s matrix given as raw CSR arrays: one line per stored entry in
    row-major order, `<original node id><sep><column><sep><value>` with value 1, or 2 on the diagonal of the nodes in
    `doubled_diagonal_nodes` (identity + ones on a self-loop, arcte.py:676-679).  `node_to_id` is the reference's
    dictionary or the array read_edge_triplets returns.  Formatted by the library (arcte_hip_write_feature_triplets)."""
    from reveal_graph_embedding_amd import _native
    n = len(indptr) - 1
    if isinstance(node_to_id, dict):
        node_ids = np.fromiter((node_to_id[i] for i in range(n)), dtype=np.int64, count=n)
    else:
        node_ids = np.asarray(node_to_id, dtype=np.int64)
    _native.write_feature_triplets(file_path, indptr, indices, node_ids, doubled_diagonal_nodes, separator)
