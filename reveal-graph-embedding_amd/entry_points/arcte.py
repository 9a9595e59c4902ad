"""Console entry point mirroring reveal_graph_embedding/entry_points/arcte.py (reference lines 12-84):
same flags and defaults, read edge list -> symmetrise -> arcte -> write feature triplets.

On one GPU the run never builds a scipy object: the edge list is parsed into flat triplet arrays, the GPU turns
them into the symmetrised CSR, the transition matrix and the seed list (arcte_hip_create_from_coo), propagates,
assembles the n x 2n pattern, and the triplet file is written straight from its row pointers and column ids."""
import argparse

import numpy as np
import scipy.sparse as spsp

from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.common import get_threads_number
from reveal_graph_embedding_amd.datautil.datarw import (read_adjacency_matrix, read_edge_triplets, write_features,
                                                        write_feature_triplets)
from reveal_graph_embedding_amd.embedding.arcte.arcte import arcte


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-i", "--input", dest="input_edge_list_path", type=str, required=True,
                        help="This is the file path of the graph in edge list format.")
    parser.add_argument("-o", "--output", dest="output_feature_path", type=str, required=True,
                        help="This is the file path of the extracted features in triplet format.")
    parser.add_argument("-s", "--separator", dest="separator", type=str, required=False, default="\t",
                        help="The character(s) separating the values in the edge list (default is tab: \"\\t\").")
    # type=bool as in the reference (:30-32): any non-empty string is True
    parser.add_argument("-u", "--undirected", dest="undirected", type=bool, required=False, default=False,
                        help="Also create the reciprocal edge for each edge in edge list.")
    parser.add_argument("-r", "--rho", dest="restart_probability", type=float, required=False, default=0.1,
                        help="The restart probability for the vertex-centric PageRank calculation.")
    parser.add_argument("-e", "--epsilon", dest="epsilon_threshold", type=float, required=False, default=1.0e-05,
                        help="The tolerance for calculating vertex-centric PageRank values.")
    parser.add_argument("-nt", "--tasks", dest="number_of_tasks", type=int, required=False, default=None,
                        help="The number of parallel tasks to create (here: an upper bound on the GPUs used).")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    number_of_tasks = args.number_of_tasks
    if number_of_tasks is None:
        number_of_tasks = get_threads_number()

    import os
    devices = _native.device_count()
    if os.environ.get("ARCTE_HIP_DEVICES"):
        devices = len(os.environ["ARCTE_HIP_DEVICES"].split(","))
    if min(devices, max(1, number_of_tasks)) == 1:
        # one GPU: flat arrays in, flat arrays out
        n, row, col, val, node_to_id = read_edge_triplets(args.input_edge_list_path, args.separator, args.undirected)
        device = int(os.environ["ARCTE_HIP_DEVICES"].split(",")[0]) if os.environ.get("ARCTE_HIP_DEVICES") else 0
        with _native.Context.from_coo(n, row, col, val, symmetrise=True, device=device) as ctx:   # reference :70-71
            ctx.run_seeds(np.sort(ctx.seed_list()), args.restart_probability, args.epsilon_threshold)
            indptr, indices = ctx.fetch_csr(with_base_block=True)
        # every stored value is 1 except the diagonal of a node with a self-loop: I + ones = 2 (arcte.py:676-679)
        loops = np.unique(row[row == col])
        write_feature_triplets(args.output_feature_path, indptr, indices, loops, args.separator, node_to_id)
        return

    adjacency_matrix, node_to_id = read_adjacency_matrix(file_path=args.input_edge_list_path,
                                                         separator=args.separator,
                                                         undirected=args.undirected)

    # Make sure we are dealing with a symmetric adjacency matrix (reference :70-71).
    adjacency_matrix = spsp.csr_matrix(adjacency_matrix)
    adjacency_matrix = (adjacency_matrix + adjacency_matrix.transpose()) / 2

    features = arcte(adjacency_matrix=adjacency_matrix,
                     rho=args.restart_probability,
                     epsilon=args.epsilon_threshold,
                     number_of_threads=number_of_tasks)
    features = spsp.csr_matrix(features)
    features.sort_indices()

    # every stored value is 1 except the diagonal of a node with a self-loop: I + ones = 2 (arcte.py:676-679)
    loops = np.flatnonzero(adjacency_matrix.diagonal() != 0)
    write_feature_triplets(args.output_feature_path, features.indptr, features.indices, loops, args.separator, node_to_id)


if __name__ == "__main__":
    main()
