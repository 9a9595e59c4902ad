"""Console entry point mirroring reveal_graph_embedding/entry_points/arcte.py (reference lines 12-84):
same flags and defaults, read edge list -> symmetrise -> arcte -> write feature triplets."""
import argparse

import scipy.sparse as spsp

from reveal_graph_embedding_amd.common import get_threads_number
from reveal_graph_embedding_amd.datautil.datarw import read_adjacency_matrix, write_features
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

    write_features(file_path=args.output_feature_path,
                   features=features,
                   separator=args.separator,
                   node_to_id=node_to_id)


if __name__ == "__main__":
    main()
