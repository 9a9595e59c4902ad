"""(GPU) Console script end to end on an R-MAT graph written as an edge-list file, next to arcte() end to end on the same
graph held as a scipy matrix: the reader (datarw.py:54-120) and the writer (:123-143) are native (csrc/arcte_io.cpp).

usage: python tools/cli_time.py NODES EDGES [REPEATS]
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hot_sweep import load_graph
from reveal_graph_embedding_amd.datautil.datarw import read_edge_triplets, write_feature_triplets
from reveal_graph_embedding_amd.embedding.arcte.arcte import arcte
from reveal_graph_embedding_amd.entry_points.arcte import main
from reveal_graph_embedding_amd import _native


def run():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    A = load_graph(n, m)
    d = tempfile.mkdtemp(prefix="arcte_cli_", dir="/tmp")
    edges, out = os.path.join(d, "edges.tsv"), os.path.join(d, "features.tsv")
    import pandas as pd
    coo = A.tocoo()
    # node ids as the file gives them: 1000 + 7 * index (the reader renumbers in first-seen order)
    pd.DataFrame({"s": 1000 + 7 * coo.row.astype(np.int64), "t": 1000 + 7 * coo.col.astype(np.int64), "w": np.ones(coo.nnz, dtype=np.int64)}
                 ).to_csv(edges, sep="\t", header=False, index=False)
    print("edge list: %d lines, %.1f MB" % (coo.nnz, os.path.getsize(edges) / 1e6), flush=True)
    best_cli = best_lib = 1e30
    parts = None
    for _ in range(repeats):
        t = time.perf_counter()
        main(["-i", edges, "-o", out, "-nt", "1"])
        best_cli = min(best_cli, time.perf_counter() - t)
        t = time.perf_counter()
        f = arcte(A, 0.1, 1e-5, 1)
        best_lib = min(best_lib, time.perf_counter() - t)
        # the console script's legs on their own
        t0 = time.perf_counter()
        nn, row, col, val, ids = read_edge_triplets(edges, "\t", False)
        t1 = time.perf_counter()
        with _native.Context.from_coo(nn, row, col, val, symmetrise=True) as ctx:
            ctx.run_seeds(np.sort(ctx.seed_list()), 0.1, 1e-5)
            indptr, indices = ctx.fetch_csr(with_base_block=True)
        t2 = time.perf_counter()
        write_feature_triplets(out, indptr, indices, np.zeros(0, np.int64), "\t", ids)
        t3 = time.perf_counter()
        legs = (t1 - t0, t2 - t1, t3 - t2)
        parts = legs if parts is None else tuple(min(a, b) for a, b in zip(parts, legs))
    print("features: %d entries, file %.1f MB" % (f.nnz, os.path.getsize(out) / 1e6))
    print("console script end to end %.3f s | arcte() end to end %.3f s | ratio %.2f" % (best_cli, best_lib, best_cli / best_lib))
    print("legs (best of %d): read + renumber %.3f s, GPU (triplets -> CSR -> seeds -> n x 2n pattern on the host) %.3f s, format + write %.3f s"
          % (repeats, parts[0], parts[1], parts[2]))
    os.remove(edges)
    os.remove(out)
    os.rmdir(d)


if __name__ == "__main__":
    run()
