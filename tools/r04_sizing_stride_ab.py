import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
os.environ["ARCTE_HIP_VERBOSE"] = "1"
from hot_sweep import load_graph
from reveal_graph_embedding_amd.embedding.arcte import arcte as A
adj = load_graph(1000000, 50000000)
A.arcte(adj, 0.1, 1e-5, 1)
for stride in (8, 16, 8, 16, 32, 8, 16, 32):
    A._SIZING_STRIDE = stride
    t = time.perf_counter()
    x = A.arcte(adj, 0.1, 1e-5, 1)
    print("STRIDE %d end-to-end %.3f s nnz %d" % (stride, time.perf_counter() - t, x.nnz), file=sys.stderr, flush=True)
    del x
