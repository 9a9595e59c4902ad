"""(GPU) arcte() with several workers against one worker, end to end, same matrix: the workers' parts are merged on the
first worker's GPU (arcte_hip_append_result + the one-pass device assembly), the reference's sum of worker matrices
(arcte.py:670-673).  On a one-GPU box the workers share the GPU (ARCTE_HIP_DEVICES=0,0,0).

usage: python tools/multi_worker_time.py NODES EDGES WORKERS
"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hot_sweep import load_graph
from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.embedding.arcte.arcte import arcte


def digest(f):
    f.sort_indices()
    h = hashlib.sha256()
    h.update(f.indptr.astype(np.int64).tobytes())
    h.update(f.indices.astype(np.int64).tobytes())
    h.update(f.data.tobytes())
    return h.hexdigest()[:16]


def main():
    n, m, workers = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    A = load_graph(n, m)
    gpus = _native.device_count()
    out = {}
    for w in (1, workers, 1, workers):
        if w > 1 and gpus < w:
            os.environ["ARCTE_HIP_DEVICES"] = ",".join(str(k % gpus) for k in range(w))
        else:
            os.environ.pop("ARCTE_HIP_DEVICES", None)
        t = time.perf_counter()
        f = arcte(A, 0.1, 1e-5, w)
        dt = time.perf_counter() - t
        out.setdefault(w, []).append((dt, digest(f), f.nnz))
        del f
    for w, runs in out.items():
        print("%d worker%s on %d GPU%s: end to end %s s; nnz %d; sha256 %s" % (
            w, "s" if w > 1 else "", min(w, gpus), "s" if min(w, gpus) > 1 else "", " / ".join("%.3f" % r[0] for r in runs), runs[0][2], runs[0][1]))
    assert len({r[1] for runs in out.values() for r in runs}) == 1, "the matrices differ"
    print("ratio (best of each): %.2f" % (min(r[0] for r in out[workers]) / min(r[0] for r in out[1])))


if __name__ == "__main__":
    main()
