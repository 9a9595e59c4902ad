"""Launch-shape sweep of the push kernel on the bench graph: wavefronts per CU x hot-table size x tiles.

usage: python tools/hot_sweep.py NODES EDGES STRIDE CONFIG [CONFIG ...]
  CONFIG = waves_per_cu:hot_cap:tiles[:waves_per_block[:flags[:lds_reserve_kb[:warm_end_rank]]]]   flags: 1 = narrow rows off   hot_cap -1 = whatever fits the LDS
           share, 0 = table off
Every configuration must return the same communities (checked by a hash of the per-seed sorted rows).
"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from reveal_graph_embedding_amd import _native
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix


def load_graph(n, m):
    import scipy.sparse as sparse
    from reveal_graph_embedding_amd.synthetic import rmat_graph
    path = "/tmp/arcte_rmat_%d_%d_0.npz" % (n, m)
    if os.path.exists(path):
        z = np.load(path)
        return sparse.csr_matrix((np.ones(z["indices"].size), z["indices"], z["indptr"]), shape=(n, n))
    a = rmat_graph(n, m, 0)
    np.savez(path[:-4] + ".tmp.npz", indptr=a.indptr, indices=a.indices)
    os.replace(path[:-4] + ".tmp.npz", path)
    return a


def result_hash(colptr, rows):
    seg = np.repeat(np.arange(colptr.size - 1), np.diff(colptr))
    order = np.lexsort((rows, seg))
    h = hashlib.sha256()
    h.update(colptr.tobytes())
    h.update(rows[order].tobytes())
    return h.hexdigest()[:16]


def main():
    n, m, stride = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    A = load_graph(n, m)
    w, od, idg = get_natural_random_walk_matrix(A)
    seeds = seed_nodes(A)[::stride]
    print("graph n=%d nnz=%d seeds=%d" % (n, A.nnz, seeds.size), flush=True)
    ref = None
    for cfg in sys.argv[4:]:
        parts = [int(x) for x in cfg.split(":")]
        wpc, cap, tiles = parts[:3]
        wpb = parts[3] if len(parts) > 3 else 1
        os.environ["ARCTE_HIP_LDS_RESERVE_KB"] = str(parts[5]) if len(parts) > 5 else "8"
        os.environ["ARCTE_HIP_NARROW"] = "0" if len(parts) > 4 and parts[4] & 1 else "1"
        os.environ["ARCTE_HIP_WARM"] = str(parts[6]) if len(parts) > 6 else "32768"
        os.environ["ARCTE_HIP_WAVES_PER_CU"] = str(wpc)
        os.environ["ARCTE_HIP_HOT"] = str(cap)
        os.environ["ARCTE_HIP_TILES"] = str(tiles)
        os.environ["ARCTE_HIP_WAVES_PER_BLOCK"] = str(wpb)
        t = time.time()
        ctx = _native.Context(w.indptr, w.indices, w.data, od, idg)
        tc = time.time() - t
        best = None
        for it in range(5):
            ctx.run_seeds(seeds, 0.1, 1e-5)
            tm = ctx.timing()
            if best is None or tm["push_ms"] < best:
                best = tm["push_ms"]
        st = ctx.stats()
        colptr, rows = ctx.fetch()
        h = result_hash(colptr, rows)
        if ref is None:
            ref = h
        byt = 52 * st["edges"] + 36 * st["pushes"] + 4 * st["enqueues"] + 36 * st["support"]
        print("waves/CU %2d hot_cap %6d tiles %d wpb %d narrow %s warm %s | slots %5d push_ms %8.2f seeds/s %8.0f Gedges/s %6.2f alg GB/s %6.0f frac %.3f "
              "dev GB %5.1f ctx %.1fs K %d occ %d hash %s %s" % (wpc, cap, tiles, wpb, ctx.info()["narrow_rows"], os.environ["ARCTE_HIP_WARM"], ctx.info()["slots"], best, seeds.size / best * 1e3,
                                                    st["edges"] / best / 1e6, byt / best / 1e6, byt / best / 1e6 / 8000, ctx.info()["device_bytes"] / 1e9,
                                                    tc, ctx.info()["hot_values_per_wave"], ctx.launch_occupancy(), h,
                                                    "OK" if h == ref else "MISMATCH"), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
