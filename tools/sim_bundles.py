"""Design study (CPU only): how well can concurrently running seeds share hub-row pushes?

Seeds are independent, so a scheduler may interleave their (fixed) push sequences freely.  A bundle of G
seeds is advanced 'smallest-degree-first': the cheapest pending push runs next, so seeds pile up in front
of hub rows and then cross them together.  With the state laid out [node][seed-in-bundle], a push shared
by k seeds touches one 16*G-byte block per edge instead of k separate 64-byte lines.
"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from reveal_graph_embedding_amd.synthetic import rmat_graph
from reveal_graph_embedding_amd.eps_randomwalk.transition import get_natural_random_walk_matrix
from reveal_graph_embedding_amd.embedding.arcte.arcte import seed_nodes

n, m = int(sys.argv[1]), int(sys.argv[2])
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
A = rmat_graph(n, m, 0)
w, od, idg = get_natural_random_walk_matrix(A)
deg = np.diff(w.indptr).astype(np.int64)
seeds = seed_nodes(A)
shard = seeds[0::8]
rng = np.random.default_rng(0)
for G in (16, 64):
    for grouping in ("consecutive", "random"):
        tot_pairs = tot_base = tot_new = 0
        share_hist = np.zeros(G + 1)
        for b in range(nb):
            if grouping == "consecutive":
                start = rng.integers(0, shard.size - G)
                members = shard[start:start + G]
            else:
                members = rng.choice(shard, size=G, replace=False)
            traces = [oracle.push_trace(w, od, idg, sd, 0.1, 1e-5) for sd in members]
            ptr = [0] * G
            while True:
                cand = [(deg[t[p]], t[p], j) for j, (t, p) in enumerate(zip(traces, ptr)) if p < len(t)]
                if not cand:
                    break
                d, u, _ = min(cand)
                mask = [j for (dd, uu, j) in cand if uu == u]
                k = len(mask)
                for j in mask:
                    ptr[j] += 1
                pairs = d * k
                tot_pairs += pairs
                tot_base += pairs * (64 + 32)            # one 64-B read + 32-B write-back per (edge, seed)
                shared = d * (2 * 16 * G)                # read+write of the whole [G] block per edge
                tot_new += min(shared, pairs * (64 + 32))
                share_hist[k] += pairs
        print("G=%d %-11s bundles=%d  pairs=%.3g  mean sharing (edge-weighted k/G)=%.3f  state bytes/pair: dense-per-seed %.1f -> bundled %.1f  (x%.2f)"
              % (G, grouping, nb, tot_pairs, (share_hist * np.arange(G + 1)).sum() / max(tot_pairs, 1) / G,
                 tot_base / tot_pairs, tot_new / tot_pairs, tot_base / tot_new), flush=True)
