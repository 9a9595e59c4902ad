"""Where does a wavefront of the push kernel spend its time?  Runs seed shard 0 of STRIDE of the R-MAT graph with
ARCTE_HIP_PROFILE=1 (s_memtime stamps around the phases of k_arcte_seeds, summed over wavefronts) at several launch shapes.

usage: python tools/phase_profile.py NODES EDGES STRIDE WAVES_PER_CU [WAVES_PER_CU ...]
"""
import os
import re
import subprocess
import sys

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from hot_sweep import load_graph
from reveal_graph_embedding_amd import _native
n, m, stride = (int(x) for x in sys.argv[1:4])
A = load_graph(n, m)
with _native.Context.from_adjacency(A.indptr, A.indices, A.data) as ctx:
    seeds = ctx.seed_list()[::stride]
    ctx.run_seeds(seeds, 0.1, 1e-5)
    ctx.run_seeds(seeds, 0.1, 1e-5)
    st = ctx.stats()
    print("STATS", seeds.size, st["pushes"], st["edges"], ctx.timing()["push_ms"], flush=True)
"""


def main():
    n, m, stride = sys.argv[1:4]
    for wpc in sys.argv[4:]:
        env = dict(os.environ, ARCTE_HIP_PROFILE="1", ARCTE_HIP_WAVES_PER_CU=wpc)
        r = subprocess.run([sys.executable, "-c", CHILD, n, m, stride], env=env, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-2000:])
            raise SystemExit(1)
        line = [l for l in r.stderr.splitlines() if "[arcte_hip profile]" in l][-1]
        v = {k: int(x) for k, x in re.findall(r"(\w+) (\d+)", line.split("|")[0])}
        cnt = {k: int(x) for k, x in re.findall(r"(\w+) (\d+)", line.split("|")[1])}
        stats = [l for l in r.stdout.splitlines() if l.startswith("STATS")][-1].split()
        seeds, pushes, edges, push_ms = int(stats[1]), int(stats[2]), int(stats[3]), float(stats[4])
        total = sum(v.values())
        slots = int(re.search(r"slots (\d+)", line).group(1))
        # ticks per millisecond from the kernel's own duration: every wavefront is busy for (almost) the whole launch
        per_ms = total / slots / push_ms
        print("waves/CU %s: %d seeds, %.1f ms, %d slots; s_memtime ticks %.0f per us of a wavefront" % (wpc, seeds, push_ms, slots, per_ms / 1e3))
        for k in ("setup", "pop_batches", "short_pushes", "long_pushes", "pop_loop_rest", "extraction", "draw"):
            extra = ""
            if k == "short_pushes":
                extra = "  %d pushes, %.2f us each" % (cnt["short_pushes"], v[k] / per_ms * 1e3 / max(cnt["short_pushes"], 1))
            if k == "long_pushes":
                extra = "  %d pushes, %.2f us each" % (cnt["long_pushes"], v[k] / per_ms * 1e3 / max(cnt["long_pushes"], 1))
            if k == "pop_batches":
                extra = "  %d batches, %.2f us each" % (cnt["pop_batches"], v[k] / per_ms * 1e3 / max(cnt["pop_batches"], 1))
            print("   %-14s %5.1f %%%s" % (k, 100.0 * v[k] / total, extra))


if __name__ == "__main__":
    main()
