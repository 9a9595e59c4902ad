#!/usr/bin/env python3
"""ARCTE hot-path benchmark: seed-vertices/sec on a synthetic R-MAT power-law graph.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run (one rank per GPU, backend nccl = RCCL).

Workload (BASELINE.json metric: "seed-vertices/sec ... 1M-node power-law graph", configs[2]/[3]):
R-MAT n=1,000,000 / 50,000,000 sampled edges (SURVEY.md 8(d) recipe), rho=0.1, eps=1e-5, float64, ALL
651 465 seeds.  The reference's degree-descending seed list is dealt round-robin over the N ranks exactly as
the reference deals it over processes (arcte.py:14-23): rank r of N runs seeds[r::N].  One STEP = one pass of
the whole hot path (effective epsilon -> exact-FIFO eps-push -> community extraction -> column-compressed
result in HBM) over every seed of the graph, inputs resident in HBM, plus -- for N > 1 -- the gather of every
rank's result on rank 0 over RCCL.  Total work is fixed as N grows ("strong"): N = 1 is configs[2], N = 8 is
configs[3].  `--shards S` (S > N) runs only shards 0..N-1 of S (a shorter step for profiling passes).

Rank 0 prints ONE JSON line; `roofline` prices the dominant kernel (k_arcte_lines) by ALGORITHMIC
bytes (52 B/edge + 36 B/push + 4 B/enqueue + 36 B/support entry, counted by the
kernel itself) over its HIP-event duration -- `frac` -- and by the push-only model of the same section
(52 B/edge + 36 B/push: the fused kernel does not move the 36 B per support entry) -- `frac_push_only`;
`cpu_baseline` times the CPU oracle (a C port of the reference's algorithm, OpenMP over seeds) on a
bounded sample of the same shard.  For N > 1 a step runs as `--sub-launches` launches over interleaved
parts of the rank's shard; the rows of a finished part travel to rank 0 while the next part runs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
KERNEL_SOURCES = ("reveal-graph-embedding_amd/csrc/arcte_kernels.hpp", "reveal-graph-embedding_amd/csrc/arcte_lines.hpp",
                  "reveal-graph-embedding_amd/csrc/arcte_prepare.hpp", "reveal-graph-embedding_amd/csrc/arcte_features.hpp",
                  "reveal-graph-embedding_amd/csrc/arcte_hip.hip")


def kernel_source_id():
    """Identifies the kernel a PMC measurement belongs to (the GPU box has no .git): hash of the device sources."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(st):
    return 52 * st["edges"] + 36 * st["pushes"] + 4 * st["enqueues"] + 36 * st["support"]


def load_graph(n, m, seed, rank, barrier):
    """R-MAT adjacency (CSR).  Generated once per node and cached under /tmp for the other ranks/runs."""
    import scipy.sparse as sparse
    from reveal_graph_embedding_amd.synthetic import rmat_graph
    path = "/tmp/arcte_rmat_%d_%d_%d.npz" % (n, m, seed)
    if not os.path.exists(path) and rank == 0:
        t = time.time()
        a = rmat_graph(n, m, seed)
        tmp = path + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, indptr=a.indptr, indices=a.indices)
        os.replace(tmp, path)
        log("[bench] generated R-MAT n=%d m=%d nnz=%d in %.1fs" % (n, m, a.nnz, time.time() - t))
    barrier()
    z = np.load(path)
    indices = z["indices"]
    return sparse.csr_matrix((np.ones(indices.size, dtype=np.float64), indices, z["indptr"]), shape=(n, n))


def cpu_baseline(w, out_degree, in_degree, shard, rho, eps, budget_s):
    from oracle import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, oracle.lib().oracle_max_threads()))
    rng = np.random.default_rng(0)
    count = min(shard.size, 8 * cores)
    while True:
        sample = np.sort(rng.choice(shard.size, size=count, replace=False))
        t = time.perf_counter()
        _, _, _, _, stats = oracle.worker(w, out_degree, in_degree, shard[sample], rho, eps, threads=cores,
                                          want_stats=True)
        dt = max(time.perf_counter() - t, 1e-3)
        if dt >= 0.6 * budget_s or count >= shard.size:
            break
        count = int(min(shard.size, count * min(max(budget_s / dt, 2.0), 16.0)))
    st = dict(pushes=int(stats[0]), edges=int(stats[1]), enqueues=int(stats[2]), support=int(stats[3]))
    return {"value": count / dt, "unit": "seeds/s", "cores": cores, "kind": "port",
            "sample": "%d seeds drawn uniformly from the step's seeds, %.1f s, OpenMP over seeds" % (count, dt),
            "algorithmic_GBps": algorithmic_bytes(st) / dt / 1e9,
            # how the C port relates to the reference's own Python path, measured in the build container on
            # identical inputs (BASELINE.md / DESIGN.md section 5; the reference cannot travel to the GPU box)
            "port_vs_reference": {"ba20000_m10_1_thread": 52, "rmat100k_2M_8_threads": 11, "unit": "x faster than the reference",
                                  "where": "build container, 8-core Xeon 2.1 GHz"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=1000000)
    ap.add_argument("--edges", type=int, default=50000000)
    ap.add_argument("--shards", type=int, default=0,
                    help="round-robin shards of the seed list, rank r runs shard r (default 0 = one shard per GPU: every seed)")
    ap.add_argument("--rho", type=float, default=0.1)
    ap.add_argument("--epsilon", type=float, default=1e-5)
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--ballast-gb", type=float, default=0.0,
                    help="experiment: hold this much device memory beside the context (how the kernel's duration depends on the device's fill)")
    ap.add_argument("--placement-tries", type=int, default=1,
                    help="contexts drawn before the run, the fastest on a calibration sample is kept; 1 (default) = take the "
                         "first, which is what arcte(), arcte_worker() and the console script do")
    ap.add_argument("--gather", choices=["rows", "counts"], default="rows",
                    help="N>1: what rank 0 collects per step (rows = the full result)")
    ap.add_argument("--sub-launches", type=int, default=0,
                    help="launches per step (default: 4 for N>1 -- the sends of a finished part overlap the next launch -- else 1)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend; gloo (host-staged gather, ranks may share a GPU) is only for "
                         "rehearsing the N>1 code path on a box with fewer GPUs than ranks")
    ap.add_argument("--variant", choices=["arcte", "pagerank", "lazy"], default="arcte",
                    help="push flavour (default: ARCTE's cumulative PageRank difference = the BASELINE metric)")
    ap.add_argument("--float32", action="store_true", help="float32 arithmetic (tolerance sweep only; not the metric)")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed steps, rank 0 merges what it gathered into arcte()'s n x 2n matrix and reports the "
                         "SHA-256 of its canonical CSR (tests compare it with the reference's own run)")
    args = ap.parse_args()

    # (dmabuf IPC is the only kind the host driver of this pool supports: RCCL's peer buffers need it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if args.shards <= 0:
        args.shards = world
    if args.shards < world:
        raise SystemExit("--shards must be >= the number of GPUs")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit("rank %d has no GPU of its own (%d visible): RCCL needs one GPU per rank" % (rank, ndev))
    gpu = local_rank % ndev
    if world > ndev:
        # The one-GPU rehearsal of the multi-rank path (--backend gloo): ranks SHARE a GPU.  The library's placement draw holds up to
        # four candidate allocations of the slot memory at once (4 x 60 GB on the 1M/50M graph); another PROCESS setting up meanwhile
        # sees a full device and sizes itself to one wavefront per CU (profiles/r04: 2 059 ms per launch).  One candidate per rank here.
        os.environ.setdefault("ARCTE_HIP_SPREAD_TRIES", "1")
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    from reveal_graph_embedding_amd import _native
    from reveal_graph_embedding_amd.distributed import gather_shards, shard_seeds

    # the adjacency matrix goes to the GPU as it is; transition matrix, degree vectors and seed list are made there
    adjacency = load_graph(args.nodes, args.edges, 0, local_rank, barrier)
    nnz = int(adjacency.nnz)
    w_indptr, w_indices = adjacency.indptr, adjacency.indices          # pattern of W = pattern of A (--verify)
    ballast = torch.empty(int(args.ballast_gb * 1e9), dtype=torch.uint8, device=dev) if args.ballast_gb > 0 else None
    if ballast is not None:
        ballast.fill_(1)

    def make_context():
        return _native.Context.from_adjacency(adjacency.indptr, adjacency.indices, adjacency.data, device=gpu, n_slots=args.slots)

    def calibrate(c):
        # kernel time of every 16th seed of this rank's shard, best of two (all of this happens before the warm-up)
        sample = shard_seeds(c.seed_list(), args.shards, rank)[::16]
        best = float("inf")
        for _ in range(2):
            c.run_seeds(sample, args.rho, args.epsilon, use_effective_epsilon=True)
            best = min(best, c.timing()["push_ms"])
        return best

    if args.placement_tries > 1:
        ctx, placement_ms = _native.fastest_context(make_context, calibrate, tries=args.placement_tries)
    else:
        ctx, placement_ms = make_context(), []          # what arcte() does: the first context (the library places its slot memory)
    del adjacency
    seeds = ctx.seed_list()
    shard = shard_seeds(seeds, args.shards, rank)
    info = ctx.info()
    state = ctx.state_info()
    placement_kept, placement_rates = ctx.placement_info()
    mem = _native.memory_info(gpu)
    mem_free, mem_total = mem["free_bytes"], mem["total_bytes"]
    log("[bench] rank %d: n=%d nnz=%d seeds=%d shard=%d slots=%d context %.1f GB + parked draw losers %.1f GB + cached %.1f GB "
        "(device: %.1f of %.1f GB in use)" % (
            rank, args.nodes, nnz, seeds.size, shard.size, info["slots"], info["device_bytes"] / 1e9, mem["parked_bytes"] / 1e9,
            mem["cached_bytes"] / 1e9, (mem_total - mem_free) / 1e9, mem_total / 1e9))

    variant = {"arcte": _native.ARCTE, "pagerank": _native.PAGERANK, "lazy": _native.LAZY_PAGERANK}[args.variant]
    run_rho = (args.rho * 0.5) / (1 - 0.5 * args.rho) if args.variant == "lazy" else args.rho   # arcte.py:109
    if args.float32:
        ctx.set_float32(True)
    from reveal_graph_embedding_amd.distributed import gather_shards_begin, gather_shards_end
    nsub = args.sub_launches if args.sub_launches > 0 else (4 if world > 1 else 1)
    parts = [np.ascontiguousarray(shard[j::nsub]) for j in range(nsub)]       # interleaved: every part has the shard's mix

    def step():
        """One pass over the rank's shard.  Returns what the step did: kernel ms, counters, emitted rows, gather time."""
        out = dict(push_ms=0.0, gather_ms=0.0, gathered_bytes=0, gathered_rows=0, rows=0,
                   stats=dict(pushes=0, edges=0, enqueues=0, support=0, candidates=0, reruns=0),
                   updates=dict(lds_updates=0, blind_line_writes=0, line_read_modify_writes=0, pushed_node_updates=0))
        pending = []
        for part in parts:
            ctx.run_seeds(part, run_rho, args.epsilon, use_effective_epsilon=True, variant=variant)
            out["push_ms"] += ctx.timing()["push_ms"]
            st_ = ctx.stats()
            for k in out["stats"]:
                out["stats"][k] += st_[k]
            up_ = ctx.state_info()
            for k in out["updates"]:
                out["updates"][k] += up_[k]
            _, total = ctx.result_sizes()
            out["rows"] += int(total)
            if world > 1:
                # the part's rows leave for rank 0 now and travel while the next part runs
                t = time.perf_counter()
                counts_t = torch.from_numpy(np.diff(ctx.colptr())).to(comm_dev)
                if args.gather == "rows":
                    rows_t = torch.empty(total, dtype=torch.int32, device=dev)
                    ctx.copy_rows_to_device(rows_t.data_ptr(), total)
                    rows_t = rows_t.to(comm_dev)
                else:
                    rows_t = torch.empty(0, dtype=torch.int32, device=comm_dev)
                pending.append(gather_shards_begin(counts_t, rows_t, dst=0))
                out["gather_ms"] += (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        for h in pending:
            got = gather_shards_end(h)
            out["gathered_bytes"] += h["bytes"]
            if got is not None:
                out["gathered_rows"] += sum(int(r.numel()) for _, r in got)
        out["gather_ms"] += (time.perf_counter() - t) * 1e3
        return out

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    push_ms = 0.0
    gather_ms = 0.0
    step_kernel_ms = []
    last = None
    for _ in range(args.steps):
        last = step()
        step_kernel_ms.append(last["push_ms"])
        push_ms += last["push_ms"]
        gather_ms += last["gather_ms"]
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        nseeds_t = torch.tensor([shard.size], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(nseeds_t, op=dist.ReduceOp.SUM)
        seeds_per_step = int(nseeds_t.item())
    else:
        seeds_per_step = int(shard.size)
    gathered_rows = last["gathered_rows"] if last else 0

    st = last["stats"]
    tm = ctx.timing()
    upd = last["updates"]
    total_rows = last["rows"]
    if world > 1:
        rows_t = torch.tensor([total_rows], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(rows_t, op=dist.ReduceOp.SUM)
        emitted_all = int(rows_t.item())
    else:
        emitted_all = int(total_rows)
    merged_sha = None
    if args.verify:
        # (after the timed region) one plain pass over the whole shard; rank 0's context takes the other ranks' parts
        # from where the transport left them and assembles arcte()'s n x 2n matrix on its device -- the path
        # distributed.arcte_distributed takes
        import hashlib
        mine = np.sort(shard)
        ctx.run_seeds(mine, run_rho, args.epsilon, use_effective_epsilon=True, variant=variant)
        if world > 1:
            _, total_v = ctx.result_sizes()
            counts_t = torch.from_numpy(np.diff(ctx.colptr())).to(comm_dev)
            rows_t = torch.empty(total_v, dtype=torch.int32, device=dev)
            ctx.copy_rows_to_device(rows_t.data_ptr(), total_v)
            got = gather_shards(counts_t, rows_t.to(comm_dev), dst=0)
            if rank == 0:
                for k in range(1, world):
                    ck, rk = got[k]
                    ctx.append_result(np.sort(shard_seeds(seeds, args.shards, k)), ck.cpu().numpy(), rk.data_ptr(), nrows=rk.numel())
        if rank == 0:
            v_indptr, v_indices = ctx.fetch_csr(True)
            h = hashlib.sha256()
            h.update(np.asarray(v_indptr, dtype=np.int64).tobytes())
            h.update(np.asarray(v_indices, dtype=np.int64).tobytes())
            merged_sha = h.hexdigest()
        ctx.run_seeds(parts[-1], run_rho, args.epsilon, use_effective_epsilon=True, variant=variant)   # (the fetch below times a part)
    # D2H of the step's result: reported beside the metric, never inside it.  Twice: into a fresh numpy array (its pages are
    # faulted in by the copy itself: tools/d2h_rate.hip, 11.6 GB/s) and into that same array again (touched pages: what a caller
    # that reuses its result buffer -- or faults it in while the GPU runs, as arcte() does -- pays)
    t = time.perf_counter()
    _, rows_host = ctx.fetch()
    fetch_cold_ms = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    ctx.fetch(out_rows=rows_host)
    fetch_ms = (time.perf_counter() - t) * 1e3
    del rows_host
    if rank == 0:
        alg = algorithmic_bytes(st)
        alg_push_only = 52 * st["edges"] + 36 * st["pushes"]          # "the eps-push-kernel-only figure"
        kernel_ms = push_ms / max(args.steps, 1)
        achieved = alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        achieved_push_only = alg_push_only / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes (tools/rocprof_passes.sh): only when they were taken from THIS
        # kernel source on this workload; a stale measurement is dropped, never reported
        traffic = None
        traffic_note = "no PMC measurement of this kernel source on this workload"
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path) and args.variant == "arcte" and not args.float32:
            try:
                pmc = json.load(open(pmc_path))
                key = "n%d_m%d_shards%d_of_%d" % (args.nodes, args.edges, world, args.shards)
                if key in pmc and pmc[key].get("kernel_source_id") == kernel_source_id():
                    # the guide's gfx950 correction: FETCH_SIZE tallies the 128-byte requests of a wide coalesced read at
                    # 64 bytes, i.e. reports half of such a stream.  Here that stream is the rows (8 bytes per traversed
                    # edge on narrow rows, 20 otherwise); the random 8-byte state reads are one 64-byte request each and
                    # are counted as they are.  WRITE_SIZE needs no correction.
                    stream = {0: 20, 1: 8, 2: 4}[info["narrow_rows"]] * st["edges"]
                    traffic = pmc[key]["fetch_bytes"] + pmc[key]["write_bytes"] + 0.5 * stream
                    traffic_note = ("FETCH_SIZE + WRITE_SIZE of %s (commit %s), plus half of the coalesced row stream (%d bytes per "
                                    "traversed edge) that FETCH_SIZE under-reports by 2x on gfx950; measured under the profiler in "
                                    "another process, stamped with this kernel source" % (
                                        pmc[key].get("source"), pmc[key].get("commit"), {0: 20, 1: 8, 2: 4}[info["narrow_rows"]]))
                elif key in pmc:
                    traffic_note = "stale: measured at kernel source %s, running %s" % (pmc[key].get("kernel_source_id"), kernel_source_id())
            except Exception:
                traffic = None
        stream_read, stream_copy = _native.stream_bandwidth(gpu)
        result = {
            "metric": "seed-vertices/sec",
            "value": seeds_per_step * args.steps / elapsed,
            "unit": "seeds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.shards == world else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.float32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": "R-MAT n=%d m=%d (nnz %d, %d seeds), rho=%g eps=%g; step = %s through eps_eff -> eps-push -> "
                            "extraction%s"
                            % (args.nodes, args.edges, nnz, seeds.size, args.rho, args.epsilon,
                               ("all %d seeds" % seeds.size + (", dealt round-robin over the degree-descending seed list, "
                                                              "rank r runs seeds[r::%d]" % world if world > 1 else ""))
                               if args.shards == world else
                               "seed shard %s of %d (round-robin over the degree-descending seed list) per GPU" % (
                                   "r" if world > 1 else "0", args.shards),
                               " + %s gather of %s on rank 0" % ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)",
                                                                  args.gather) if world > 1 else ""),
                "variant": args.variant,
                "seeds_per_step": seeds_per_step,
                "shards": args.shards,
                "slots_per_gpu": info["slots"], "waves_per_cu": info["waves_per_cu"],
                "hot_values_per_wave": info["hot_values_per_wave"], "narrow_rows": info["narrow_rows"],
                "warm_end_rank": info["warm_end_rank"],
                "state": {k: state[k] for k in ("line_state", "lines_per_slot", "pushed_capacity", "candidate_capacity", "slot_bytes",
                                                "bitmap_lds_bytes", "lds_bytes_per_wave", "lines_region_b", "region_b_indirect", "region_b_pool_lines")},
                "kernel_source_id": kernel_source_id(),
                # what the process holds on the device once the context exists: the context's own buffers, the losers of the
                # slot-memory draw that stay allocated (at most one) and buffers kept from destroyed contexts
                "device_memory": {"context_bytes": info["device_bytes"], "parked_bytes": mem["parked_bytes"],
                                  "cached_bytes": mem["cached_bytes"], "in_use_bytes_after_create": mem_total - mem_free,
                                  "total_bytes": mem_total},
                "placement_tries": len(placement_ms), "placement_calibration_ms": [round(x, 2) for x in placement_ms],
                # the library's own draw of the slot memory at context creation (every caller gets it): G updates/s of the
                # probe on each candidate allocation, and which one was kept
                "slot_memory_probe_gups": [round(x, 2) for x in placement_rates], "slot_memory_kept": placement_kept,
                "emitted_rows_rank0": int(total_rows), "emitted_rows_all_ranks": emitted_all, "merged_sha256": merged_sha,
                "gathered_rows_rank0": int(gathered_rows),
                "sub_launches": nsub, "gather_ms_per_step_rank0": gather_ms / max(args.steps, 1),
                "gathered_bytes_per_step_rank0": int(last["gathered_bytes"]) if last else 0,
                "per_seed": {k: st[k] / max(shard.size, 1) for k in ("pushes", "edges", "enqueues", "support", "candidates")},
                "reruns": st["reruns"],
                "eps_kernel_ms": tm["eps_ms"], "compact_ms": tm["compact_ms"], "call_ms": tm["call_ms"],
                "result_d2h_ms_rank0": fetch_ms, "result_d2h_into_untouched_pages_ms_rank0": fetch_cold_ms,
                "pcie_inclusive_seeds_per_s_rank0": shard.size / ((elapsed / max(args.steps, 1)) + fetch_ms * 1e-3),
            },
            "roofline": {
                "bound": "hbm",
                # (the name rocprofv3 prints: MODE, VAR, ROWS (0 wide, 1 narrow, 2 packed), TAIL, PROF, LT, WPE, STAGE, IND)
                #  WPE = 4: the 128-VGPR build of ARCTE's worker on packed rows, launched with more than twelve wavefronts per CU)
                "kernel": ("void (anonymous namespace)::k_arcte_lines<0, %d, %d, %s, false, 1, %d, false, %s>((anonymous namespace)::PushParams, "
                           "(anonymous namespace)::LineParams)" % (variant, info["narrow_rows"],
                                                                   "true" if state["lines_region_b"] else "false",
                                                                   4 if (info["waves_per_cu"] > 12 and info["narrow_rows"] == 2 and variant == 0) else 1,
                                                                   "true" if state["region_b_indirect"] else "false")) if state["line_state"] else
                          "k_arcte_seeds<0, %d, %s, %d, %s%s>" % (variant, "float" if args.float32 else "double", info["tiles"],
                                                                  "true" if info["hot_values_per_wave"] else "false",
                                                                  ", true" if info["narrow_rows"] and info["hot_values_per_wave"] else ""),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "achieved_push_only": achieved_push_only, "frac_push_only": achieved_push_only / HBM_PEAK_GBS,
                "algorithmic_bytes_push_only_per_launch": alg_push_only,
                "traffic": traffic, "traffic_note": traffic_note,
                "measured_stream_peak": {"read_GBps": stream_read, "copy_GBps": stream_copy,
                                         "frac_of_read": achieved / stream_read if stream_read > 0 else None,
                                         "how": "arcte_hip_stream_bandwidth: 16 B/lane sweep over 4 GiB on this box, best of 3"},
                "algorithmic_bytes_per_launch": alg, "kernel_ms_per_launch": kernel_ms,
                "kernel_ms_each_launch": [round(x, 3) for x in step_kernel_ms],
                # how the last launch's state updates were served (counted by the kernel), per traversed edge
                "updates_per_edge": {k: upd[k] / max(st["edges"], 1) for k in ("lds_updates", "blind_line_writes",
                                                                              "line_read_modify_writes", "pushed_node_updates")},
            },
        }
        if args.variant != "arcte":
            # the 52/36/4/36-byte model prices ARCTE's push (s and r per edge); the PageRank flavours move less
            result["roofline"]["note"] = "algorithmic-byte model is ARCTE's; indicative only for this flavour"
        if world == 1 and args.cpu_seconds > 0 and args.variant == "arcte" and not args.float32:
            import scipy.sparse as sparse
            t_indptr, t_indices, t_data, out_degree, in_degree = ctx.transition()
            w = sparse.csr_matrix((t_data, t_indices, t_indptr), shape=(args.nodes, args.nodes))
            result["cpu_baseline"] = cpu_baseline(w, out_degree, in_degree, shard, args.rho, args.epsilon, args.cpu_seconds)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
