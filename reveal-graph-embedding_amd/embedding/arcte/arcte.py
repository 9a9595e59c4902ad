"""Mirror of reveal_graph_embedding/embedding/arcte/arcte.py for the ARCTE driver
(reference lines 14-50, 279-388, 591-688).  The propagation, the effective-epsilon rule and the
community extraction all run in HIP kernels behind reveal_graph_embedding_amd._native."""
import itertools
import os
import threading

import numpy as np
import scipy.sparse as sparse

from reveal_graph_embedding_amd import _native


def parallel_chunks(l, n):
    for thread_id in range(n):
        yield roundrobin_chunks(l, n, thread_id)


def roundrobin_chunks(l, n, id):
    l_c = iter(l)
    x = list(itertools.islice(l_c, id, None, n))
    if len(x):
        return x


def calculate_epsilon_effective(rho, epsilon, seed_degree, neighbor_degrees, mean_degree):
    """
    Semi-automatic effective epsilon threshold calculation (reference arcte.py:26-50).

    Evaluated by the same kernel arcte_worker uses (numpy's pairwise summation order for the mean, device log); no
    context, no slots.  rho and mean_degree are unused, as in the reference.
    """
    nd = np.ascontiguousarray(neighbor_degrees, dtype=np.float64).reshape(-1)
    if nd.size == 0:
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    return _native.epsilon_effective_scalar(epsilon, seed_degree, nd)


def _seed_matrix(n, seeds, colptr, rows):
    """n x n CSR of ones with column seeds[k] = rows[colptr[k]:colptr[k+1]] (reference arcte.py:379-388).
    With ascending seeds the column-compressed result IS a CSC matrix, and one linear-time transpose gives
    the canonical CSR; any other order goes through COO like the reference does."""
    seeds = np.asarray(seeds, dtype=np.int64)
    if seeds.size == 0 or np.all(np.diff(seeds) > 0):
        counts = np.zeros(n + 1, dtype=np.int64)
        counts[seeds + 1] = np.diff(colptr)
        indptr = np.cumsum(counts)
        index_dtype = np.int32 if max(n, int(indptr[-1])) < 2 ** 31 else np.int64
        features = sparse.csc_matrix((np.ones(rows.size, dtype=np.float64), rows.astype(index_dtype, copy=False),
                                      indptr.astype(index_dtype)), shape=(n, n))
        return features.tocsr()
    cols = np.repeat(seeds, np.diff(colptr))
    features = sparse.coo_matrix((np.ones(rows.size, dtype=np.float64), (rows.astype(np.int64), cols)), shape=(n, n))
    return sparse.csr_matrix(features)


def _set_self_loop_values(features, loop_nodes):
    """The device-assembled pattern stores ONE diagonal entry for a node with a self-loop; the reference's
    identity + ones (arcte.py:676-679) makes that value 2.0."""
    for i in loop_nodes:
        lo, hi = features.indptr[i], features.indptr[i + 1]
        features.data[lo + np.searchsorted(features.indices[lo:hi], i)] = 2.0
    return features


def _finish_base_values(features, rw_transition):
    """(host-assembled matrices already carry the 2.0 of I + ones)"""
    if rw_transition is None:
        return features
    n = rw_transition.shape[0]
    row_of = np.repeat(np.arange(n), np.diff(rw_transition.indptr))
    return _set_self_loop_values(features, row_of[rw_transition.indices == row_of])


class _OnesInBackground:
    """float64 ones of the given length, written by a few threads (ndarray.fill releases the GIL) while the caller
    waits for the device."""

    def __init__(self, size, threads=8):
        self._out = np.empty(size, dtype=np.float64)
        threads = max(1, min(threads, size >> 22))              # one thread per 32 MB at least
        step = -(-size // threads) if size else 0
        self._threads = [threading.Thread(target=self._out[k * step:(k + 1) * step].fill, args=(1.0,))
                         for k in range(threads)] if size else []
        for t in self._threads:
            t.start()

    def result(self):
        for t in self._threads:
            t.join()
        return self._out


class _HostArraysInBackground:
    """The two large host arrays of the result matrix -- the values (float64 ones, arcte.py:381) and the column ids the device
    will copy out -- allocated for `capacity` entries and written / faulted in by background threads WHILE the GPU runs the
    rest of the seeds: a device-to-host copy into pages that have never been touched runs at a fifth of the PCIe rate
    (tools/d2h_rate.hip: 11.6 against 53.5 GB/s), and np.ones(nnz) alone took 0.48 s of arcte()'s 1.05 s on the 1M-node graph."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.ones = np.empty(self.capacity, dtype=np.float64)
        self.indices = np.empty(max(self.capacity, 1), dtype=np.int32)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = max(1, min(24, cores - 1, self.capacity >> 22))
        step = -(-self.capacity // threads) if self.capacity else 0
        self._threads = []
        for k in range(threads if self.capacity else 0):
            self._threads.append(threading.Thread(target=self._fill, args=(k * step, min(self.capacity, (k + 1) * step))))
        for t in self._threads:
            t.start()

    def _fill(self, lo, hi):
        self.ones[lo:hi].fill(1.0)
        self.indices[lo:hi].fill(0)

    def wait(self):
        for t in self._threads:
            t.join()


# a run of at least this many seeds is split into a sizing part and the rest (below)
_SPLIT_MIN_SEEDS = 65536
_SIZING_STRIDE = 8


def _features_of_run(ctx, variant, iterate_nodes, rho, epsilon, with_base_block, pattern):
    """Run `iterate_nodes` on the context and return the reference's matrix: n x n local communities
    (arcte.py:379-388) or, with the base block, arcte()'s n x 2n [I + pattern | local] (arcte.py:676-683).
    `pattern()` supplies the n x n pattern of ones for the host-assembly fallback only."""
    iterate_nodes = np.asarray(iterate_nodes, dtype=np.int64).reshape(-1)
    number_of_nodes = ctx.n
    laziness_factor = 0.5
    if variant == _native.LAZY_PAGERANK:
        rho = (rho*(0.5))/(1-(0.5*rho))          # lazy_rho, reference arcte.py:109
    unique = with_base_block or np.unique(iterate_nodes).size == iterate_nodes.size
    result_order = iterate_nodes                    # the seeds in the order the context lists their columns
    if unique and iterate_nodes.size >= _SPLIT_MIN_SEEDS and not os.environ.get("ARCTE_HIP_NO_SPLIT_RUN"):
        # Large runs go in TWO launches: every eighth seed first (the list's own mix: its emitted rows times eight size the
        # result within a per cent), then the others -- and while those run, background threads allocate, fill and fault in the
        # host arrays the result will land in.  The device assembly + copy-out that follows finds touched pages.
        import time
        verbose = bool(os.environ.get("ARCTE_HIP_VERBOSE"))
        t0 = time.perf_counter()
        sizing = iterate_nodes[::_SIZING_STRIDE]
        keep = np.ones(iterate_nodes.size, dtype=bool)
        keep[::_SIZING_STRIDE] = False
        result_order = np.concatenate([iterate_nodes[keep], sizing])          # (an appended run lists its own seeds first)
        ctx.run_seeds(sizing, rho, epsilon, use_effective_epsilon=True, variant=variant, laziness_factor=laziness_factor)
        _, rows_first = ctx.result_sizes()
        base_entries = ctx.result_csr_size(with_base_block) - rows_first
        estimate = base_entries + int(rows_first * (iterate_nodes.size / max(sizing.size, 1)) * 1.04) + (1 << 20)
        t1 = time.perf_counter()
        host = _HostArraysInBackground(estimate)
        try:
            ctx.run_seeds(iterate_nodes[keep], rho, epsilon, use_effective_epsilon=True, variant=variant,
                          laziness_factor=laziness_factor, append=True)
            t2 = time.perf_counter()
        finally:
            host.wait()
        t3 = time.perf_counter()
        nnz = ctx.result_csr_size(with_base_block)
        if nnz <= host.capacity:
            try:
                indptr, indices = ctx.fetch_csr(with_base_block, out_indices=host.indices)
                if verbose:
                    import sys
                    print("[arcte] sizing part %.3f s, the rest %.3f s, waited %.3f s for the host arrays (%d entries, estimate %d), "
                          "device assembly + copy-out %.3f s" % (t1 - t0, t2 - t1, t3 - t2, nnz, host.capacity, time.perf_counter() - t3),
                          file=sys.stderr, flush=True)
            except _native.ArcteHipError as e:
                if e.code != -3:                  # ARCTE_HIP_ECAPACITY: too many entries for the device assembly
                    raise
            else:
                width = 2 * number_of_nodes if with_base_block else number_of_nodes
                index_dtype = np.int32 if max(width, indices.size) < 2 ** 31 else np.int64
                return sparse.csr_matrix((host.ones[:indices.size], indices.astype(index_dtype, copy=False),
                                          indptr.astype(index_dtype)), shape=(number_of_nodes, width))
        del host
    else:
        ctx.run_seeds(iterate_nodes, rho, epsilon, use_effective_epsilon=True, variant=variant,
                      laziness_factor=laziness_factor)
    if unique:
        # the device sorts the (row, seed) pairs into the CSR the reference builds via COO (arcte.py:379-388); the
        # host meanwhile writes the matrix's values (all ones, arcte.py:381) -- 7 GB on the 1M-node graph
        ones = None
        try:
            ones = _OnesInBackground(ctx.result_csr_size(with_base_block))
            indptr, indices = ctx.fetch_csr(with_base_block)
        except _native.ArcteHipError as e:
            if ones is not None:
                ones.result()
            if e.code != -3:                      # ARCTE_HIP_ECAPACITY: too many entries for the device assembly
                raise
        else:
            width = 2 * number_of_nodes if with_base_block else number_of_nodes
            index_dtype = np.int32 if max(width, indices.size) < 2 ** 31 else np.int64
            return sparse.csr_matrix((ones.result()[:indices.size], indices.astype(index_dtype, copy=False),
                                      indptr.astype(index_dtype)), shape=(number_of_nodes, width))
    colptr, rows = ctx.fetch()
    local = _seed_matrix(number_of_nodes, result_order, colptr, rows)
    if not with_base_block:
        return local
    base = sparse.csr_matrix(sparse.eye(number_of_nodes, number_of_nodes, dtype=np.float64)) + pattern()
    return _finish_base_values(sparse.hstack([base, local]).tocsr(), None)


def _worker(variant, iterate_nodes, indices_c, indptr_c, data_c, out_degree, in_degree, rho, epsilon, device):
    number_of_nodes = out_degree.size
    with _native.Context(indptr_c, indices_c, data_c, out_degree, in_degree, device=device) as ctx:
        return _features_of_run(ctx, variant, iterate_nodes, rho, epsilon, False, None)


def arcte_worker(iterate_nodes, indices_c, indptr_c, data_c, out_degree, in_degree, rho, epsilon, device=0):
    """
    Local community features of the seeds in `iterate_nodes` (reference arcte.py:279-388): n x n CSR of
    ones whose column j holds the local community of seed j.  Takes the flat CSR arrays of the
    random-walk matrix exactly as the reference's pool workers do; `device` picks the GPU.
    """
    return _worker(_native.ARCTE, iterate_nodes, indices_c, indptr_c, data_c, out_degree, in_degree, rho, epsilon,
                   device)


def arcte_with_pagerank_worker(iterate_nodes, indices_c, indptr_c, data_c, out_degree, in_degree, rho, epsilon,
                               device=0):
    """The PageRank-push flavour of arcte_worker (reference arcte.py:166-276)."""
    return _worker(_native.PAGERANK, iterate_nodes, indices_c, indptr_c, data_c, out_degree, in_degree, rho,
                   epsilon, device)


def arcte_with_lazy_pagerank_worker(iterate_nodes, indices_c, indptr_c, data_c, out_degree, in_degree, rho, epsilon,
                                    device=0):
    """The lazy-PageRank-push flavour of arcte_worker (reference arcte.py:53-163; lazy_rho of :109, laziness 0.5)."""
    return _worker(_native.LAZY_PAGERANK, iterate_nodes, indices_c, indptr_c, data_c, out_degree, in_degree, rho,
                   epsilon, device)


def seed_nodes(adjacency_matrix):
    """Seed list of the reference (arcte.py:610-617): nodes whose pattern in-count exceeds 1, by
    descending count.  Ties are ordered by node id here (the reference's unstable argsort leaves
    them unspecified; the order only decides which worker handles a seed)."""
    a = sparse.csr_matrix(adjacency_matrix)
    edge_count_vector = np.bincount(a.indices, minlength=a.shape[1]).astype(np.int64)
    iterate_nodes = np.where(edge_count_vector > 1)[0]
    order = np.argsort(-edge_count_vector[iterate_nodes], kind="stable")
    return iterate_nodes[order]


_VARIANT_OF = {arcte_worker: _native.ARCTE, arcte_with_pagerank_worker: _native.PAGERANK,
               arcte_with_lazy_pagerank_worker: _native.LAZY_PAGERANK}


def arcte_with_pagerank(adjacency_matrix, rho, epsilon, number_of_threads=None):
    """arcte() with PageRank pushes (reference arcte.py:490-588): same driver, different worker."""
    return _arcte_driver(adjacency_matrix, rho, epsilon, number_of_threads, arcte_with_pagerank_worker)


def arcte_with_lazy_pagerank(adjacency_matrix, rho, epsilon, number_of_threads=None):
    """arcte() with lazy PageRank pushes (reference arcte.py:391-489): same driver, different worker."""
    return _arcte_driver(adjacency_matrix, rho, epsilon, number_of_threads, arcte_with_lazy_pagerank_worker)


def arcte(adjacency_matrix, rho, epsilon, number_of_threads=None):
    """
    Extracts local community features for all graph nodes based on the partitioning of node-centric
    similarity vectors (reference arcte.py:591-688).

    Inputs:  - A in R^(nxn): adjacency matrix (any scipy sparse format).
             - rho: restart probability.
             - epsilon: approximation threshold.
             - number_of_threads: the reference's process count.  Here it bounds the number of GPUs
               used (None = every visible GPU); seeds are dealt round-robin over them exactly like
               the reference deals them over its processes (arcte.py:14-23, 650-651).

    Outputs: - X in R^(nx2n): CSR; columns [0, n) are the base communities I + pattern(A),
               columns [n, 2n) the ARCTE local communities.
    """
    return _arcte_driver(adjacency_matrix, rho, epsilon, number_of_threads, arcte_worker)


def _arcte_driver(adjacency_matrix, rho, epsilon, number_of_threads, worker):
    adjacency_matrix = sparse.csr_matrix(adjacency_matrix, dtype=np.float64)
    number_of_nodes = adjacency_matrix.shape[0]

    n_gpus = _native.device_count()
    if n_gpus < 1:
        raise _native.ArcteHipError(-2, "no HIP device visible (there is no CPU fallback)")
    devices = list(range(n_gpus))
    if os.environ.get("ARCTE_HIP_DEVICES"):
        # explicit placement, one worker per listed device id (ids may repeat: several workers on one GPU)
        devices = [int(x) for x in os.environ["ARCTE_HIP_DEVICES"].split(",")]
    if number_of_threads is not None:
        devices = devices[:max(1, int(number_of_threads))]
    n_gpus = len(devices)
    variant = _VARIANT_OF[worker]

    def pattern():
        ones = adjacency_matrix.copy()
        ones.data = np.ones_like(ones.data, dtype=np.float64)
        return ones

    def find_self_loops():
        row_of = np.repeat(np.arange(number_of_nodes), np.diff(adjacency_matrix.indptr))
        return row_of[adjacency_matrix.indices == row_of]

    # (0.2 s of numpy on the 1M-node graph, off the critical path: it runs while the GPU does)
    loops_box = []
    loops_thread = threading.Thread(target=lambda: loops_box.append(find_self_loops()))
    loops_thread.start()

    def self_loops():
        loops_thread.join()
        return loops_box[0]

    # The adjacency matrix goes to the GPU as it is: the transition matrix, both degree vectors and the seed list
    # (arcte.py:608-617) are made there (arcte_hip_create_from_adjacency) and never come back.
    if n_gpus == 1:
        with _native.Context.from_adjacency(adjacency_matrix.indptr, adjacency_matrix.indices, adjacency_matrix.data,
                                            device=devices[0]) as ctx:
            # the library orders the work heaviest-first by itself; ascending ids make the result a CSC matrix.
            # The whole n x 2n pattern [I + pattern(A) | local communities] (arcte.py:676-683) is assembled on the
            # device; the only values that are not 1 are the diagonal entries of nodes with a self-loop (I + ones).
            features = _features_of_run(ctx, variant, np.sort(ctx.seed_list()), rho, epsilon, True, pattern)
        return _set_self_loop_values(features, self_loops())

    # More than one GPU: every worker prepares the graph on its GPU and runs its round-robin chunk of the seed list
    # (arcte.py:650-666).  The reference's parent then sums the workers' matrices (:670-673); every seed owns its column,
    # so here the first worker's context takes the others' rows straight from their GPUs (arcte_hip_append_result) and
    # assembles the whole n x 2n matrix on its device, exactly as the one-GPU path does.
    contexts = [None] * n_gpus
    chunks = [None] * n_gpus
    errors = []

    def work(k):
        try:
            ctx = _native.Context.from_adjacency(adjacency_matrix.indptr, adjacency_matrix.indices, adjacency_matrix.data,
                                                 device=devices[k])
            contexts[k] = ctx
            chunk = roundrobin_chunks(ctx.seed_list(), n_gpus, k)          # arcte.py:650-651
            chunks[k] = np.sort(np.asarray(chunk if chunk is not None else [], dtype=np.int64))
            run_rho = (rho*(0.5))/(1-(0.5*rho)) if variant == _native.LAZY_PAGERANK else rho      # arcte.py:109
            ctx.run_seeds(chunks[k], run_rho, epsilon, use_effective_epsilon=True, variant=variant, laziness_factor=0.5)
        except BaseException as e:  # surfaced below; the reference drops worker errors silently
            errors.append(e)

    try:
        threads = [threading.Thread(target=work, args=(k,)) for k in range(n_gpus)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        first = contexts[0]
        for k in range(1, n_gpus):
            _, total = contexts[k].result_sizes()
            first.append_result(chunks[k], np.diff(contexts[k].colptr()), contexts[k].result_device_rows(), nrows=total)
            contexts[k].close()
            contexts[k] = None
        ones = _OnesInBackground(first.result_csr_size(True))
        try:
            indptr, indices = first.fetch_csr(True)
        except _native.ArcteHipError as e:
            ones.result()
            if e.code != -3:                      # ARCTE_HIP_ECAPACITY: too many entries for the device assembly
                raise
            colptr, rows = first.fetch()
            seeds_all = np.concatenate(chunks)
            local = _seed_matrix(number_of_nodes, seeds_all, colptr, rows)
            base = sparse.csr_matrix(sparse.eye(number_of_nodes, number_of_nodes, dtype=np.float64)) + pattern()
            return sparse.hstack([base, local]).tocsr()                  # arcte.py:676-683
    finally:
        for ctx in contexts:
            if ctx is not None:
                ctx.close()
    width = 2 * number_of_nodes
    index_dtype = np.int32 if max(width, indices.size) < 2 ** 31 else np.int64
    features = sparse.csr_matrix((ones.result()[:indices.size], indices.astype(index_dtype, copy=False), indptr.astype(index_dtype)),
                                 shape=(number_of_nodes, width))
    return _set_self_loop_values(features, self_loops())
