/*
 * arcte_hip.h -- C ABI of the MI355X (gfx950) ARCTE hot path.
 *
 * The reference (MKLab-ITI/reveal-graph-embedding) has no FFI of its own for this
 * path: its native seam is the flat-array signature of arcte_worker().  Every entry
 * point below names the reference interface it replaces (paths relative to
 * reveal_graph_embedding/ in the reference tree).  INTEGRATION.md shows the ctypes
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ARCTE_HIP_E* code otherwise;
 *     arcte_hip_last_error() then holds a message (thread-local, never NULL);
 *   - host arrays are borrowed for the duration of the call only; device memory
 *     belongs to the context; no exceptions cross the boundary;
 *   - one context per GPU, calls on one context must be serialised by the caller;
 *   - node ids are int32 (n < 2^31), CSR row pointers int64, values float64;
 *   - there is NO CPU fallback: without a HIP device every compute entry fails.
 */
#ifndef ARCTE_HIP_H
#define ARCTE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARCTE_HIP_OK 0
#define ARCTE_HIP_EINVAL (-1)   /* bad argument */
#define ARCTE_HIP_EHIP (-2)     /* a HIP runtime call failed (no device, out of memory, ...) */
#define ARCTE_HIP_ECAPACITY (-3) /* a seed did not fit the queue/output capacity even after growing */
#define ARCTE_HIP_ESTATE (-4)   /* call order (e.g. fetch before run) */
#define ARCTE_HIP_EGRAPH (-5)   /* input the reference itself cannot process (see arcte_hip_run_seeds) */

typedef struct arcte_hip_ctx arcte_hip_ctx;

/* Version of this ABI (bumped on any signature change). */
int arcte_hip_abi_version(void);

/* Message of the last failing call on this thread. */
const char *arcte_hip_last_error(void);

/* Number of visible HIP devices (0 on a machine without a GPU; does not initialise one). */
int arcte_hip_device_count(int *count);

/*
 * Upload the random-walk transition matrix to `device` and allocate the per-wavefront
 * propagation slots.  Inputs are exactly what get_natural_random_walk_matrix returns
 * (eps_randomwalk/transition.py:43-99) and what arcte_worker receives
 * (embedding/arcte/arcte.py:279-286): CSR (indptr[n+1], indices[nnz] ascending inside a
 * row, data[nnz]) of W = D_out^-1 A, weighted out_degree[n], in_degree[n].
 * A slot is the scratch of one wavefront (one seed in flight): n_slots = 0 picks 12 wavefronts per compute unit -- 14 on graphs
 * whose rows pack into one 32-bit word per edge, 16 of those when region B's lines are indirect (n > ~2.5 M);
 * ARCTE_HIP_WAVES_PER_CU overrides -- of the LINE state (arcte_hip_state_info: 8 bytes per node and slot in strided
 * 64-byte lines, a touched-line bitmap + the hottest nodes' values in LDS), fewer when the slot memory would take the
 * device past 65 % full; ARCTE_HIP_STATE=dense selects the round-2 state (32-byte entries, 6 wavefronts per CU);
 * the CU's LDS is divided among its resident wavefronts for the hot table.  queue_capacity = 0 picks
 * min(2^20, max(4096, n/16 rounded up to a power of two)) ring entries (the FIFO of similarity.py:180 is
 * unbounded; an overflowing seed is re-run with a 4x larger ring, never dropped).  A row that stores the same
 * column twice is rejected with ARCTE_HIP_EINVAL (scipy's sum_duplicates() removes such entries).
 */
int arcte_hip_create(int device, int64_t n, int64_t nnz,
                     const int64_t *indptr, const int32_t *indices, const double *data,
                     const double *out_degree, const double *in_degree,
                     int64_t n_slots, int64_t queue_capacity, arcte_hip_ctx **out);

int arcte_hip_destroy(arcte_hip_ctx *ctx);

/* arcte_hip_destroy keeps the context's large slot buffers (tens of GB) in a process-wide cache for the next context
 * of the same shape on that device: freeing and re-allocating them costs seconds in the runtime.  This returns the
 * cached memory to the driver. */
int arcte_hip_trim(void);

/*
 * get_natural_random_walk_matrix (eps_randomwalk/transition.py:43-99) and the seed ordering of arcte()
 * (embedding/arcte/arcte.py:610-617) ON THE DEVICE, followed by arcte_hip_create's slot set-up: the caller hands
 * over the ADJACENCY matrix as arcte() holds it after csr_matrix(A) (CSR, any column order inside a row, no
 * duplicate columns) and never builds W, the degree vectors or the seed list on the host.  Rounding follows scipy:
 * out_degree = A.sum(axis=1) is data[first] + numpy-pairwise(rest) over the stored row, in_degree = A.sum(axis=0)
 * a left fold over the column in storage order, zero rows divide by 1 (transition.py:58), columns end up ascending
 * (transition.py:65).  nnz < 2^31.
 */
int arcte_hip_create_from_adjacency(int device, int64_t n, int64_t nnz,
                                    const int64_t *indptr, const int32_t *indices, const double *data,
                                    int64_t n_slots, int64_t queue_capacity, arcte_hip_ctx **out);

/*
 * The same from edge-list triplets (what datautil/datarw.py:54-120 read_adjacency_matrix returns: row, col, value
 * with ids already renumbered): duplicates are summed like csr_matrix(coo) does (in input order here; scipy's order
 * for three or more copies of one position is unspecified), and with symmetrise = 1 the matrix becomes
 * (A + A^T)/2 as in entry_points/arcte.py:70-71 -- all on the device.  nnz < 2^30.
 */
int arcte_hip_create_from_coo(int device, int64_t n, int64_t nnz,
                              const int32_t *row, const int32_t *col, const double *val, int symmetrise,
                              int64_t n_slots, int64_t queue_capacity, arcte_hip_ctx **out);

/* Sizes of the graph a context holds: nodes, stored transitions, and the length of arcte()'s seed list
 * (nodes whose pattern in-count exceeds 1, arcte.py:617).  Any pointer may be NULL. */
int arcte_hip_graph_sizes(arcte_hip_ctx *ctx, int64_t *n, int64_t *nnz, int64_t *nseeds);

/* Copy W (indptr[n+1], indices[nnz], data[nnz]) and the degree vectors to the host: the 3-tuple
 * get_natural_random_walk_matrix returns (transition.py:99).  Any pointer may be NULL. */
int arcte_hip_fetch_transition(arcte_hip_ctx *ctx, int64_t *indptr, int32_t *indices, double *data,
                               double *out_degree, double *in_degree);

/* arcte()'s seed list (arcte.py:610-617): node ids by descending pattern in-count, count > 1 only; ties, which the
 * reference's unstable argsort leaves unspecified, are in ascending node order.  seeds has nseeds entries. */
int arcte_hip_fetch_seed_list(arcte_hip_ctx *ctx, int64_t *seeds);

/*
 * calculate_epsilon_effective (embedding/arcte/arcte.py:26-50) for each seed, with the
 * seed's weighted out-degree and its neighbours' out-degrees as at arcte.py:340.
 * The neighbour mean uses numpy's pairwise summation order.
 */
int arcte_hip_epsilon_effective(arcte_hip_ctx *ctx, const int64_t *seeds, int64_t nseeds,
                                double epsilon, double *eps_out);

/* The same rule for ONE seed given as the reference's function takes it (arcte.py:26: seed degree and the vector of
 * its neighbours' degrees), without a context: the scalar call surface of calculate_epsilon_effective. */
int arcte_hip_epsilon_effective_scalar(int device, double epsilon, double seed_degree,
                                       const double *neighbor_degrees, int64_t n_neighbors, double *eps_out);

/*
 * The loop body of arcte_worker (embedding/arcte/arcte.py:337-376) for every seed:
 * effective epsilon -> fast_approximate_cumulative_pagerank_difference
 * (eps_randomwalk/similarity.py:149-222, pushes of eps_randomwalk/push.py:41-64) ->
 * degree normalisation -> threshold select -> emit.  use_effective_epsilon = 0 feeds the
 * raw `epsilon` to every seed instead (the older cython_opt driver's behaviour).
 * Results stay on the device until fetched.  A seed whose closed neighbourhood is not
 * contained in its support (zero-weight edges; the reference mis-indexes there,
 * arcte.py:359-360) fails the call with ARCTE_HIP_EGRAPH.
 */
int arcte_hip_run_seeds(arcte_hip_ctx *ctx, const int64_t *seeds, int64_t nseeds,
                        double rho, double epsilon, int use_effective_epsilon);

/*
 * The same loop for the reference's PageRank-flavoured workers.  variant 0 = arcte_hip_run_seeds;
 * variant 1 = arcte_with_pagerank_worker (embedding/arcte/arcte.py:166-276: pushes of
 * eps_randomwalk/push.py:4-17 driven by similarity.py:11-63); variant 2 =
 * arcte_with_lazy_pagerank_worker (arcte.py:53-163: push.py:20-38, similarity.py:66-146 with its
 * self re-push loops; laziness_factor as at similarity.py:75, and `rho` is what the worker hands
 * down, i.e. the caller applies lazy_rho of arcte.py:109).  Both guard the extraction with the
 * intersection test of arcte.py:129-133: a seed whose closed neighbourhood is not inside the support
 * (or that has a self-loop) emits nothing instead of failing.
 */
int arcte_hip_run_seeds_variant(arcte_hip_ctx *ctx, const int64_t *seeds, int64_t nseeds,
                                double rho, double epsilon, int use_effective_epsilon,
                                int variant, double laziness_factor);

/*
 * The same for ONE MORE part of a seed list: the completed run's result stays on the context and this run's columns join it
 * (the result then lists the new seeds first, the earlier ones behind them; the CSR assembly orders columns by seed id).  The
 * reference's counterpart is the sum over the chunks of a worker (arcte.py:384-386) -- every seed owns its column, so the sum is
 * a concatenation.  Counters and timings of arcte_hip_run_stats / _counters / _timing add up over the parts.
 * ARCTE_HIP_ESTATE without a completed (non-centrality) run to append to.
 */
int arcte_hip_run_seeds_append(arcte_hip_ctx *ctx, const int64_t *seeds, int64_t nseeds, double rho, double epsilon,
                               int use_effective_epsilon, int variant, double laziness_factor);

/*
 * The loop of arcte_and_centrality (embedding/arcte/cython_opt/arcte.pyx:165-217), the reference's older
 * single-process driver, for the nodes in [node_begin, node_end): every node WITH out-edges is a seed, in index order;
 * the propagation runs with the RAW epsilon; s/in_degree of every support node is added to a centrality vector in
 * seed order (per node a left fold, exactly the reference's sequence of additions: the contributions of a batch of
 * seeds are sorted by (node, seed) on the device and folded in that order by one wavefront per node); the community is
 * everything at or above the smallest value inside the closed neighbourhood, emitted iff it has more members than that
 * neighbourhood (a set: a self-loop does not count twice).  Where a node outside the neighbourhood TIES with that
 * smallest value the reference's own answer depends on numpy's unstable argsort (arcte.pyx:194-208); this takes it.
 * Nodes of the range without out-edges get centrality 1.0 (arcte.pyx:210).  Results: arcte_hip_fetch_result (one
 * entry per seed of the range, in order), arcte_hip_fetch_centrality, arcte_hip_features_from_result (columns
 * numbered by a running counter over the emitting seeds, arcte.pyx:213-215; base block = identity + W, see there).
 * A sub-range gives the partial sums of that range (one rank of a sharded run: add the vectors, then concatenate).
 */
int arcte_hip_run_centrality(arcte_hip_ctx *ctx, int64_t node_begin, int64_t node_end, double rho, double epsilon);
/* centrality[n] of the last arcte_hip_run_centrality. */
int arcte_hip_fetch_centrality(arcte_hip_ctx *ctx, double *centrality);

/* Sizes of the last run: number of seeds and total emitted (row, seed) pairs. */
int arcte_hip_result_sizes(arcte_hip_ctx *ctx, int64_t *nseeds, int64_t *total_rows);

/*
 * Copy the last run to the host in column-compressed form: rows[colptr[k] .. colptr[k+1])
 * are the members of seed k's local community (empty when the reference emits nothing,
 * arcte.py:370).  Any of the output pointers may be NULL.  eps_used[k] is the threshold
 * the seed ran with, nop[k] the return value of
 * fast_approximate_cumulative_pagerank_difference.
 */
int arcte_hip_fetch_result(arcte_hip_ctx *ctx, int64_t *colptr, int32_t *rows,
                           double *eps_used, int64_t *nop);

/*
 * The last run as a ROW-compressed matrix, assembled on the device (the (member, seed) pairs are laid out in
 * ascending seed order and stable-sorted by their row id alone: log2(n) key bits, any number of entries): row =
 * node id, column = seed id, columns ascending inside a row -- the CSR that arcte_worker returns
 * (arcte.py:379-388).  With
 * with_base_block = 1 the columns are shifted by n and the base-community block I + pattern(W) of
 * arcte.py:676-679 is merged in front, i.e. the result is arcte()'s n x 2n feature pattern (arcte.py:683); a
 * node with a self-loop gets ONE diagonal entry there (the caller stores 2.0 for it, as the reference's I + ones
 * does).  Call arcte_hip_result_csr_size first; indptr has n+1 entries, indices at least that many entries;
 * *nnz_out receives the number of stored entries.  A run that listed a seed twice is refused (ARCTE_HIP_EINVAL).
 */
int arcte_hip_result_csr_size(arcte_hip_ctx *ctx, int with_base_block, int64_t *nnz);
int arcte_hip_fetch_result_csr(arcte_hip_ctx *ctx, int with_base_block, int64_t *indptr, int32_t *indices,
                               int64_t *nnz_out);

/* Device addresses of the last run's rows (int32[total_rows]) for a device-side gather
 * (RCCL); valid until the next run on this context. */
int arcte_hip_result_device_rows(arcte_hip_ctx *ctx, void **rows_dev);

/* Copy the last run's rows (int32[total_rows]) into caller-owned DEVICE memory on the same GPU
 * (e.g. a tensor that a RCCL collective will send); device-to-device, no host round trip. */
int arcte_hip_copy_result_rows_to_device(arcte_hip_ctx *ctx, void *dst_dev, int64_t capacity_rows);

/*
 * Work counters of the last run, summed over its seeds:
 * stats[0] pushes, [1] edges traversed, [2] enqueues, [3] support entries,
 * [4] seed re-runs after a queue/output overflow, [5] kernel launches.
 */
int arcte_hip_run_stats(arcte_hip_ctx *ctx, int64_t stats[6]);

/* Extensible form of the above: out[0..n) = pushes, edges, enqueues, support, re-runs, launches, candidates
 * (nodes the extraction had to examine: those whose s/in_degree reached the lower bound of the threshold), rows that
 * a helper wavefront took half of (launch shape ARCTE_HIP_COOP=1, else 0); entries past the known counters are zero. */
int arcte_hip_run_counters(arcte_hip_ctx *ctx, int64_t *out, int n);

/*
 * Device time of the last run in milliseconds, from HIP events recorded on the
 * context's stream: ms[0] effective-epsilon kernel, [1] push/extract kernel(s),
 * [2] compaction, [3] whole call (host wall clock, includes the small host scans).
 */
int arcte_hip_run_timing(arcte_hip_ctx *ctx, double ms[4]);

/*
 * fast_approximate_cumulative_pagerank_difference (eps_randomwalk/similarity.py:149-222)
 * on caller-owned dense s[n], r[n]: both are uploaded as they are, the seed entries are
 * set to 1, the propagation runs with the raw `epsilon`, and both are written back.
 * *nop receives the number of pushes.
 */
int arcte_hip_similarity_slice(arcte_hip_ctx *ctx, int64_t seed, double rho, double epsilon,
                               double *s, double *r, int64_t *nop);

/* fast_approximate_personalized_pagerank (similarity.py:11-63, variant 1) and
 * lazy_approximate_personalized_pagerank (similarity.py:66-146, variant 2) with the same contract
 * (only r[seed] is set to 1 there); variant 0 = arcte_hip_similarity_slice. */
int arcte_hip_similarity_slice_variant(arcte_hip_ctx *ctx, int64_t seed, double rho, double epsilon,
                                       int variant, double laziness_factor,
                                       double *s, double *r, int64_t *nop);

/*
 * The same vectors out of the PRODUCTION kernel (k_arcte_lines, what arcte_hip_run_seeds launches): ONE seed runs
 * through the worker loop as in arcte_hip_run_seeds_variant and, when its FIFO has run dry, the kernel gathers the
 * seed's state from every level it lives on (on-chip values, the strided lines of regions A and B, the pushed-state
 * array) into dense s[n] and r[n] by node id, starting from zeros as arcte_worker does (arcte.py:337-338): what
 * fast_approximate_cumulative_pagerank_difference (similarity.py:149-222; variants: :11-146) leaves with its caller.
 * use_effective_epsilon as in arcte_hip_run_seeds.  nop (may be NULL) receives the number of pushes.  A debug / parity
 * entry: the run's result (one column) can be fetched afterwards like any other run's.  ARCTE_HIP_ESTATE when the
 * context runs the dense-state kernel.
 */
int arcte_hip_seed_state(arcte_hip_ctx *ctx, int64_t seed, double rho, double epsilon, int use_effective_epsilon,
                         int variant, double laziness_factor, double *s, double *r, int64_t *nop);

/*
 * cumulative_pagerank_difference_limit_push (eps_randomwalk/push.py:41-64): one push of
 * `push_node` over (w_i, a_i) on caller-owned dense s[n], r[n].  Context-free.
 */
int arcte_hip_push(int device, int64_t n, double *s, double *r,
                   const double *w_i, const int32_t *a_i, int64_t deg,
                   int64_t push_node, double rho);

/* pagerank_limit_push (push.py:4-17, variant 1) and pagerank_lazy_push (push.py:20-38, variant 2). */
int arcte_hip_push_variant(int device, int64_t n, double *s, double *r,
                           const double *w_i, const int32_t *a_i, int64_t deg,
                           int64_t push_node, double rho, int variant, double laziness_factor);

/*
 * Arithmetic of the propagation kernels on this context: 0 = float64 (default; the reference's type, the
 * only one held to bit-exact parity), 1 = float32 (BASELINE.json configs[4] tolerance sweep: 16-byte state
 * entries, float weights/degrees; the effective epsilon is still computed in float64 and rounded once).
 * Takes effect from the next run; switching clears the per-slot state.
 */
int arcte_hip_set_float32(arcte_hip_ctx *ctx, int enable);

/*
 * ---- Feature matrices on the device and the weighting the reference applies right after arcte() --------------------
 * (experiments/utility.py:66 and :101-104).  A features object is a CSR matrix (int64 row pointers, int32 column ids,
 * float64 values) that lives in HBM; the operations below stream over it in place, so arcte()'s n x 2n result never
 * has to visit the host between extraction and weighting.
 */
typedef struct arcte_hip_features arcte_hip_features;

/* arcte()'s feature matrix of the last run, kept on the device: the pattern of arcte_hip_fetch_result_csr with the
 * reference's values (1.0; 2.0 on the diagonal of a node with a self-loop, arcte.py:676-679). */
int arcte_hip_features_from_result(arcte_hip_ctx *ctx, int with_base_block, arcte_hip_features **out);
/* Any CSR matrix from the host (e.g. a feature matrix made elsewhere). */
int arcte_hip_features_upload(int device, int64_t n_rows, int64_t n_cols, int64_t nnz,
                              const int64_t *indptr, const int32_t *indices, const double *data,
                              arcte_hip_features **out);
int arcte_hip_features_destroy(arcte_hip_features *f);
int arcte_hip_features_sizes(arcte_hip_features *f, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
/* Copy to the host; any pointer may be NULL. */
int arcte_hip_features_fetch(arcte_hip_features *f, int64_t *indptr, int32_t *indices, double *data);
/* features[rows, :] as a new object (the train / test split of experiments/utility.py:94-97). */
int arcte_hip_features_select_rows(arcte_hip_features *f, const int64_t *rows, int64_t n_selected,
                                   arcte_hip_features **out);

/* normalize_columns (embedding/common.py:49-67): every column with more than one stored entry is divided by
 * sqrt(log(number of stored entries)). */
int arcte_hip_features_normalize_columns(arcte_hip_features *f);
/* normalize_rows (embedding/common.py:29-46): sklearn normalize(norm="l2"); the squares of a row are summed in storage
 * order, zero rows stay. */
int arcte_hip_features_normalize_rows(arcte_hip_features *f);
/* chi2_contingency_matrix + peak_snr_weight_aggregation (embedding/community_weighting.py:11-84).  Y is the binarised
 * label matrix in CSR form (y_indptr[n_rows+1], y_indices = class ids of each row), as LabelBinarizer leaves it
 * (:19-21; the caller expands a binary problem to two classes).  contingency_out (n_classes x n_cols, row-major) may be
 * NULL; weights_out has n_cols entries. */
int arcte_hip_features_chi2_psnr_weights(arcte_hip_features *f, const int64_t *y_indptr, const int32_t *y_indices,
                                         int64_t n_classes, double *contingency_out, double *weights_out);
/* peak_snr_weight_aggregation (embedding/community_weighting.py:48-84) alone, on a host n_classes x n_cols matrix. */
int arcte_hip_peak_snr_weights(int device, int64_t n_classes, int64_t n_cols, const double *contingency,
                               double *weights_out);
/* community_weighting (embedding/community_weighting.py:87-125) for ONE matrix (the reference runs the same lines on
 * X_train and X_test): columns with more than one stored entry are multiplied by log(1 + w) (0 when w == 0),
 * zeros are eliminated, rows are l2-normalised. */
int arcte_hip_features_community_weighting(arcte_hip_features *f, const double *community_weights);

/*
 * The file formats either side of the path (entry_points/arcte.py:63-84), natively.
 *
 * arcte_hip_edge_list_read replaces read_adjacency_matrix (datautil/datarw.py:54-120) up to the scipy wrapper: every
 * line is `line.strip().split(separator)` (common.py:36-49); a line whose first field begins with '#' is skipped;
 * fields 0 / 1 are the integer node ids, field 2 the float weight; ids are renumbered in first-seen order, source before
 * target (:87-92); with `undirected` every non-loop edge is followed by its reciprocal (:105-109); duplicate edges stay
 * duplicate triplets (arcte_hip_create_from_coo sums them like csr_matrix(coo) does).  A malformed line fails with
 * ARCTE_HIP_EINVAL and its line number (the reference raises ValueError / IndexError there).  No GPU involved.
 * _sizes gives the number of nodes and of triplets, _fetch copies them out (row / col int32, val float64, node_ids[new
 * id] = original id: the reference's node_to_id; any pointer may be NULL), _destroy frees the list.
 */
typedef struct arcte_hip_edge_list arcte_hip_edge_list;
int arcte_hip_edge_list_read(const char *path, const char *separator, int undirected, arcte_hip_edge_list **out);
int arcte_hip_edge_list_sizes(arcte_hip_edge_list *el, int64_t *n_nodes, int64_t *n_triplets);
int arcte_hip_edge_list_fetch(arcte_hip_edge_list *el, int32_t *row, int32_t *col, double *val, int64_t *node_ids);
int arcte_hip_edge_list_destroy(arcte_hip_edge_list *el);
/* write_features (datautil/datarw.py:123-143) for arcte()'s matrix given as CSR arrays (what
 * arcte_hip_fetch_result_csr returns): one line per stored entry in row-major order,
 * `<node_ids[row]><separator><column><separator><int(value)>`; every value is 1 except the diagonal entry of the nodes
 * listed in doubled_diagonal, which is 2 (identity + ones on a self-loop, embedding/arcte/arcte.py:676-679). */
int arcte_hip_write_feature_triplets(const char *path, int64_t n_rows, const int64_t *indptr, const int32_t *indices,
                                     const int64_t *node_ids, const int64_t *doubled_diagonal, int64_t n_doubled,
                                     const char *separator);

/* The reference's parent process SUMS its workers' n x n matrices (embedding/arcte/arcte.py:670-673); every seed owns
 * its column, so the sum is a concatenation.  This entry appends another worker's result -- its seeds, their community
 * sizes and the members (int32 node ids, concatenated; `rows` may point to host memory or to memory of any GPU of this
 * process: hipMemcpyDefault) -- to the completed run of `ctx`, after which arcte_hip_result_csr_size /
 * arcte_hip_fetch_result_csr / arcte_hip_features_from_result assemble the matrix of ALL parts on ctx's GPU, as they do
 * for a one-GPU run.  A seed must not appear in two parts. */
int arcte_hip_append_result(arcte_hip_ctx *ctx, const int64_t *seeds, const int64_t *counts, int64_t nseeds, const void *rows,
                            int64_t nrows);

/*
 * Measurement helper (no counterpart in the reference; SURVEY.md 8(d) asks for the on-box streaming rate beside
 * the 8 TB/s spec figure): the rate of a coalesced 16-byte-per-lane read sweep and of a copy (read + write bytes)
 * over `bytes` of device memory, in GB/s, best of three.
 */
int arcte_hip_stream_bandwidth(int device, int64_t bytes, double *read_gbps, double *copy_gbps);

/* 1 when the library was built with `make AB=1`: the launch shapes that lost their A/B in rounds 2-3 (ARCTE_HIP_TILES=2/4,
 * ARCTE_HIP_STAGE_ROWS, ARCTE_HIP_COOP helper wavefronts, more than twelve wavefronts per CU on the 128-VGPR build) and the
 * instrumented instantiation (ARCTE_HIP_PROFILE) exist; 0 (the default build): those knobs are ignored. */
int arcte_hip_has_ab_builds(void);

/* Device memory this PROCESS holds outside any context (no counterpart in the reference): info[0] bytes of placement-draw
 * losers kept allocated on `device` (at most ARCTE_HIP_PARK_MAX = 1 buffer per device; returned by arcte_hip_trim, by a draw
 * of another shape, by a growing context and whenever an allocation fails), [1] bytes of destroyed contexts' large buffers
 * kept for the next context of the same shape, [2] free and [3] total bytes of the device as the runtime reports them. */
int arcte_hip_memory_info(int device, int64_t info[4]);

/* Diagnostic: how many workgroups of the float64 ARCTE propagation kernel the runtime's occupancy query admits per
 * compute unit with the context's launch shape (the slot count assumes info[7] / info[4] of them). */
int arcte_hip_launch_occupancy(arcte_hip_ctx *ctx, int *workgroups_per_cu);

/* Properties of the context: info[0] slots, [1] queue capacity, [2] device bytes held,
 * [3] compute units, [4] wavefronts per workgroup of the propagation kernel, [5] values of the LDS-resident hot
 * table per wavefront (0 = table off), [6] 64-edge tiles per push iteration, [7] wavefronts per compute unit the
 * LDS is divided among, [8] 1 when the propagation streams narrow rows (every row's weights are one number and every
 * in_degree is exact in float32 -- unweighted graphs: 8 instead of 20 bytes per traversed edge), 2 when it streams PACKED
 * rows (line state only: ONE 32-bit word per edge -- the target's rank and, above it, its integer in_degree; the few
 * highest-ranked nodes' in_degrees come from a float32 table; ARCTE_HIP_PACK=0 switches it off), [9] end rank of the
 * warm table (nodes ranked between [5] and [9] keep their state in a compact per-slot array; 0 = off). */
int arcte_hip_info(arcte_hip_ctx *ctx, int64_t info[10]);

/* Where the per-seed state lives (no counterpart in the reference, whose s and r are two dense numpy vectors per
 * worker, arcte.py:325-326, re-zeroed per seed, :337-338).  info[0] 1: line state (nodes named by rank, one float64
 * per node and slot in strided 64-byte lines, touched-line bitmap in LDS, pushed nodes in a compact {r, s} array),
 * 0: dense 32-byte entries with epoch tags; [1] lines per slot (M: a slot holds 8 M values); [2] entries of the
 * pushed-state array per slot; [3] entries of the candidate list per slot; [4] bytes of slot scratch held;
 * [5] bytes of LDS the bitmap takes per wavefront; [6] bytes of LDS a wavefront may claim; [7] lines per slot of region B
 * (the ranks beyond 8 M: touched-bits in a per-slot bitmap in global memory; 0 = every rank is covered by the LDS bitmap);
 * of the last run: [8] updates of on-chip values, [9] blind whole-line writes (first touch of a line),
 * [10] read-modify-writes of a line, [11] updates of pushed nodes; [12] 1 when region B's lines are indirect (an 8-byte
 * entry per line + a pool of lines per slot), [13] pool lines per slot. */
int arcte_hip_state_info(arcte_hip_ctx *ctx, int64_t info[14]);

/* The draws of the slot memory's placement made when the context was created (no counterpart in the reference): the
 * propagation kernel's speed depends on how its per-seed state is laid over the physical memory, which hipMalloc leaves
 * to chance, so a context large enough to care allocates up to ARCTE_HIP_PLACEMENT_TRIES (8) candidates, measures a few
 * milliseconds of random read-modify-writes on each, stops when one is 10 % faster than another and keeps the fastest.  *drawn = candidates measured (0: no draw),
 * *kept = index of the one in use, rates[i] = G updates/s of candidate i (up to `capacity` of them). */
int arcte_hip_placement_info(arcte_hip_ctx *ctx, int *kept, double *rates, int capacity, int *drawn);

#ifdef __cplusplus
}
#endif
#endif /* ARCTE_HIP_H */
