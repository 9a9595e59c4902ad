// Device-side graph preparation for the ARCTE hot path (gfx950): the steps the reference runs in scipy before the
// first seed -- edge list -> CSR, symmetrisation (entry_points/arcte.py:70-71), get_natural_random_walk_matrix
// (eps_randomwalk/transition.py:43-99) and the seed list of arcte() (embedding/arcte/arcte.py:610-617).
// Included by arcte_hip.hip only.
//
// Rounding is part of the contract (the weighted / directed fixtures pin it), so the reductions keep scipy's order:
//   out_degree = A.sum(axis=1)  -> np.add.reduceat over the stored row: data[first] + pairwise(data[first+1:])
//                                  (numpy copies the first element of a segment and pairwise-sums the rest);
//   in_degree  = A.sum(axis=0)  -> ones @ A = csc_matvec on the transpose: a left fold over the stored entries
//                                  of a column in row-major storage order, starting from 0.0;
//   W.data     = data / out_degree[row] (zero rows divide by 1, transition.py:58), then sort_indices().
#pragma once

#include "arcte_kernels.hpp"

namespace {

// (the numpy pairwise sum of a plain array by one wavefront is pw_sum_plain in arcte_kernels.hpp: one evaluator for both users)

// transition.py:55 + :58: out_degree[i] = A.sum(axis=1)[i] (np.add.reduceat order), zero -> 1.  One wavefront per row.
__global__ __launch_bounds__(BLOCK) void k_out_degree(const int64_t *indptr, const double *data, int64_t n, double *out_degree)
{
    __shared__ EpsShared sh[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (i >= n) return;
    const int64_t b = indptr[i], e = indptr[i + 1];
    double sum = 0.0;
    if (e > b) {
        const double rest = pw_sum_plain(data, b + 1, e - b - 1, sh[wave], lane);
        sum = (e - b > 1) ? data[b] + rest : data[b];
    }
    if (lane == 0) out_degree[i] = (sum == 0.0) ? 1.0 : sum;
}

// transition.py:56: in_degree[c] = left fold, from 0.0, over column c's stored entries in row-major storage order
// (`vals` holds them grouped by column by a STABLE sort of the storage order).  One thread per column: the fold
// is a dependent chain by definition.
__global__ void k_in_degree(const int64_t *colptr, const double *vals, int64_t n, double *in_degree)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double acc = 0.0;
    for (int64_t k = colptr[c]; k < colptr[c + 1]; k++) acc += vals[k];
    in_degree[c] = acc;
}

// transition.py:61-63: W.data = A.data / out_degree[row].  One wavefront per row.
__global__ __launch_bounds__(BLOCK) void k_row_scale(const int64_t *indptr, const double *out_degree, int64_t n, double *data)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= n) return;
    const double d = out_degree[i];
    for (int64_t k = indptr[i] + lane; k < indptr[i + 1]; k += WAVE) data[k] = data[k] / d;
}

// sorted (row << 32 | column) keys: flag[0] = 1 when two neighbours are equal (a row stores a column twice)
__global__ void k_adjacent_equal(const uint64_t *keys, int64_t m, int32_t *flag)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k + 1 < m && keys[k] == keys[k + 1]) flag[0] = 1;
}

// flags[0]: a column index out of [0, n); flags[1]: some row is not strictly ascending (unsorted or duplicate)
__global__ __launch_bounds__(BLOCK) void k_check_rows(const int64_t *indptr, const int32_t *indices, int64_t n, int32_t *flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t b = indptr[i], e = indptr[i + 1];
    bool bad_range = false, bad_order = false;
    for (int64_t k = b + lane; k < e; k += WAVE) {
        const int32_t c = indices[k];
        bad_range |= (c < 0 || c >= n);
        if (k + 1 < e) bad_order |= (indices[k + 1] <= c);
    }
    if (bad_range) flags[0] = 1;
    if (bad_order) flags[1] = 1;
}

// sort keys (row << 32 | col) of a CSR's stored entries; one wavefront per row
__global__ __launch_bounds__(BLOCK) void k_csr_keys(const int64_t *indptr, const int32_t *indices, int64_t n, uint64_t *keys)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= n) return;
    for (int64_t k = indptr[i] + lane; k < indptr[i + 1]; k += WAVE) keys[k] = ((uint64_t)(uint32_t)i << 32) | (uint32_t)indices[k];
}

// COO triplets -> keys; with `mirror` the transposed copy is appended behind the nnz originals (A + A^T)
__global__ void k_coo_keys(const int32_t *row, const int32_t *col, const double *val, int64_t nnz, int mirror, int64_t n,
                           uint64_t *keys, double *vals, int32_t *flags)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int32_t r = row[k], c = col[k];
    if (r < 0 || r >= n || c < 0 || c >= n) { flags[0] = 1; return; }
    keys[k] = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
    vals[k] = val[k];
    if (mirror) {
        keys[nnz + k] = ((uint64_t)(uint32_t)c << 32) | (uint32_t)r;
        vals[nnz + k] = val[k];
    }
}

// sorted keys -> head flags (1 where a new (row, col) starts); the inclusive scan of the flags numbers the unique entries
__global__ void k_head_flags(const uint64_t *keys, int64_t m, int64_t *head)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) head[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}

// One thread per unique entry: sum its duplicates in sorted (= input, the sort is stable) order, scale, write the
// canonical CSR arrays.  `pos` = inclusive scan of the head flags.
__global__ void k_merge_duplicates(const uint64_t *keys, const double *vals, const int64_t *pos, int64_t m, double scale,
                                   int32_t *indices, double *data, int32_t *rows)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    if (k != 0 && keys[k] == keys[k - 1]) return;
    double acc = vals[k];
    for (int64_t j = k + 1; j < m && keys[j] == keys[k]; j++) acc += vals[j];
    const int64_t o = pos[k] - 1;
    indices[o] = (int32_t)(uint32_t)keys[k];
    rows[o] = (int32_t)(uint32_t)(keys[k] >> 32);
    data[o] = acc * scale;
}

// indptr[r] = first stored entry whose row is >= r (rows ascending)
__global__ void k_rows_to_indptr(const int32_t *rows, int64_t nnz, int64_t n, int64_t *indptr)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n) return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (rows[mid] < r) lo = mid + 1; else hi = mid;
    }
    indptr[r] = lo;
}

__global__ void k_split_keys(const uint64_t *keys, int64_t m, int32_t *indices)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) indices[k] = (int32_t)(uint32_t)keys[k];
}

__global__ void k_u32_to_i64(const uint32_t *in, int64_t n, int64_t *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// arcte.py:617: the seeds are the nodes whose pattern in-count exceeds 1
__global__ void k_count_seeds(const uint32_t *count, int64_t n, unsigned long long *nseeds)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool is_seed = i < n && count[i] > 1;
    const uint64_t m = __ballot(is_seed);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(nseeds, (unsigned long long)__popcll(m));
}

}  // namespace
