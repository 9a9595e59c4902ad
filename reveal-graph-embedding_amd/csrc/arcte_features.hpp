// Device-resident feature matrices and the weighting the reference applies to them right after arcte()
// (SURVEY.md 8(f)4): embedding/common.py:29-67 (normalize_columns, normalize_rows) and
// embedding/community_weighting.py:11-125 (chi-squared contingency, peak-SNR aggregation, community_weighting).
// Included by arcte_hip.hip only.  All of it is bandwidth-bound streaming over a CSR (one pass per operation).
#pragma once

#include "arcte_kernels.hpp"

namespace {

// stored entries per column ("document frequency", common.py:60 / community_weighting.py:95: getcol(j).data.size)
__global__ void k_feat_column_counts(const int32_t *indices, int64_t nnz, uint32_t *count)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) atomicAdd(count + indices[k], 1u);
}

// common.py:61-63: scale[j] = 1/sqrt(log(df)) is applied as a division by sqrt(log(df)) for df > 1
__global__ void k_feat_idf(const uint32_t *count, int64_t ncols, double *divisor)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < ncols) divisor[j] = count[j] > 1 ? sqrt(log((double)count[j])) : 1.0;
}

__global__ void k_feat_divide_columns(const int32_t *indices, const double *divisor, int64_t nnz, double *data)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) data[k] = data[k] / divisor[indices[k]];
}

// community_weighting.py:96-103: columns with more than one stored entry are multiplied by log(1 + w) (0 when w == 0)
__global__ void k_feat_reinforcement(const uint32_t *count, const double *weights, int64_t ncols, double *factor)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    double f = 1.0;
    if (count[j] > 1) f = (weights[j] == 0.0) ? 0.0 : log(1.0 + weights[j]);
    factor[j] = f;
}

__global__ void k_feat_multiply_columns(const int32_t *indices, const double *factor, int64_t nnz, double *data)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) data[k] = data[k] * factor[indices[k]];
}

// common.py:40 / community_weighting.py:118-119: sklearn normalize(norm="l2") -- the squares of a row are summed in
// storage order (sklearn/utils/sparsefuncs_fast.pyx), zero rows stay.  One wavefront per row: coalesced loads, the
// sum as the same left fold (ordered_add64).
__global__ __launch_bounds__(BLOCK) void k_feat_normalize_rows(const int64_t *indptr, int64_t nrows, double *data)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= nrows) return;
    const int64_t b = indptr[i], e = indptr[i + 1];
    double acc = 0.0;
    for (int64_t k0 = b; k0 < e; k0 += WAVE) {
        const int64_t k = k0 + lane;
        const double x = k < e ? data[k] : 0.0;
        const int64_t left = e - k0;
        acc = left >= WAVE ? ordered_add64(acc, x * x) : ordered_add_n(acc, x * x, (int)left);
    }
    if (acc == 0.0) return;
    const double norm = sqrt(acc);
    for (int64_t k = b + lane; k < e; k += WAVE) data[k] = data[k] / norm;
}

// eliminate_zeros() (community_weighting.py:115-116): keep[k] = data[k] != 0
__global__ void k_feat_nonzero_flags(const double *data, int64_t nnz, int64_t *keep)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) keep[k] = data[k] != 0.0 ? 1 : 0;
}

// `pos` = inclusive scan of keep: surviving entry k moves to pos[k]-1; the row pointers move with it
__global__ void k_feat_compact(const int32_t *indices, const double *data, const int64_t *pos, int64_t nnz, int32_t *indices_out,
                               double *data_out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int64_t prev = k ? pos[k - 1] : 0;
    if (pos[k] != prev) { indices_out[prev] = indices[k]; data_out[prev] = data[k]; }
}

__global__ void k_feat_compact_indptr(const int64_t *indptr, const int64_t *pos, int64_t nrows, int64_t *indptr_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nrows) return;
    const int64_t b = indptr[i];
    indptr_out[i] = b ? pos[b - 1] : 0;
}

// X[rows]: row lengths, then the copy (one wavefront per selected row)
__global__ void k_feat_selected_lengths(const int64_t *indptr, const int64_t *rows, int64_t nsel, int64_t *len)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nsel) len[i] = indptr[rows[i] + 1] - indptr[rows[i]];
}

__global__ __launch_bounds__(BLOCK) void k_feat_copy_rows(const int64_t *indptr, const int32_t *indices, const double *data, const int64_t *rows,
                                                          int64_t nsel, const int64_t *indptr_out, int32_t *indices_out, double *data_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= nsel) return;
    const int64_t b = indptr[rows[i]], e = indptr[rows[i] + 1], o = indptr_out[i];
    for (int64_t k = b + lane; k < e; k += WAVE) { indices_out[o + (k - b)] = indices[k]; data_out[o + (k - b)] = data[k]; }
}

// community_weighting.py:24 observed = Y^T X with X's values replaced by ones (:12-13): every (class of row i,
// feature of row i) pair counts one.  Integer-valued sums: exact in any order.  One wavefront per row of X.
__global__ __launch_bounds__(BLOCK) void k_chi2_observed(const int64_t *x_indptr, const int32_t *x_indices, const int64_t *y_indptr,
                                                         const int32_t *y_indices, int64_t nrows, int64_t ncols, double *observed,
                                                         double *class_count)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= nrows) return;
    const int64_t xb = x_indptr[i], xe = x_indptr[i + 1];
    for (int64_t c = y_indptr[i]; c < y_indptr[i + 1]; c++) {
        const int64_t cls = y_indices[c];
        if (lane == 0) atomicAdd(class_count + cls, 1.0);
        for (int64_t k = xb + lane; k < xe; k += WAVE) atomicAdd(observed + cls * ncols + x_indices[k], 1.0);
    }
}

// community_weighting.py:28-43: chi2[c][f] = (observed - expected)^2 / expected, expected = P(class c) * count(f),
// a zero expectation divides by 1
__global__ void k_chi2_statistic(const double *class_count, const uint32_t *feature_count, int64_t nrows, int64_t nclasses,
                                 int64_t ncols, double *m)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nclasses * ncols) return;
    const int64_t c = t / ncols, f = t % ncols;
    const double class_prob = class_count[c] / (double)nrows;            // Y.mean(axis=0)
    double expected = class_prob * (double)feature_count[f];
    double d = m[t] - expected;
    d = d * d;
    if (expected == 0.0) expected = 1.0;
    m[t] = d / expected;
}

// community_weighting.py:51-55: np.var of each class row (two-pass: mean, then mean of squared deviations); one
// workgroup per class, fixed-shape tree reductions (deterministic)
__global__ __launch_bounds__(256) void k_psnr_row_variance(const double *m, int64_t ncols, double *variance)
{
    __shared__ double red[256];
    const double *row = m + (int64_t)blockIdx.x * ncols;
    double acc = 0.0;
    for (int64_t f = threadIdx.x; f < ncols; f += 256) { const double x = row[f]; acc += (x != x) ? 0.0 : x; }   // NaN -> 0 (:49)
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const double mean = red[0] / (double)ncols;
    __syncthreads();
    acc = 0.0;
    for (int64_t f = threadIdx.x; f < ncols; f += 256) { double x = row[f]; x = (x != x) ? 0.0 : x; acc += (x - mean) * (x - mean); }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) variance[blockIdx.x] = red[0] / (double)ncols;
}

// community_weighting.py:57-68: per feature, over the classes with a positive statistic: (max - min)/sigma for two or
// more, max/sigma for one, 0 for none
__global__ void k_psnr_weights(const double *m, int64_t nclasses, int64_t ncols, const double *variance, double *weights)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= ncols) return;
    double vsum = 0.0;
    for (int64_t c = 0; c < nclasses; c++) vsum += variance[c];
    const double sigma = sqrt(vsum / (double)nclasses);                  // :55
    double mx = 0.0, mn = 0.0;
    int cnt = 0;
    for (int64_t c = 0; c < nclasses; c++) {
        double x = m[c * ncols + f];
        if (x != x) x = 0.0;
        if (x > 0.0) {
            if (cnt == 0) { mx = x; mn = x; }
            else { mx = x > mx ? x : mx; mn = x < mn ? x : mn; }
            cnt++;
        }
    }
    double w = 0.0;
    if (cnt > 1) w = (mx - mn) / sigma;
    else if (cnt == 1) w = mx / sigma;
    weights[f] = w;
}

}  // namespace
