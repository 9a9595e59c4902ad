// Device code of the ARCTE hot path for gfx950 (wave64): effective-epsilon kernel, the per-seed
// propagation + extraction kernel (templated on mode, push flavour and value type) and the small
// utility kernels.  Included by arcte_hip.hip only; see that file's header for the design.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

// ---------------------------------------------------------------------------------------------
// device helpers (wave64)
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = WAVE * WAVES_PER_BLOCK;

enum SeedStatus : int32_t {
    ST_OK = 0,
    ST_QUEUE_OVERFLOW = 1,
    ST_OUTPUT_OVERFLOW = 2,
    ST_MISSING_BASE = 3,
    ST_RUNAWAY = 4,   // push cap hit: every wave must reach an exit (guards against a non-converging input)
    ST_CONTRIB_OVERFLOW = 5,   // MODE 2: the centrality contributions of this batch do not fit their arena
};

__device__ __forceinline__ int lane_below(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

__device__ __forceinline__ uint64_t bcast_u64(uint64_t v)
{
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}


__device__ __forceinline__ double shfl_f64(double v, int src) { return __shfl(v, src, WAVE); }
__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src)
{
    int lo = __shfl((int)(uint32_t)v, src, WAVE);
    int hi = __shfl((int)(uint32_t)((uint64_t)v >> 32), src, WAVE);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__device__ __forceinline__ double wave_min(double x)
{
    for (int o = 32; o > 0; o >>= 1) {
        double y = __shfl_xor(x, o, WAVE);
        x = (y < x) ? y : x;
    }
    return x;
}
__device__ __forceinline__ double wave_max(double x)
{
    for (int o = 32; o > 0; o >>= 1) {
        double y = __shfl_xor(x, o, WAVE);
        x = (y > x) ? y : x;
    }
    return x;
}

struct GraphDev {
    int64_t n;
    const int64_t *indptr;
    const int32_t *indices;
    const double *data;
    const double *out_degree;
    const double *in_degree;
    const double *edge_in_degree;   // in_degree[indices[k]] stored with the edge: streams with the row
    const float *data_f, *in_degree_f, *edge_in_degree_f;   // float32 copies (NULL until float32 is switched on)
};

// ---------------------------------------------------------------------------------------------
// a4: calculate_epsilon_effective (arcte.py:26-50), one wavefront per seed
// ---------------------------------------------------------------------------------------------

struct PwFrame {
    int64_t lo;
    int64_t n;
    double left;
    int32_t stage;
    int32_t pad;
};

struct EpsShared {
    double leaf[128];
    PwFrame frames[40];
};

// numpy's pairwise float64 sum (oracle: np_pairwise) of a[lo .. lo + n), a[k] = at(k), by one wavefront.  One
// evaluator for both users: the neighbour-degree mean of calculate_epsilon_effective (at(k) = out_degree[indices[k]],
// min/max of a folded in: MINMAX) and the row sums of get_natural_random_walk_matrix (arcte_prepare.hpp: a plain array).
// Leaf (n <= 128): numpy's eight strided partial sums, combined pairwise, then the remainder left to right.
template <bool MINMAX, typename At>
__device__ double pw_leaf_t(At at, int64_t lo, int64_t n, double *leaf, int lane, double &amin, double &amax)
{
    for (int i = lane; i < n; i += WAVE) {
        const double a = at(lo + i);
        leaf[i] = a;
        if (MINMAX) {
            amin = (a < amin) ? a : amin;
            amax = (a > amax) ? a : amax;
        }
    }
    // same-wave LDS write -> read: the DS queue is in order, the compiler keeps the dependency
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    double res;
    if (n < 8) {
        res = 0.0;
        for (int i = 0; i < n; i++) res += leaf[i];
    } else {
        int64_t nfull = n - (n % 8);
        double r = 0.0;
        if (lane < 8) {
            r = leaf[lane];
            for (int64_t i = 8 + lane; i < nfull; i += 8) r += leaf[i];
        }
        double r0 = shfl_f64(r, 0), r1 = shfl_f64(r, 1), r2 = shfl_f64(r, 2), r3 = shfl_f64(r, 3);
        double r4 = shfl_f64(r, 4), r5 = shfl_f64(r, 5), r6 = shfl_f64(r, 6), r7 = shfl_f64(r, 7);
        res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (int64_t i = nfull; i < n; i++) res += leaf[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return res;
}

// the recursion above the leaves, evaluated with an explicit frame stack in LDS
template <bool MINMAX, typename At>
__device__ double pw_sum_wave_t(At at, int64_t lo, int64_t n, EpsShared &S, int lane, double &amin, double &amax)
{
    int sp = 0;
    double ret = 0.0;
    if (lane == 0) { S.frames[0].lo = lo; S.frames[0].n = n; S.frames[0].stage = 0; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    while (sp >= 0) {
        const int64_t flo = S.frames[sp].lo;
        const int64_t fn = S.frames[sp].n;
        const int stage = S.frames[sp].stage;
        if (fn <= 128) {
            ret = pw_leaf_t<MINMAX>(at, flo, fn, S.leaf, lane, amin, amax);
            sp--;
            continue;
        }
        int64_t n2 = fn / 2;
        n2 -= n2 % 8;
        if (stage == 0) {
            if (lane == 0) {
                S.frames[sp].stage = 1;
                S.frames[sp + 1].lo = flo; S.frames[sp + 1].n = n2; S.frames[sp + 1].stage = 0;
            }
            sp++;
        } else if (stage == 1) {
            if (lane == 0) {
                S.frames[sp].left = ret;
                S.frames[sp].stage = 2;
                S.frames[sp + 1].lo = flo + n2; S.frames[sp + 1].n = fn - n2; S.frames[sp + 1].stage = 0;
            }
            sp++;
        } else {
            ret = S.frames[sp].left + ret;
            sp--;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    return ret;
}

// a[i] = out_degree[indices[lo + i]], min/max of a folded in
__device__ double pw_sum_wave(const GraphDev &g, int64_t lo, int64_t n, EpsShared &S, int lane, double &amin, double &amax)
{
    return pw_sum_wave_t<true>([&](int64_t k) -> double { return g.out_degree[g.indices[k]]; }, lo, n, S, lane, amin, amax);
}

// a plain array
__device__ double pw_sum_plain(const double *a, int64_t lo, int64_t n, EpsShared &S, int lane)
{
    double unused_min = 0.0, unused_max = 0.0;
    return pw_sum_wave_t<false>([&](int64_t k) -> double { return a[k]; }, lo, n, S, lane, unused_min, unused_max);
}

// arcte.py:32-48 from the reduced neighbour degrees
__device__ __forceinline__ double epsilon_from_stats(double epsilon, double ds, double sum, int64_t m, double amin, double amax)
{
    double mean = sum / (double)m;                                            // arcte.py:32
    double e = (epsilon * log(1 + ds)) / log(1 + mean);                       // :35
    // :39-40  max/min over i of 1/(ds*a_i): correctly rounded * and / are monotone, so the
    // extrema sit at the extrema of a_i
    double emax = 1 / (ds * amin);
    double emin = 1 / (ds * amax);
    if (m == 0) { emax = -INFINITY; emin = INFINITY; }
    if (e > emax) e = emax;                                                   // :45-48
    else if (e < emin) e = (emin + e) / 2;
    return e;
}

// Rows of at least EPS_BIG_ROW neighbours go to k_epsilon_effective_big: one wavefront would walk them leaf by
// leaf (the 127 441-neighbour hub of the 1M/50M graph alone took 3 ms, the whole kernel's duration).
constexpr int64_t EPS_BIG_ROW = 4096;
constexpr int EPS_BIG_WAVES = 16;      // = 2^4 subtrees of numpy's recursion

__global__ __launch_bounds__(BLOCK) void k_epsilon_effective(GraphDev g, const int32_t *seeds, int64_t nseeds,
                                                             double epsilon, double *eps_out)
{
    __shared__ EpsShared sh[WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (k >= nseeds) return;
    const int32_t seed = seeds[k];
    const int64_t b = g.indptr[seed];
    const int64_t m = g.indptr[seed + 1] - b;
    if (m >= EPS_BIG_ROW) return;                    // handled by k_epsilon_effective_big
    double amin = INFINITY, amax = -INFINITY;
    const double sum = pw_sum_wave(g, b, m, sh[wave], lane, amin, amax);
    amin = wave_min(amin);
    amax = wave_max(amax);
    if (lane == 0) eps_out[k] = epsilon_from_stats(epsilon, g.out_degree[seed], sum, m, amin, amax);
}

// One workgroup of 16 wavefronts per big row: wavefront w evaluates subtree w of the fourth level of numpy's
// pairwise recursion (the split points are the recursion's own), the partial sums are combined in the
// recursion's order, so the result is bit-identical to the one-wavefront evaluation.
__global__ __launch_bounds__(EPS_BIG_WAVES * WAVE) void k_epsilon_effective_big(GraphDev g, const int32_t *seeds,
                                                                              const int32_t *big_pos, int64_t nbig,
                                                                              double epsilon, double *eps_out)
{
    __shared__ EpsShared sh[EPS_BIG_WAVES];
    __shared__ double part[EPS_BIG_WAVES], pmin[EPS_BIG_WAVES], pmax[EPS_BIG_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t k = big_pos[blockIdx.x];
    const int32_t seed = seeds[k];
    const int64_t b = g.indptr[seed];
    const int64_t m = g.indptr[seed + 1] - b;
    int64_t lo = b, n = m;
    for (int level = 3; level >= 0; level--) {       // m >= 4096 keeps every node of the first four levels > 128
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        if ((wave >> level) & 1) { lo += n2; n -= n2; }
        else n = n2;
    }
    double amin = INFINITY, amax = -INFINITY;
    const double sum = pw_sum_wave(g, lo, n, sh[wave], lane, amin, amax);
    amin = wave_min(amin);
    amax = wave_max(amax);
    if (lane == 0) { part[wave] = sum; pmin[wave] = amin; pmax[wave] = amax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[EPS_BIG_WAVES];
        for (int i = 0; i < EPS_BIG_WAVES; i++) t[i] = part[i];
        for (int width = EPS_BIG_WAVES; width > 1; width >>= 1)
            for (int i = 0; i < width / 2; i++) t[i] = t[2 * i] + t[2 * i + 1];      // left + right at every level
        double mn = pmin[0], mx = pmax[0];
        for (int i = 1; i < EPS_BIG_WAVES; i++) { mn = pmin[i] < mn ? pmin[i] : mn; mx = pmax[i] > mx ? pmax[i] : mx; }
        eps_out[k] = epsilon_from_stats(epsilon, g.out_degree[seed], t[0], m, mn, mx);
    }
}

// ---------------------------------------------------------------------------------------------
// a2 + a3 + a5: per-seed FIFO propagation and community extraction
// ---------------------------------------------------------------------------------------------

// Dense per-slot state, one 32-byte sector per node.  An entry is live only while its epoch equals
// the slot's current epoch (one epoch per seed), so "s[:] = 0; r[:] = 0" (arcte.py:337-338) costs
// nothing and first touches are recognised without a separate bitmap.
template <typename T> struct EntryT;
template <> struct __attribute__((aligned(32))) EntryT<double> {
    double r;
    double s;
    double d;         // in_degree of the node (threshold tests and the degree normalisation)
    uint32_t epoch;
    uint32_t pad;
};
// float32 flavour (BASELINE.json configs[4] tolerance sweep): one 16-byte entry per node
template <> struct __attribute__((aligned(16))) EntryT<float> {
    float r;
    float s;
    float d;
    uint32_t epoch;
};
typedef EntryT<double> Entry;
static_assert(sizeof(Entry) == 32, "Entry must be one 32-byte sector");
static_assert(sizeof(EntryT<float>) == 16, "float entry must be 16 bytes");

template <typename T> struct LoT { T r, s; };
template <typename T> struct HiT { T d; uint32_t epoch; };

__device__ __forceinline__ LoT<double> load_lo(const EntryT<double> *e) { double2 t = *reinterpret_cast<const double2 *>(e); return {t.x, t.y}; }
__device__ __forceinline__ HiT<double> load_hi(const EntryT<double> *e)
{
    double2 t = *(reinterpret_cast<const double2 *>(e) + 1);
    return {t.x, (uint32_t)(uint64_t)__double_as_longlong(t.y)};
}
__device__ __forceinline__ void store_lo(EntryT<double> *e, double r, double s) { *reinterpret_cast<double2 *>(e) = make_double2(r, s); }
__device__ __forceinline__ void store_hi(EntryT<double> *e, double d, uint32_t epoch)
{
    *(reinterpret_cast<double2 *>(e) + 1) = make_double2(d, __longlong_as_double((long long)(uint64_t)epoch));
}
__device__ __forceinline__ LoT<float> load_lo(const EntryT<float> *e) { float2 t = *reinterpret_cast<const float2 *>(e); return {t.x, t.y}; }
__device__ __forceinline__ HiT<float> load_hi(const EntryT<float> *e)
{
    float2 t = *(reinterpret_cast<const float2 *>(e) + 1);
    return {t.x, __float_as_uint(t.y)};
}
__device__ __forceinline__ void store_lo(EntryT<float> *e, float r, float s) { *reinterpret_cast<float2 *>(e) = make_float2(r, s); }
__device__ __forceinline__ void store_hi(EntryT<float> *e, float d, uint32_t epoch)
{
    *(reinterpret_cast<float2 *>(e) + 1) = make_float2(d, __uint_as_float(epoch));
}

// the float64 / float32 views of the graph's value arrays
template <typename T> struct GraphValues { const T *data, *in_degree, *edge_in_degree; };
template <typename T> __device__ __forceinline__ GraphValues<T> graph_values(const GraphDev &g);
template <> __device__ __forceinline__ GraphValues<double> graph_values<double>(const GraphDev &g) { return {g.data, g.in_degree, g.edge_in_degree}; }
template <> __device__ __forceinline__ GraphValues<float> graph_values<float>(const GraphDev &g) { return {g.data_f, g.in_degree_f, g.edge_in_degree_f}; }

template <typename T> __device__ __forceinline__ T shfl_real(T v, int src);
template <> __device__ __forceinline__ double shfl_real<double>(double v, int src) { return shfl_f64(v, src); }
template <> __device__ __forceinline__ float shfl_real<float>(float v, int src) { return __shfl(v, src, WAVE); }
template <typename T> __device__ __forceinline__ T wave_min_real(T x)
{
    for (int o = 32; o > 0; o >>= 1) {
        T y = __shfl_xor(x, o, WAVE);
        x = (y < x) ? y : x;
    }
    return x;
}
// margin of the candidate bound: far above one rounding error of the type, far below any real gap
template <typename T> __device__ __forceinline__ T cand_margin();
template <> __device__ __forceinline__ double cand_margin<double>() { return 1.0 - 0x1p-40; }
template <> __device__ __forceinline__ float cand_margin<float>() { return 1.0f - 0x1p-16f; }

// Ring entry of the per-seed FIFO (similarity.py:180): the node, its index in the LDS-resident hot table
// (HOT_NONE when it has none) and its in_degree, all known when the node is enqueued, so a pop needs one
// 16-byte load before it can test r/in_degree.
struct __attribute__((aligned(16))) QEntry {
    int32_t v;
    uint32_t h;
    double d;
};
static_assert(sizeof(QEntry) == 16, "QEntry must be 16 bytes");
constexpr uint32_t HOT_NONE = 0xFFFFu;

// Sentinel of the hot table: "this node's state has moved to the dense HBM state" (it was pushed, or it is the
// seed).  A NaN with a payload no arithmetic on finite inputs produces.
template <typename T> __device__ __forceinline__ T hot_moved();
template <> __device__ __forceinline__ double hot_moved<double>() { return __longlong_as_double(0x7FF8DEAD0000BEEFll); }
template <> __device__ __forceinline__ float hot_moved<float>() { return __uint_as_float(0x7FC0BEEFu); }
__device__ __forceinline__ bool is_moved(double x) { return __double_as_longlong(x) == 0x7FF8DEAD0000BEEFll; }
__device__ __forceinline__ bool is_moved(float x) { return __float_as_uint(x) == 0x7FC0BEEFu; }

// Warm table (HBM, one compact array per slot): the nodes ranked [hotK, warmK2) -- behind the LDS table in the same
// degree order.  One {value, epoch tag} pair per node (two T-sized words: 16 bytes in float64), with the LDS table's
// rule: the value stands for r == s until the node is pushed, a pushed node moves to the dense entry and leaves the
// sentinel.  Nothing else is different from the dense state except the ADDRESS: a slot's warm nodes sit in
// (warmK2 - hotK) * 16 bytes instead of being strewn over its 32 n bytes, so the lines the chip's last-level cache
// keeps for a slot are lines the next deposits hit again (tools/rmw_wall2.hip: the same read-modify-write runs at
// 22-27 G/s on 1-0.5 GB of private tables against 18.6 G/s on the dense layout).
template <typename T> struct WarmT { T x; T tag; };
__device__ __forceinline__ double warm_tag(uint32_t epoch, double) { return __longlong_as_double((long long)(uint64_t)epoch); }
__device__ __forceinline__ float warm_tag(uint32_t epoch, float) { return __uint_as_float(epoch); }
__device__ __forceinline__ bool warm_live(double tag, uint32_t epoch) { return (uint64_t)__double_as_longlong(tag) == (uint64_t)epoch; }
__device__ __forceinline__ bool warm_live(float tag, uint32_t epoch) { return __float_as_uint(tag) == epoch; }
__device__ __forceinline__ WarmT<double> load_warm(const WarmT<double> *e) { double2 t = *reinterpret_cast<const double2 *>(e); return {t.x, t.y}; }
__device__ __forceinline__ WarmT<float> load_warm(const WarmT<float> *e) { float2 t = *reinterpret_cast<const float2 *>(e); return {t.x, t.y}; }
__device__ __forceinline__ void store_warm(WarmT<double> *e, double x, uint32_t epoch) { *reinterpret_cast<double2 *>(e) = make_double2(x, warm_tag(epoch, 0.0)); }
__device__ __forceinline__ void store_warm(WarmT<float> *e, float x, uint32_t epoch) { *reinterpret_cast<float2 *>(e) = make_float2(x, warm_tag(epoch, 0.0f)); }

struct PushParams {
    GraphDev g;
    // hot table (LDS): nodes ranked by pattern in-degree; edge_hot[k] = rank of indices[k] (HOT_NONE beyond the
    // ranked prefix), node_hot[v] likewise per node; a wavefront keeps ranks < hotK on chip
    const uint16_t *edge_hot;
    const uint16_t *node_hot;
    uint32_t hotK;
    void *warm;        // [slots][warmN] WarmT<T>, indexed by rank; ranks [hotK, warmK2) are in use
    uint32_t warmK2;   // <= hotK switches the warm table off
    uint32_t warmN;    // entries per slot
    // work list
    const int32_t *work_pos;   // positions into seeds/eps/out arrays for this launch (NULL = identity)
    int64_t nwork;
    unsigned long long *work_counter;
    const int32_t *seeds;
    const double *eps;
    double one_minus_rho;
    double rho;        // PageRank flavours: s[u] += rho*r[u]
    double lazy;       // lazy flavour: laziness factor
    // per-slot scratch
    void *state;       // [slots][n] EntryT<T>
    uint32_t *slot_epoch;   // [slots] last epoch used by the slot
    QEntry *queue;     // [slots][qcap]
    int32_t *sup;      // [slots][n]   candidate list (see cand_thr)
    uint32_t qcap;     // power of two
    int32_t max_pushes; // per-seed cap, see ST_RUNAWAY
    // outputs
    int32_t *raw;      // raw row arena, allocation order
    unsigned long long rawcap;
    unsigned long long *raw_cursor;
    int64_t *out_off;
    int32_t *out_cnt;
    int32_t *status;
    int32_t *nop;
    unsigned long long *stats;   // [0] pushes [1] edges [2] enqueues [3] support [4] failed seeds [5] candidates [7] split rows
    // MODE 2: (node << contrib_shift | seed - contrib_seed_base, s/in_degree) of every support node of every seed of
    // the batch, for the centrality accumulation (the seeds of a batch are a block of ascending node ids)
    uint64_t *contrib_key;
    double *contrib_val;
    unsigned long long contrib_cap;
    unsigned long long *contrib_cursor;
    int64_t contrib_seed_base;
    int contrib_shift;
    // COOP: a second wavefront per seed walks the second half of long rows (see CoopShared)
    QEntry *hqueue;         // [slots][qcap] staging ring of the helper's enqueues
    int64_t coop_min;       // rows of at least this many edges are split
    // PROF: where the wavefronts' time goes (s_memtime ticks summed over wavefronts): [0] seed set-up, [1] pop batches
    // (queue entries, r, row bounds), [2] pushes of rows that fit one step, [3] longer rows, [4] the rest of the pop loop
    // (re-reads, re-tests), [5] extraction, [6] short pushes, [7] long pushes, [8] pop batches, [9] drawing work
    unsigned long long *prof;
};

// Lane j's double as a wavefront-uniform value (j uniform)
__device__ __forceinline__ double lane_value(double x, int j)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), j);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
    return __hiloint2double(hi, lo);
}

// acc + x[lane 0] + x[lane 1] + ... in exactly that order: the left fold a sequential CPU loop performs, fed by
// coalesced 64-value loads.  The chain is one dependent add per value whichever way it is laid out; this way the
// memory side streams.
__device__ __forceinline__ double ordered_add64(double acc, double x)
{
#pragma unroll
    for (int j = 0; j < WAVE; j++) acc += lane_value(x, j);
    return acc;
}

__device__ __forceinline__ double ordered_add_n(double acc, double x, int count)       // count uniform, < 64
{
    for (int j = 0; j < count; j++) acc += lane_value(x, j);
    return acc;
}

// The contributions of a batch, sorted by (node, seed): where each node's run begins and ends (both 0 = no run;
// the caller zeroes the arrays)
__global__ void k_contribution_bounds(const uint64_t *keys, int64_t m, int shift, int64_t *first, int64_t *last)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const uint64_t v = keys[k] >> shift;
    if (k == 0 || (keys[k - 1] >> shift) != v) first[v] = k;
    if (k == m - 1 || (keys[k + 1] >> shift) != v) last[v] = k + 1;
}

// arcte.pyx:190-191: centrality += s_norm, seed after seed.  One wavefront per node folds the node's run into
// centrality[node] in ascending seed order -- per node exactly the reference's sequence of additions.
__global__ __launch_bounds__(BLOCK) void k_apply_contributions(const double *vals, const int64_t *first, const int64_t *last, int64_t n,
                                                               double *centrality)
{
    const int lane = threadIdx.x & 63;
    const int64_t v = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (v >= n) return;
    const int64_t b = first[v], e = last[v];
    if (b == e) return;
    double acc = centrality[v];
    for (int64_t k0 = b; k0 < e; k0 += WAVE) {
        const int64_t k = k0 + lane;
        const double x = k < e ? vals[k] : 0.0;
        const int64_t left = e - k0;
        acc = left >= WAVE ? ordered_add64(acc, x) : ordered_add_n(acc, x, (int)left);
    }
    if (lane == 0) centrality[v] = acc;
}

// arcte.pyx:210: nodes that were no seeds (no out-edges) get centrality 1
__global__ void k_centrality_non_seeds(const int64_t *indptr, int64_t n, double *centrality)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && indptr[i + 1] == indptr[i]) centrality[i] = 1.0;
}

// COOP launch shape: one workgroup of TWO wavefronts per seed.  Wavefront 0 (the leader) runs the algorithm exactly as
// the one-wavefront kernel does; wavefront 1 (the helper) owns nothing: it waits at a barrier and, for a row of
// coop_min edges or more, walks the row's second half while the leader walks the first.  Exact because the targets of
// one row are distinct nodes (distinct state entries, distinct LDS values) and each receives one deposit; the helper
// stages the nodes it would enqueue and the leader appends them behind its own, which is the row's edge order.  What
// it buys is LDS: the table belongs to the SEED, and half as many seeds share a CU at the same number of wavefronts.
// All control flow is the leader's: the helper follows commands left in LDS, every barrier is met by both.
struct CoopShared {
    unsigned long long seed_wk;     // leader -> helper, per seed: work item, arena cursor at draw time
    unsigned long long seed_cur;
    double c, r_self, cand_thr, w_row;  // leader -> helper, per split row
    int64_t rb, re;
    int32_t row_cmd, u;
    int32_t nsup;                   // candidate count while a row is split (both append through it)
    uint32_t h_cnt;                 // helper -> leader: staged enqueues, support growth, ring overflow
    int32_t h_first, h_ok;
};
constexpr int32_t COOP_ROW = 1, COOP_DONE = 2;

// Registers of one step of the push pipeline: the row data of TILES x 64 edges, and the state gathered for them
template <typename T, int TILES> struct RowStage {
    bool a[TILES];
    int32_t v[TILES];
    uint32_t hh[TILES];
    T w[TILES], d[TILES];
};
template <typename T, int TILES> struct EntStage {
    LoT<T> l[TILES];
    HiT<T> h[TILES];
    T x[TILES];
    bool chip[TILES];
    bool warm[TILES];
};

// MODE 0: full arcte_worker body (extract).  MODE 1: similarity slice only on the dense vectors the
// host placed in slot 0 (k_state_from_dense), left there for k_state_to_dense.  MODE 2: the body of
// arcte_and_centrality (embedding/arcte/cython_opt/arcte.pyx:168-217): MODE 0 plus the centrality contributions.
// VAR 0: cumulative PageRank difference (push.py:41-64, similarity.py:149-222) -- ARCTE proper.
// VAR 1: PageRank limit push (push.py:4-17, similarity.py:11-63).
// VAR 2: lazy PageRank push (push.py:20-38, similarity.py:66-146) with its self re-push loops.
// HOT: the state of the hotK nodes of highest pattern in-degree lives in LDS, one table of hotK values per
// wavefront, zeroed per seed -- exactly the reference's s[:] = 0; r[:] = 0 (arcte.py:337-338).  Until a node is
// pushed its r and s are the SAME number (both receive every deposit, push.py:63-64, and only a push separates
// them, push.py:59), so one value per node is enough (for the PageRank flavours s stays 0 until the push);
// a node that IS pushed -- and the seed -- moves to the dense HBM state and leaves the sentinel behind.  Every
// deposit to an on-chip node is an LDS read-modify-write instead of a random HBM one: with 2 560-5 120 values per
// wavefront that is 20-30 % of the traversed edges of the 1M/50M graph (tools/hot_share.py).
// NARROW (float64 only): every row's transition weights are one and the same number (an unweighted graph: 1/out_degree)
// and every in_degree is exactly representable in float32, so a push streams 10 bytes per edge (index, float in_degree,
// hot rank) instead of 22 -- the weight comes from the row's first entry, the in_degree widens back to the same double.
template <int MODE, int VAR, typename T, int TILES, bool HOT, bool NARROW = false, bool COOP = false, bool PROF = false>
__global__ __launch_bounds__(BLOCK) void k_arcte_seeds(PushParams P)
{
    unsigned long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&]() -> unsigned long long { return PROF ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ULL; };
    static_assert(!COOP || (MODE == 0 && VAR == 0 && HOT && sizeof(T) == 8), "the helper wavefront exists for ARCTE's worker in float64");
    extern __shared__ __attribute__((aligned(16))) unsigned char hot_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t slot = COOP ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const bool helper = COOP && wave == 1;
    const GraphDev &g = P.g;
    const GraphValues<T> gv = graph_values<T>(g);
    EntryT<T> *__restrict__ st = reinterpret_cast<EntryT<T> *>(P.state) + slot * g.n;
    // (the helper's "queue" is its staging ring: the row walk below appends to whatever q, head and tail say)
    QEntry *__restrict__ q = helper ? P.hqueue + slot * (int64_t)P.qcap : P.queue + slot * (int64_t)P.qcap;
    const QEntry *__restrict__ hq = COOP ? P.hqueue + slot * (int64_t)P.qcap : nullptr;
    int32_t *__restrict__ sup = P.sup + slot * g.n;
    const uint32_t qmask = P.qcap - 1;
    const T omr = (T)P.one_minus_rho;
    const uint32_t K = HOT ? P.hotK : 0u;
    T *hot = reinterpret_cast<T *>(hot_raw) + (COOP ? (size_t)0 : (size_t)wave * K);
    CoopShared *S = reinterpret_cast<CoopShared *>(hot_raw + (size_t)K * sizeof(T));      // COOP only: behind the table
    const uint32_t K2 = (HOT && P.warmK2 > K) ? P.warmK2 : K;          // warm ranks: [K, K2)
    WarmT<T> *__restrict__ wm = reinterpret_cast<WarmT<T> *>(P.warm) + slot * (int64_t)P.warmN;   // indexed by rank
    uint32_t epoch = P.slot_epoch[slot];

    // Dynamic seed queue: lane 0 draws the next work item, the wave broadcasts it.  The
    // wave_barrier (convergent, emits nothing) keeps LLVM's jump threading from fusing this
    // lane-0 block with the lane-0 block that ends the previous iteration -- that fusion turned the
    // loop divergent (lanes 1..63 ran ahead without lane 0 and never left it).
    unsigned long long coop_cur = 0;     // COOP: the arena cursor the leader saw when it drew the seed
    auto next_work = [&]() -> unsigned long long {
        __builtin_amdgcn_wave_barrier();
        unsigned long long w = 0;
        if (COOP) {
            if (!helper && lane == 0) {
                S->seed_wk = atomicAdd(P.work_counter, 1ULL);
                S->seed_cur = __hip_atomic_load(P.raw_cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            w = bcast_u64(S->seed_wk);
            coop_cur = bcast_u64(S->seed_cur);
            __syncthreads();            // both have read: the leader may draw again
            return w;
        }
        if (lane == 0) w = atomicAdd(P.work_counter, 1ULL);
        return bcast_u64(w);
    };
    // r of node u whose hot rank is h: on chip unless it has moved
    auto read_r = [&](int32_t u, uint32_t h) -> T {
        if (HOT && h < K) {
            const T x = hot[h];
            if (!is_moved(x)) return x;
        } else if (HOT && h < K2) {
            const WarmT<T> e = load_warm(wm + h);
            if (warm_live(e.tag, epoch) && !is_moved(e.x)) return e.x;
        }
        return st[u].r;
    };
    // (the extra bound is insurance only: a wave can draw at most nwork items, so a loop that ever ran past that
    //  would be a compiler-induced divergence like the one described above, and must still terminate)
    unsigned long long drawn = 0;
    unsigned long long t_mark = tick();
    for (unsigned long long wk = next_work(); wk < (unsigned long long)P.nwork && drawn <= (unsigned long long)P.nwork;
         wk = next_work(), drawn++) {
        if (PROF) { const unsigned long long t = tick(); prof[9] += t - t_mark; t_mark = t; }
        const int32_t pos = P.work_pos ? P.work_pos[wk] : (int32_t)wk;
        const int32_t seed = P.seeds[pos];
        const T eps = (T)P.eps[pos];
        if (MODE != 1) {
            // the arena is already full: this seed is re-run by the host after the arena is drained
            unsigned long long cur = 0;
            if (COOP) cur = coop_cur;
            else {
                if (lane == 0) cur = __hip_atomic_load(P.raw_cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cur = bcast_u64(cur);
            }
            if (cur > P.rawcap) {
                if (lane == 0 && !helper) {
                    P.status[pos] = ST_OUTPUT_OVERFLOW;
                    P.out_cnt[pos] = 0;
                    P.out_off[pos] = 0;
                    P.nop[pos] = 0;
                    atomicAdd(&P.stats[4], 1ULL);
                }
                continue;
            }
        }
        epoch++;                           // every entry of the previous seed is stale from here on
        if (HOT && !helper) {
            // s[:] = 0; r[:] = 0 for the on-chip nodes
            for (uint32_t i = lane; i < K; i += WAVE) hot[i] = T(0);
        }

        uint32_t head = 0, tail = 0;       // ring counters (wave-uniform)
        int32_t nsup = 0;          // candidates
        int32_t nfirst = 0;        // nodes with s != 0 (the support of the similarity slice)
        T cand_thr = T(0);
        int32_t npush = 0;
        unsigned long long nedges = 0;
        bool ok = true, runaway = false;
        bool coop_row = helper;      // a split row is being walked: candidates are counted in LDS

        // row data of one pipeline step: TILES x 64 edges from `base`, clamped to the row's last edge `re - 1`.
        // Every lane issues every load of a step, lanes beyond the row's end at a clamped or dummy address (the
        // row's last edge, the seed's own state entry: lines that are in cache anyway).  Loads under an
        // exec-masked branch would leave the number of outstanding loads unknown to the compiler, which then
        // waits for ALL of them (vmcnt(0)) before the first use and serialises the pipeline.
        auto load_row_at = [&](int64_t base, int64_t re, T w_row, RowStage<T, TILES> &R) {
#pragma unroll
            for (int t = 0; t < TILES; t++) {
                const int64_t k = base + t * WAVE + lane;
                R.a[t] = k < re;
                const int64_t kk = R.a[t] ? k : re - 1;
                R.v[t] = g.indices[kk];
                if (NARROW) { R.w[t] = w_row; R.d[t] = (T)g.edge_in_degree_f[kk]; }
                else { R.w[t] = gv.data[kk]; R.d[t] = gv.edge_in_degree[kk]; }
                R.hh[t] = HOT ? (uint32_t)P.edge_hot[kk] : HOT_NONE;
            }
        };

        // ---- the scatter of one push over the edges [rb, re) of u's row (push.py:60-64) with the ordered enqueue of
        //      similarity.py:194-196 / :214-216: c is what every neighbour receives per unit of transition weight, r_self
        //      r[u] right after the push bookkeeping (what a self-loop lane adds to).
        auto walk = [&](int32_t u, T c, T r_self, int64_t rb, int64_t re, T w_row, bool do_enqueue) __attribute__((always_inline)) {
            // The row is walked TILES x 64 edges at a time as a three-stage software pipeline: the row data (index,
            // weight, in_degree, hot rank) of step i+2 and the state gathers of step i+1 are in flight while step i
            // is added, stored and enqueued.  Legal because the targets of one row are distinct (CSR columns are
            // unique), so a later step's gathers never read what an earlier step of the SAME push stores; across
            // pushes program order holds (the next push's gathers are issued after this push's last store).
            RowStage<T, TILES> Ra, Rb, Rc;
            EntStage<T, TILES> Ea, Eb;
            auto load_row = [&](int64_t base, RowStage<T, TILES> &R) { load_row_at(base, re, w_row, R); };
            auto gather = [&](const RowStage<T, TILES> &R, EntStage<T, TILES> &E) {
#pragma unroll
                for (int t = 0; t < TILES; t++) {
                    // nodes without a place in the LDS table go to HBM at once (warm table or dense state), the others
                    // after a look at the table
                    const bool in_lds = HOT && R.hh[t] < K;
                    const bool in_warm = HOT && !in_lds && R.hh[t] < K2;
                    const bool cold = R.a[t] && !in_lds && !in_warm;
                    const EntryT<T> *e = cold ? st + R.v[t] : st + seed;
                    // (a warm entry has the size of an entry's first half: one load serves either)
                    const void *first = (R.a[t] && in_warm) ? (const void *)(wm + R.hh[t]) : (const void *)e;
                    E.l[t] = load_lo(reinterpret_cast<const EntryT<T> *>(first));
                    E.h[t] = load_hi(e);
                    E.chip[t] = false;
                    E.warm[t] = false;
                    E.x[t] = T(0);
                }
                if (HOT && K2 > K) {
#pragma unroll
                    for (int t = 0; t < TILES; t++) {
                        if (R.a[t] && R.hh[t] >= K && R.hh[t] < K2) {
                            // {value, tag} arrived as (lo.r, lo.s); a moved node is fetched from the dense state
                            const bool live = warm_live(E.l[t].s, epoch);
                            const bool moved = live && is_moved(E.l[t].r);
                            E.warm[t] = !moved;
                            E.x[t] = live ? E.l[t].r : T(0);
                            if (moved) { E.l[t] = load_lo(st + R.v[t]); E.h[t] = load_hi(st + R.v[t]); }
                        }
                    }
                }
                if (HOT) {
#pragma unroll
                    for (int t = 0; t < TILES; t++) {
                        if (R.a[t] && R.hh[t] < K) {
                            E.x[t] = hot[R.hh[t]];
                            E.chip[t] = !is_moved(E.x[t]);
                            if (!E.chip[t]) { E.l[t] = load_lo(st + R.v[t]); E.h[t] = load_hi(st + R.v[t]); }
                        }
                    }
                }
            };
            auto process = [&](const RowStage<T, TILES> &R, const EntStage<T, TILES> &E) {
#pragma unroll
                for (int t = 0; t < TILES; t++) {
                    const bool act = R.a[t];
                    const bool warm = HOT && E.warm[t];
                    const bool chip = (HOT && E.chip[t]) || warm;        // one value stands for r == s
                    const int32_t v = R.v[t];
                    const T w = R.w[t];
                    const T dv = R.d[t];
                    const LoT<T> lo = E.l[t];
                    const bool live = chip || E.h[t].epoch == epoch;
                    const T p = c * w;                                  // push.py:62 / :17 / :38
                    // (an on-chip node is never u itself: u has just moved)
                    const T r_old = chip ? E.x[t] : (live ? ((v != u) ? lo.r : r_self) : T(0));   // a self-loop sees r[u] as just set
                    const T s_old = chip ? ((VAR == 0) ? E.x[t] : T(0)) : (live ? lo.s : T(0));
                    const T r_new = r_old + p;                          // push.py:64
                    const T s_new = (VAR == 0) ? s_old + p : s_old;     // push.py:63 (ARCTE only)
                    if (act) {
                        if (warm) store_warm(wm + R.hh[t], r_new, epoch);
                        else if (chip) hot[R.hh[t]] = r_new;            // == s_new for ARCTE; s stays 0 otherwise
                        else {
                            store_lo(st + v, r_new, s_new);
                            if (!live) store_hi(st + v, dv, epoch);
                        }
                    }
                    if (VAR == 0) {
                        // Candidate list: every node whose s/in_degree has reached cand_thr, a lower bound of
                        // the final selection threshold (s only grows, so each node crosses once).  It replaces
                        // the full touched list: extraction only has to look at candidates.
                        const T bar = cand_thr * dv;
                        const bool cross = act && (s_new > T(0) && s_new >= bar) && !(s_old > T(0) && s_old >= bar);
                        const uint64_t mc = __ballot(cross);
                        if (COOP && coop_row) {
                            if (mc) {
                                int32_t at = 0;
                                if (lane == 0) at = atomicAdd(&S->nsup, (int32_t)__popcll(mc));
                                at = __shfl(at, 0, WAVE);
                                if (cross) sup[at + lane_below(mc)] = v;
                            }
                        } else {
                            if (cross) sup[nsup + lane_below(mc)] = v;
                            nsup += __popcll(mc);
                        }
                        nfirst += __popcll(__ballot(act && s_old == T(0) && s_new != T(0)));   // support of s grows
                    }
                    if (!do_enqueue) continue;
                    const bool enq = act && (r_new / dv >= eps);             // similarity.py:194/214
                    const uint64_t me = __ballot(enq);
                    const uint32_t cnt = __popcll(me);
                    if (cnt) {
                        if (tail - head + cnt > P.qcap) { ok = false; }
                        else {
                            if (enq) {
                                QEntry e;
                                e.v = v; e.h = R.hh[t]; e.d = (double)dv;
                                q[(tail + lane_below(me)) & qmask] = e;
                            }
                            tail += cnt;
                        }
                    }
                }
            };
            constexpr int64_t STEP = TILES * WAVE;
            if (re - rb > STEP) {
                load_row(rb, Ra);
                load_row(rb + STEP, Rb);
                gather(Ra, Ea);
                for (int64_t base = rb; base < re; base += STEP) {
                    // (row data first: the wait for it at the top of the next turn then leaves this turn's gathers
                    //  in flight instead of draining them)
                    load_row(base + 2 * STEP, Rc);
                    gather(Rb, Eb);
                    process(Ra, Ea);
                    if (!ok) break;
                    Ra = Rb; Ea = Eb; Rb = Rc;
                }
            } else if (re > rb) {
                // a row that fits one step (most pushes, a minority of the edges): nothing to overlap
                load_row(rb, Ra);
                gather(Ra, Ea);
                process(Ra, Ea);
            }
        };

        // ---- one push of node u (push.py:41-64) followed by the ordered enqueue of
        //      similarity.py:194-196 / :214-216.  `ru` is r[u] at pop time, `hu` u's hot rank, `du` its in_degree.
        auto push = [&](int32_t u, uint32_t hu, T du, T ru, int64_t rb, int64_t re, bool do_enqueue) {
            const unsigned long long t_push = tick();
            T c;            // what every neighbour receives per unit of transition weight
            T r_self;       // r[u] right after the push bookkeeping (what a self-loop lane adds to)
            // u still on chip: this push is its first, so s[u] == r[u] == ru (ARCTE) or s[u] == 0 (PageRank
            // flavours); it moves to the HBM state now, where the code below finds it as it finds any other node
            bool on_chip = false;
            if (HOT && hu < K) {
                on_chip = !is_moved(hot[hu]);
                if (on_chip && lane == 0) {
                    hot[hu] = hot_moved<T>();
                    store_lo(st + u, ru, (VAR == 0) ? ru : T(0));
                    store_hi(st + u, du, epoch);
                }
            } else if (HOT && hu < K2) {
                // the same for a node of the warm table (u was deposited to in this epoch: its warm entry is live)
                const WarmT<T> e = load_warm(wm + hu);
                on_chip = warm_live(e.tag, epoch) && !is_moved(e.x);
                if (on_chip && lane == 0) {
                    store_warm(wm + hu, hot_moved<T>(), epoch);
                    store_lo(st + u, ru, (VAR == 0) ? ru : T(0));
                    store_hi(st + u, du, epoch);
                }
            }
            if (VAR == 0) {
                c = omr * ru;                                    // push.py:56
                r_self = T(0);
                if (lane == 0) st[u].r = T(0);                    // push.py:59
            } else {
                const T A = (T)P.rho * ru;                     // push.py:10 / :29
                if (VAR == 1) { c = omr * ru; r_self = T(0); }                                  // push.py:11,15
                else { c = omr * (1 - (T)P.lazy) * ru; r_self = omr * (T)P.lazy * (ru); }            // push.py:30-31
                bool grew = false;
                if (lane == 0) {
                    const T s_old = on_chip ? T(0) : st[u].s;   // u is live: it was deposited to, or is the seed
                    const T s_new = s_old + A;              // push.py:14 / :34
                    store_lo(st + u, r_self, s_new);             // push.py:15 / :35
                    grew = s_old == T(0) && s_new != T(0);
                    if (grew) sup[nsup] = u;                     // s is non-zero exactly at pushed nodes
                }
                const int g1 = __popcll(__ballot(grew));
                nsup += g1;
                nfirst += g1;
            }
            const T w_row = (NARROW && re > rb) ? gv.data[rb] : T(0);
            if (COOP && re - rb >= P.coop_min) {
                // split: the helper takes [mid, re), in whole steps
                constexpr int64_t STEP = TILES * WAVE;
                const int64_t mid = rb + ((re - rb) / 2 + STEP - 1) / STEP * STEP;
                if (lane == 0) {
                    S->u = u; S->c = (double)c; S->r_self = (double)r_self; S->cand_thr = (double)cand_thr; S->w_row = (double)w_row;
                    S->rb = mid; S->re = re; S->nsup = nsup; S->row_cmd = COOP_ROW;
                    atomicAdd(&P.stats[7], 1ULL);                  // split rows
                }
                coop_row = true;
                __syncthreads();                                   // (A) the helper starts
                walk(u, c, r_self, rb, mid, w_row, do_enqueue);
                __syncthreads();                                   // (B) both halves are stored
                coop_row = false;
                nsup = S->nsup;
                nfirst += S->h_first;
                if (!S->h_ok) ok = false;
                const uint32_t hcnt = S->h_cnt;
                if (ok && hcnt) {
                    if (tail - head + hcnt > P.qcap) ok = false;
                    else {
                        for (uint32_t i = lane; i < hcnt; i += WAVE) q[(tail + i) & qmask] = hq[i];
                        tail += hcnt;
                    }
                }
            } else {
                walk(u, c, r_self, rb, re, w_row, do_enqueue);
            }
            npush++;
            nedges += (unsigned long long)(re - rb);
            if (npush >= P.max_pushes) { ok = false; runaway = true; }
            if (PROF) {
                // (the time of a push is taken out of the phase it interrupts: t_mark moves forward by it)
                const unsigned long long dt = tick() - t_push;
                const bool is_long = re - rb > (int64_t)TILES * WAVE;
                prof[is_long ? 3 : 2] += dt;
                prof[is_long ? 7 : 6] += 1;
                t_mark += dt;
            }
        };

        if (helper) {
            // serve the split rows of this seed until the leader is done with it
            for (;;) {
                __syncthreads();                                   // (A)
                if (S->row_cmd == COOP_DONE) break;
                head = 0; tail = 0; nfirst = 0; ok = true;
                cand_thr = (T)S->cand_thr;
                walk((int32_t)bcast_u64((uint64_t)(uint32_t)S->u), (T)S->c, (T)S->r_self, (int64_t)bcast_u64((uint64_t)S->rb),
                     (int64_t)bcast_u64((uint64_t)S->re), (T)S->w_row, true);
                if (lane == 0) { S->h_cnt = tail; S->h_first = nfirst; S->h_ok = ok ? 1 : 0; }
                __syncthreads();                                   // (B)
            }
            continue;
        }
        // ---- similarity.py:176-192: s[seed] = r[seed] = 1, one unconditional push
        const int64_t seed_b = g.indptr[seed], seed_e = g.indptr[seed + 1];
        const T seed_d = gv.in_degree[seed];
        if (lane == 0) {
            if (VAR == 0) store_lo(st + seed, T(1), T(1));         // similarity.py:176-177
            else if (MODE != 1) store_lo(st + seed, T(1), T(0));   // similarity.py:26 / :85: only r[seed] = 1
            else st[seed].r = T(1);                                //   (MODE 1: the caller's s[seed] stays)
            if (MODE != 1) store_hi(st + seed, seed_d, epoch);   // MODE 1: the host made every entry live
            if (VAR == 0) sup[0] = seed;
            if (HOT) {
                const uint32_t hs = P.node_hot[seed];
                if (hs < K) hot[hs] = hot_moved<T>();              // the seed's state is the HBM entry just written
                else if (hs < K2) store_warm(wm + hs, hot_moved<T>(), epoch);
            }
        }
        nsup = (VAR == 0) ? 1 : 0;
        nfirst = (VAR == 0) ? 1 : 0;
        if (MODE == 0 && VAR == 0) {
            // Lower bound of the selection threshold (arcte.py:358-360): the threshold is the minimum of
            // s/in_degree over the closed neighbourhood at the END; s never decreases, so the minimum
            // right after the first push (s[b] = c*w_b, s[seed] >= 1) bounds it from below.  Scaled down a
            // hair so that the cheap product test s >= cand_thr*d admits everything the exact division does.
            T lb = T(1) / seed_d;
            const T c0 = omr * T(1);
            for (int64_t k = seed_b + lane; k < seed_e; k += WAVE) {
                const T x = (c0 * gv.data[k]) / gv.edge_in_degree[k];
                lb = (x < lb) ? x : lb;
            }
            cand_thr = wave_min_real<T>(lb) * cand_margin<T>();
        }
        // (PageRank flavours: s is non-zero only at pushed nodes; the candidate list is the pushed nodes)
        if (PROF) { const unsigned long long t = tick(); prof[0] += t - t_mark; t_mark = t; }
        push(seed, HOT_NONE, seed_d, T(1), seed_b, seed_e, true);
        if (VAR == 2) {
            // similarity.py:108-116: re-push the seed while it stays above the threshold, no enqueue
            while (ok) {
                const T ru2 = st[seed].r;
                if (!(ru2 / seed_d >= eps)) break;
                push(seed, HOT_NONE, seed_d, ru2, seed_b, seed_e, false);
            }
        }

        // ---- similarity.py:199-216: FIFO with duplicates.  Up to 64 queue entries are taken per
        //      batch; r/in_degree of all of them is tested in parallel and the first passing entry
        //      (in FIFO order) is pushed; entries before it are no-op pops.  r of the not yet
        //      consumed entries is re-read after every push, so every test sees r at its pop time.
        while (ok && head != tail) {
            const uint32_t navail = tail - head;
            const uint32_t bn = navail < (uint32_t)WAVE ? navail : (uint32_t)WAVE;
            const bool valid = (uint32_t)lane < bn;
            int32_t u_l = 0;
            uint32_t h_l = HOT_NONE;
            T r_l = T(0), d_l = T(1);
            int64_t rb_l = 0, re_l = 0;
            if (valid) {
                const QEntry e = q[(head + lane) & qmask];
                u_l = e.v;
                h_l = e.h;
                d_l = (T)e.d;
                r_l = read_r(u_l, h_l);       // queued nodes were deposited to in this epoch: live
                rb_l = g.indptr[u_l];
                re_l = g.indptr[u_l + 1];
            }
            head += bn;    // the batch lives in registers from here on
            int consumed = 0;
            bool pass = valid && (r_l / d_l >= eps);                                  // similarity.py:204
            if (PROF) {
                // (the ballot makes the wavefront wait for the loads of the batch)
                const unsigned long long any = __ballot(pass);
                asm volatile("" ::"s"(any));
                const unsigned long long t = tick();
                prof[1] += t - t_mark; t_mark = t; prof[8] += 1;
            }
            for (;;) {
                const uint64_t m = __ballot(pass && lane >= consumed);
                if (m == 0) break;
                const int i = __ffsll((unsigned long long)m) - 1;
                const int32_t u = __shfl(u_l, i, WAVE);
                const uint32_t hu = (uint32_t)__shfl((int)h_l, i, WAVE);
                const T du = shfl_real<T>(d_l, i);
                consumed = i + 1;
                // r of a passing entry can only have grown since it was read -- unless the node was pushed in
                // between (the queue holds duplicates): read it again, it is this entry's pop time now
                const T ru = read_r(u, hu);
                if (!(ru / du >= eps)) {
                    if (lane == i) pass = false;
                    continue;
                }
                const int64_t rb = shfl_i64(rb_l, i);
                const int64_t re = shfl_i64(re_l, i);
                push(u, hu, du, ru, rb, re, true);
                if (VAR == 2) {
                    // similarity.py:136-144: re-push the same node while it stays above the threshold
                    while (ok) {
                        const T ru2 = st[u].r;
                        if (!(ru2 / du >= eps)) break;
                        push(u, hu, du, ru2, rb, re, false);
                    }
                }
                if (!ok) break;
                // re-test the entries that did not pass: the push may have lifted them over the threshold (a passing
                // entry is re-read when its turn comes), so every test sees r at its pop time
                if (valid && lane >= consumed && !pass) {
                    r_l = read_r(u_l, h_l);
                    pass = r_l / d_l >= eps;
                }
            }
        }

        if (PROF) { const unsigned long long t = tick(); prof[4] += t - t_mark; t_mark = t; }
        // ---- arcte.py:352-376: degree-normalise, threshold = min over the closed neighbourhood,
        //      select everything at or above it, emit iff larger than the base community.
        //      One pass over the candidate list: selected nodes are compacted in place, then copied.
        int32_t sta = ok ? ST_OK : (runaway ? ST_RUNAWAY : ST_QUEUE_OVERFLOW);
        int32_t emitted = 0, support = 0;
        const int32_t ncand = nsup;
        unsigned long long off = 0;
        if (MODE != 1 && ok) {
            const int64_t sb = g.indptr[seed], se = g.indptr[seed + 1];
            T thr = st[seed].s / st[seed].d;
            bool miss = st[seed].s == T(0), selfloop = false;
            for (int64_t k = sb + lane; k < se; k += WAVE) {
                const int32_t v = g.indices[k];
                selfloop |= (v == seed);
                T sv = T(0);
                bool chip = false;
                if (HOT) {
                    const uint32_t hv = P.edge_hot[k];
                    if (hv < K) {
                        const T x = hot[hv];
                        chip = !is_moved(x);
                        if (chip) sv = (VAR == 0) ? x : T(0);
                    } else if (hv < K2) {
                        const WarmT<T> e = load_warm(wm + hv);
                        const bool live = warm_live(e.tag, epoch);
                        chip = !(live && is_moved(e.x));         // never touched (s = 0) or one value for r == s
                        if (chip) sv = (live && VAR == 0) ? e.x : T(0);
                    }
                }
                if (!chip) {
                    const LoT<T> lo = load_lo(st + v);
                    const HiT<T> hi = load_hi(st + v);
                    sv = (hi.epoch == epoch) ? lo.s : T(0);
                }
                miss |= (sv == T(0));
                const T x = sv / gv.edge_in_degree[k];
                thr = (x < thr) ? x : thr;
            }
            thr = wave_min_real<T>(thr);
            const bool missing = __ballot(miss) != 0;
            const bool any_selfloop = __ballot(selfloop) != 0;
            // MODE 2 (arcte.pyx:125-241): the candidate list holds EVERY node of the support (cand_thr = 0), each of
            // them hands (node, seed, s/in_degree) to the centrality accumulation (arcte.pyx:190-191); a closed
            // neighbourhood that is not inside the support emits nothing (the scan of :200-208 never completes)
            // instead of failing; the neighbourhood is a set there (:196-198): a self-loop does not count twice.
            unsigned long long coff = 0;
            bool contribute = false;
            if (MODE == 2) {
                if (lane == 0) coff = atomicAdd(P.contrib_cursor, (unsigned long long)nsup);
                coff = bcast_u64(coff);
                contribute = coff + (unsigned long long)nsup <= P.contrib_cap;
                if (!contribute) sta = ST_CONTRIB_OVERFLOW;
            }
            const int64_t base_size = (MODE == 2) ? (se - sb) + (any_selfloop ? 0 : 1) : (se - sb) + 1;
            if (MODE != 2 && VAR == 0 && missing) sta = ST_MISSING_BASE;
            else if (VAR != 0 && (missing || any_selfloop)) {
                // arcte.py:129-133: the PageRank flavours skip a seed whose closed neighbourhood is not inside
                // the support; intersect1d de-duplicates, so a seed with a self-loop never passes the guard
                support = nfirst;
            } else {
                int32_t cnt = 0;
                for (int32_t i0 = 0; i0 < nsup; i0 += WAVE) {
                    const int32_t i = i0 + lane;
                    bool sel = false;
                    int32_t v = 0;
                    if (i < nsup) {
                        v = sup[i];
                        bool chip = false;
                        T sv = T(0), dv = T(1);
                        if (HOT) {
                            const uint32_t hv = P.node_hot[v];
                            if (hv < K) {
                                const T x = hot[hv];
                                chip = !is_moved(x);
                                // (candidates of the PageRank flavours were pushed, so they are never on chip)
                                if (chip) { sv = (VAR == 0) ? x : T(0); dv = gv.in_degree[v]; }
                            } else if (hv < K2) {
                                const WarmT<T> e = load_warm(wm + hv);
                                const bool live = warm_live(e.tag, epoch);
                                chip = !(live && is_moved(e.x));
                                if (chip) { sv = (live && VAR == 0) ? e.x : T(0); dv = gv.in_degree[v]; }
                            }
                        }
                        if (!chip) { sv = st[v].s; dv = st[v].d; }
                        const T xn = sv / dv;
                        sel = xn >= thr;                                      // arcte.py:363-367
                        if (MODE == 2) {
                            sel = sel && !missing;
                            if (contribute) {
                                P.contrib_key[coff + i] = ((uint64_t)(uint32_t)v << P.contrib_shift) | (uint64_t)(seed - P.contrib_seed_base);
                                P.contrib_val[coff + i] = (double)xn;
                            }
                        }
                    }
                    const uint64_t ms = __ballot(sel);
                    if (sel) sup[cnt + lane_below(ms)] = v;               // in place: cnt <= i0
                    cnt += __popcll(ms);
                }
                support = nfirst;
                if ((int64_t)cnt > base_size && sta == ST_OK) {                       // arcte.py:370 / arcte.pyx:211
                    if (lane == 0) off = atomicAdd(P.raw_cursor, (unsigned long long)cnt);
                    off = bcast_u64(off);
                    if (off + (unsigned long long)cnt > P.rawcap) sta = ST_OUTPUT_OVERFLOW;
                    else {
                        for (int32_t i = lane; i < cnt; i += WAVE) P.raw[off + i] = sup[i];
                        emitted = cnt;
                    }
                }
            }
        }
        if (lane == 0) {
            P.status[pos] = sta;
            P.out_cnt[pos] = emitted;
            P.out_off[pos] = (int64_t)off;
            P.nop[pos] = npush;
            if (sta == ST_OK) {
                atomicAdd(&P.stats[0], (unsigned long long)npush);
                atomicAdd(&P.stats[1], nedges);
                atomicAdd(&P.stats[2], (unsigned long long)tail);
                atomicAdd(&P.stats[3], (unsigned long long)support);
                atomicAdd(&P.stats[5], (unsigned long long)ncand);
            } else {
                atomicAdd(&P.stats[4], 1ULL);
            }
        }
        if (COOP) {
            if (lane == 0) S->row_cmd = COOP_DONE;
            __syncthreads();                                       // (A): releases the helper into the next seed
        }
        if (PROF) { const unsigned long long t = tick(); prof[5] += t - t_mark; t_mark = t; }
    }
    if (lane == 0 && !helper) P.slot_epoch[slot] = epoch;
    if (PROF && lane == 0 && P.prof) {
#pragma unroll
        for (int k = 0; k < 10; k++) atomicAdd(P.prof + k, prof[k]);
    }
}

// copy per-seed segments src[src_off[p] .. +cnt[p]) -> dst[dst_off[p] ..), one wavefront per segment;
// p = work_pos[k] (identity when NULL)
__global__ __launch_bounds__(BLOCK) void k_gather_segments(const int32_t *src, const int64_t *src_off, const int32_t *cnt,
                                                           const int64_t *dst_off, int32_t *dst, const int32_t *work_pos,
                                                           int64_t nseg)
{
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (k >= nseg) return;
    const int64_t p = work_pos ? work_pos[k] : k;
    const int32_t c = cnt[p];
    const int32_t *s = src + src_off[p];
    int32_t *d = dst + dst_off[p];
    for (int32_t i = lane; i < c; i += WAVE) d[i] = s[i];
}

// caller's dense s, r -> slot 0, every entry live in the epoch the slice kernel is about to use
template <typename T>
__global__ void k_state_from_dense(const double *s, const double *r, const double *in_degree, void *state,
                                   const uint32_t *slot_epoch, int64_t n)
{
    EntryT<T> *st = reinterpret_cast<EntryT<T> *>(state);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        store_lo(st + i, (T)r[i], (T)s[i]);
        store_hi(st + i, (T)in_degree[i], slot_epoch[0] + 1);
    }
}

// slot 0 -> dense s, r (slot_epoch[0] is the epoch the slice kernel just used)
template <typename T>
__global__ void k_state_to_dense(const void *state, const uint32_t *slot_epoch, double *s, double *r, int64_t n)
{
    const EntryT<T> *st = reinterpret_cast<const EntryT<T> *>(state);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const LoT<T> lo = load_lo(st + i);
        const bool live = load_hi(st + i).epoch == slot_epoch[0];
        r[i] = live ? (double)lo.r : 0.0;
        s[i] = live ? (double)lo.s : 0.0;
    }
}

__global__ void k_to_float(const double *in, float *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// ---- result assembly: (row, column) pairs -> stable sort by ROW -> CSR ----------------------------------------
// The pairs are laid out so that inside every row they already stand in ascending column order (base block in
// CSR order, then the local block seed by seed in ascending seed id); a STABLE radix sort on the row id alone
// (log2 n bits: three 8-bit passes at n = 1M instead of eight over a 64-bit key) then yields the canonical CSR.
// local block: members of the seed whose ascending rank is j become pairs (member, col_offset + seed id) at
// dst_off[j]; seg_of[j] = position of that seed in the run.
__global__ __launch_bounds__(BLOCK) void k_pairs_local(const int32_t *rows, const int64_t *colptr, const int32_t *seeds,
                                                       const int32_t *seg_of, const int64_t *dst_off, int64_t nseeds,
                                                       uint32_t col_offset, uint32_t *key_row, uint32_t *val_col)
{
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (j >= nseeds) return;
    const int64_t k = seg_of[j];
    const int64_t b = colptr[k], e = colptr[k + 1];
    const uint32_t col = col_offset + (uint32_t)seeds[k];
    const int64_t o = dst_off[j];
    for (int64_t i = b + lane; i < e; i += WAVE) {
        key_row[o + (i - b)] = (uint32_t)rows[i];
        val_col[o + (i - b)] = col;
    }
}

// base block I + pattern(W) (arcte.py:676-679): row i gets its stored columns with i itself merged in at its sorted
// place; when the row already stores i (self-loop) the identity entry is dropped here and the host doubles that
// value instead.  Row i writes deg(i) + 1 pairs at indptr[i] + i; a dropped identity entry leaves row id `n`,
// which sorts behind every real row and is cut off by the caller.
__global__ __launch_bounds__(BLOCK) void k_pairs_base(const int64_t *indptr, const int32_t *indices, int64_t n, uint32_t *key_row,
                                                      uint32_t *val_col)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t b = indptr[i], e = indptr[i + 1];
    const int64_t o = b + i;
    bool loop = false;
    int64_t below = 0;       // stored columns smaller than i (rows are ascending): the identity entry's place
    for (int64_t k = b + lane; k < e; k += WAVE) {
        const int32_t c = indices[k];
        loop |= (c == (int32_t)i);
        below += (c < (int32_t)i) ? 1 : 0;
    }
    const bool any_loop = __ballot(loop) != 0;
    for (int off = 32; off > 0; off >>= 1) below += __shfl_xor((long long)below, off, WAVE);
    for (int64_t k = b + lane; k < e; k += WAVE) {
        const int32_t c = indices[k];
        const int64_t pos = (k - b) + ((!any_loop && c > (int32_t)i) ? 1 : 0);
        key_row[o + pos] = (uint32_t)i;
        val_col[o + pos] = (uint32_t)c;
    }
    if (lane == 0) {
        const int64_t pos = any_loop ? (e - b) : below;
        key_row[o + pos] = any_loop ? (uint32_t)n : (uint32_t)i;
        val_col[o + pos] = (uint32_t)i;
    }
}

// sorted row ids -> indptr[r] = first position whose row is >= r, for r in [0, n]
__global__ void k_rows_to_indptr_u32(const uint32_t *rows, int64_t nkeys, int64_t n, int64_t *indptr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n) return;
    int64_t lo = 0, hi = nkeys;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (rows[mid] < (uint32_t)t) lo = mid + 1; else hi = mid;
    }
    indptr[t] = lo;
}

// work-order keys: seeds with a big row first (their first push and their threshold pass walk the whole row)
__global__ void k_front_keys(uint64_t *keys, const int32_t *pos, int64_t npos)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npos) keys[pos[i]] = 0;
}

// ---- hot table: rank the nodes by pattern in-degree ---------------------------------------------------------
// number of stored edges that point at each node
__global__ void k_column_counts(const int32_t *indices, int64_t nnz, uint32_t *count)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) atomicAdd(count + indices[k], 1u);
}

// sort keys: descending count (the radix sort is ascending and stable, so ties keep the node order)
__global__ void k_rank_keys(const uint32_t *count, int64_t n, uint32_t *keys, int32_t *ids)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = ~count[i]; ids[i] = (int32_t)i; }
}

// node_hot[v] = rank of v when it is among the `ranked` nodes of highest count (node_hot was filled with HOT_NONE)
__global__ void k_assign_hot(const int32_t *sorted_ids, int64_t ranked, uint16_t *node_hot)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ranked) node_hot[sorted_ids[i]] = (uint16_t)i;
}

// the rank of every stored edge's target: streams with the row like the weight and the in_degree
__global__ void k_edge_hot(const int32_t *indices, const uint16_t *node_hot, uint16_t *edge_hot, int64_t nnz)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) edge_hot[k] = node_hot[indices[k]];
}

// May the propagation stream the narrow rows (see k_arcte_seeds)?  flags[0]: some row holds two different weights;
// flags[1]: some in_degree does not survive the round trip through float32.  One wavefront per row.
__global__ __launch_bounds__(BLOCK) void k_check_narrow(const int64_t *indptr, const double *data, const double *in_degree, int64_t n,
                                                        int32_t *flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t b = indptr[i], e = indptr[i + 1];
    bool differs = false;
    if (e > b) {
        const double w0 = data[b];
        for (int64_t k = b + lane; k < e; k += WAVE) differs |= (__double_as_longlong(data[k]) != __double_as_longlong(w0));
    }
    if (differs) flags[0] = 1;
    if (lane == 0 && (double)(float)in_degree[i] != in_degree[i]) flags[1] = 1;
}

// in_degree[indices[k]] for every stored edge
__global__ void k_edge_in_degree(const int32_t *indices, const double *in_degree, double *out, int64_t nnz)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) out[k] = in_degree[indices[k]];
}

// ---- streaming-rate probes (bench.py reports the roofline fraction against the measured rate too) ----------
__global__ __launch_bounds__(256) void k_stream_read(const uint4 *src, int64_t n16, unsigned long long *sink)
{
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ULL);
}

__global__ __launch_bounds__(256) void k_stream_copy(const uint4 *src, uint4 *dst, int64_t n16)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// push.py:41-64 (variant 0), :4-17 (variant 1), :20-38 (variant 2) on dense device vectors, one workgroup
__global__ __launch_bounds__(BLOCK) void k_single_push(double *s, double *r, const double *w_i, const int32_t *a_i,
                                                       int64_t deg, int64_t push_node, double rho, double one_minus_rho,
                                                       int variant, double lazy)
{
    __shared__ double commute;
    if (threadIdx.x == 0) {
        const double ru = r[push_node];
        if (variant == 0) {
            commute = one_minus_rho * ru;
            r[push_node] = 0.0;
        } else {
            s[push_node] += rho * ru;
            if (variant == 1) { commute = one_minus_rho * ru; r[push_node] = 0.0; }
            else { commute = one_minus_rho * (1 - lazy) * ru; r[push_node] = one_minus_rho * lazy * (ru); }
        }
    }
    __syncthreads();
    __threadfence_block();
    const double c = commute;
    for (int64_t k = threadIdx.x; k < deg; k += BLOCK) {
        const int32_t v = a_i[k];
        const double p = c * w_i[k];
        if (variant == 0) s[v] += p;
        r[v] += p;
    }
}

}  // namespace
