// ARCTE eps-truncated absorbing-random-walk propagation for MI355X (gfx950 / CDNA4): host side and C ABI.
//
// One 64-lane wavefront owns one seed at a time ("slot"): it runs the reference's strictly
// sequential FIFO of similarity.py:149-222 exactly, and spends its 64 lanes on the edges of
// the row being pushed (push.py:62-64: distinct targets, no conflicts).  Thousands of slots
// are in flight per GPU, so the chip is kept busy by seed-level parallelism while every seed keeps the
// reference's operation order -- which is what makes the output sparsity pattern bit-exact.
//
// Where a seed's state lives (csrc/arcte_lines.hpp, the default since round 3): nodes are named by rank; the
// highest ranks keep one value in LDS, the others one float64 in strided 64-byte lines of the slot's memory whose
// first touch is a blind whole-line write arbitrated by an LDS bitmap; pushed nodes move to a compact {r, s} array.
// The dense-state kernel of rounds 1-2 (csrc/arcte_kernels.hpp: 32-byte {r, s, in_degree, epoch} entries) serves
// arcte_hip_similarity_slice, float32 and the A/B knobs; its slots are made on demand.
//
// Arithmetic notes (all pinned by tests against the CPU oracle):
//   - built with -ffp-contract=off: p = c*w then s+p, r+p are separate IEEE operations as in
//     push.py:62-64, never an FMA;
//   - thresholds use true IEEE division r/in_degree >= eps (similarity.py:204,214);
//   - the neighbour-degree mean of calculate_epsilon_effective follows numpy's pairwise order.
//
// C ABI: include/arcte_hip.h.  No CPU fallback lives here.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "arcte_hip.h"

#include "arcte_kernels.hpp"
#include "arcte_lines.hpp"
#include "arcte_prepare.hpp"
#include "arcte_features.hpp"

namespace {

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

thread_local std::string g_err = "";

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ARCTE_HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

// (slot-memory candidates that lost their draw are parked, not freed: see draw_slot_memory; they are returned when an
//  allocation fails)
bool free_parked_buffers();

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    hipError_t alloc(size_t c)
    {
        release();
        count = c;
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(c, 1) * sizeof(T));
        if (e != hipSuccess && free_parked_buffers()) {
            (void)hipGetLastError();
            e = hipMalloc((void **)&p, std::max<size_t>(c, 1) * sizeof(T));
        }
        if (e == hipSuccess) capacity = std::max<size_t>(c, 1);
        return e;
    }
    // grow-only variant for per-run buffers: a repeated run of the same size allocates nothing
    size_t capacity = 0;
    hipError_t reserve(size_t c)
    {
        if (p && c <= capacity) {
            count = c;
            return hipSuccess;
        }
        hipError_t e = alloc(c);
        if (e == hipSuccess) capacity = std::max<size_t>(c, 1);
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
        capacity = 0;
    }
    size_t bytes() const { return count * sizeof(T); }
};

// releases a temporary DevBuf when the scope is left early (HIP_TRY returns); a buffer that changed hands (p = nullptr)
// is left alone
template <typename T>
struct ScopedRelease {
    DevBuf<T> &b;
    explicit ScopedRelease(DevBuf<T> &buf) : b(buf) {}
    ~ScopedRelease() { b.release(); }
};

// The slot buffers are tens of GB.  hipFree of such a buffer is deferred by the runtime and the NEXT large hipMalloc
// pays for it -- 1.7 to 6 s per allocation on the MI355X boxes (tools/alloc_time.hip) -- which would dominate a
// second arcte() call in the same process.  So a context hands its big buffers to a small process-wide cache when it
// is destroyed and the next context of the same shape takes them back (it clears them anyway);
// arcte_hip_trim() returns the memory to the driver.
struct BigCacheEntry { int device; void *p; size_t bytes; };
std::mutex g_big_mutex;
std::vector<BigCacheEntry> g_big_cache;
// Slot-memory candidates that lost a placement draw (draw_slot_memory): kept allocated, so that the allocator cannot hand
// the same slow memory out again and no deferred hipFree has to be paid for by the next hipMalloc (2.2 s measured);
// returned by arcte_hip_trim() and whenever an allocation fails.  g_best_probe: the fastest probe rate seen per
// (device, size) in this process -- a candidate that reaches it is taken at once.
std::vector<BigCacheEntry> g_parked;
// line-state contexts of this process alive per device, with the slot memory they plan to hold: a context that SHARES its
// GPU with another LARGE one (several workers on one device) takes packed slots and the first allocation it gets -- drawing
// candidates of tens of GB beside another context's slots is what made three workers on one GPU 2.3x slower than one
// (profiles/r04/multi_worker_1m.txt, first record).  Small contexts (a forgotten handle of a test graph) do not count.
struct LiveLines { int device; const void *ctx; size_t planned; };
std::vector<LiveLines> g_live_lines;
// registers ctx's plan and says whether another context of this device plans (or holds) 1 GB of slot memory or more
bool plan_slot_memory(int device, const void *ctx, size_t planned)
{
    std::lock_guard<std::mutex> lock(g_big_mutex);
    bool shared = false, found = false;
    for (auto &e : g_live_lines) {
        if (e.ctx == ctx) { e.planned = planned; found = true; }
        else if (e.device == device && e.planned >= ((size_t)1 << 30)) shared = true;
    }
    if (!found) g_live_lines.push_back({device, ctx, planned});
    return shared;
}
void forget_slot_memory_plan(const void *ctx)
{
    std::lock_guard<std::mutex> lock(g_big_mutex);
    for (size_t i = 0; i < g_live_lines.size(); i++)
        if (g_live_lines[i].ctx == ctx) { g_live_lines.erase(g_live_lines.begin() + (long)i); break; }
}
struct BestProbe { int device; size_t bytes; double rate; };
std::vector<BestProbe> g_best_probe;

bool free_parked_buffers()
{
    std::lock_guard<std::mutex> lock(g_big_mutex);
    const bool any = !g_parked.empty();
    int current = 0;
    const bool have_current = hipGetDevice(&current) == hipSuccess;
    for (auto &e : g_parked) {
        (void)hipSetDevice(e.device);
        (void)hipFree(e.p);
    }
    g_parked.clear();
    if (any && have_current) (void)hipSetDevice(current);        // (the caller's allocation goes on)
    return any;
}

// the parked buffers of one device: all of them, or (keep_bytes > 0) those of any other size -- a draw of another shape
// has no use for them and they would sit on memory the new slots want
void free_parked_on(int device, size_t keep_bytes = 0)
{
    std::lock_guard<std::mutex> lock(g_big_mutex);
    for (size_t i = 0; i < g_parked.size();) {
        if (g_parked[i].device == device && (keep_bytes == 0 || g_parked[i].bytes != keep_bytes)) {
            (void)hipFree(g_parked[i].p);          // (callers have set the device)
            g_parked.erase(g_parked.begin() + (long)i);
        } else i++;
    }
}

size_t parked_bytes_on(int device)
{
    std::lock_guard<std::mutex> lock(g_big_mutex);
    size_t b = 0;
    for (const auto &e : g_parked)
        if (e.device == device) b += e.bytes;
    return b;
}
constexpr size_t BIG_BUFFER = (size_t)256 << 20;

template <typename T>
hipError_t alloc_cached(DevBuf<T> &b, size_t count, int device)
{
    b.release();
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (bytes >= BIG_BUFFER) {
        std::lock_guard<std::mutex> lock(g_big_mutex);
        for (size_t i = 0; i < g_big_cache.size(); i++)
            if (g_big_cache[i].device == device && g_big_cache[i].bytes == bytes) {
                b.p = (T *)g_big_cache[i].p;
                b.count = count;
                b.capacity = std::max<size_t>(count, 1);
                g_big_cache.erase(g_big_cache.begin() + (long)i);
                return hipSuccess;
            }
        // nothing of that size: whatever is cached for this device is of a stale shape
        for (size_t i = 0; i < g_big_cache.size();)
            if (g_big_cache[i].device == device) { (void)hipFree(g_big_cache[i].p); g_big_cache.erase(g_big_cache.begin() + (long)i); }
            else i++;
    }
    return b.alloc(count);
}

template <typename T>
void release_cached(DevBuf<T> &b, int device)
{
    const size_t bytes = b.capacity * sizeof(T);
    if (b.p && bytes >= BIG_BUFFER) {
        std::lock_guard<std::mutex> lock(g_big_mutex);
        if (g_big_cache.size() < 16) {
            g_big_cache.push_back({device, (void *)b.p, bytes});
            b.p = nullptr;
            b.count = 0;
            b.capacity = 0;
            return;
        }
    }
    b.release();
}

// The slot memory of the line state: a hipMalloc buffer that is parked in the process-wide cache between contexts.
// (Composing it from physical chunks mapped at chunk-aligned virtual addresses -- hipMemCreate / hipMemAddressReserve /
// hipMemMap -- was tried in round 3 to take the chance out of how virtual and physical addresses are aligned to each
// other: with 64 MB and 1 GB chunks the kernel's duration stayed on the same two levels, with 256 MB chunks three runs
// out of three returned WRONG results -- a node's value read back as untouched -- so that path is not in the library.)
struct SlotMem {
    char *p = nullptr;
    size_t size = 0;
    DevBuf<char> plain;
    size_t bytes() const { return size; }
    hipError_t alloc(size_t bytes, int device)
    {
        release(device);
        size = bytes;
        if (bytes == 0) return hipSuccess;
        hipError_t r = alloc_cached(plain, bytes, device);
        p = plain.p;
        return r;
    }
    void release(int device)
    {
        release_cached(plain, device);
        plain.release();
        p = nullptr;
        size = 0;
    }
};

int32_t max_pushes_limit()
{
    if (const char *env = getenv("ARCTE_HIP_MAX_PUSHES")) {
        long long v = atoll(env);
        if (v > 0 && v < (1ll << 31)) return (int32_t)v;
    }
    return 1 << 26;
}

uint32_t next_pow2(uint64_t x);
// The FIFO holds a few hundred entries for typical seeds; an overflowing seed is re-run with a 4x ring.
uint32_t default_queue_capacity(int64_t n);

uint32_t next_pow2(uint64_t x)
{
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return (uint32_t)std::min<uint64_t>(p, 1ull << 31);
}

uint32_t default_queue_capacity(int64_t n)
{
    return std::min<uint32_t>(1u << 20, std::max<uint32_t>(4096u, next_pow2((uint64_t)std::max<int64_t>(n / 16, 1))));
}

}  // namespace

struct arcte_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    int cus = 0;
    int64_t n = 0, nnz = 0;
    DevBuf<int64_t> indptr;
    DevBuf<int32_t> indices;
    DevBuf<double> data, out_degree, in_degree, edge_in_degree;
    DevBuf<float> data_f, in_degree_f, edge_in_degree_f;
    int float32 = 0;        // arithmetic type of the propagation kernels
    int state_is_f32 = 0;   // layout the state buffer currently holds
    // slots
    int64_t slots = 0;
    uint32_t qcap = 0;
    DevBuf<Entry> state;
    DevBuf<uint32_t> slot_epoch;
    DevBuf<QEntry> queue;
    DevBuf<int32_t> sup;
    // hot table (LDS-resident state of the highest-degree nodes) and the launch shape that goes with it
    DevBuf<uint16_t> node_hot, edge_hot;
    DevBuf<char> warm;           // [slots][warm_n] {value, tag} pairs (16 bytes each): the ranks behind the LDS table
    int64_t warm_n = 0;          // entries per slot
    uint32_t warm_k2 = 0;        // ranks below this have a warm entry (0: off)
    int64_t hot_ranked = 0;      // nodes that carry a rank (<= HOT_NONE)
    // arcte_and_centrality (arcte.pyx:125-241)
    DevBuf<uint64_t> contrib_key;
    DevBuf<double> contrib_val, centrality;
    int64_t contrib_seed_base = 0;          // the running batch of arcte_hip_run_centrality: key = node << shift | seed - base
    int contrib_shift = 32;
    DevBuf<uint64_t> contrib_key_sorted;    // sort output of one batch; kept with the context like the arena itself
    DevBuf<double> contrib_val_sorted;
    DevBuf<char> contrib_temp;
    DevBuf<int64_t> run_first, run_last;    // per node: its run in the sorted batch
    int centrality_run = 0;      // the last run was arcte_hip_run_centrality: columns are numbered by a running counter
    DevBuf<int32_t> ranked_ids;  // every node by descending pattern in-count, ties by node id (stable)
    int64_t nseeds_all = 0;      // arcte.py:617: how many of them have an in-count above 1 = the seed list
    int waves_per_block = 1;     // wavefronts per workgroup of k_arcte_seeds
    int coop = 0;                // a helper wavefront per seed walks the second half of long rows (CoopShared)
    int64_t coop_min = 512;      // rows of at least this many edges are split
    DevBuf<QEntry> hqueue;       // [slots][qcap] the helpers' staging rings
    DevBuf<unsigned long long> prof;   // ARCTE_HIP_PROFILE=1: phase ticks of the push kernel (PushParams::prof)
    int waves_per_cu = 0;        // resident wavefronts per CU the slot count was sized for
    int tiles = 2;               // 64-edge tiles per push iteration
    int narrow = 0;              // uniform row weights + float32-exact in_degrees: the push streams 10 bytes per edge
    uint64_t seeds_since_clear = 0;
    std::vector<int32_t> row_len;   // host copy of the row lengths: the work order is heaviest seed first
    // line state (arcte_lines.hpp): nodes named by rank, one float64 per node and slot in strided 64-byte lines, a
    // touched-line bitmap in LDS, pushed nodes in a compact {r, s} array
    int lines = 0;                  // the context runs k_arcte_lines (float64 worker / centrality runs)
    int dense_auto = 0;             // the caller asked for an automatic slot count (dense slots are made on demand)
    int64_t want_slots = 0, want_queue = 0;   // what the caller asked for at creation
    DevBuf<uint32_t> edge_rank, node_rank;   // (edge_rank: the packed words when `pack` is set)
    DevBuf<float> in_degree_rf;              // packed rows: float32 in_degree by rank (the in_degrees a word has no room for)
    int pack = 0;                            // rows stream as ONE 32-bit word per edge (arcte_lines.hpp, ROWS == 2)
    uint32_t rank_bits = 0;
    int64_t pack_escapes = 0, pack_escape_end = 0;   // nodes whose in_degree is looked up, one past the last such rank
    DevBuf<int64_t> rowspan;
    DevBuf<double> in_degree_r;
    int64_t l_slots = 0;
    uint32_t l_M = 0, l_Mshift = 0, l_MB = 0, l_MBshift = 0, l_qcap = 0, l_pcap = 0, l_scap = 0;
    int l_waves_per_cu = 0;
    // ONE block per slot for what a wavefront touches all the time, [region A values | ring | candidate list | pushed
    // state | region B bits], and one for region B's values; both sizes are powers of two (lines_layout)
    SlotMem l_block, l_blockb;
    size_t l_block_bytes = 0, l_blockb_bytes = 0, l_off_queue = 0, l_off_sup = 0, l_off_ps = 0, l_off_gbm = 0, l_off_b = 0;
    int l_ind = 0;                // region B's lines are indirect (arcte_lines.hpp, IND): bidx table + pool instead of 8 bytes per node
    uint32_t l_pool = 0;          // pool lines per slot
    DevBuf<uint32_t> l_gen;       // [slots] last seed generation of every slot
    size_t l_spread = 0;          // > 0: ONE allocation, a slot every max(l_spread, what it needs) bytes, region B behind its hot block
    DevBuf<unsigned long long> l_stats;
    int64_t line_stats[4] = {0, 0, 0, 0};   // last run: LDS updates, blind line writes, read-modify-writes, updates of pushed nodes
    std::vector<double> placement_probe;    // G updates/s of every candidate allocation of the slot memory, in draw order
    int placement_kept = -1;
    int registered = 0;                     // listed in g_live_lines
    int shared_device = 0;                  // another large line-state context of this process lives on the device
    DevBuf<double> dump_s, dump_r;          // arcte_hip_seed_state: the one seed's dense s and r (LineParams::dump_s)
    int dump_on = 0;
    // per-run
    int64_t run_nseeds = -1;
    DevBuf<int32_t> seeds_d, work_pos, out_cnt, status, nop_d;
    DevBuf<double> eps_d;
    DevBuf<int64_t> out_off, dst_off;
    DevBuf<unsigned long long> counters;   // [0] work counter [1] raw cursor [2..6] stats
    DevBuf<int32_t> raw, rows_final;
    DevBuf<uint64_t> sort_keys, sort_keys_in;
    int64_t eps_big_count = 0;
    DevBuf<int32_t> sort_iota, eps_big_pos;
    DevBuf<char> sort_temp;
    int64_t raw_for_seeds = 0;
    int64_t rows_reserve_extra = 0;   // rows a running call leaves room for behind its own (arcte_hip_run_seeds_append)
    int64_t final_rows = 0;
    std::vector<int64_t> colptr;
    int64_t stats[6] = {0, 0, 0, 0, 0, 0};
    int64_t candidates = 0;
    int64_t split_rows = 0;      // rows a helper wavefront took half of (ARCTE_HIP_COOP)
    double ms[4] = {0, 0, 0, 0};

    GraphDev graph() const
    {
        GraphDev g;
        g.n = n;
        g.indptr = indptr.p;
        g.indices = indices.p;
        g.data = data.p;
        g.out_degree = out_degree.p;
        g.in_degree = in_degree.p;
        g.edge_in_degree = edge_in_degree.p;
        g.data_f = data_f.p;
        g.in_degree_f = in_degree_f.p;
        g.edge_in_degree_f = edge_in_degree_f.p;
        return g;
    }
    size_t device_bytes() const
    {
        return indptr.bytes() + indices.bytes() + data.bytes() + out_degree.bytes() + in_degree.bytes() + edge_in_degree.bytes() + data_f.bytes() + in_degree_f.bytes() + edge_in_degree_f.bytes() + state.bytes() + slot_epoch.bytes() +
               node_hot.bytes() + edge_hot.bytes() + warm.bytes() +
               queue.bytes() + sup.bytes() + seeds_d.bytes() + work_pos.bytes() + out_cnt.bytes() + status.bytes() +
               nop_d.bytes() + eps_d.bytes() + out_off.bytes() + dst_off.bytes() + raw.bytes() + rows_final.bytes() +
               edge_rank.bytes() + node_rank.bytes() + rowspan.bytes() + in_degree_r.bytes() + in_degree_rf.bytes() + slot_bytes_lines();
    }
    size_t slot_bytes_lines() const { return l_block.bytes() + l_blockb.bytes(); }
    size_t slot_bytes_dense() const { return state.bytes() + sup.bytes() + queue.bytes() + hqueue.bytes() + warm.bytes(); }
};

namespace {

int alloc_slots(arcte_hip_ctx *c, int64_t slots, uint32_t qcap)
{
    HIP_TRY(alloc_cached(c->state, (size_t)slots * c->n, c->device));
    HIP_TRY(c->slot_epoch.alloc((size_t)slots));
    HIP_TRY(alloc_cached(c->sup, (size_t)slots * c->n, c->device));
    HIP_TRY(alloc_cached(c->queue, (size_t)slots * qcap, c->device));
    c->hqueue.release();
    if (c->coop) HIP_TRY(c->hqueue.alloc((size_t)slots * qcap));
    // warm table: one entry per rank in [0, warm_k2) per slot (the first hotK of them lie unused under the LDS table:
    // the LDS share depends on the arithmetic type, the allocation does not)
    release_cached(c->warm, c->device);
    c->warm_n = 0;
    if (c->warm_k2 > 0) {
        c->warm_n = c->warm_k2;
        HIP_TRY(alloc_cached(c->warm, (size_t)slots * c->warm_n * 16, c->device));
        HIP_TRY(hipMemsetAsync(c->warm.p, 0, c->warm.bytes(), c->stream));
    }
    HIP_TRY(hipMemsetAsync(c->state.p, 0, c->state.bytes(), c->stream));
    HIP_TRY(hipMemsetAsync(c->slot_epoch.p, 0, c->slot_epoch.bytes(), c->stream));
    c->seeds_since_clear = 0;
    c->state_is_f32 = c->float32;
    c->slots = slots;
    c->qcap = qcap;
    return 0;
}

int grow_queue(arcte_hip_ctx *c)
{
    if (c->qcap >= (1u << 30)) return fail(ARCTE_HIP_ECAPACITY, "FIFO ring cannot grow past 2^30 entries");
    uint32_t nq = c->qcap * 4;
    // keep the footprint bounded: fewer, deeper slots once rings get large
    int64_t slots = c->slots;
    const int wpb = c->waves_per_block;
    while (slots > wpb && (size_t)slots * nq * sizeof(QEntry) > ((size_t)16 << 30)) slots /= 2;
    slots = std::max<int64_t>(wpb, slots - slots % wpb);
    c->queue.release();
    if (slots != c->slots) {
        c->state.release();
        c->slot_epoch.release();
        c->sup.release();
        return alloc_slots(c, slots, nq);
    }
    HIP_TRY(c->queue.alloc((size_t)slots * nq));
    if (c->coop) HIP_TRY(c->hqueue.alloc((size_t)slots * nq));
    c->qcap = nq;
    return 0;
}

constexpr size_t LDS_PER_CU = 160 * 1024;     // gfx950

int env_int(const char *name, int fallback)
{
    if (const char *e = getenv(name)) {
        char *end = nullptr;
        long v = strtol(e, &end, 10);
        if (end != e) return (int)v;
    }
    return fallback;
}

// ARCTE_HIP_STAGE_ROWS=1 (A/B of round 3, no gain): exists in `make AB=1` builds only
int stage_rows_on()
{
#ifdef ARCTE_HIP_AB_BUILDS
    return env_int("ARCTE_HIP_STAGE_ROWS", 0) != 0;
#else
    return 0;
#endif
}

// Values of the LDS-resident hot table per wavefront: the CU's LDS divided by the wavefronts resident on it
// (1 KiB allocation granularity), never more than there are ranked nodes.  ARCTE_HIP_HOT=0 switches the table
// off (A/B), any other number caps it.
uint32_t hot_values_per_wave(const arcte_hip_ctx *c, size_t value_bytes)
{
    const int cap = env_int("ARCTE_HIP_HOT", -1);
    if (cap == 0 || c->hot_ranked <= 0 || c->waves_per_cu <= 0) return 0;
    // A share that fills the LDS to the last byte does not reliably leave room for the intended number of
    // workgroups per CU (launch-shape sweeps: 3 x 52 KiB, 5 x 32 KiB, 6 x 26 KiB and, on some boxes, 4 x 40 KiB ran
    // like one workgroup fewer); 8 KiB are left unclaimed.
    const size_t reserve = (size_t)std::max(0, env_int("ARCTE_HIP_LDS_RESERVE_KB", 8)) * 1024;
    // (with helpers a seed has two wavefronts: half as many tables per CU, and CoopShared behind each)
    const size_t shares = (size_t)std::max(1, c->coop ? c->waves_per_cu / 2 : c->waves_per_cu);
    size_t per_wave = (LDS_PER_CU - std::min(reserve, LDS_PER_CU / 2)) / shares;
    per_wave = per_wave / 1024 * 1024;
    if (c->coop) per_wave -= 256;
    uint64_t k = per_wave / value_bytes;
    k = std::min<uint64_t>(k, (uint64_t)c->hot_ranked);
    if (cap > 0) k = std::min<uint64_t>(k, (uint64_t)cap);
    return (uint32_t)(k - k % 4);
}

// ---- line state (arcte_lines.hpp) -----------------------------------------------------------------------------
// bytes of this device in the process-wide buffer cache (buffers of destroyed contexts): hipMemGetInfo does not count
// them as free, but the next allocation of this process takes them back or frees them (alloc_cached).  Parked losers of
// a placement draw are NOT in this number: they stay allocated, so whoever sizes slot memory must count them as used
// (round-3 advisor: counted as free they let a re-allocation push the device past the fill level where the kernel falls
// off an edge, profiles/r03/device_fill_8m.txt).
size_t cached_bytes_on(int device)
{
    std::lock_guard<std::mutex> lock(g_big_mutex);
    size_t b = 0;
    for (const auto &e : g_big_cache)
        if (e.device == device) b += e.bytes;
    return b;
}

// LDS one wavefront of k_arcte_lines may claim: the CU's share, at most what one workgroup may have
size_t lines_lds_per_wave(const arcte_hip_ctx *c)
{
    const size_t reserve = (size_t)std::max(0, env_int("ARCTE_HIP_LDS_RESERVE_KB", 8)) * 1024;
    const size_t shares = (size_t)std::max(1, c->l_waves_per_cu);
    size_t per_wave = (LDS_PER_CU - std::min(reserve, LDS_PER_CU / 2)) / shares;
    per_wave = per_wave / 1024 * 1024;
    return std::min<size_t>(per_wave, (size_t)64 * 1024);
}

// values of the LDS level: what the touched-line bitmap (M bits) leaves of the wavefront's share
uint32_t lines_hot_values(const arcte_hip_ctx *c)
{
    const int cap = env_int("ARCTE_HIP_HOT", -1);
    if (cap == 0) return 0;
    const size_t per_wave = lines_lds_per_wave(c), bitmap = c->l_M / 8 + (stage_rows_on() ? 2560 : 0);
    if (per_wave <= bitmap) return 0;
    uint64_t k = (per_wave - bitmap) / sizeof(double);
    k = std::min<uint64_t>(k, (uint64_t)c->n);
    if (cap > 0) k = std::min<uint64_t>(k, (uint64_t)cap);
    return (uint32_t)k;
}

// The layout of a slot's memory.  A wavefront's every access is a random one into its own slot, so the cost of address
// translation is decided by how a slot's OFTEN touched bytes sit in the address space -- measured, ms per 81 434 seeds of
// the 1M/50M graph over fresh contexts of one process (profiles/r03/context_lottery_*.txt):
//   five arrays, one per kind (values, ring, candidates, pushed state, bits), slot after slot in each:  77 ... 96
//   [line][slot]: line l of every slot side by side, a slot spread over ALL pages:                      96 ... 100
//   one block per slot of a whole number of 2 MB pages (14 MB):                                         76 / 86 / 98
//   one block per slot whose size is a power of two (16 MB), i.e. every slot aligned to its size:       81.6 (11 of 12)
// Translations are cached for aligned power-of-two ranges; a slot that starts at an odd multiple of 2 MB is served by
// 2 MB ranges, one that starts at a multiple of its size by one range.  So: the often touched parts -- region A's values,
// the ring, the candidate list, the pushed-state array, region B's touched-bits -- form ONE block of power-of-two size
// (8 MB with the default capacities), and region B's values, touched by a few per cent of the updates, another.
// That is the PACKED layout; a context that is large enough and finds the room spreads its slots over one allocation
// instead (l_spread, see setup_lines): the same hot block at the start of every slot's stride, region B behind it.
struct LinesLayout { size_t off_queue, off_sup, off_ps, off_gbm, off_b, block, blockb; };
LinesLayout lines_layout(const arcte_hip_ctx *c, uint32_t qcap, uint32_t pcap, uint32_t scap)
{
    auto up = [](size_t x, size_t a) { return (x + a - 1) / a * a; };
    LinesLayout y;
    size_t o = ((size_t)c->l_M << 3) * sizeof(double);
    y.off_queue = o; o += up((size_t)qcap * sizeof(QEntry), 256);
    y.off_sup = o;   o += up((size_t)scap * sizeof(int32_t), 256);
    y.off_ps = o;    o += up((size_t)pcap * sizeof(double2), 256);
    y.off_gbm = o;   o += up((size_t)(c->l_ind ? 0 : (c->l_MB >> 5)) * sizeof(uint32_t), 256);      // (indirect lines carry no touched-bits: the entry's generation is the claim)
    const bool pow2 = env_int("ARCTE_HIP_SLOT_POW2", 1) != 0;         // 0: whole 2 MB pages only (A/B)
    const bool split = env_int("ARCTE_HIP_SLOT_SPLIT", 1) != 0;       // 1: region B's values in an allocation of their own (A/B)
    // region B per slot: eight float64 per line, or (indirect lines) an 8-byte entry per line + the pool's lines
    const size_t bytes_b = c->l_ind ? (size_t)c->l_MB * sizeof(uint64_t) + (size_t)c->l_pool * 64
                                    : ((size_t)c->l_MB << 3) * sizeof(double);   // (MB is a power of two)
    auto pow2_size = [&](size_t x) { size_t p = 4096; while (p < x) p <<= 1; return p; };
    if (c->l_spread) {
        // slots SPREAD over the device (see setup_lines): the hot block at the start of the slot's stride, region B's values
        // behind it, the rest of the stride unused
        const size_t hot = pow2 ? pow2_size(o) : up(o, (size_t)2 << 20);
        y.off_b = hot;
        y.block = std::max(up(hot + bytes_b, (size_t)2 << 20), c->l_spread);
        y.blockb = 0;
        return y;
    }
    y.off_b = o;
    if (!split) o += bytes_b;
    y.block = pow2 ? pow2_size(o) : up(o, (size_t)2 << 20);
    y.blockb = split ? bytes_b : 0;
    return y;
}

size_t lines_bytes_per_slot(const arcte_hip_ctx *c, uint32_t qcap, uint32_t pcap, uint32_t scap)
{
    const LinesLayout y = lines_layout(c, qcap, pcap, scap);
    return y.block + y.blockb;
}

// The slot memory's placement decides the propagation kernel's speed (two main levels 23 % apart),
// and hipMalloc leaves it to chance.  So a context that is large enough to care draws up to ARCTE_HIP_PLACEMENT_TRIES (8;
// ARCTE_HIP_SPREAD_TRIES = 4 of the much larger spread layout, of which the third and fourth are drawn only while every candidate so far sits on the slow level) candidate allocations -- alive at the same time, so that
// they are different memory --, runs k_probe_slots on each (a few milliseconds of the kernel's own access pattern) and
// keeps the fastest; the others are parked (g_parked).  The probe sees levels of memory (20, 22, 24 and 26 G updates/s
// on the 1M/50M graph's slots; the push kernel runs 93 / 85 / 78 / 72 ms per 81 434 seeds on them): the draw stops as
// soon as it holds a candidate 10 % faster than another one.  Every caller gets this -- arcte(), the console script,
// bench.py alike.  A draw ends when the device has no room for another candidate.
int draw_slot_memory(arcte_hip_ctx *c, size_t slots, size_t block)
{
    const size_t bytes = slots * block;
    c->placement_probe.clear();
    c->placement_kept = -1;
    int tries = std::max(1, std::min(16, env_int("ARCTE_HIP_PLACEMENT_TRIES", 8)));
    // (a draw costs ~0.1 s per candidate: only for graphs whose runs are long enough to repay it)
    if (bytes < ((size_t)std::max(1, env_int("ARCTE_HIP_PLACEMENT_MIN_MB", 1024)) << 20) || slots < 256 ||
        c->n < (int64_t)env_int("ARCTE_HIP_PLACEMENT_MIN_NODES", 262144))
        tries = 1;
    if (c->shared_device) tries = 1;          // another large context of this process lives on the device: see g_live_lines
    if (tries == 1) {
        HIP_TRY(c->l_block.alloc(bytes, c->device));
        return 0;
    }
    if (c->l_spread) tries = std::min(tries, std::max(1, env_int("ARCTE_HIP_SPREAD_TRIES", 4)));      // (candidates of 50 GB and more)
    free_parked_on(c->device, bytes);          // losers of a draw of another shape: of no use to this one
    // A draw may cost this much allocation time before it settles for a candidate off the slow level (on some boxes one hipMalloc
    // of tens of GB in three or four takes 1.5-3.7 s, profiles/r04/draw_by_position.txt).  3 s: a process that runs graphs of this
    // size spends longer reading them, and a level is worth 6 % of every run that follows; 400 ms let one bench run settle for a
    // single candidate at 23.5 (0.352).
    const double alloc_budget_s = std::max(0, env_int("ARCTE_HIP_DRAW_ALLOC_MS", 3000)) * 1e-3;
    const double level_good = 0.1 * std::max(1, env_int("ARCTE_HIP_DRAW_GOOD_X10", 250)), level_ok = 0.1 * std::max(1, env_int("ARCTE_HIP_DRAW_OK_X10", 230));
    double alloc_spent_s = 0.0;
    std::vector<SlotMem> cand((size_t)tries);
    DevBuf<unsigned long long> sink;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int best = -1;
    double known_best = 0.0;
    {
        std::lock_guard<std::mutex> lock(g_big_mutex);
        for (const auto &b : g_best_probe)
            if (b.device == c->device && b.bytes == bytes) known_best = b.rate;
    }
    int rc = [&]() -> int {
        HIP_TRY(sink.alloc(1));
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        const uint32_t span = (uint32_t)(std::min<size_t>(block, ((size_t)c->l_M << 3) * sizeof(double)) / sizeof(double));
        for (int t = 0; t < tries; t++) {
            if (t > 0) {
                size_t free_b = 0, total_b = 0;
                HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                if (free_b + cached_bytes_on(c->device) < bytes + bytes / 2 + ((size_t)24 << 30)) break;      // no room for another candidate
                // (nor one that would leave the device past 65 % full with its loser parked: the edge of profiles/r03/device_fill_8m.txt)
                if (t >= 2 && total_b - std::min(total_b, free_b) + bytes > total_b / 100 * 80) break;
                // (the allocation-time budget ends a draw that holds a candidate off the slow level; a draw that holds only slow
                //  ones goes on -- 750 against 590-630 ms per launch is worth seconds of hipMalloc)
                if (alloc_spent_s > alloc_budget_s && best >= 0 && c->placement_probe[(size_t)best] >= level_ok) break;
            }
            const auto ta = std::chrono::steady_clock::now();
            if (cand[(size_t)t].alloc(bytes, c->device) != hipSuccess) { (void)hipGetLastError(); cand[(size_t)t].release(c->device); break; }
            const double alloc_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
            alloc_spent_s += alloc_s;
            if (env_int("ARCTE_HIP_VERBOSE", 0))
                fprintf(stderr, "[arcte_hip] slot memory candidate %d: %.1f GB allocated in %.3f s at %p (address mod 64 MB: %llu MB)\n", t, bytes / 1e9, alloc_s,
                        (void *)cand[(size_t)t].p, (unsigned long long)(((uintptr_t)cand[(size_t)t].p >> 20) & 63));
            // (on some boxes a large hipMalloc that follows frees takes seconds -- profiles/r03/first_call_1m.txt; the draw goes on
            //  all the same: ending it there left the first call of arcte() as slow as before -- the output buffers' allocations
            //  are as slow -- and the context without its protection against the slow level)
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {          // (the first pass faults the translations in)
                HIP_TRY(hipEventRecord(e0, c->stream));
                hipLaunchKernelGGL(k_probe_slots, dim3((unsigned)slots), dim3(WAVE), 0, c->stream, cand[(size_t)t].p, (int64_t)block, span, 128, sink.p);
                HIP_TRY(hipEventRecord(e1, c->stream));
                HIP_TRY(hipEventSynchronize(e1));
                HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            }
            const double rate = (double)slots * WAVE * 128 / (ms * 1e-3) / 1e9;
            c->placement_probe.push_back(rate);
            if (env_int("ARCTE_HIP_VERBOSE", 0)) fprintf(stderr, "[arcte_hip] slot memory candidate %d probes at %.2f G updates/s\n", t, rate);
            if (best < 0 || rate > c->placement_probe[(size_t)best]) best = t;
            const double slowest = *std::min_element(c->placement_probe.begin(), c->placement_probe.end());
            // (packed slots: two classes, 20 and 24 -- a fast one in hand is enough; spread slots have a top level at 26 that a
            //  20 / 24 pair does not show yet)
            if (!c->l_spread && c->placement_probe[(size_t)best] >= 1.1 * slowest) break;
            // Spread slots: the probe's levels are 20, 22, 24 and 26 G updates/s on every box so far (the push kernel: ~750, 680, 630 and
            // 590 ms per launch of the 1M/50M graph).  A candidate on the top level ends the draw at once (one allocation, nothing
            // parked).  Below it the draw goes on, to ARCTE_HIP_SPREAD_TRIES (4) candidates: WHERE an allocation lands decides its
            // level, and later ones land better -- six processes with four forced candidates each on one box
            // (profiles/r04/draw_by_position.txt): the first candidate on the slow level six times of six, the best of three
            // 25.7 / 23.9 / 25.3 / 24.2 / 24.0 / 24.2, the best of four 25.7 / 23.9 / 25.3 / 25.7 / 25.2 / 25.3.  The price is hipMalloc
            // time on boxes where one allocation in three or four takes seconds: past ARCTE_HIP_DRAW_ALLOC_MS a draw that holds a
            // candidate off the slow level stops (the check above).
            if (c->l_spread && c->placement_probe[(size_t)best] >= level_good) break;
            if (known_best > 0.0 && c->placement_probe[(size_t)best] >= 0.97 * known_best) break;   // as good as this process has seen
        }
        if (best < 0) return fail(ARCTE_HIP_EHIP, "no memory for the propagation slots");
        return 0;
    }();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    sink.release();
    if (rc) {
        for (auto &m : cand) { m.plain.release(); m.p = nullptr; m.size = 0; }
        return rc;
    }
    c->l_block.release(c->device);
    c->l_block = cand[(size_t)best];          // (shallow: the candidate's buffer changes hands)
    cand[(size_t)best].plain.p = nullptr;
    cand[(size_t)best].p = nullptr;
    {
        // Of the losers at most ARCTE_HIP_PARK_MAX (1) stay allocated per device -- the SLOWEST: the allocator cannot hand
        // that memory out again and one deferred hipFree less is paid for by the next hipMalloc -- the others go back to the
        // driver before the context exists (round 3 kept them all: 155 GB held for a 1M-node graph).  The best rate is
        // remembered for the next context of this shape.
        // ... and none when the kept buffer and a parked one together would hold more than 35 % of the device: the 14 / 16
        // wavefronts per CU of round 4 make candidates of 60-69 GB, and two of those beside the output buffers of a whole launch
        // of the 8M-node graph put the device past its fill edge (5 611 against ~3 700 ms per launch, profiles/r04/occupancy_sweeps.txt)
        size_t total_dev = 0, free_dev = 0;
        (void)hipMemGetInfo(&free_dev, &total_dev);
        const int park_max = (total_dev && 2 * bytes > total_dev / 100 * 35) ? 0 : std::max(0, env_int("ARCTE_HIP_PARK_MAX", 1));
        std::vector<std::pair<double, size_t>> losers;          // (probe rate, candidate)
        for (size_t i = 0; i < cand.size(); i++)
            if (cand[i].plain.p) losers.push_back({i < c->placement_probe.size() ? c->placement_probe[i] : 0.0, i});
        std::sort(losers.begin(), losers.end());
        std::lock_guard<std::mutex> lock(g_big_mutex);
        int parked_here = 0;
        for (const auto &e : g_parked) parked_here += e.device == c->device;
        for (const auto &l : losers) {
            SlotMem &m = cand[l.second];
            if (parked_here < park_max) { g_parked.push_back({c->device, (void *)m.plain.p, m.plain.capacity}); parked_here++; }
            else (void)hipFree(m.plain.p);
            m.plain.p = nullptr; m.plain.count = 0; m.plain.capacity = 0; m.p = nullptr; m.size = 0;
        }
        bool found = false;
        for (auto &b : g_best_probe)
            if (b.device == c->device && b.bytes == bytes) { b.rate = std::max(b.rate, c->placement_probe[(size_t)best]); found = true; }
        if (!found) g_best_probe.push_back({c->device, bytes, c->placement_probe[(size_t)best]});
        if (env_int("ARCTE_HIP_VERBOSE", 0))
            fprintf(stderr, "[arcte_hip] slot memory: %d candidates probed, kept %d, %d buffer(s) parked on this device\n", (int)c->placement_probe.size(), best, parked_here);
    }
    c->placement_kept = best;
    return 0;
}

int alloc_lines(arcte_hip_ctx *c, int64_t slots, uint32_t qcap, uint32_t pcap, uint32_t scap)
{
    // (nothing is cleared: a value is only ever read after a bitmap said its line was written by this seed)
    const LinesLayout y = lines_layout(c, qcap, pcap, scap);
    {
        int rd = draw_slot_memory(c, (size_t)slots, y.block);
        if (rd) return rd;
    }
    HIP_TRY(c->l_blockb.alloc((size_t)slots * y.blockb, c->device));
    if (const int poison = env_int("ARCTE_HIP_POISON", -1); poison >= 0) {
        // test hook: the slot memory starts as garbage of the caller's choice -- nothing may depend on what it held
        HIP_TRY(hipMemsetAsync(c->l_block.p, poison, c->l_block.bytes(), c->stream));
        if (c->l_blockb.bytes()) HIP_TRY(hipMemsetAsync(c->l_blockb.p, poison, c->l_blockb.bytes(), c->stream));
    }
    c->l_block_bytes = y.block;
    c->l_blockb_bytes = y.blockb;
    if (c->l_ind && c->l_MB) {
        // indirect lines: an entry is valid when it carries the generation of the seed in hand; generations count from 1
        // per slot and allocation, so the entries start as zeros (the one thing in a slot that is ever cleared: once)
        char *bpart = y.blockb ? c->l_blockb.p : c->l_block.p + y.off_b;
        HIP_TRY(hipMemset2DAsync(bpart, y.blockb ? y.blockb : y.block, 0, (size_t)c->l_MB * sizeof(uint64_t), (size_t)slots, c->stream));
        HIP_TRY(c->l_gen.alloc((size_t)slots));
        HIP_TRY(hipMemsetAsync(c->l_gen.p, 0, c->l_gen.bytes(), c->stream));
    }
    c->l_off_queue = y.off_queue; c->l_off_sup = y.off_sup; c->l_off_ps = y.off_ps; c->l_off_gbm = y.off_gbm; c->l_off_b = y.off_b;
    if (!c->l_stats.p) HIP_TRY(c->l_stats.alloc(4));
    c->l_slots = slots;
    c->l_qcap = qcap;
    c->l_pcap = pcap;
    c->l_scap = scap;
    return 0;
}

// a seed ran out of ring, pushed-state or candidate entries: four times as many of those (never more than a seed can
// need), fewer slots if that is what the memory allows
int grow_lines(arcte_hip_ctx *c, bool queue_over, bool pushed_over, bool sup_over, bool pool_over = false)
{
    uint32_t qcap = c->l_qcap, pcap = c->l_pcap, scap = c->l_scap;
    const uint32_t node_cap = next_pow2((uint64_t)c->n);
    if (queue_over) {
        if (qcap >= (1u << 30)) return fail(ARCTE_HIP_ECAPACITY, "FIFO ring cannot grow past 2^30 entries");
        qcap *= 4;
    }
    if (pushed_over) {
        if (pcap >= node_cap) return fail(ARCTE_HIP_ECAPACITY, "pushed-state array already holds every node");
        pcap = std::min<uint32_t>(node_cap, pcap * 4);
    }
    if (sup_over) {
        if (scap >= node_cap) return fail(ARCTE_HIP_ECAPACITY, "candidate list already holds every node");
        scap = std::min<uint32_t>(node_cap, scap * 4);
    }
    if (pool_over) {
        // Every CLAIM of region B takes a pool line, not every line: a claim that finds its line claimed leaves its candidate
        // unused (3 % of the claims on the graphs region B exists for; nearly all of them on a 2 000-node graph whose region B is
        // forced).  A pool that outgrows the dense lines has no reason to exist: unless indirect lines were asked for
        // (ARCTE_HIP_B_INDIRECT=1: the pool keeps growing, to 2^26 lines), slots whose pool passes TWICE the dense lines get dense
        // lines, which cannot overflow.
        if (c->l_pool >= (1u << 26)) return fail(ARCTE_HIP_ECAPACITY, "region B's pool cannot grow past 2^26 lines");
        // (four times while a pool is small, twice from 8 MB on: every slot pays for what the heaviest seed of the graph needs)
        c->l_pool *= c->l_pool >= (1u << 17) ? 2 : 4;
        if (c->l_pool > 2 * (uint64_t)c->l_MB && env_int("ARCTE_HIP_B_INDIRECT", -1) != 1) { c->l_ind = 0; c->l_gen.release(); }
    }
    // (the slot memory changes its shape: losers of the old shape's draw are of no use any more, and left allocated they
    //  would push the device's fill past what setup_lines budgeted for)
    free_parked_on(c->device);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    free_b += c->slot_bytes_lines() + cached_bytes_on(c->device);
    if (const int mb = env_int("ARCTE_HIP_TEST_GROW_FREE_MB", 0); mb > 0) free_b = std::min<size_t>(free_b, (size_t)mb << 20);      // test hook: a full device
    // (fewer slots by half a wavefront per CU at a time, as setup_lines does; round 4's first version halved them and a whole launch
    //  of the 8M-node graph went from 4 096 slots to 2 048 for 2 % of memory: tools/whole_launch_probe.py)
    int64_t slots = c->l_slots;
    const size_t per_slot = lines_bytes_per_slot(c, qcap, pcap, scap);
    while (slots > c->cus && (size_t)slots * per_slot > free_b / 4 * 3) slots -= std::max(1, c->cus / 2);
    while (slots > 1 && (size_t)slots * per_slot > free_b / 4 * 3) slots = (slots + 1) / 2;
    c->l_waves_per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(32, (slots + c->cus - 1) / c->cus));
    c->l_block.release(c->device);
    c->l_blockb.release(c->device);
    return alloc_lines(c, slots, qcap, pcap, scap);
}

template <int MODE, int VAR>
int launch_lines_v(arcte_hip_ctx *c, const PushParams &P, const LineParams &L, int blocks, size_t lds)
{
    auto go = [&](auto kernel) -> int {
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(WAVE), lds, c->stream, P, L);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    const bool tail = c->l_MB > 0;
    // One 64-edge tile per pipeline step is the default: 149 VGPRs, so three wavefronts per SIMD instead of two (186 with
    // two tiles), and the kernel wants wavefronts in flight more than it wants long steps (1M/50M graph, ms per 81 434
    // seeds: two tiles at 8 per CU 91.1; one tile at 9 / 10 / 11 / 12 per CU 89.5 / 81.4 / 78.3 / 76.4).
    // ARCTE_HIP_TILES=2 / 4 keep the longer steps for ARCTE's worker on narrow rows (A/B).
    // rows: 0 wide, 1 narrow (rank + float32 in_degree), 2 packed (one word per edge; c->edge_rank holds the packed words then)
    const int rows = c->pack ? 2 : (c->narrow ? 1 : 0);
    if (tail && c->l_ind) {         // region B's lines indirect
        if (rows == 2 && MODE == 0 && VAR == 0 && c->l_waves_per_cu > 12) return go(k_arcte_lines<0, 0, 2, true, false, 1, 4, false, true>);
        if (rows == 2) return go(k_arcte_lines<MODE, VAR, 2, true, false, 1, 1, false, true>);
        return rows == 1 ? go(k_arcte_lines<MODE, VAR, 1, true, false, 1, 1, false, true>) : go(k_arcte_lines<MODE, VAR, 0, true, false, 1, 1, false, true>);
    }
#ifdef ARCTE_HIP_AB_BUILDS          // the arms that lost their A/B (profiles/r03): `make AB=1` builds them, the knobs below select them (narrow rows: ARCTE_HIP_PACK=0)
    if (rows == 1 && c->tiles == 4 && MODE == 0 && VAR == 0) return tail ? go(k_arcte_lines<0, 0, 1, true, false, 4>) : go(k_arcte_lines<0, 0, 1, false, false, 4>);
    if (rows == 1 && c->tiles == 2 && MODE == 0 && VAR == 0 && c->l_waves_per_cu > 8)   // two tiles at three wavefronts per SIMD (168 VGPRs)
        return tail ? go(k_arcte_lines<0, 0, 1, true, false, 2, 3>) : go(k_arcte_lines<0, 0, 1, false, false, 2, 3>);
    if (rows == 1 && c->tiles == 2 && MODE == 0 && VAR == 0) return tail ? go(k_arcte_lines<0, 0, 1, true, false, 2>) : go(k_arcte_lines<0, 0, 1, false, false, 2>);
    if (rows == 1 && MODE == 0 && VAR == 0 && stage_rows_on())       // A/B: row data of the steps in flight staged through LDS
        return tail ? go(k_arcte_lines<0, 0, 1, true, false, 1, 1, true>) : go(k_arcte_lines<0, 0, 1, false, false, 1, 1, true>);
    if (rows == 1 && MODE == 0 && VAR == 0 && c->l_waves_per_cu > 12)      // four wavefronts per SIMD: the compiler spills to fit 128 VGPRs
        return tail ? go(k_arcte_lines<0, 0, 1, true, false, 1, 4>) : go(k_arcte_lines<0, 0, 1, false, false, 1, 4>);
#endif
    // more than twelve wavefronts per CU (ARCTE_HIP_WAVES_PER_CU; ARCTE's worker on packed rows): the build that fits four
    // wavefronts per SIMD (128 VGPRs, the compiler spills the rest).  An experiment knob: large graphs, whose rows are short, are
    // sensitive to the wavefronts in flight (8M-node graph: 0.184 / 0.222 at 9 / 12 per CU, profiles/r04/large_graph_sweep.txt)
    if (rows == 2 && MODE == 0 && VAR == 0 && c->l_waves_per_cu > 12) {
        if (tail && c->l_ind) return go(k_arcte_lines<0, 0, 2, true, false, 1, 4, false, true>);
        return tail ? go(k_arcte_lines<0, 0, 2, true, false, 1, 4>) : go(k_arcte_lines<0, 0, 2, false, false, 1, 4>);
    }
    if (rows == 2) return tail ? go(k_arcte_lines<MODE, VAR, 2, true, false, 1>) : go(k_arcte_lines<MODE, VAR, 2, false, false, 1>);
    if (rows == 1) return tail ? go(k_arcte_lines<MODE, VAR, 1, true, false, 1>) : go(k_arcte_lines<MODE, VAR, 1, false, false, 1>);
    return tail ? go(k_arcte_lines<MODE, VAR, 0, true, false, 1>) : go(k_arcte_lines<MODE, VAR, 0, false, false, 1>);
}

// mode 0: arcte_worker's loop (any push flavour); mode 2: arcte_and_centrality's (ARCTE's own push)
int launch_lines(arcte_hip_ctx *c, PushParams P, int64_t nwork, int variant, int mode)
{
    LineParams L;
    L.edge_rank = c->edge_rank.p;
    L.in_degree_rf = c->in_degree_rf.p;
    L.rank_bits = c->rank_bits;
    L.node_rank = c->node_rank.p;
    L.ranked_ids = c->ranked_ids.p;
    L.rowspan = c->rowspan.p;
    L.in_degree_r = c->in_degree_r.p;
    // a slot's parts (the kernel addresses region A's and region B's values through ONE index: line * 8 + place, region
    // B's lines continuing region A's)
    char *block = c->l_block.p;
    L.vals = reinterpret_cast<double *>(block);
    {
        // region B's part of a slot: in an allocation of its own (packed layout) or behind the hot block; with indirect
        // lines the entries come first and the pool's lines are what the kernel addresses as region B's values
        char *bpart = c->l_blockb_bytes ? c->l_blockb.p : block + c->l_off_b;
        const size_t bstride = c->l_blockb_bytes ? c->l_blockb_bytes : c->l_block_bytes;
        const size_t entries = c->l_ind ? (size_t)c->l_MB * sizeof(uint64_t) : 0;
        L.bidx = reinterpret_cast<uint64_t *>(bpart);
        L.bidx_stride = (int64_t)(bstride / sizeof(uint64_t));
        L.bgen = c->l_gen.p;
        L.pool_cap = c->l_pool;
        L.vals_b = reinterpret_cast<double *>(bpart + entries) - ((int64_t)c->l_M << 3);      // indexed from region A's first line
        L.valsb_stride = (int64_t)(bstride / sizeof(double));
    }
    L.ps = reinterpret_cast<double2 *>(block + c->l_off_ps);
    L.sup = reinterpret_cast<int32_t *>(block + c->l_off_sup);
    L.gbm = reinterpret_cast<uint32_t *>(block + c->l_off_gbm);
    L.M = c->l_M;
    L.Mshift = c->l_Mshift;
    L.MB = c->l_MB;
    L.MBshift = c->l_MBshift;
    L.vals_stride = (int64_t)(c->l_block_bytes / sizeof(double));
    L.ps_stride = (int64_t)(c->l_block_bytes / sizeof(double2));
    L.sup_stride = (int64_t)(c->l_block_bytes / sizeof(int32_t));
    L.q_stride = (int64_t)(c->l_block_bytes / sizeof(QEntry));
    L.gbm_stride = (int64_t)(c->l_block_bytes / sizeof(uint32_t));
    L.pcap = c->l_pcap;
    L.scap = c->l_scap;
    L.K = lines_hot_values(c);
    L.lstats = c->l_stats.p;
    L.dump_s = c->dump_on ? c->dump_s.p : nullptr;
    L.dump_r = c->dump_on ? c->dump_r.p : nullptr;
    P.queue = reinterpret_cast<QEntry *>(c->l_block.p + c->l_off_queue);
    P.qcap = c->l_qcap;
    const size_t lds = (size_t)L.K * sizeof(double) + c->l_M / 8 + (stage_rows_on() ? 2560 : 0);
    int blocks = (int)std::min<int64_t>(c->l_slots, std::max<int64_t>(nwork, 1));
    // (only ARCTE's worker on packed rows has the build that fits more than twelve wavefronts per CU; the others run on the first
    //  twelve slots per CU of a context that has more)
    if (c->l_waves_per_cu > 12 && !(c->pack && mode == 0 && variant == 0)) blocks = std::min<int>(blocks, 12 * c->cus);
    if (mode == 2) return launch_lines_v<2, 0>(c, P, L, blocks, lds);
#ifdef ARCTE_HIP_AB_BUILDS          // ARCTE_HIP_PROFILE=1: the instrumented instantiation (tools/phase_profile.py)
    if (c->prof.p && c->narrow && variant == 0 && !c->l_ind) {
        auto kernel = c->pack ? (c->l_MB > 0 ? k_arcte_lines<0, 0, 2, true, true, 1> : k_arcte_lines<0, 0, 2, false, true, 1>)
                              : (c->l_MB > 0 ? k_arcte_lines<0, 0, 1, true, true, 1> : k_arcte_lines<0, 0, 1, false, true, 1>);
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(WAVE), lds, c->stream, P, L);
        HIP_TRY(hipGetLastError());
        return 0;
    }
#endif
    if (variant == 1) return launch_lines_v<0, 1>(c, P, L, blocks, lds);
    if (variant == 2) return launch_lines_v<0, 2>(c, P, L, blocks, lds);
    return launch_lines_v<0, 0>(c, P, L, blocks, lds);
}

template <typename K>
int launch_with_lds(K kernel, int blocks, int threads, size_t lds, hipStream_t stream, const PushParams &P)
{
    if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, stream, P);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int MODE, int VAR, typename T>
int launch_seeds_v(arcte_hip_ctx *c, PushParams P, int64_t nwork)
{
    P.edge_hot = c->edge_hot.p;
    P.node_hot = c->node_hot.p;
    P.hotK = 0;
    P.warm = (void *)c->warm.p;
    P.warmK2 = c->warm_k2;
    P.warmN = (uint32_t)c->warm_n;
    if constexpr (MODE == 1) {
        // works on the dense vectors the host placed in slot 0: exactly one wavefront may run, all state in HBM
        return launch_with_lds(k_arcte_seeds<MODE, VAR, T, 2, false>, 1, WAVE, 0, c->stream, P);
    } else {
    const int wpb = c->waves_per_block;
    const int64_t waves = std::min<int64_t>(c->slots, std::max<int64_t>(nwork, 1));
    const int blocks = (int)((waves + wpb - 1) / wpb);
    P.hotK = hot_values_per_wave(c, sizeof(T));
    const size_t lds = (size_t)wpb * P.hotK * sizeof(T);
    if (P.hotK == 0) return launch_with_lds(k_arcte_seeds<MODE, VAR, T, 2, false>, blocks, wpb * WAVE, lds, c->stream, P);
#ifdef ARCTE_HIP_AB_BUILDS
    if constexpr (std::is_same<T, double>::value && MODE == 0 && VAR == 0) {
        if (c->prof.p && c->narrow && !c->coop && c->tiles == 2)
            return launch_with_lds(k_arcte_seeds<0, 0, double, 2, true, true, false, true>, blocks, wpb * WAVE, lds, c->stream, P);
        if (c->coop) {
            // one workgroup of two wavefronts per seed: leader + helper
            const int seeds_in_flight = (int)std::min<int64_t>(c->slots, std::max<int64_t>(nwork, 1));
            const size_t lds2 = (size_t)P.hotK * sizeof(T) + sizeof(CoopShared);
            if (c->narrow) return launch_with_lds(k_arcte_seeds<0, 0, double, 2, true, true, true>, seeds_in_flight, 2 * WAVE, lds2, c->stream, P);
            return launch_with_lds(k_arcte_seeds<0, 0, double, 2, true, false, true>, seeds_in_flight, 2 * WAVE, lds2, c->stream, P);
        }
    }
    if constexpr (std::is_same<T, double>::value)
        if (c->narrow && c->tiles == 4) return launch_with_lds(k_arcte_seeds<MODE, VAR, T, 4, true, true>, blocks, wpb * WAVE, lds, c->stream, P);
    if (c->tiles == 4) return launch_with_lds(k_arcte_seeds<MODE, VAR, T, 4, true>, blocks, wpb * WAVE, lds, c->stream, P);
#endif
    if constexpr (std::is_same<T, double>::value)
        if (c->narrow) return launch_with_lds(k_arcte_seeds<MODE, VAR, T, 2, true, true>, blocks, wpb * WAVE, lds, c->stream, P);
    return launch_with_lds(k_arcte_seeds<MODE, VAR, T, 2, true>, blocks, wpb * WAVE, lds, c->stream, P);
    }
}

// MODE 2 exists for ARCTE's own push in float64 only (arcte.pyx has no other flavour)
int launch_centrality(arcte_hip_ctx *c, PushParams P, int64_t nwork)
{
    P.edge_hot = c->edge_hot.p;
    P.node_hot = c->node_hot.p;
    P.warm = (void *)c->warm.p;
    P.warmK2 = c->warm_k2;
    P.warmN = (uint32_t)c->warm_n;
    const int wpb = c->waves_per_block;
    const int64_t waves = std::min<int64_t>(c->slots, std::max<int64_t>(nwork, 1));
    const int blocks = (int)((waves + wpb - 1) / wpb);
    P.hotK = hot_values_per_wave(c, sizeof(double));
    const size_t lds = (size_t)wpb * P.hotK * sizeof(double);
    if (P.hotK == 0) return launch_with_lds(k_arcte_seeds<2, 0, double, 2, false>, blocks, wpb * WAVE, lds, c->stream, P);
    if (c->narrow) return launch_with_lds(k_arcte_seeds<2, 0, double, 2, true, true>, blocks, wpb * WAVE, lds, c->stream, P);
    return launch_with_lds(k_arcte_seeds<2, 0, double, 2, true>, blocks, wpb * WAVE, lds, c->stream, P);
}

template <int MODE, typename T>
int launch_seeds_t(arcte_hip_ctx *c, const PushParams &P, int64_t nwork, int variant)
{
    if (variant == 1) return launch_seeds_v<MODE, 1, T>(c, P, nwork);
    if (variant == 2) return launch_seeds_v<MODE, 2, T>(c, P, nwork);
    return launch_seeds_v<MODE, 0, T>(c, P, nwork);
}

// The state buffer holds either 32-byte float64 entries or 16-byte float32 entries; stale bytes of the other
// layout could alias a live epoch, so a switch of arithmetic clears it.
int prepare_precision(arcte_hip_ctx *c)
{
    if (c->float32 && !c->data_f.p) {
        HIP_TRY(c->data_f.alloc(c->nnz));
        if (!c->edge_in_degree_f.p) HIP_TRY(c->edge_in_degree_f.alloc(c->nnz));
        HIP_TRY(c->in_degree_f.alloc(c->n));
        const int tb = 256;
        if (c->nnz) {
            hipLaunchKernelGGL(k_to_float, dim3((unsigned)((c->nnz + tb - 1) / tb)), dim3(tb), 0, c->stream, c->data.p, c->data_f.p, c->nnz);
            hipLaunchKernelGGL(k_to_float, dim3((unsigned)((c->nnz + tb - 1) / tb)), dim3(tb), 0, c->stream, c->edge_in_degree.p,
                               c->edge_in_degree_f.p, c->nnz);
        }
        hipLaunchKernelGGL(k_to_float, dim3((unsigned)((c->n + tb - 1) / tb)), dim3(tb), 0, c->stream, c->in_degree.p, c->in_degree_f.p, c->n);
        HIP_TRY(hipGetLastError());
    }
    if (c->state_is_f32 != c->float32 && c->state.p) {
        if (c->warm.p) HIP_TRY(hipMemsetAsync(c->warm.p, 0, c->warm.bytes(), c->stream));
        HIP_TRY(hipMemsetAsync(c->state.p, 0, c->state.bytes(), c->stream));
        HIP_TRY(hipMemsetAsync(c->slot_epoch.p, 0, c->slot_epoch.bytes(), c->stream));
        c->seeds_since_clear = 0;
        c->state_is_f32 = c->float32;
    }
    return 0;
}

template <int MODE>
int launch_seeds(arcte_hip_ctx *c, const PushParams &P, int64_t nwork, int variant)
{
    if (c->float32) return launch_seeds_t<MODE, float>(c, P, nwork, variant);
    return launch_seeds_t<MODE, double>(c, P, nwork, variant);
}


// ---------------------------------------------------------------------------------------------
// context construction: begin (device, stream, graph buffers) -> graph arrays on the device -> finish
// ---------------------------------------------------------------------------------------------
int ctx_begin(int device, int64_t n, int64_t nnz, arcte_hip_ctx **out)
{
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(ARCTE_HIP_EHIP, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    arcte_hip_ctx *c = new arcte_hip_ctx();
    c->device = device;
    c->n = n;
    c->nnz = nnz;
    int rc = [&]() -> int {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        c->cus = prop.multiProcessorCount;
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(c->indptr.alloc(n + 1));
        HIP_TRY(c->indices.alloc(nnz));
        HIP_TRY(c->data.alloc(nnz));
        HIP_TRY(c->out_degree.alloc(n));
        HIP_TRY(c->in_degree.alloc(n));
        HIP_TRY(c->counters.alloc(16));
        return 0;
    }();
    if (rc) {
        std::string keep = g_err;
        arcte_hip_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return 0;
}

// Pattern in-count of every node and the nodes in descending-count order (stable: ties keep the node order).
// Serves the seed list of arcte() (arcte.py:610-617) and the ranks of the hot table.
int rank_nodes(arcte_hip_ctx *c, DevBuf<uint32_t> &count)
{
    const int64_t n = c->n, nnz = c->nnz;
    DevBuf<uint32_t> keys_in, keys_out;
    DevBuf<int32_t> ids_in;
    DevBuf<char> temp;
    int rk = [&]() -> int {
        HIP_TRY(count.alloc(n));
        HIP_TRY(keys_in.alloc(n));
        HIP_TRY(keys_out.alloc(n));
        HIP_TRY(ids_in.alloc(n));
        HIP_TRY(c->ranked_ids.alloc(n));
        HIP_TRY(hipMemsetAsync(count.p, 0, count.bytes(), c->stream));
        HIP_TRY(hipMemsetAsync(c->counters.p, 0, 8 * sizeof(unsigned long long), c->stream));
        const int tb = 256;
        if (nnz) hipLaunchKernelGGL(k_column_counts, dim3((unsigned)((nnz + tb - 1) / tb)), dim3(tb), 0, c->stream, c->indices.p, nnz, count.p);
        hipLaunchKernelGGL(k_rank_keys, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, count.p, n, keys_in.p, ids_in.p);
        hipLaunchKernelGGL(k_count_seeds, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, count.p, n, c->counters.p);
        HIP_TRY(hipGetLastError());
        size_t temp_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in.p, keys_out.p, ids_in.p, c->ranked_ids.p, (int)n, 0, 32, c->stream));
        HIP_TRY(temp.alloc(temp_bytes));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys_in.p, keys_out.p, ids_in.p, c->ranked_ids.p, (int)n, 0, 32, c->stream));
        unsigned long long ns = 0;
        HIP_TRY(hipMemcpyAsync(&ns, c->counters.p, sizeof(ns), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->nseeds_all = (int64_t)ns;
        return 0;
    }();
    keys_in.release(); keys_out.release(); ids_in.release(); temp.release();
    return rk;
}

// The caller's CSR arrays as they arrived on the device (arcte_hip_create: the arcte_worker seam): every column index
// inside [0, n), no column stored twice in a row -- the lanes of a push own one target each (push.py:62-64 on a CSR
// row), a repeated column would make two of them race; scipy's sum_duplicates() (or get_natural_random_walk_matrix)
// removes them.  The row ORDER is the caller's and stays: it is the FIFO's enqueue order (similarity.py:194-196).
int validate_rows_on_device(arcte_hip_ctx *c)
{
    const int64_t n = c->n, nnz = c->nnz;
    DevBuf<int32_t> flags;
    DevBuf<uint64_t> keys_in, keys_out;
    DevBuf<char> temp;
    int rc = [&]() -> int {
        int32_t fl[2] = {0, 0};
        HIP_TRY(flags.alloc(2));
        HIP_TRY(hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), c->stream));
        const int row_blocks = (int)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
        hipLaunchKernelGGL(k_check_rows, dim3(row_blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->indices.p, n, flags.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(fl, flags.p, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (fl[0]) return fail(ARCTE_HIP_EINVAL, "column index out of range");
        if (!fl[1] || nnz < 2) return 0;               // every row strictly ascending: no duplicates
        // some row is not ascending: unsorted (fine) or a repeated column -- sort a copy of the (row, column) keys
        HIP_TRY(keys_in.alloc(nnz));
        HIP_TRY(keys_out.alloc(nnz));
        hipLaunchKernelGGL(k_csr_keys, dim3(row_blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->indices.p, n, keys_in.p);
        size_t tb = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, keys_in.p, keys_out.p, (size_t)nnz, 0, 64, c->stream));
        HIP_TRY(temp.alloc(tb));
        HIP_TRY(hipcub::DeviceRadixSort::SortKeys(temp.p, tb, keys_in.p, keys_out.p, (size_t)nnz, 0, 64, c->stream));
        HIP_TRY(hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), c->stream));
        hipLaunchKernelGGL(k_adjacent_equal, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, c->stream, keys_out.p, nnz, flags.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(fl, flags.p, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (fl[0]) return fail(ARCTE_HIP_EINVAL, "a row stores the same column twice (call sum_duplicates() first)");
        return 0;
    }();
    flags.release(); keys_in.release(); keys_out.release(); temp.release();
    return rc;
}

// Slots of the dense-state kernel.  Its launch shape: the kernel is bound by the chip's random-access rate into the
// per-slot HBM state, and the LDS-resident hot table takes 20-40 % of those accesses away -- the more the fewer
// wavefronts share a CU's LDS, while fewer wavefronts keep fewer accesses in flight (interleaved A/B on the 1M/50M graph,
// profiles/r02/ab_interleaved_*.txt, ms per 81 434 seeds: 4 per CU 104.8, 5: 95.0, 6: 93.1, 7: 93.2, 8: 92.0, 10: 97.7,
// 12: 94.9 -- flat from 6 to 8; 6 needs the least memory of those).
int setup_dense(arcte_hip_ctx *c, int64_t n_slots, int64_t queue_capacity)
{
    const int64_t n = c->n;
    const int wpb = c->waves_per_block;
    int64_t slots = n_slots;
    if (slots <= 0) {
        c->waves_per_cu = std::max(1, std::min(env_int("ARCTE_HIP_WAVES_PER_CU", c->coop ? 8 : 6), 32));
        slots = (int64_t)c->waves_per_cu * c->cus;
        if (c->coop) slots = (int64_t)std::max(1, c->waves_per_cu / 2) * c->cus;       // a slot = a seed = two wavefronts
        // keep the slot scratch within a fixed share of the device (buffers parked in the cache are ours to reuse)
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        free_b += cached_bytes_on(c->device);
        size_t per_slot = (size_t)n * (sizeof(Entry) + sizeof(int32_t)) + (size_t)default_queue_capacity(n) * sizeof(QEntry) * (c->coop ? 2 : 1) + (size_t)c->warm_k2 * 16;
        while (slots > c->cus && (size_t)slots * per_slot > free_b / 4 * 3) slots -= c->cus / 2;
    }
    slots = std::max<int64_t>(wpb, (slots + wpb - 1) / wpb * wpb);
    c->waves_per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(32, (slots + c->cus - 1) / c->cus)) * (c->coop ? 2 : 1);
    uint32_t qcap = queue_capacity > 0 ? next_pow2((uint64_t)queue_capacity) : default_queue_capacity(n);
    if (qcap < (uint32_t)WAVE) qcap = WAVE;
    return alloc_slots(c, slots, qcap);
}

// The dense state is made when something asks for it: a similarity slice (one slot is enough), a float32 run, a
// context created with ARCTE_HIP_STATE=dense / ARCTE_HIP_COOP (the caller's slot count).
int ensure_dense(arcte_hip_ctx *c, bool full)
{
    if (c->slots > 0 && (!full || c->dense_auto)) return 0;
    if (c->slots > 0) {
        release_cached(c->state, c->device); release_cached(c->sup, c->device); release_cached(c->queue, c->device);
    }
    int r = full ? setup_dense(c, c->want_slots, c->want_queue) : setup_dense(c, c->waves_per_block, c->want_queue);
    if (r) return r;
    c->dense_auto = full ? 1 : 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// Slots of the line-state kernel: M lines of eight float64 per slot, a ring, the pushed-state array, the candidate list.
int setup_lines(arcte_hip_ctx *c, uint32_t M)
{
    const int64_t n = c->n;
    c->l_M = M;
    c->l_Mshift = 0;
    while ((1u << c->l_Mshift) < M) c->l_Mshift++;
    // region B: the ranks the LDS bitmap does not reach (>= 8 M), their touched-bits in global memory
    c->l_MB = 0;
    c->l_MBshift = 0;
    if ((int64_t)M * 8 < n) {
        c->l_MB = std::max<uint32_t>(128u, next_pow2((uint64_t)((n - (int64_t)M * 8 + 7) / 8)));
        while ((1u << c->l_MBshift) < c->l_MB) c->l_MBshift++;
    }
    // (a seed of the 1M/50M graph enqueues 210 nodes; the ring grows by four and the seed is re-run when it overflows)
    uint32_t qcap = c->want_queue > 0 ? next_pow2((uint64_t)c->want_queue) : std::min<uint32_t>(default_queue_capacity(n), 1u << 15);
    if (qcap < (uint32_t)WAVE) qcap = WAVE;
    const uint32_t node_cap = next_pow2((uint64_t)n);
    // a seed of the 1M/50M graph pushes 190 distinct nodes (p99 600, the heaviest a few thousand: tools/line_study.py) and
    // lists 1 300 candidates (the heaviest over 65 536); what does not fit is re-run with four times the room (grow_lines)
    const uint32_t pcap = std::min<uint32_t>(node_cap, (uint32_t)std::max(64, env_int("ARCTE_HIP_PUSHED", 4096)));
    const uint32_t scap = std::min<uint32_t>(node_cap, (uint32_t)std::max(64, env_int("ARCTE_HIP_CANDIDATES", 262144)));
    // region B dense (8 bytes per node and slot) or indirect (8 bytes per LINE + a pool of lines: arcte_lines.hpp, IND)?
    // ARCTE_HIP_B_INDIRECT = 1 / 0 decides; by default (below) indirect only when the dense region would cost wavefronts
    {
        // Round 4, with the claim that hands out the pool line (ms per launch, dense against indirect, one box each: n = 4M 2 314 /
        // 2 339 with 129 / 52 GB of slot memory; n = 8M 4 574 / 4 256 with 193 / 58 GB; n = 16M: dense does not fit twelve wavefronts
        // per CU): indirect from 16 MB of dense region B per slot on (n > ~2.5 M), dense below (n = 1M: 4 MB per slot).
        const int want = env_int("ARCTE_HIP_B_INDIRECT", -1);
        const size_t dense_b = ((size_t)c->l_MB << 3) * sizeof(double);
        c->l_ind = c->l_MB > 0 && (want == 1 || (want < 0 && dense_b >= ((size_t)std::max(1, env_int("ARCTE_HIP_B_INDIRECT_MIN_MB", 16)) << 20)));
        c->l_pool = std::min<uint32_t>(c->l_MB, (uint32_t)std::max(64, env_int("ARCTE_HIP_B_POOL", 32768)));
    }
    int64_t slots = c->want_slots;
    const bool auto_slots = slots <= 0;
    if (auto_slots) {
        // Wavefronts per CU.  Twelve is what the 156-VGPR kernel admits (three per SIMD).  ARCTE's worker on packed rows also exists
        // as a 128-VGPR build (four per SIMD; the compiler spills 76-144 bytes per lane), and on graphs whose rows are short the
        // wavefronts in flight are worth more than the registers (profiles/r04/occupancy_sweeps.txt, frac of 8 TB/s):
        //   1M/50M   12: 0.365-0.382   14: 0.375-0.386 (-2 % time at equal memory level)   16: 0.373 / 0.357 (8 / 4 KB of touched-bits)
        //   4M/100M  12: 0.285         14: 0.304                                           16: 0.317 / 0.301-0.305
        //   8M/100M  12: 0.226         14: 0.245                                           16: 0.245 / 0.257-0.260
        //   16M/200M 12: 0.227         14: 0.242                                           16: - / 0.243
        // (8M and 16M: launches of a quarter / a sixteenth of the seeds.)  A WHOLE launch of the 8M graph makes the pools of region B
        // grow to what its heaviest seeds claim (262 144 lines = 16 MB per slot; 56 502 of the 3 142 632 seeds outgrow the first
        // 32 768), and keeps the gain as long as grow_lines keeps the slots: 3 730-3 809 ms per launch (0.259-0.253) with sixteen per
        // CU against 4 230-4 530 (0.228-0.213) with twelve, one context each (tools/whole_launch_probe.py).  (The first sweeps of the
        // 16M graph said 0.155-0.172 for 14 / 16: grow_lines HALVED the slots then, once a pool had grown.)
        // So: fourteen while region B is dense (n < ~2.5 M), sixteen with indirect region B -- fewer when the memory says so, below --
        // and twelve for every graph without packed rows: the other instantiations need their 140-190 VGPRs.
        int waves_default = 12;
        if (c->pack) waves_default = !c->l_ind ? 14 : 16;
        c->l_waves_per_cu = std::max(1, std::min(env_int("ARCTE_HIP_WAVES_PER_CU", waves_default), 32));
        slots = (int64_t)c->l_waves_per_cu * c->cus;
    }
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    free_b += cached_bytes_on(c->device);
    // What the slot memory may take: with it the device stays under 65 % full.  (8M-node graph, 3 072 slots of 71 MB: the
    // kernel takes 510 ms per 392 829 seeds with 225 GB of the 309 GB in use when the slots are allocated and 790-800 ms
    // with 227 GB or more, whoever holds the other bytes; slots that are re-allocated after the output buffers exist -- a
    // seed outgrew its pushed-state array -- fall off the same edge at 219 GB: profiles/r03/device_fill_8m.txt.)
    const size_t used_b = total_b > free_b ? total_b - free_b : 0;
    const size_t budget = std::min<size_t>(free_b / 20 * 17, total_b / 100 * 65 > used_b ? total_b / 100 * 65 - used_b : 0);
    // Slots SPREAD over the device.  The kernel's speed follows how far apart the slots' often touched bytes lie in the
    // device's memory (tools/mem_class_map.hip, G random updates/s of 3 072 wavefronts, each inside 2 MB of its slot):
    // slots packed into 12 GB 20.3 or 24.3 by allocation (the lottery of DESIGN.md section 5), the same at the start of a
    // 200 GB allocation 20.3 every time, one slot every 16 MB of 48 GB 24.2, every 32 MB of 96 GB 26.7 and every 64 MB of
    // 200 GB 25.9 on one box; the push kernel on the 1M/50M graph: 72.3 ms per 81 434 seeds with a slot every 16 MB (probe
    // 26.2) against 75.5-78.5 packed on its fast class (24).  So a context that is large enough to care takes ONE
    // allocation with a slot every ARCTE_HIP_SLOT_SPREAD_MB (16) and leaves the bytes between the slots unused, when the
    // device has the room; the packed layout is what remains otherwise (several contexts on one GPU).  The levels exist
    // for spread slots too (another box: 23.8 / 23.9 / 20.0 in three processes), so the placement draw stays, over fewer
    // candidates.
    // what this context is about to ask for, and whether it shares the device (decided once, before any allocation)
    if (!c->registered) {
        c->shared_device = plan_slot_memory(c->device, c, (size_t)slots * lines_bytes_per_slot(c, qcap, pcap, scap)) ? 1 : 0;
        c->registered = 1;
    }
    auto decide_spread = [&]() {
        c->l_spread = 0;
        const int spread_mb = env_int("ARCTE_HIP_SLOT_SPREAD_MB", 16);
        if (spread_mb <= 0 || slots < 256 || n < (int64_t)env_int("ARCTE_HIP_PLACEMENT_MIN_NODES", 262144)) return;
        if (c->shared_device) return;          // packed slots
        c->l_spread = 1;                                          // (1: behind one another without padding)
        const size_t needed = lines_bytes_per_slot(c, qcap, pcap, scap);
        size_t stride = (size_t)spread_mb << 20;
        while (stride > needed && (size_t)slots * stride > std::min<size_t>(budget, total_b / 100 * 40)) stride >>= 1;
        c->l_spread = stride > needed ? stride : ((size_t)slots * needed >= ((size_t)64 << 30) ? needed : 0);
    };
    decide_spread();
    if (auto_slots) {
        const int64_t asked = slots;
        while (slots > c->cus && (size_t)slots * lines_bytes_per_slot(c, qcap, pcap, scap) > budget) slots -= c->cus / 2;
        // Indirect lines cost region B's updates one more request each and save 7/8 of its memory: 8M-node graph, ms per
        // 392 829 seeds: dense with the 2 560 slots that fit 536, indirect with all 3 072 slots 560-593; 4M: 269 against 303.
        // So dense as long as it leaves ten wavefronts per CU, indirect beyond (n > ~10 M on 288 GB).
        if (c->l_MB > 0 && !c->l_ind && env_int("ARCTE_HIP_B_INDIRECT", -1) < 0 && slots < std::min<int64_t>(asked, 10 * (int64_t)c->cus)) {
            c->l_ind = 1;
            slots = asked;
            decide_spread();
            while (slots > c->cus && (size_t)slots * lines_bytes_per_slot(c, qcap, pcap, scap) > budget) slots -= c->cus / 2;
        }
    }
    slots = std::max<int64_t>(1, slots);
    c->l_waves_per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(32, (slots + c->cus - 1) / c->cus));
    int r = (c->l_spread && env_int("ARCTE_HIP_TEST_SPREAD_FAILS", 0)) ? fail(ARCTE_HIP_EHIP, "test hook: no room for spread slots")
                                                                           : alloc_lines(c, slots, qcap, pcap, scap);
    if (r && c->l_spread) {
        // the room that was there a moment ago has gone (another context of this process, another process): packed slots
        const size_t needed_packed = [&] { c->l_spread = 0; return lines_bytes_per_slot(c, qcap, pcap, scap); }();
        if (auto_slots)
            while (slots > c->cus && (size_t)slots * needed_packed > budget) slots -= c->cus / 2;
        c->l_waves_per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(32, (slots + c->cus - 1) / c->cus));
        c->l_block.release(c->device);
        c->l_blockb.release(c->device);
        r = alloc_lines(c, slots, qcap, pcap, scap);
    }
    return r;
}

// Everything after the transition matrix and the degree vectors are on the device (indptr, indices, data,
// out_degree, in_degree): per-edge in_degree, node ranking, hot-table ranks, launch shape, slots.
int ctx_finish(arcte_hip_ctx *c, int64_t n_slots, int64_t queue_capacity)
{
    const int64_t n = c->n, nnz = c->nnz;
    if (c->row_len.size() != (size_t)n) {
        std::vector<int64_t> ip((size_t)n + 1);
        HIP_TRY(hipMemcpyAsync(ip.data(), c->indptr.p, (n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->row_len.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) c->row_len[i] = (int32_t)std::min<int64_t>(ip[i + 1] - ip[i], INT32_MAX);
    }
    HIP_TRY(c->edge_in_degree.alloc(nnz));
    if (nnz) {
        hipLaunchKernelGGL(k_edge_in_degree, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, c->stream, c->indices.p,
                           c->in_degree.p, c->edge_in_degree.p, nnz);
        HIP_TRY(hipGetLastError());
    }
    // ---- narrow rows?  (unweighted graphs: one weight per row; integer degrees: exact in float32)
    c->narrow = 0;
    if (nnz && env_int("ARCTE_HIP_NARROW", 1)) {
        DevBuf<int32_t> flags;
        int32_t fl[2] = {1, 1};
        int rn = [&]() -> int {
            HIP_TRY(flags.alloc(2));
            HIP_TRY(hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), c->stream));
            hipLaunchKernelGGL(k_check_narrow, dim3((unsigned)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)), dim3(BLOCK), 0, c->stream, c->indptr.p,
                               c->data.p, c->in_degree.p, n, flags.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(fl, flags.p, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            return 0;
        }();
        flags.release();
        if (rn) return rn;
        if (!fl[0] && !fl[1]) {
            HIP_TRY(c->edge_in_degree_f.alloc(nnz));
            hipLaunchKernelGGL(k_to_float, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, c->stream, c->edge_in_degree.p, c->edge_in_degree_f.p, nnz);
            HIP_TRY(hipGetLastError());
            c->narrow = 1;
        }
    }
    // ---- hot table: the ranks stream with the rows.  All on the device; the host never sees the ranking.
    {
        const int64_t ranked = std::min<int64_t>(n, (int64_t)HOT_NONE);
        HIP_TRY(c->node_hot.alloc(n));
        HIP_TRY(c->edge_hot.alloc(nnz));
        HIP_TRY(hipMemsetAsync(c->node_hot.p, 0xFF, c->node_hot.bytes(), c->stream));
        DevBuf<uint32_t> cnt;
        int rk = rank_nodes(c, cnt);
        cnt.release();
        if (rk) return rk;
        const int tb = 256;
        hipLaunchKernelGGL(k_assign_hot, dim3((unsigned)((ranked + tb - 1) / tb)), dim3(tb), 0, c->stream, c->ranked_ids.p, ranked, c->node_hot.p);
        if (nnz) hipLaunchKernelGGL(k_edge_hot, dim3((unsigned)((nnz + tb - 1) / tb)), dim3(tb), 0, c->stream, c->indices.p, c->node_hot.p, c->edge_hot.p, nnz);
        HIP_TRY(hipGetLastError());
        c->hot_ranked = ranked;
    }
    // ---- rank space (k_arcte_lines names nodes by rank): rank of every node and of every edge's target, row bounds
    //      and in_degree by rank
    {
        const int tb = 256;
        HIP_TRY(c->node_rank.alloc(n));
        HIP_TRY(c->edge_rank.alloc(nnz));
        HIP_TRY(c->rowspan.alloc(2 * n));
        HIP_TRY(c->in_degree_r.alloc(n));
        hipLaunchKernelGGL(k_node_rank, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, c->ranked_ids.p, n, c->node_rank.p);
        hipLaunchKernelGGL(k_rank_space, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, c->ranked_ids.p, c->indptr.p, c->in_degree.p, n,
                           c->rowspan.p, c->in_degree_r.p);
        HIP_TRY(hipGetLastError());
        // Packed rows (round 4): on graphs with narrow rows the in_degree of an edge's target is an integer, and all but the
        // highest-ranked nodes' fit beside the rank in ONE 32-bit word: 4 instead of 8 bytes per traversed edge.  The nodes
        // whose in_degree does not fit (or is no integer) are looked up in a float32 table by rank; packed only when those
        // lie among the first million ranks (a table part that stays cached).  ARCTE_HIP_PACK=0: the 8-byte stream.
        c->pack = 0;
        c->rank_bits = 1;
        while (c->rank_bits < 31 && ((int64_t)1 << c->rank_bits) < n) c->rank_bits++;
        // (test hook: more bits for the rank than the graph needs leave fewer for the in_degree, so that small graphs
        //  exercise the table lookup too)
        const int forced_bits = env_int("ARCTE_HIP_PACK_RANK_BITS", 0);
        if (forced_bits > (int)c->rank_bits && forced_bits <= 30) c->rank_bits = (uint32_t)forced_bits;
        if (c->narrow && nnz && (c->rank_bits <= 26 || forced_bits > 0) && env_int("ARCTE_HIP_PACK", 1)) {
            DevBuf<unsigned long long> st;
            unsigned long long h[2] = {0, 0};
            int rp = [&]() -> int {
                HIP_TRY(st.alloc(2));
                HIP_TRY(hipMemsetAsync(st.p, 0, 2 * sizeof(unsigned long long), c->stream));
                HIP_TRY(c->in_degree_rf.alloc(n));
                hipLaunchKernelGGL(k_in_degree_rf, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, c->in_degree_r.p, n, c->rank_bits,
                                   c->in_degree_rf.p, st.p);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(h, st.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                return 0;
            }();
            st.release();
            if (rp) return rp;
            c->pack_escapes = (int64_t)h[0];
            c->pack_escape_end = (int64_t)h[1];
            c->pack = c->pack_escape_end <= (int64_t)env_int("ARCTE_HIP_PACK_TABLE_RANKS", 1 << 20);
            if (!c->pack) c->in_degree_rf.release();
        }
        if (nnz && c->pack)
            hipLaunchKernelGGL(k_edge_pack, dim3((unsigned)((nnz + tb - 1) / tb)), dim3(tb), 0, c->stream, c->indices.p, c->node_rank.p, c->in_degree_r.p,
                               c->rank_bits, c->edge_rank.p, nnz);
        else if (nnz)
            hipLaunchKernelGGL(k_edge_rank, dim3((unsigned)((nnz + tb - 1) / tb)), dim3(tb), 0, c->stream, c->indices.p, c->node_rank.p, c->edge_rank.p, nnz);
        HIP_TRY(hipGetLastError());
    }
    // ---- knobs of the dense-state kernel (k_arcte_seeds: similarity slices, float32, helpers, ARCTE_HIP_STATE=dense)
    c->warm_k2 = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(env_int("ARCTE_HIP_WARM", 32768), c->hot_ranked));
    if (env_int("ARCTE_HIP_HOT", -1) == 0) c->warm_k2 = 0;
    c->waves_per_block = std::max(1, std::min(env_int("ARCTE_HIP_WAVES_PER_BLOCK", 1), WAVES_PER_BLOCK));
#ifdef ARCTE_HIP_AB_BUILDS
    c->coop = env_int("ARCTE_HIP_COOP", 0) != 0 && env_int("ARCTE_HIP_HOT", -1) != 0;
#else
    c->coop = 0;          // (helper wavefronts, two / four tiles per step, staged rows, the profile: `make AB=1`)
#endif
    c->coop_min = std::max(2 * 128, env_int("ARCTE_HIP_COOP_MIN", 512));
    if (c->coop) c->waves_per_block = 1;
    c->tiles = env_int("ARCTE_HIP_TILES", 1);
    if (c->tiles != 2 && c->tiles != 4) c->tiles = 1;
#ifndef ARCTE_HIP_AB_BUILDS
    c->tiles = 1;
#endif
    c->want_slots = n_slots;
    c->want_queue = queue_capacity;
    // ---- which state?  Lines (arcte_lines.hpp) unless the dense state is asked for.  The LDS bitmap covers the
    //      8 M highest-ranked nodes, M = ARCTE_HIP_LINES_LDS at most.  On the 1M/50M graph 99.4 % of the traversed edges
    //      point at the 524 288 highest-ranked nodes (tools/line_study.py); at twelve wavefronts per CU (12 KB of LDS
    //      each) a bitmap of 8 KB + 512 on-chip values beats 4 KB + 1 024 and 2 KB + 1 280 (ms per 81 434 seeds, interleaved
    //      processes on one box: 78.2 / 83.0 / 85.6; 4M-node graph 282 / 290), and 16 KB leaves no wavefront its share
    const char *state_env = getenv("ARCTE_HIP_STATE");
    //      Round 4 (profiles/r04/occupancy_sweeps.txt): on graphs of 6 M nodes and more (sparser rows, region B's lines indirect,
    //      sixteen wavefronts per CU: setup_lines) 4 KB of touched-bits + 640 on-chip values beat 8 KB + 128: 0.257-0.260 against
    //      0.245 at 8M (16M: 0.243 with 4 KB); at 4M 8 KB still wins (0.317 against 0.301-0.305).
    const int lines_default = (c->pack && n >= 6000000) ? 32768 : 65536;
    const uint32_t lines_lds = next_pow2((uint64_t)std::max(64, env_int("ARCTE_HIP_LINES_LDS", lines_default)));
    const uint32_t M = std::min<uint32_t>(lines_lds, std::max<uint32_t>(64u, next_pow2((uint64_t)((n + 7) / 8))));
    c->lines = !(state_env && state_env[0] == 'd') && !c->coop;
    int r = c->lines ? setup_lines(c, M) : setup_dense(c, n_slots, queue_capacity);
    if (r) return r;
    if (!c->lines) c->dense_auto = 1;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// get_natural_random_walk_matrix (transition.py:43-99) on the device.  On entry c->indptr / indices / data hold the
// ADJACENCY matrix as stored (any column order inside a row); on exit they hold W = D_out^-1 A with ascending
// columns, and out_degree / in_degree the weighted degrees, rounded as scipy rounds them (arcte_prepare.hpp).
int transition_on_device(arcte_hip_ctx *c)
{
    const int64_t n = c->n, nnz = c->nnz;
    const int tb = 256;
    const int row_blocks = (int)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    DevBuf<int32_t> flags, idx_sorted;
    DevBuf<uint32_t> count;
    DevBuf<int64_t> colptr;
    DevBuf<double> vals_by_col;
    DevBuf<uint64_t> keys_in, keys_out;
    DevBuf<double> data_sorted;
    DevBuf<char> temp;
    int rc = [&]() -> int {
        int32_t fl[2] = {0, 0};
        HIP_TRY(flags.alloc(2));
        HIP_TRY(hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), c->stream));
        hipLaunchKernelGGL(k_check_rows, dim3(row_blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->indices.p, n, flags.p);
        HIP_TRY(hipMemcpyAsync(fl, flags.p, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (fl[0]) return fail(ARCTE_HIP_EINVAL, "column index out of range");
        // transition.py:55-58 (sums are taken in STORAGE order, before sort_indices)
        hipLaunchKernelGGL(k_out_degree, dim3(row_blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->data.p, n, c->out_degree.p);
        // transition.py:56: stable sort of the stored entries by column, then one left fold per column
        HIP_TRY(count.alloc(n));
        HIP_TRY(colptr.alloc(n + 1));
        HIP_TRY(vals_by_col.alloc(nnz));
        HIP_TRY(idx_sorted.alloc(nnz));
        HIP_TRY(hipMemsetAsync(count.p, 0, count.bytes(), c->stream));
        HIP_TRY(hipMemsetAsync(colptr.p, 0, sizeof(int64_t), c->stream));
        if (nnz) hipLaunchKernelGGL(k_column_counts, dim3((unsigned)((nnz + tb - 1) / tb)), dim3(tb), 0, c->stream, c->indices.p, nnz, count.p);
        hipLaunchKernelGGL(k_u32_to_i64, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, count.p, n, colptr.p + 1);
        HIP_TRY(hipGetLastError());
        size_t tb1 = 0, tb2 = 0;
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tb1, colptr.p + 1, colptr.p + 1, (int)n, c->stream));
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, reinterpret_cast<const uint32_t *>(c->indices.p),
                                                   reinterpret_cast<uint32_t *>(idx_sorted.p), c->data.p, vals_by_col.p, (int)nnz, 0, 32, c->stream));
        HIP_TRY(temp.alloc(std::max(tb1, tb2)));
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(temp.p, tb1, colptr.p + 1, colptr.p + 1, (int)n, c->stream));
        if (nnz)
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, tb2, reinterpret_cast<const uint32_t *>(c->indices.p),
                                                       reinterpret_cast<uint32_t *>(idx_sorted.p), c->data.p, vals_by_col.p, (int)nnz, 0, 32, c->stream));
        hipLaunchKernelGGL(k_in_degree, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, colptr.p, vals_by_col.p, n, c->in_degree.p);
        HIP_TRY(hipGetLastError());
        vals_by_col.release(); idx_sorted.release();
        if (fl[1]) {
            // transition.py:65 sort_indices(): order the stored entries by (row, column)
            HIP_TRY(keys_in.alloc(nnz));
            HIP_TRY(keys_out.alloc(nnz));
            HIP_TRY(data_sorted.alloc(nnz));
            hipLaunchKernelGGL(k_csr_keys, dim3(row_blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->indices.p, n, keys_in.p);
            size_t tb3 = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb3, keys_in.p, keys_out.p, c->data.p, data_sorted.p, (int)nnz, 0, 64, c->stream));
            if (tb3 > temp.count) HIP_TRY(temp.alloc(tb3));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, tb3, keys_in.p, keys_out.p, c->data.p, data_sorted.p, (int)nnz, 0, 64, c->stream));
            hipLaunchKernelGGL(k_split_keys, dim3((unsigned)((nnz + tb - 1) / tb)), dim3(tb), 0, c->stream, keys_out.p, nnz, c->indices.p);
            HIP_TRY(hipMemcpyAsync(c->data.p, data_sorted.p, nnz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), c->stream));
            hipLaunchKernelGGL(k_check_rows, dim3(row_blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->indices.p, n, flags.p);
            HIP_TRY(hipMemcpyAsync(fl, flags.p, sizeof(fl), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (fl[1]) return fail(ARCTE_HIP_EINVAL, "a row stores the same column twice (call sum_duplicates() first)");
        }
        // transition.py:61-63
        hipLaunchKernelGGL(k_row_scale, dim3(row_blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->out_degree.p, n, c->data.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }();
    flags.release(); idx_sorted.release(); count.release(); colptr.release(); vals_by_col.release();
    keys_in.release(); keys_out.release(); data_sorted.release(); temp.release();
    return rc;
}

}  // namespace

// (arcte_io.cpp reports its errors through the same thread-local message)
extern "C" __attribute__((visibility("hidden"))) int arcte_io_set_error(int code, const char *msg) { return fail(code, msg ? msg : ""); }

extern "C" {

int arcte_hip_abi_version(void) { return 9; }

const char *arcte_hip_last_error(void) { return g_err.c_str(); }

int arcte_hip_device_count(int *count)
{
    if (!count) return fail(ARCTE_HIP_EINVAL, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *count = c;
    return 0;
}

int arcte_hip_create(int device, int64_t n, int64_t nnz, const int64_t *indptr, const int32_t *indices,
                     const double *data, const double *out_degree, const double *in_degree, int64_t n_slots,
                     int64_t queue_capacity, arcte_hip_ctx **out)
{
    if (!out) return fail(ARCTE_HIP_EINVAL, "out is NULL");
    *out = nullptr;
    if (n <= 0 || n >= (int64_t)1 << 31 || nnz < 0) return fail(ARCTE_HIP_EINVAL, "n must be in [1, 2^31), nnz >= 0");
    if (!indptr || !out_degree || !in_degree || (nnz > 0 && (!indices || !data)))
        return fail(ARCTE_HIP_EINVAL, "NULL graph array");
    if (indptr[0] != 0 || indptr[n] != nnz) return fail(ARCTE_HIP_EINVAL, "indptr does not span [0, nnz]");
    for (int64_t i = 0; i < n; i++)
        if (indptr[i + 1] < indptr[i]) return fail(ARCTE_HIP_EINVAL, "indptr is not monotone");
    {
        // (one branch-free sweep: argument errors are reported before any device work; the check that needs memory and a
        //  sort -- no column twice in a row -- runs on the device, validate_rows_on_device)
        uint32_t bad = 0;
        for (int64_t k = 0; k < nnz; k++) bad |= (uint32_t)(indices[k] < 0) | (uint32_t)((int64_t)indices[k] >= n);
        if (bad) return fail(ARCTE_HIP_EINVAL, "column index out of range");
    }
    arcte_hip_ctx *c = nullptr;
    int rc = ctx_begin(device, n, nnz, &c);
    if (rc) return rc;
    c->row_len.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) c->row_len[i] = (int32_t)std::min<int64_t>(indptr[i + 1] - indptr[i], INT32_MAX);
    rc = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(c->indptr.p, indptr, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        if (nnz) {
            HIP_TRY(hipMemcpyAsync(c->indices.p, indices, nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->data.p, data, nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipMemcpyAsync(c->out_degree.p, out_degree, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->in_degree.p, in_degree, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        int rv = validate_rows_on_device(c);
        if (rv) return rv;
        return ctx_finish(c, n_slots, queue_capacity);
    }();
    if (rc) {
        std::string keep = g_err;
        arcte_hip_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return 0;
}


// ---- device-side graph preparation -------------------------------------------------------------

int arcte_hip_create_from_adjacency(int device, int64_t n, int64_t nnz, const int64_t *indptr, const int32_t *indices,
                                    const double *data, int64_t n_slots, int64_t queue_capacity, arcte_hip_ctx **out)
{
    if (!out) return fail(ARCTE_HIP_EINVAL, "out is NULL");
    *out = nullptr;
    if (n <= 0 || n >= (int64_t)1 << 31 || nnz < 0 || nnz >= (int64_t)1 << 31) return fail(ARCTE_HIP_EINVAL, "n must be in [1, 2^31), nnz in [0, 2^31)");
    if (!indptr || (nnz > 0 && (!indices || !data))) return fail(ARCTE_HIP_EINVAL, "NULL graph array");
    if (indptr[0] != 0 || indptr[n] != nnz) return fail(ARCTE_HIP_EINVAL, "indptr does not span [0, nnz]");
    for (int64_t i = 0; i < n; i++)
        if (indptr[i + 1] < indptr[i]) return fail(ARCTE_HIP_EINVAL, "indptr is not monotone");
    arcte_hip_ctx *c = nullptr;
    int rc = ctx_begin(device, n, nnz, &c);
    if (rc) return rc;
    c->row_len.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) c->row_len[i] = (int32_t)std::min<int64_t>(indptr[i + 1] - indptr[i], INT32_MAX);
    rc = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(c->indptr.p, indptr, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        if (nnz) {
            HIP_TRY(hipMemcpyAsync(c->indices.p, indices, nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->data.p, data, nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        int r = transition_on_device(c);
        if (r) return r;
        return ctx_finish(c, n_slots, queue_capacity);
    }();
    if (rc) {
        std::string keep = g_err;
        arcte_hip_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return 0;
}

int arcte_hip_create_from_coo(int device, int64_t n, int64_t nnz, const int32_t *row, const int32_t *col, const double *val,
                              int symmetrise, int64_t n_slots, int64_t queue_capacity, arcte_hip_ctx **out)
{
    if (!out) return fail(ARCTE_HIP_EINVAL, "out is NULL");
    *out = nullptr;
    if (n <= 0 || n >= (int64_t)1 << 31 || nnz < 0 || nnz >= (int64_t)1 << 30) return fail(ARCTE_HIP_EINVAL, "n must be in [1, 2^31), nnz in [0, 2^30)");
    if (nnz > 0 && (!row || !col || !val)) return fail(ARCTE_HIP_EINVAL, "NULL triplet array");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(ARCTE_HIP_EHIP, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    const int tb = 256;
    DevBuf<int32_t> row_d, col_d, flags, rows_u, idx_u;
    DevBuf<double> val_d, vals, vals_sorted, data_u;
    DevBuf<uint64_t> keys, keys_sorted;
    DevBuf<int64_t> head;
    DevBuf<char> temp;
    arcte_hip_ctx *c = nullptr;
    // Stage 1: the triplets become a canonical CSR -- csr_matrix(coo) sums duplicate entries (here: in input order).
    // Stage 2 (symmetrise): (A + A^T)/2 of entry_points/arcte.py:70-71: the canonical entries and their mirror
    // images are merged the same way (at most two per position, a + b is exact in either order) and halved.
    auto canonicalise = [&](const uint64_t *keys_in, const double *vals_in, int64_t m, double scale, int64_t *unique_out) -> int {
        // sort (stable) -> head flags -> positions -> merged entries in rows_u / idx_u / data_u
        HIP_TRY(keys_sorted.reserve(m));
        HIP_TRY(vals_sorted.reserve(m));
        HIP_TRY(head.reserve(m));
        size_t t1 = 0, t2 = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, keys_in, keys_sorted.p, vals_in, vals_sorted.p, (int)m, 0, 64, 0));
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, t2, head.p, head.p, (int)m, 0));
        HIP_TRY(temp.reserve(std::max(t1, t2)));
        if (m == 0) { *unique_out = 0; return 0; }
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, t1, keys_in, keys_sorted.p, vals_in, vals_sorted.p, (int)m, 0, 64, 0));
        hipLaunchKernelGGL(k_head_flags, dim3((unsigned)((m + tb - 1) / tb)), dim3(tb), 0, 0, keys_sorted.p, m, head.p);
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(temp.p, t2, head.p, head.p, (int)m, 0));
        int64_t uniq = 0;
        HIP_TRY(hipMemcpy(&uniq, head.p + (m - 1), sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_TRY(rows_u.reserve(uniq));
        HIP_TRY(idx_u.reserve(uniq));
        HIP_TRY(data_u.reserve(uniq));
        hipLaunchKernelGGL(k_merge_duplicates, dim3((unsigned)((m + tb - 1) / tb)), dim3(tb), 0, 0, keys_sorted.p, vals_sorted.p, head.p, m,
                           scale, idx_u.p, data_u.p, rows_u.p);
        HIP_TRY(hipGetLastError());
        *unique_out = uniq;
        return 0;
    };
    int rc = [&]() -> int {
        HIP_TRY(row_d.alloc(nnz));
        HIP_TRY(col_d.alloc(nnz));
        HIP_TRY(val_d.alloc(nnz));
        HIP_TRY(flags.alloc(2));
        HIP_TRY(keys.alloc(2 * nnz));
        HIP_TRY(vals.alloc(2 * nnz));
        HIP_TRY(hipMemset(flags.p, 0, 2 * sizeof(int32_t)));
        if (nnz) {
            HIP_TRY(hipMemcpy(row_d.p, row, nnz * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(col_d.p, col, nnz * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(val_d.p, val, nnz * sizeof(double), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_coo_keys, dim3((unsigned)((nnz + tb - 1) / tb)), dim3(tb), 0, 0, row_d.p, col_d.p, val_d.p, nnz, 0, n,
                               keys.p, vals.p, flags.p);
            HIP_TRY(hipGetLastError());
        }
        int32_t fl[2] = {0, 0};
        HIP_TRY(hipMemcpy(fl, flags.p, sizeof(fl), hipMemcpyDeviceToHost));
        if (fl[0]) return fail(ARCTE_HIP_EINVAL, "triplet index out of range");
        int64_t uniq = 0;
        int r = canonicalise(keys.p, vals.p, nnz, 1.0, &uniq);
        if (r) return r;
        if (symmetrise && uniq) {
            // the canonical entries (rows_u, idx_u, data_u) and their mirror images
            HIP_TRY(keys.reserve(2 * uniq));
            HIP_TRY(vals.reserve(2 * uniq));
            hipLaunchKernelGGL(k_coo_keys, dim3((unsigned)((uniq + tb - 1) / tb)), dim3(tb), 0, 0, rows_u.p, idx_u.p, data_u.p, uniq, 1, n,
                               keys.p, vals.p, flags.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            r = canonicalise(keys.p, vals.p, 2 * uniq, 0.5, &uniq);
            if (r) return r;
        }
        r = ctx_begin(device, n, uniq, &c);
        if (r) return r;
        hipLaunchKernelGGL(k_rows_to_indptr, dim3((unsigned)((n + 1 + tb - 1) / tb)), dim3(tb), 0, 0, rows_u.p, uniq, n, c->indptr.p);
        HIP_TRY(hipGetLastError());
        if (uniq) {
            HIP_TRY(hipMemcpy(c->indices.p, idx_u.p, uniq * sizeof(int32_t), hipMemcpyDeviceToDevice));
            HIP_TRY(hipMemcpy(c->data.p, data_u.p, uniq * sizeof(double), hipMemcpyDeviceToDevice));
        }
        HIP_TRY(hipDeviceSynchronize());
        return 0;
    }();
    row_d.release(); col_d.release(); val_d.release(); flags.release(); rows_u.release(); idx_u.release();
    vals.release(); vals_sorted.release(); data_u.release(); keys.release(); keys_sorted.release(); head.release(); temp.release();
    if (!rc) {
        rc = transition_on_device(c);
        if (!rc) rc = ctx_finish(c, n_slots, queue_capacity);
    }
    if (rc) {
        std::string keep = g_err;
        if (c) arcte_hip_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return 0;
}

int arcte_hip_graph_sizes(arcte_hip_ctx *c, int64_t *n, int64_t *nnz, int64_t *nseeds)
{
    if (!c) return fail(ARCTE_HIP_EINVAL, "ctx is NULL");
    if (n) *n = c->n;
    if (nnz) *nnz = c->nnz;
    if (nseeds) *nseeds = c->nseeds_all;
    return 0;
}

int arcte_hip_fetch_transition(arcte_hip_ctx *c, int64_t *indptr, int32_t *indices, double *data, double *out_degree,
                               double *in_degree)
{
    if (!c) return fail(ARCTE_HIP_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (indptr) HIP_TRY(hipMemcpy(indptr, c->indptr.p, (c->n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (indices && c->nnz) HIP_TRY(hipMemcpy(indices, c->indices.p, c->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (data && c->nnz) HIP_TRY(hipMemcpy(data, c->data.p, c->nnz * sizeof(double), hipMemcpyDeviceToHost));
    if (out_degree) HIP_TRY(hipMemcpy(out_degree, c->out_degree.p, c->n * sizeof(double), hipMemcpyDeviceToHost));
    if (in_degree) HIP_TRY(hipMemcpy(in_degree, c->in_degree.p, c->n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int arcte_hip_fetch_seed_list(arcte_hip_ctx *c, int64_t *seeds)
{
    if (!c || (!seeds && c->nseeds_all)) return fail(ARCTE_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<int32_t> tmp((size_t)std::max<int64_t>(c->nseeds_all, 1));
    if (c->nseeds_all) HIP_TRY(hipMemcpy(tmp.data(), c->ranked_ids.p, c->nseeds_all * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < c->nseeds_all; k++) seeds[k] = tmp[k];
    return 0;
}

int arcte_hip_destroy(arcte_hip_ctx *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->registered) {
        forget_slot_memory_plan(c);
        c->registered = 0;
    }
    c->indptr.release(); c->indices.release(); c->data.release(); c->out_degree.release(); c->in_degree.release();
    release_cached(c->state, c->device); release_cached(c->sup, c->device); release_cached(c->queue, c->device); release_cached(c->warm, c->device);
    release_cached(c->contrib_key, c->device); release_cached(c->contrib_val, c->device);
    release_cached(c->contrib_key_sorted, c->device); release_cached(c->contrib_val_sorted, c->device);
    c->contrib_temp.release(); c->run_first.release(); c->run_last.release();
    c->l_block.release(c->device);
    c->l_blockb.release(c->device);
    c->l_stats.release();
    c->l_gen.release();
    c->dump_s.release(); c->dump_r.release();
    c->edge_rank.release(); c->node_rank.release(); c->rowspan.release(); c->in_degree_r.release(); c->in_degree_rf.release();
    c->state.release(); c->slot_epoch.release(); c->warm.release(); c->contrib_key.release(); c->contrib_val.release(); c->centrality.release(); c->ranked_ids.release(); c->node_hot.release(); c->edge_hot.release(); c->edge_in_degree.release(); c->data_f.release(); c->in_degree_f.release(); c->edge_in_degree_f.release(); c->queue.release(); c->hqueue.release(); c->prof.release(); c->sup.release();
    c->seeds_d.release(); c->work_pos.release(); c->out_cnt.release(); c->status.release(); c->nop_d.release();
    c->eps_d.release(); c->out_off.release(); c->dst_off.release(); c->counters.release();
    // (the output arena and the result rows are GBs: freed and allocated again by the next context's run they cost that run a
    //  hipMalloc behind deferred frees -- 1.85 s against 0.90 in profiles/r03/first_call_1m.txt -- so they are cached like the slots)
    release_cached(c->raw, c->device); release_cached(c->rows_final, c->device); c->sort_keys.release(); c->sort_iota.release(); c->sort_temp.release(); c->eps_big_pos.release(); c->sort_keys_in.release();
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

static int upload_seeds(arcte_hip_ctx *c, const int64_t *seeds, int64_t nseeds)
{
    std::vector<int32_t> s32((size_t)std::max<int64_t>(nseeds, 1));
    for (int64_t k = 0; k < nseeds; k++) {
        if (seeds[k] < 0 || seeds[k] >= c->n) return fail(ARCTE_HIP_EINVAL, "seed id out of range");
        s32[k] = (int32_t)seeds[k];
    }
    HIP_TRY(c->seeds_d.reserve(nseeds));
    HIP_TRY(c->eps_d.reserve(nseeds));
    if (nseeds) HIP_TRY(hipMemcpy(c->seeds_d.p, s32.data(), nseeds * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

static int launch_eps(arcte_hip_ctx *c, const int64_t *seeds, int64_t nseeds, double epsilon)
{
    if (nseeds == 0) return 0;
    int blocks = (int)((nseeds + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    hipLaunchKernelGGL(k_epsilon_effective, dim3(blocks), dim3(BLOCK), 0, c->stream, c->graph(), c->seeds_d.p, nseeds,
                       epsilon, c->eps_d.p);
    HIP_TRY(hipGetLastError());
    // rows too long for one wavefront get a workgroup each
    std::vector<int32_t> big;
    for (int64_t k = 0; k < nseeds; k++)
        if (c->row_len[(size_t)seeds[k]] >= EPS_BIG_ROW) big.push_back((int32_t)k);
    c->eps_big_count = (int64_t)big.size();
    if (!big.empty()) {
        HIP_TRY(c->eps_big_pos.reserve(big.size()));
        HIP_TRY(hipMemcpyAsync(c->eps_big_pos.p, big.data(), big.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_epsilon_effective_big, dim3((unsigned)big.size()), dim3(EPS_BIG_WAVES * WAVE), 0, c->stream,
                           c->graph(), c->seeds_d.p, c->eps_big_pos.p, (int64_t)big.size(), epsilon, c->eps_d.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));     // `big` is a host temporary
    }
    return 0;
}

int arcte_hip_epsilon_effective(arcte_hip_ctx *c, const int64_t *seeds, int64_t nseeds, double epsilon, double *eps_out)
{
    if (!c || nseeds < 0 || (nseeds && (!seeds || !eps_out))) return fail(ARCTE_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    c->run_nseeds = -1;
    int r = upload_seeds(c, seeds, nseeds);
    if (r) return r;
    r = launch_eps(c, seeds, nseeds, epsilon);
    if (r) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (nseeds) HIP_TRY(hipMemcpy(eps_out, c->eps_d.p, nseeds * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

// mode 0: arcte_worker's loop.  mode 2: the loop of arcte_and_centrality (the caller has reset the contribution
// cursor counters[8]); returns RC_RETRY_BATCH (> 0, nothing of the batch may be used) when a seed has to be re-run --
// its contributions could otherwise be counted twice -- after growing whatever was too small.
constexpr int RC_RETRY_BATCH = 1, RC_CONTRIB_FULL = 2;

int arcte_hip_epsilon_effective_scalar(int device, double epsilon, double seed_degree, const double *neighbor_degrees, int64_t m,
                                       double *eps_out)
{
    if (!eps_out || m < 0 || (m && !neighbor_degrees)) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (m == 0) return fail(ARCTE_HIP_EGRAPH, "zero-size array to reduction operation maximum which has no identity (arcte.py:39)");
    if (m >= ((int64_t)1 << 31) - 1) return fail(ARCTE_HIP_EINVAL, "too many neighbours");
    HIP_TRY(hipSetDevice(device));
    // a star: node 0 carries seed_degree, nodes 1..m the neighbour degrees -- the shapes the bulk kernels walk
    std::vector<int64_t> indptr_h((size_t)m + 2, m);
    indptr_h[0] = 0;
    std::vector<int32_t> indices_h((size_t)m);
    for (int64_t k = 0; k < m; k++) indices_h[(size_t)k] = (int32_t)(k + 1);
    std::vector<double> od_h((size_t)m + 1);
    od_h[0] = seed_degree;
    memcpy(od_h.data() + 1, neighbor_degrees, (size_t)m * sizeof(double));
    DevBuf<int64_t> indptr_d;
    DevBuf<int32_t> indices_d, seed_d, big_d;
    DevBuf<double> od_d, out_d;
    int rc = [&]() -> int {
        HIP_TRY(indptr_d.alloc(m + 2));
        HIP_TRY(indices_d.alloc(m));
        HIP_TRY(od_d.alloc(m + 1));
        HIP_TRY(out_d.alloc(1));
        HIP_TRY(seed_d.alloc(1));
        HIP_TRY(big_d.alloc(1));
        HIP_TRY(hipMemcpy(indptr_d.p, indptr_h.data(), (m + 2) * sizeof(int64_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(indices_d.p, indices_h.data(), m * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(od_d.p, od_h.data(), (m + 1) * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(seed_d.p, 0, sizeof(int32_t)));
        HIP_TRY(hipMemset(big_d.p, 0, sizeof(int32_t)));
        GraphDev g = {};
        g.n = m + 1;
        g.indptr = indptr_d.p;
        g.indices = indices_d.p;
        g.out_degree = od_d.p;
        if (m >= EPS_BIG_ROW)
            hipLaunchKernelGGL(k_epsilon_effective_big, dim3(1), dim3(EPS_BIG_WAVES * WAVE), 0, 0, g, seed_d.p, big_d.p, (int64_t)1, epsilon, out_d.p);
        else
            hipLaunchKernelGGL(k_epsilon_effective, dim3(1), dim3(BLOCK), 0, 0, g, seed_d.p, (int64_t)1, epsilon, out_d.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(eps_out, out_d.p, sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    }();
    indptr_d.release(); indices_d.release(); seed_d.release(); big_d.release(); od_d.release(); out_d.release();
    return rc;
}

static int run_seeds_impl(arcte_hip_ctx *c, const int64_t *seeds, int64_t nseeds, double rho, double epsilon,
                          int use_effective_epsilon, int variant, double lazy, int mode = 0)
{
    if (!c || nseeds < 0 || (nseeds && !seeds)) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (variant < 0 || variant > 2) return fail(ARCTE_HIP_EINVAL, "variant must be 0 (ARCTE), 1 (PageRank) or 2 (lazy PageRank)");
    if (nseeds >= ((int64_t)1 << 31)) return fail(ARCTE_HIP_EINVAL, "too many seeds for one call");
    HIP_TRY(hipSetDevice(c->device));
    auto t0 = std::chrono::steady_clock::now();
    c->run_nseeds = -1;
    c->centrality_run = 0;
    if (mode == 2 && (variant != 0 || c->float32)) return fail(ARCTE_HIP_EINVAL, "the centrality driver runs ARCTE's own push in float64");
    const bool use_lines = c->lines && !c->float32;      // (float32 is the tolerance sweep of the dense-state kernel)
    if (!use_lines) {
        int rd = ensure_dense(c, true);
        if (rd) return rd;
    }
    {
        int rp = prepare_precision(c);
        if (rp) return rp;
    }
    for (auto &x : c->line_stats) x = 0;
    c->final_rows = 0;
    for (auto &s : c->stats) s = 0;
    c->candidates = 0;
    c->split_rows = 0;
    for (auto &m : c->ms) m = 0;
    int r = upload_seeds(c, seeds, nseeds);
    if (r) return r;
    // epochs are 32-bit and advance by one per seed and slot: clear long before any slot can wrap
    c->seeds_since_clear += (uint64_t)nseeds;
    if (c->seeds_since_clear > (1ull << 31) && c->state.p) {
        if (c->warm.p) HIP_TRY(hipMemsetAsync(c->warm.p, 0, c->warm.bytes(), c->stream));
        HIP_TRY(hipMemsetAsync(c->state.p, 0, c->state.bytes(), c->stream));
        HIP_TRY(hipMemsetAsync(c->slot_epoch.p, 0, c->slot_epoch.bytes(), c->stream));
        c->seeds_since_clear = (uint64_t)nseeds;
    }
    HIP_TRY(c->out_cnt.reserve(nseeds));
    HIP_TRY(c->status.reserve(nseeds));
    HIP_TRY(c->nop_d.reserve(nseeds));
    HIP_TRY(c->out_off.reserve(nseeds));
    HIP_TRY(c->dst_off.reserve(nseeds));
    HIP_TRY(c->work_pos.reserve(nseeds));
    c->colptr.assign((size_t)nseeds + 1, 0);
    if (nseeds == 0) {
        c->run_nseeds = 0;
        return 0;
    }

    // a4
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    if (use_effective_epsilon) {
        r = launch_eps(c, seeds, nseeds, epsilon);
        if (r) return r;
    } else {
        std::vector<double> e((size_t)nseeds, epsilon);
        HIP_TRY(hipMemcpyAsync(c->eps_d.p, e.data(), nseeds * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));

    // raw arena: grows on demand (seeds that did not fit are re-run)
    if (c->raw.count < (size_t)c->n || c->raw_for_seeds < nseeds) {
        release_cached(c->raw, c->device);
        c->raw_for_seeds = nseeds;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        // 4096 rows per seed is ~3x what power-law graphs emit on average; the arena is reused by later
        // runs and re-filled (failed seeds re-run) when it is still too small
        size_t want = std::max<size_t>((size_t)c->n, std::min<size_t>((size_t)nseeds * 4096, (size_t)1 << 33));       // (3.1 M seeds of the 8M graph emit 5.9 G rows)
        want = std::min(want, std::max<size_t>((size_t)c->n, free_b / 4 / sizeof(int32_t)));
        if (const char *env = getenv("ARCTE_HIP_ARENA_ROWS")) {   // test hook: force a small arena
            long long v = atoll(env);
            if (v > 0) want = std::max<size_t>((size_t)c->n, (size_t)v);
        }
        HIP_TRY(alloc_cached(c->raw, want, c->device));          // (a destroyed context of this shape left its arena in the cache)
    }

    if (use_effective_epsilon) {
        // calculate_epsilon_effective reduces over the seed's out-neighbours (arcte.py:32,39-40): with none
        // numpy raises "zero-size array to reduction operation maximum which has no identity"
        for (int64_t k = 0; k < nseeds; k++)
            if (c->row_len[(size_t)seeds[k]] == 0)
                return fail(ARCTE_HIP_EGRAPH, "seed " + std::to_string(seeds[k]) +
                                                  " has no out-neighbours: the effective epsilon is undefined (the reference raises at arcte.py:39)");
    }
    // Work order: heaviest seed first, whatever order the caller listed the seeds in; results stay in the caller's
    // order.  With effective epsilons the weight of a seed is known almost exactly beforehand: the traversed edges
    // rank like 1/eps_eff (Spearman 0.95 on the config-1 graph, against 0.75 for the seed's degree, the reference's
    // own ordering key, arcte.py:614-616 -- the heaviest seeds are LOW-degree nodes with a small epsilon).  So the
    // first launch runs in ascending-epsilon order, sorted on the device; with a raw epsilon the degree decides.
    std::vector<int32_t> work((size_t)nseeds), next;   // positions (into seeds[]) still to run
    for (int64_t k = 0; k < nseeds; k++) work[k] = (int32_t)k;
    bool sorted_on_device = false;
    if (use_effective_epsilon) {
        int rs = [&]() -> int {
            HIP_TRY(c->sort_keys.reserve(nseeds));
            HIP_TRY(c->sort_iota.reserve(nseeds));
            HIP_TRY(hipMemcpyAsync(c->sort_iota.p, work.data(), nseeds * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            size_t temp_bytes = 0;
            // keys: the epsilons' bit patterns (positive doubles order like their bits); seeds with a big row get
            // key 0 = first: few pushes, but their first push and their threshold pass walk 10^4-10^5 edges with one
            // wavefront, which must not happen at the very end of the launch
            HIP_TRY(c->sort_keys_in.reserve(nseeds));
            HIP_TRY(hipMemcpyAsync(c->sort_keys_in.p, c->eps_d.p, nseeds * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
            if (c->eps_big_count > 0)   // A/B on one box: 156.6 vs 159.4 ms per bench launch
                hipLaunchKernelGGL(k_front_keys, dim3((unsigned)((c->eps_big_count + 255) / 256)), dim3(256), 0, c->stream,
                                   c->sort_keys_in.p, c->eps_big_pos.p, c->eps_big_count);
            const uint64_t *keys_in = c->sort_keys_in.p;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in, c->sort_keys.p, c->sort_iota.p, c->work_pos.p,
                                                       (int)nseeds, 0, 64, c->stream));
            HIP_TRY(c->sort_temp.reserve(temp_bytes));
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->sort_temp.p, temp_bytes, keys_in, c->sort_keys.p, c->sort_iota.p,
                                                       c->work_pos.p, (int)nseeds, 0, 64, c->stream));
            return 0;
        }();
        if (rs) return rs;
        sorted_on_device = true;
    } else {
        std::stable_sort(work.begin(), work.end(), [&](int32_t a, int32_t b) {
            return c->row_len[(size_t)seeds[a]] > c->row_len[(size_t)seeds[b]];
        });
    }
    std::vector<int32_t> status_h((size_t)nseeds), cnt_h((size_t)nseeds);
    std::vector<int64_t> dst_h((size_t)nseeds, 0);
    std::vector<int64_t> seg_start((size_t)nseeds, 0);   // where each seed's rows sit in rows_final (launch order)
    bool identity = false;
    int64_t final_used = 0;
    double ms_push = 0, ms_compact = 0;
    if (env_int("ARCTE_HIP_PROFILE", 0) != 0) {
        if (!c->prof.p) HIP_TRY(c->prof.alloc(16));
        HIP_TRY(hipMemsetAsync(c->prof.p, 0, c->prof.bytes(), c->stream));
    } else c->prof.release();
    int launches = 0;
    while (!work.empty()) {
        const int64_t nwork = (int64_t)work.size();
        if (!identity && !(sorted_on_device && launches == 0))
            HIP_TRY(hipMemcpyAsync(c->work_pos.p, work.data(), nwork * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(c->counters.p, 0, 8 * sizeof(unsigned long long), c->stream));     // (the contribution cursor [8] runs on)
        HIP_TRY(hipMemsetAsync(c->counters.p + 9, 0, sizeof(unsigned long long), c->stream));     // split rows
        PushParams P;
        P.g = c->graph();
        P.work_pos = identity ? nullptr : c->work_pos.p;
        P.nwork = nwork;
        P.work_counter = c->counters.p + 0;
        P.seeds = c->seeds_d.p;
        P.eps = c->eps_d.p;
        P.one_minus_rho = 1 - rho;
        P.rho = rho;
        P.lazy = lazy;
        P.state = (void *)c->state.p;
        P.slot_epoch = c->slot_epoch.p;
        P.queue = c->queue.p;
        P.hqueue = c->hqueue.p;
        P.coop_min = c->coop_min;
        P.prof = c->prof.p;
        P.sup = c->sup.p;
        P.qcap = c->qcap;
        P.max_pushes = max_pushes_limit();
        P.raw = c->raw.p;
        P.rawcap = c->raw.count;
        P.raw_cursor = c->counters.p + 1;
        P.out_off = c->out_off.p;
        P.out_cnt = c->out_cnt.p;
        P.status = c->status.p;
        P.nop = c->nop_d.p;
        P.stats = c->counters.p + 2;
        P.contrib_key = c->contrib_key.p;
        P.contrib_val = c->contrib_val.p;
        P.contrib_cap = c->contrib_key.count;
        P.contrib_cursor = c->counters.p + 8;
        P.contrib_seed_base = c->contrib_seed_base;
        P.contrib_shift = c->contrib_shift;
        if (use_lines) HIP_TRY(hipMemsetAsync(c->l_stats.p, 0, c->l_stats.bytes(), c->stream));
        HIP_TRY(hipEventRecord(c->ev[2], c->stream));
        if (use_lines) r = launch_lines(c, P, nwork, variant, mode);
        else r = (mode == 2) ? launch_centrality(c, P, nwork) : launch_seeds<0>(c, P, nwork, variant);
        if (r) return r;
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        launches++;
        unsigned long long cnt8[10], lst[4] = {0, 0, 0, 0};
        if (use_lines) HIP_TRY(hipMemcpyAsync(lst, c->l_stats.p, sizeof(lst), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(cnt8, c->counters.p, sizeof(cnt8), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(status_h.data(), c->status.p, nseeds * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(cnt_h.data(), c->out_cnt.p, nseeds * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
        ms_push += ms;
        for (int i = 0; i < 4; i++) c->stats[i] += (int64_t)cnt8[2 + i];
        for (int i = 0; i < 4; i++) c->line_stats[i] += (int64_t)lst[i];
        c->candidates += (int64_t)cnt8[7];
        c->split_rows += (int64_t)cnt8[9];

        // finished seeds of this launch: their rows are appended to rows_final in caller (position) order
        next.clear();
        bool queue_over = false, out_over = false, contrib_over = false, pushed_over = false, sup_over = false, pool_over = false;
        int64_t add = 0;
        std::vector<int32_t> by_pos(work);
        if (!std::is_sorted(by_pos.begin(), by_pos.end())) std::sort(by_pos.begin(), by_pos.end());      // (the first launch lists every position in order)
        for (int32_t pos : by_pos) {
            const int32_t st = status_h[pos];
            dst_h[pos] = final_used + add;
            if (st == ST_OK) {
                seg_start[pos] = dst_h[pos];
                c->colptr[(size_t)pos + 1] = cnt_h[pos];
                add += cnt_h[pos];
            } else if (st == ST_RUNAWAY) {
                return fail(ARCTE_HIP_ECAPACITY, "seed " + std::to_string(seeds[pos]) + ": push cap reached (" +
                                                     std::to_string(max_pushes_limit()) + " pushes without converging)");
            } else if (st == ST_MISSING_BASE) {
                return fail(ARCTE_HIP_EGRAPH, "seed " + std::to_string(seeds[pos]) +
                                                  ": closed neighbourhood not contained in the support (zero-weight edge?)");
            } else {
                queue_over |= (st == ST_QUEUE_OVERFLOW);
                out_over |= (st == ST_OUTPUT_OVERFLOW);
                contrib_over |= (st == ST_CONTRIB_OVERFLOW);
                pushed_over |= (st == ST_PUSHED_OVERFLOW);
                sup_over |= (st == ST_SUP_OVERFLOW);
                pool_over |= (st == ST_POOL_OVERFLOW);
                next.push_back(pos);
            }
        }
        if (mode == 2 && !next.empty()) {
            if (contrib_over) return RC_CONTRIB_FULL;
            if (use_lines && (queue_over || pushed_over || sup_over || pool_over)) {
                r = grow_lines(c, queue_over, pushed_over, sup_over, pool_over);
                if (r) return r;
            } else if (queue_over) {
                r = grow_queue(c);
                if (r) return r;
            }
            if (out_over) {
                if (c->raw.count >= ((size_t)1 << 33)) return fail(ARCTE_HIP_ECAPACITY, "the output arena cannot grow further");
                size_t want = c->raw.count * 4;
                HIP_TRY(c->raw.alloc(want));
            }
            return RC_RETRY_BATCH;
        }
        if (add > 0) {
            if ((size_t)(final_used + add) > c->rows_final.count) {
                DevBuf<int32_t> bigger;
                ScopedRelease<int32_t> bigger_guard(bigger);
                size_t want = std::max<size_t>((size_t)(final_used + add + c->rows_reserve_extra), c->rows_final.count * 2);
                HIP_TRY(alloc_cached(bigger, want, c->device));
                if (final_used)
                    HIP_TRY(hipMemcpyAsync(bigger.p, c->rows_final.p, final_used * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                c->rows_final.release();
                c->rows_final = bigger;
                bigger.p = nullptr;
            }
            HIP_TRY(hipMemcpyAsync(c->dst_off.p, dst_h.data(), nseeds * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipEventRecord(c->ev[4], c->stream));
            int blocks = (int)((nwork + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
            // failed seeds wrote a zero count, so they copy nothing
            hipLaunchKernelGGL(k_gather_segments, dim3(blocks), dim3(BLOCK), 0, c->stream, c->raw.p, c->out_off.p,
                               c->out_cnt.p, c->dst_off.p, c->rows_final.p, P.work_pos, nwork);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c->ev[5], c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipEventElapsedTime(&ms, c->ev[4], c->ev[5]));
            ms_compact += ms;
            final_used += add;
        }
        if (!next.empty()) {
            c->stats[4] += (int64_t)next.size();
            if (use_lines && (queue_over || pushed_over || sup_over || pool_over)) {
                r = grow_lines(c, queue_over, pushed_over, sup_over, pool_over);
                if (r) return r;
            } else if (queue_over) {
                r = grow_queue(c);
                if (r) return r;
            }
            if (out_over && next.size() == (size_t)nwork) {
                // not a single seed fitted: the arena itself is too small
                if (c->raw.count >= ((size_t)1 << 33)) return fail(ARCTE_HIP_ECAPACITY, "no seed fits the output arena");
                size_t want = c->raw.count * 4;
                HIP_TRY(c->raw.alloc(want));
            }
        }
        work.swap(next);
        identity = false;
    }
    c->stats[5] = launches;

    // colptr in seed order; rows_final is in launch order -> one more gather when there were re-runs
    for (int64_t k = 0; k < nseeds; k++) c->colptr[k + 1] += c->colptr[k];
    if (launches > 1 && final_used > 0) {
        DevBuf<int32_t> ordered;
        HIP_TRY(ordered.alloc(final_used));
        std::vector<int32_t> cnt_all((size_t)nseeds);
        for (int64_t k = 0; k < nseeds; k++) cnt_all[k] = (int32_t)(c->colptr[k + 1] - c->colptr[k]);
        HIP_TRY(hipMemcpyAsync(c->out_cnt.p, cnt_all.data(), nseeds * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->out_off.p, seg_start.data(), nseeds * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->dst_off.p, c->colptr.data(), nseeds * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        int blocks = (int)((nseeds + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
        hipLaunchKernelGGL(k_gather_segments, dim3(blocks), dim3(BLOCK), 0, c->stream, c->rows_final.p, c->out_off.p,
                           c->out_cnt.p, c->dst_off.p, ordered.p, (const int32_t *)nullptr, nseeds);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->rows_final.release();
        c->rows_final = ordered;
        ordered.p = nullptr;
    }
    c->final_rows = final_used;
    c->run_nseeds = nseeds;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->ms[0] = ms;
    c->ms[1] = ms_push;
    c->ms[2] = ms_compact;
    c->ms[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (c->prof.p) {
        // ARCTE_HIP_PROFILE=1 (a study aid, tools/phase_profile.py): s_memtime ticks per phase, summed over wavefronts
        unsigned long long p[10];
        HIP_TRY(hipMemcpy(p, c->prof.p, sizeof(p), hipMemcpyDeviceToHost));
        fprintf(stderr, "[arcte_hip profile] ticks: setup %llu pop_batches %llu short_pushes %llu long_pushes %llu pop_loop_rest %llu extraction %llu draw %llu | "
                        "counts: short_pushes %llu long_pushes %llu pop_batches %llu | push_ms %.3f slots %lld\n",
                p[0], p[1], p[2], p[3], p[4], p[5], p[9], p[6], p[7], p[8], ms_push, (long long)(use_lines ? c->l_slots : c->slots));
    }
    return 0;
}

int arcte_hip_run_seeds(arcte_hip_ctx *c, const int64_t *seeds, int64_t nseeds, double rho, double epsilon,
                        int use_effective_epsilon)
{
    return run_seeds_impl(c, seeds, nseeds, rho, epsilon, use_effective_epsilon, 0, 0.0);
}

int arcte_hip_run_seeds_variant(arcte_hip_ctx *c, const int64_t *seeds, int64_t nseeds, double rho, double epsilon,
                                int use_effective_epsilon, int variant, double laziness_factor)
{
    return run_seeds_impl(c, seeds, nseeds, rho, epsilon, use_effective_epsilon, variant, laziness_factor);
}

int arcte_hip_run_centrality(arcte_hip_ctx *c, int64_t node_begin, int64_t node_end, double rho, double epsilon)
{
    if (!c || node_begin < 0 || node_end > c->n || node_begin > node_end) return fail(ARCTE_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    c->run_nseeds = -1;
    std::vector<int64_t> seeds;                                        // arcte.pyx:165: nodes with out-edges, in index order
    for (int64_t i = node_begin; i < node_end; i++)
        if (c->row_len[(size_t)i] > 0) seeds.push_back(i);
    const int64_t ns = (int64_t)seeds.size();
    HIP_TRY(c->centrality.alloc(c->n));
    HIP_TRY(hipMemsetAsync(c->centrality.p, 0, c->centrality.bytes(), c->stream));
    // contribution arena of one batch: (node << shift | seed - first seed of the batch, value) pairs, sorted and folded per batch
    if (!c->contrib_key.p) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        size_t cap = std::min<size_t>((size_t)1 << 29, free_b / 8 / 16);       // larger batches sort slower (measured: 2^29 1.36 s, 2^30 1.53 s, 2^31 1.88 s)
        if (const char *env = getenv("ARCTE_HIP_CONTRIB_ENTRIES")) {   // test hook: force small batches
            long long v = atoll(env);
            if (v > 0) cap = (size_t)v;
        }
        cap = std::max<size_t>(cap, (size_t)c->n);
        HIP_TRY(alloc_cached(c->contrib_key, cap, c->device));
        HIP_TRY(alloc_cached(c->contrib_val, cap, c->device));
    }
    DevBuf<int32_t> all_rows;
    DevBuf<uint64_t> &key_sorted = c->contrib_key_sorted;
    DevBuf<double> &val_sorted = c->contrib_val_sorted;
    DevBuf<int64_t> &run_first = c->run_first, &run_last = c->run_last;
    DevBuf<char> &temp = c->contrib_temp;
    std::vector<int64_t> all_colptr((size_t)ns + 1, 0);
    int64_t stats_sum[6] = {0, 0, 0, 0, 0, 0}, cand_sum = 0, rows_used = 0;
    double ms_sum[4] = {0, 0, 0, 0};
    auto t0 = std::chrono::steady_clock::now();
    auto bits_for = [](uint64_t values) { int b = 1; while (b < 63 && ((uint64_t)1 << b) < values) b++; return b; };
    int rc = [&]() -> int {
        const int node_bits = bits_for((uint64_t)c->n);
        if (!run_first.p) {
            HIP_TRY(run_first.alloc(c->n));
            HIP_TRY(run_last.alloc(c->n));
        }
        double per_seed = (double)std::min<int64_t>(c->n, 4096);
        int64_t pos = 0;
        while (pos < ns) {
            int64_t batch = std::max<int64_t>(1, std::min<int64_t>(ns - pos, (int64_t)((double)c->contrib_key.count / per_seed)));
            unsigned long long m = 0;
            for (;;) {
                // the seeds of a batch are ascending node ids: the sort key carries their offset in the batch's id range,
                // so the radix sort runs over log2(range) + log2(n) bits instead of 32 + log2(n)
                c->contrib_seed_base = seeds[(size_t)pos];
                c->contrib_shift = bits_for((uint64_t)(seeds[(size_t)(pos + batch - 1)] - seeds[(size_t)pos] + 1));
                HIP_TRY(hipMemsetAsync(c->counters.p + 8, 0, sizeof(unsigned long long), c->stream));
                int r = run_seeds_impl(c, seeds.data() + pos, batch, rho, epsilon, 0, 0, 0.0, 2);
                if (r < 0) return r;
                if (r == RC_CONTRIB_FULL) {
                    if (batch == 1) {
                        size_t want = c->contrib_key.count * 2;
                        HIP_TRY(alloc_cached(c->contrib_key, want, c->device));
                        HIP_TRY(alloc_cached(c->contrib_val, want, c->device));
                    } else batch = std::max<int64_t>(1, batch / 2);
                    continue;
                }
                if (r == RC_RETRY_BATCH) continue;
                break;
            }
            HIP_TRY(hipMemcpy(&m, c->counters.p + 8, sizeof(m), hipMemcpyDeviceToHost));
            if (m) {
                const int key_bits = c->contrib_shift + node_bits;
                if (key_sorted.count < c->contrib_key.count) {         // one allocation of the arena's size, not one per batch
                    HIP_TRY(alloc_cached(key_sorted, c->contrib_key.count, c->device));
                    HIP_TRY(alloc_cached(val_sorted, c->contrib_key.count, c->device));
                }
                size_t tb = 0;
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, c->contrib_key.p, key_sorted.p, c->contrib_val.p, val_sorted.p, (size_t)m, 0,
                                                           key_bits, c->stream));
                HIP_TRY(temp.reserve(tb));
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, tb, c->contrib_key.p, key_sorted.p, c->contrib_val.p, val_sorted.p, (size_t)m, 0,
                                                           key_bits, c->stream));
                HIP_TRY(hipMemsetAsync(run_first.p, 0, run_first.bytes(), c->stream));
                HIP_TRY(hipMemsetAsync(run_last.p, 0, run_last.bytes(), c->stream));
                hipLaunchKernelGGL(k_contribution_bounds, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, key_sorted.p, (int64_t)m,
                                   c->contrib_shift, run_first.p, run_last.p);
                hipLaunchKernelGGL(k_apply_contributions, dim3((unsigned)((c->n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)), dim3(BLOCK), 0, c->stream,
                                   val_sorted.p, run_first.p, run_last.p, c->n, c->centrality.p);
                HIP_TRY(hipGetLastError());
            }
            // this batch's communities behind the earlier ones
            if (c->final_rows) {
                if ((size_t)(rows_used + c->final_rows) > all_rows.count) {
                    DevBuf<int32_t> bigger;
                    ScopedRelease<int32_t> bigger_guard(bigger);
                    HIP_TRY(bigger.alloc(std::max<size_t>((size_t)(rows_used + c->final_rows), all_rows.count * 2)));
                    if (rows_used) HIP_TRY(hipMemcpyAsync(bigger.p, all_rows.p, rows_used * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    all_rows.release();
                    all_rows = bigger;
                    bigger.p = nullptr;
                }
                HIP_TRY(hipMemcpyAsync(all_rows.p + rows_used, c->rows_final.p, c->final_rows * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
            }
            for (int64_t k = 0; k < batch; k++) all_colptr[(size_t)(pos + k) + 1] = rows_used + c->colptr[(size_t)k + 1];
            rows_used += c->final_rows;
            for (int i = 0; i < 6; i++) stats_sum[i] += c->stats[i];
            cand_sum += c->candidates;
            for (int i = 0; i < 3; i++) ms_sum[i] += c->ms[i];
            HIP_TRY(hipStreamSynchronize(c->stream));
            per_seed = std::max(64.0, 1.25 * (double)m / (double)batch);
            pos += batch;
        }
        if (node_end > node_begin)
            hipLaunchKernelGGL(k_centrality_non_seeds, dim3((unsigned)((node_end - node_begin + 255) / 256)), dim3(256), 0, c->stream,
                               c->indptr.p + node_begin, node_end - node_begin, c->centrality.p + node_begin);
        HIP_TRY(hipGetLastError());
        int r = upload_seeds(c, seeds.data(), ns);
        if (r) return r;
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }();
    if (rc) { all_rows.release(); c->run_nseeds = -1; return rc; }
    c->rows_final.release();
    c->rows_final = all_rows;
    all_rows.p = nullptr;
    c->final_rows = rows_used;
    c->colptr.swap(all_colptr);
    for (int i = 0; i < 6; i++) c->stats[i] = stats_sum[i];
    c->candidates = cand_sum;
    for (int i = 0; i < 3; i++) c->ms[i] = ms_sum[i];
    c->ms[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->run_nseeds = ns;
    c->centrality_run = 1;
    return 0;
}

int arcte_hip_fetch_centrality(arcte_hip_ctx *c, double *centrality)
{
    if (!c || !centrality) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0 || !c->centrality_run) return fail(ARCTE_HIP_ESTATE, "no completed centrality run on this context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(centrality, c->centrality.p, c->n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int arcte_hip_result_sizes(arcte_hip_ctx *c, int64_t *nseeds, int64_t *total_rows)
{
    if (!c) return fail(ARCTE_HIP_EINVAL, "ctx is NULL");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    if (nseeds) *nseeds = c->run_nseeds;
    if (total_rows) *total_rows = c->final_rows;
    return 0;
}

int arcte_hip_fetch_result(arcte_hip_ctx *c, int64_t *colptr, int32_t *rows, double *eps_used, int64_t *nop)
{
    if (!c) return fail(ARCTE_HIP_EINVAL, "ctx is NULL");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t ns = c->run_nseeds;
    if (colptr) memcpy(colptr, c->colptr.data(), (ns + 1) * sizeof(int64_t));
    if (rows && c->final_rows)
        HIP_TRY(hipMemcpy(rows, c->rows_final.p, c->final_rows * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (eps_used && ns) HIP_TRY(hipMemcpy(eps_used, c->eps_d.p, ns * sizeof(double), hipMemcpyDeviceToHost));
    if (nop && ns) {
        std::vector<int32_t> tmp((size_t)ns);
        HIP_TRY(hipMemcpy(tmp.data(), c->nop_d.p, ns * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < ns; k++) nop[k] = tmp[k];
    }
    return 0;
}

int arcte_hip_result_csr_size(arcte_hip_ctx *c, int with_base_block, int64_t *nnz)
{
    if (!c || !nnz) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    *nnz = c->final_rows + (with_base_block ? c->nnz + c->n : 0);
    return 0;
}

// The last run as CSR on the device: indptr_d[n+1], cols (uint32 column ids, the first *valid of them).
// Columns: seed id (arcte(): every seed owns its column) or, after arcte_hip_run_centrality, a running counter over the
// seeds that emitted a community (arcte.pyx:213-215); *n_local_cols receives the width of the local block.
static int assemble_csr_device(arcte_hip_ctx *c, int with_base_block, DevBuf<int64_t> &indptr_d, DevBuf<uint32_t> &cols, int64_t *valid_out,
                               int64_t *n_local_cols = nullptr)
{
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t n = c->n, ns = c->run_nseeds;
    const int64_t nbase = with_base_block ? c->nnz + n : 0;
    const int64_t nkeys = nbase + c->final_rows;
    if (const char *env = getenv("ARCTE_HIP_MAX_SORT_KEYS")) {      // test hook: force the host-assembly fallback
        long long v = atoll(env);
        if (v > 0 && nkeys >= v) return fail(ARCTE_HIP_ECAPACITY, "too many entries for the device assembly (test hook): assemble on the host instead");
    }
    // the seeds in ascending id order: the local pairs are written seed by seed in that order, so every row's
    // columns are ascending before the (stable) sort by row
    std::vector<int32_t> seeds_h((size_t)std::max<int64_t>(ns, 1)), seg_of((size_t)std::max<int64_t>(ns, 1));
    std::vector<int64_t> dst((size_t)std::max<int64_t>(ns, 1));
    if (ns) HIP_TRY(hipMemcpy(seeds_h.data(), c->seeds_d.p, ns * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < ns; k++) seg_of[(size_t)k] = (int32_t)k;
    std::stable_sort(seg_of.begin(), seg_of.begin() + ns, [&](int32_t a, int32_t b) { return seeds_h[(size_t)a] < seeds_h[(size_t)b]; });
    for (int64_t j = 1; j < ns; j++)
        if (seeds_h[(size_t)seg_of[(size_t)j]] == seeds_h[(size_t)seg_of[(size_t)j - 1]])
            return fail(ARCTE_HIP_EINVAL, "the run listed a seed twice: its column would hold duplicate entries");
    {
        int64_t o = nbase;
        for (int64_t j = 0; j < ns; j++) {
            dst[(size_t)j] = o;
            o += c->colptr[(size_t)seg_of[(size_t)j] + 1] - c->colptr[(size_t)seg_of[(size_t)j]];
        }
    }
    DevBuf<uint32_t> key_a, key_b, val_a;
    DevBuf<int64_t> colptr_d, dst_d;
    DevBuf<int32_t> seg_d, colid_d;
    DevBuf<char> temp;
    std::vector<int32_t> colid_h;
    int64_t local_cols = n;
    if (c->centrality_run) {
        colid_h.assign((size_t)std::max<int64_t>(ns, 1), 0);
        int32_t next_col = 0;
        for (int64_t k = 0; k < ns; k++)
            if (c->colptr[(size_t)k + 1] > c->colptr[(size_t)k]) colid_h[(size_t)k] = next_col++;
        local_cols = next_col;
    }
    if (n_local_cols) *n_local_cols = local_cols;
    const bool verbose = env_int("ARCTE_HIP_VERBOSE", 0) != 0;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    int rc = [&]() -> int {
        if (c->centrality_run) {
            HIP_TRY(colid_d.alloc(ns));
            if (ns) HIP_TRY(hipMemcpyAsync(colid_d.p, colid_h.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        }
        // The sort's large buffers come from (and go back to) the process-wide cache: a second arcte() call of the same shape
        // allocates nothing large.  That matters beyond the allocation's own time: memory given back to the driver -- the losers of
        // the slot-memory draw, 60 GB each -- is cleared before it is handed out again, and the next LARGE hipMalloc waits for that
        // (3 x 60 GB freed: 5.1 s; tools/free_cost_probe.hip, profiles/r04/free_cost_probe.txt).  It was these allocations that
        // waited, in the first or the second call (profiles/r04/first_call_1m.txt).
        HIP_TRY(alloc_cached(key_a, (size_t)nkeys, c->device));
        HIP_TRY(alloc_cached(key_b, (size_t)nkeys, c->device));
        HIP_TRY(alloc_cached(val_a, (size_t)nkeys, c->device));
        HIP_TRY(alloc_cached(cols, (size_t)nkeys, c->device));
        HIP_TRY(indptr_d.alloc(n + 1));
        HIP_TRY(colptr_d.alloc(ns + 1));
        HIP_TRY(dst_d.alloc(ns));
        HIP_TRY(seg_d.alloc(ns));
        const double t_alloc = since();
        HIP_TRY(hipMemcpyAsync(colptr_d.p, c->colptr.data(), (ns + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        if (ns) {
            HIP_TRY(hipMemcpyAsync(dst_d.p, dst.data(), ns * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(seg_d.p, seg_of.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        }
        if (with_base_block) {
            int blocks = (int)((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
            hipLaunchKernelGGL(k_pairs_base, dim3(blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->indices.p, n, key_a.p, val_a.p);
        }
        if (ns && c->final_rows) {
            int blocks = (int)((ns + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
            hipLaunchKernelGGL(k_pairs_local, dim3(blocks), dim3(BLOCK), 0, c->stream, c->rows_final.p, colptr_d.p,
                               c->centrality_run ? colid_d.p : c->seeds_d.p, seg_d.p, dst_d.p, ns, with_base_block ? (uint32_t)n : 0u, key_a.p,
                               val_a.p);
        }
        HIP_TRY(hipGetLastError());
        int end_bit = 1;                      // row ids 0 .. n (n = dropped identity entries, sorted to the end)
        while (end_bit < 32 && ((uint64_t)1 << end_bit) <= (uint64_t)n) end_bit++;
        size_t temp_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, key_a.p, key_b.p, val_a.p, cols.p, (size_t)nkeys, 0, end_bit, c->stream));
        HIP_TRY(alloc_cached(temp, temp_bytes, c->device));
        if (nkeys) HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, key_a.p, key_b.p, val_a.p, cols.p, (size_t)nkeys, 0, end_bit, c->stream));
        const int tb = 256;
        hipLaunchKernelGGL(k_rows_to_indptr_u32, dim3((unsigned)((n + 1 + tb - 1) / tb)), dim3(tb), 0, c->stream, key_b.p, nkeys, n, indptr_d.p);
        HIP_TRY(hipGetLastError());
        int64_t valid = 0;                    // pairs with row id n (dropped identity entries) sit behind row n-1
        HIP_TRY(hipMemcpyAsync(&valid, indptr_d.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        *valid_out = valid;
        if (verbose)
            fprintf(stderr, "[arcte_hip] CSR assembly of %lld pairs: %.1f ms to the buffers (host order of the seeds + device allocations), %.1f ms pairs + sort + indptr\n",
                    (long long)nkeys, t_alloc, since() - t_alloc);
        return 0;
    }();
    const double t_done = since();
    release_cached(key_a, c->device); release_cached(key_b, c->device); release_cached(val_a, c->device); release_cached(temp, c->device);
    colptr_d.release(); dst_d.release(); seg_d.release(); colid_d.release();
    if (rc) release_cached(cols, c->device);
    if (verbose) fprintf(stderr, "[arcte_hip] CSR assembly: %.1f ms to give the sort's buffers back\n", since() - t_done);
    return rc;
}

int arcte_hip_fetch_result_csr(arcte_hip_ctx *c, int with_base_block, int64_t *indptr, int32_t *indices, int64_t *nnz_out)
{
    if (!c || !indptr || !nnz_out) return fail(ARCTE_HIP_EINVAL, "bad argument");
    DevBuf<int64_t> indptr_d;
    DevBuf<uint32_t> cols;
    int64_t valid = 0;
    int rc = assemble_csr_device(c, with_base_block, indptr_d, cols, &valid);
    if (!rc) rc = [&]() -> int {
        if (valid && !indices) return fail(ARCTE_HIP_EINVAL, "indices is NULL");
        HIP_TRY(hipMemcpy(indptr, indptr_d.p, (c->n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
        // the column ids go out in pieces, so that the pageable destination is faulted in while the next piece moves
        // (Round 4: a destination pinned by hipHostRegister, taken in one piece, gains nothing -- 881 M column ids in 115.7 ms either
        //  way on a box whose link moves 30.5 GB/s, whichever NUMA node touched the pages first: profiles/r04/e2e_arcte_1m.txt)
        const int64_t piece = (int64_t)64 << 20;
        const auto t0 = std::chrono::steady_clock::now();
        for (int64_t o = 0; o < valid; o += piece)
            HIP_TRY(hipMemcpy(indices + o, cols.p + o, std::min(piece, valid - o) * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (env_int("ARCTE_HIP_VERBOSE", 0))
            fprintf(stderr, "[arcte_hip] CSR copy-out of %lld column ids: %.1f ms\n", (long long)valid,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        *nnz_out = valid;
        return 0;
    }();
    indptr_d.release(); release_cached(cols, c->device);
    return rc;
}

int arcte_hip_result_device_rows(arcte_hip_ctx *c, void **rows_dev)
{
    if (!c || !rows_dev) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    *rows_dev = c->rows_final.p;
    return 0;
}

int arcte_hip_copy_result_rows_to_device(arcte_hip_ctx *c, void *dst_dev, int64_t capacity_rows)
{
    if (!c || (!dst_dev && capacity_rows)) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    if (capacity_rows < c->final_rows) return fail(ARCTE_HIP_EINVAL, "destination too small");
    HIP_TRY(hipSetDevice(c->device));
    if (c->final_rows) {
        HIP_TRY(hipMemcpyAsync(dst_dev, c->rows_final.p, c->final_rows * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int arcte_hip_append_result(arcte_hip_ctx *c, const int64_t *seeds, const int64_t *counts, int64_t nseeds, const void *rows, int64_t nrows)
{
    if (!c || nseeds < 0 || nrows < 0 || (nseeds && (!seeds || !counts)) || (nrows && !rows)) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context to append to");
    HIP_TRY(hipSetDevice(c->device));
    int64_t sum = 0;
    std::vector<int32_t> s32((size_t)std::max<int64_t>(nseeds, 1));
    if (c->centrality_run && nseeds) {
        // arcte.pyx:213-215 numbers the columns by a running counter over the seeds in node order: a part joins a centrality
        // run only as the NEXT node block (the distributed driver appends the ranks' blocks in rank order)
        int32_t last = -1;
        if (c->run_nseeds > 0) HIP_TRY(hipMemcpy(&last, c->seeds_d.p + (c->run_nseeds - 1), sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < nseeds; k++) {
            if (seeds[k] <= (k ? seeds[k - 1] : (int64_t)last))
                return fail(ARCTE_HIP_EINVAL, "a part appended to a centrality run must continue it in ascending node order");
        }
    }
    for (int64_t k = 0; k < nseeds; k++) {
        if (seeds[k] < 0 || seeds[k] >= c->n) return fail(ARCTE_HIP_EINVAL, "seed id out of range");
        if (counts[k] < 0) return fail(ARCTE_HIP_EINVAL, "negative community size");
        s32[(size_t)k] = (int32_t)seeds[k];
        sum += counts[k];
    }
    if (sum != nrows) return fail(ARCTE_HIP_EINVAL, "the community sizes do not add up to the number of rows");
    const int64_t ns0 = c->run_nseeds, rows0 = c->final_rows;
    if (ns0 + nseeds >= ((int64_t)1 << 31)) return fail(ARCTE_HIP_EINVAL, "too many seeds for one result");
    // grow the two device arrays, keeping what they hold
    if ((size_t)(ns0 + nseeds) > c->seeds_d.capacity) {
        DevBuf<int32_t> bigger;
        ScopedRelease<int32_t> bigger_guard(bigger);
        HIP_TRY(bigger.alloc(std::max<size_t>((size_t)(ns0 + nseeds), c->seeds_d.capacity * 2)));
        if (ns0) HIP_TRY(hipMemcpy(bigger.p, c->seeds_d.p, ns0 * sizeof(int32_t), hipMemcpyDeviceToDevice));
        c->seeds_d.release();
        c->seeds_d = bigger;
        bigger.p = nullptr;
    }
    c->seeds_d.count = (size_t)(ns0 + nseeds);
    if ((size_t)(rows0 + nrows) > c->rows_final.capacity) {
        DevBuf<int32_t> bigger;
        ScopedRelease<int32_t> bigger_guard(bigger);
        HIP_TRY(bigger.alloc(std::max<size_t>((size_t)(rows0 + nrows), c->rows_final.capacity * 2)));
        if (rows0) HIP_TRY(hipMemcpy(bigger.p, c->rows_final.p, rows0 * sizeof(int32_t), hipMemcpyDeviceToDevice));
        c->rows_final.release();
        c->rows_final = bigger;
        bigger.p = nullptr;
    }
    c->rows_final.count = (size_t)(rows0 + nrows);
    if (nseeds) HIP_TRY(hipMemcpy(c->seeds_d.p + ns0, s32.data(), nseeds * sizeof(int32_t), hipMemcpyHostToDevice));
    // (hipMemcpyDefault: the rows may lie in host memory or on any device of this process)
    if (nrows) HIP_TRY(hipMemcpy(c->rows_final.p + rows0, rows, nrows * sizeof(int32_t), hipMemcpyDefault));
    c->colptr.resize((size_t)(ns0 + nseeds) + 1);
    for (int64_t k = 0; k < nseeds; k++) c->colptr[(size_t)(ns0 + k) + 1] = c->colptr[(size_t)(ns0 + k)] + counts[k];
    c->run_nseeds = ns0 + nseeds;
    c->final_rows = rows0 + nrows;
    return 0;
}

// A second (third, ...) part of a seed list on one context: the completed run's result stays and this run's columns join it, as
// if arcte_hip_append_result had been handed them -- what lets a caller size and fault in its host arrays from the first part
// while the rest of the seeds runs (embedding/arcte/arcte.py).  The reference's counterpart is the sum of the chunk matrices
// (arcte.py:384-386, 670-673); every seed owns its column, so the sum is a concatenation.
int arcte_hip_run_seeds_append(arcte_hip_ctx *c, const int64_t *seeds, int64_t nseeds, double rho, double epsilon,
                               int use_effective_epsilon, int variant, double laziness_factor)
{
    if (!c) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context to append to");
    if (c->centrality_run) return fail(ARCTE_HIP_ESTATE, "the columns of a centrality run are numbered by a running counter: parts cannot be appended");
    HIP_TRY(hipSetDevice(c->device));
    // the completed part steps aside: seeds and sizes on the host, rows where they are
    const int64_t ns0 = c->run_nseeds, rows0 = c->final_rows;
    std::vector<int32_t> s32((size_t)std::max<int64_t>(ns0, 1));
    if (ns0) HIP_TRY(hipMemcpy(s32.data(), c->seeds_d.p, ns0 * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int64_t> seeds0((size_t)ns0), counts0((size_t)ns0);
    for (int64_t k = 0; k < ns0; k++) {
        seeds0[(size_t)k] = s32[(size_t)k];
        counts0[(size_t)k] = c->colptr[(size_t)k + 1] - c->colptr[(size_t)k];
    }
    int64_t stats0[6], cand0 = c->candidates, split0 = c->split_rows, lines0[4];
    double ms0[4];
    for (int i = 0; i < 6; i++) stats0[i] = c->stats[i];
    for (int i = 0; i < 4; i++) { lines0[i] = c->line_stats[i]; ms0[i] = c->ms[i]; }
    DevBuf<int32_t> rows_prev = c->rows_final;
    ScopedRelease<int32_t> rows_prev_guard(rows_prev);
    c->rows_final = DevBuf<int32_t>();
    c->rows_reserve_extra = rows0;
    int rc = run_seeds_impl(c, seeds, nseeds, rho, epsilon, use_effective_epsilon, variant, laziness_factor);
    c->rows_reserve_extra = 0;
    if (rc) return rc;
    rc = arcte_hip_append_result(c, seeds0.data(), counts0.data(), ns0, rows_prev.p, rows0);
    if (rc) { c->run_nseeds = -1; return rc; }
    for (int i = 0; i < 6; i++) c->stats[i] += stats0[i];
    for (int i = 0; i < 4; i++) { c->line_stats[i] += lines0[i]; c->ms[i] += ms0[i]; }
    c->candidates += cand0;
    c->split_rows += split0;
    return 0;
}

int arcte_hip_run_stats(arcte_hip_ctx *c, int64_t stats[6])
{
    if (!c || !stats) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    for (int i = 0; i < 6; i++) stats[i] = c->stats[i];
    return 0;
}

int arcte_hip_run_counters(arcte_hip_ctx *c, int64_t *out, int n)
{
    if (!c || !out || n < 0) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    const int64_t all[8] = {c->stats[0], c->stats[1], c->stats[2], c->stats[3], c->stats[4], c->stats[5], c->candidates, c->split_rows};
    for (int i = 0; i < n; i++) out[i] = i < 8 ? all[i] : 0;
    return 0;
}

int arcte_hip_run_timing(arcte_hip_ctx *c, double ms[4])
{
    if (!c || !ms) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->run_nseeds < 0) return fail(ARCTE_HIP_ESTATE, "no completed run on this context");
    for (int i = 0; i < 4; i++) ms[i] = c->ms[i];
    return 0;
}

static int similarity_slice_impl(arcte_hip_ctx *c, int64_t seed, double rho, double epsilon, int variant, double lazy,
                                 double *s, double *r, int64_t *nop)
{
    if (!c || !s || !r) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (variant < 0 || variant > 2) return fail(ARCTE_HIP_EINVAL, "variant must be 0 (ARCTE), 1 (PageRank) or 2 (lazy PageRank)");
    if (seed < 0 || seed >= c->n) return fail(ARCTE_HIP_EINVAL, "seed id out of range");
    HIP_TRY(hipSetDevice(c->device));
    c->run_nseeds = -1;
    {
        // works on the caller's dense s and r (any content): the dense-state kernel on slot 0
        int rd0 = ensure_dense(c, false);
        if (rd0) return rd0;
        int rp = prepare_precision(c);
        if (rp) return rp;
    }
    const int64_t n = c->n;
    DevBuf<double> sd, rd;
    DevBuf<int32_t> seed_d, small;     // small: cnt, status, nop
    DevBuf<int64_t> off_d;
    DevBuf<double> eps1;
    int rc = [&]() -> int {
        HIP_TRY(sd.alloc(n));
        HIP_TRY(rd.alloc(n));
        HIP_TRY(seed_d.alloc(1));
        HIP_TRY(small.alloc(3));
        HIP_TRY(off_d.alloc(1));
        HIP_TRY(eps1.alloc(1));
        int32_t s32 = (int32_t)seed;
        HIP_TRY(hipMemcpy(seed_d.p, &s32, sizeof(s32), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(eps1.p, &epsilon, sizeof(double), hipMemcpyHostToDevice));
        for (;;) {
            HIP_TRY(hipMemcpyAsync(sd.p, s, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(rd.p, r, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
            const int tb = 256;
            if (c->float32)
                hipLaunchKernelGGL(k_state_from_dense<float>, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, sd.p, rd.p,
                                   c->in_degree.p, (void *)c->state.p, c->slot_epoch.p, n);
            else
                hipLaunchKernelGGL(k_state_from_dense<double>, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream, sd.p, rd.p,
                                   c->in_degree.p, (void *)c->state.p, c->slot_epoch.p, n);
            HIP_TRY(hipMemsetAsync(c->counters.p, 0, 8 * sizeof(unsigned long long), c->stream));
            PushParams P;
            P.g = c->graph();
            P.work_pos = nullptr;
            P.nwork = 1;
            P.work_counter = c->counters.p + 0;
            P.seeds = seed_d.p;
            P.eps = eps1.p;
            P.one_minus_rho = 1 - rho;
            P.rho = rho;
            P.lazy = lazy;
            P.state = (void *)c->state.p;
            P.slot_epoch = c->slot_epoch.p;
            P.queue = c->queue.p;
        P.hqueue = c->hqueue.p;
        P.coop_min = c->coop_min;
        P.prof = c->prof.p;
            P.sup = c->sup.p;
            P.qcap = c->qcap;
            P.max_pushes = max_pushes_limit();
                P.raw = nullptr;
            P.rawcap = 0;
            P.raw_cursor = c->counters.p + 1;
            P.out_off = off_d.p;
            P.out_cnt = small.p + 0;
            P.status = small.p + 1;
            P.nop = small.p + 2;
            P.stats = c->counters.p + 2;
            int r2 = launch_seeds<1>(c, P, 1, variant);
            if (r2) return r2;
            int32_t h[3];
            HIP_TRY(hipMemcpyAsync(h, small.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (h[1] == ST_RUNAWAY) {
                return fail(ARCTE_HIP_ECAPACITY, "push cap reached after " + std::to_string(h[2]) + " pushes");
            }
            if (h[1] == ST_QUEUE_OVERFLOW) {
                // the epoch has moved on, so slot 0 is clean again; retry with a deeper ring
                r2 = grow_queue(c);
                if (r2) return r2;
                continue;
            }
            if (c->float32)
                hipLaunchKernelGGL(k_state_to_dense<float>, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream,
                                   (const void *)c->state.p, c->slot_epoch.p, sd.p, rd.p, n);
            else
                hipLaunchKernelGGL(k_state_to_dense<double>, dim3((unsigned)((n + tb - 1) / tb)), dim3(tb), 0, c->stream,
                                   (const void *)c->state.p, c->slot_epoch.p, sd.p, rd.p, n);
            HIP_TRY(hipMemcpyAsync(s, sd.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(r, rd.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            // nothing to clean: the next seed on slot 0 runs in a new epoch
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (nop) *nop = h[2];
            return 0;
        }
    }();
    sd.release(); rd.release(); seed_d.release(); small.release(); off_d.release(); eps1.release();
    return rc;
}

int arcte_hip_similarity_slice(arcte_hip_ctx *c, int64_t seed, double rho, double epsilon, double *s, double *r,
                               int64_t *nop)
{
    return similarity_slice_impl(c, seed, rho, epsilon, 0, 0.0, s, r, nop);
}

int arcte_hip_similarity_slice_variant(arcte_hip_ctx *c, int64_t seed, double rho, double epsilon, int variant,
                                       double laziness_factor, double *s, double *r, int64_t *nop)
{
    return similarity_slice_impl(c, seed, rho, epsilon, variant, laziness_factor, s, r, nop);
}

int arcte_hip_seed_state(arcte_hip_ctx *c, int64_t seed, double rho, double epsilon, int use_effective_epsilon, int variant,
                         double laziness_factor, double *s, double *r, int64_t *nop)
{
    if (!c || !s || !r) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (seed < 0 || seed >= c->n) return fail(ARCTE_HIP_EINVAL, "seed id out of range");
    if (!c->lines || c->float32) return fail(ARCTE_HIP_ESTATE, "this context does not run the line-state kernel (ARCTE_HIP_STATE=dense or float32)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->dump_s.reserve((size_t)c->n));
    HIP_TRY(c->dump_r.reserve((size_t)c->n));
    // (a seed that outgrows a capacity is run again by run_seeds_impl: the last run's dump stands)
    c->dump_on = 1;
    const int rc = run_seeds_impl(c, &seed, 1, rho, epsilon, use_effective_epsilon, variant, laziness_factor);
    c->dump_on = 0;
    if (rc) return rc;
    HIP_TRY(hipMemcpy(s, c->dump_s.p, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r, c->dump_r.p, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost));
    if (nop) {
        int32_t h = 0;
        HIP_TRY(hipMemcpy(&h, c->nop_d.p, sizeof(h), hipMemcpyDeviceToHost));
        *nop = h;
    }
    return 0;
}

static int push_impl(int device, int64_t n, double *s, double *r, const double *w_i, const int32_t *a_i, int64_t deg,
                     int64_t push_node, double rho, int variant, double lazy)
{
    if (variant < 0 || variant > 2) return fail(ARCTE_HIP_EINVAL, "variant must be 0 (ARCTE), 1 (PageRank) or 2 (lazy PageRank)");
    if (!s || !r || n <= 0 || deg < 0 || (deg && (!w_i || !a_i))) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (push_node < 0 || push_node >= n) return fail(ARCTE_HIP_EINVAL, "push_node out of range");
    for (int64_t k = 0; k < deg; k++)
        if (a_i[k] < 0 || a_i[k] >= n) return fail(ARCTE_HIP_EINVAL, "adjacent node out of range");
    HIP_TRY(hipSetDevice(device));
    DevBuf<double> sd, rd, wd;
    DevBuf<int32_t> ad;
    int rc = [&]() -> int {
        HIP_TRY(sd.alloc(n));
        HIP_TRY(rd.alloc(n));
        HIP_TRY(wd.alloc(deg));
        HIP_TRY(ad.alloc(deg));
        HIP_TRY(hipMemcpy(sd.p, s, n * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(rd.p, r, n * sizeof(double), hipMemcpyHostToDevice));
        if (deg) {
            HIP_TRY(hipMemcpy(wd.p, w_i, deg * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(ad.p, a_i, deg * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(k_single_push, dim3(1), dim3(BLOCK), 0, 0, sd.p, rd.p, wd.p, ad.p, deg, push_node, rho, 1 - rho,
                           variant, lazy);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(s, sd.p, n * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(r, rd.p, n * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    }();
    sd.release(); rd.release(); wd.release(); ad.release();
    return rc;
}

int arcte_hip_push(int device, int64_t n, double *s, double *r, const double *w_i, const int32_t *a_i, int64_t deg,
                   int64_t push_node, double rho)
{
    return push_impl(device, n, s, r, w_i, a_i, deg, push_node, rho, 0, 0.0);
}

int arcte_hip_push_variant(int device, int64_t n, double *s, double *r, const double *w_i, const int32_t *a_i,
                           int64_t deg, int64_t push_node, double rho, int variant, double laziness_factor)
{
    return push_impl(device, n, s, r, w_i, a_i, deg, push_node, rho, variant, laziness_factor);
}

// ---- device-resident feature matrices ----------------------------------------------------------

struct arcte_hip_features {
    int device = 0;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    DevBuf<int64_t> indptr;
    DevBuf<int32_t> indices;
    DevBuf<double> data;
    DevBuf<uint32_t> col_count;             // stored entries per column of the CURRENT pattern (valid while col_count_valid)
    bool col_count_valid = false;
};

// Stored entries per column; kept with the matrix until its pattern changes (every weighting step asks for them)
static int column_counts(arcte_hip_features *f)
{
    if (f->col_count_valid) return 0;
    HIP_TRY(f->col_count.alloc(f->n_cols));
    HIP_TRY(hipMemset(f->col_count.p, 0, f->col_count.bytes()));
    if (f->nnz) hipLaunchKernelGGL(k_feat_column_counts, dim3((unsigned)((f->nnz + 255) / 256)), dim3(256), 0, 0, f->indices.p, f->nnz, f->col_count.p);
    HIP_TRY(hipGetLastError());
    f->col_count_valid = true;
    return 0;
}

int arcte_hip_features_destroy(arcte_hip_features *f)
{
    if (!f) return 0;
    (void)hipSetDevice(f->device);
    f->indptr.release(); f->indices.release(); f->data.release(); f->col_count.release();
    delete f;
    return 0;
}

int arcte_hip_features_upload(int device, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *indptr, const int32_t *indices,
                              const double *data, arcte_hip_features **out)
{
    if (!out) return fail(ARCTE_HIP_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_rows < 0 || n_cols <= 0 || n_cols >= ((int64_t)1 << 31) || nnz < 0 || !indptr || (nnz && (!indices || !data)))
        return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (indptr[0] != 0 || indptr[n_rows] != nnz) return fail(ARCTE_HIP_EINVAL, "indptr does not span [0, nnz]");
    for (int64_t k = 0; k < nnz; k++)
        if (indices[k] < 0 || indices[k] >= n_cols) return fail(ARCTE_HIP_EINVAL, "column index out of range");
    HIP_TRY(hipSetDevice(device));
    arcte_hip_features *f = new arcte_hip_features();
    f->device = device; f->n_rows = n_rows; f->n_cols = n_cols; f->nnz = nnz;
    int rc = [&]() -> int {
        HIP_TRY(f->indptr.alloc(n_rows + 1));
        HIP_TRY(f->indices.alloc(nnz));
        HIP_TRY(f->data.alloc(nnz));
        HIP_TRY(hipMemcpy(f->indptr.p, indptr, (n_rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        if (nnz) {
            HIP_TRY(hipMemcpy(f->indices.p, indices, nnz * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(f->data.p, data, nnz * sizeof(double), hipMemcpyHostToDevice));
        }
        return 0;
    }();
    if (rc) { std::string keep = g_err; arcte_hip_features_destroy(f); g_err = keep; return rc; }
    *out = f;
    return 0;
}

// value of every stored entry of arcte()'s matrix: 1.0, and 2.0 on the diagonal of a node with a self-loop (I + ones)
__global__ __launch_bounds__(BLOCK) void k_feat_values_of_result(const int64_t *w_indptr, const int32_t *w_indices, const int64_t *indptr,
                                                                 const int32_t *indices, int64_t n, int with_base_block, double *data)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t b = indptr[i], e = indptr[i + 1];
    for (int64_t k = b + lane; k < e; k += WAVE) data[k] = 1.0;
    if (!with_base_block) return;
    bool loop = false;
    for (int64_t k = w_indptr[i] + lane; k < w_indptr[i + 1]; k += WAVE) loop |= (w_indices[k] == (int32_t)i);
    if (__ballot(loop) == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int64_t k = b + lane; k < e; k += WAVE)
        if (indices[k] == (int32_t)i) data[k] = 2.0;
}

// arcte_and_centrality's matrix (arcte.pyx:219-228): local communities hold ones; the base block is identity + the
// matrix the reference holds under the name adjacency_matrix at that point -- which is W (cython_opt/transition.pyx:19
// normalises its argument in place), so entry (i, c) = W[i, c] + (1 if c == i)
__global__ __launch_bounds__(BLOCK) void k_feat_values_identity_plus_w(const int64_t *w_indptr, const int32_t *w_indices, const double *w_data,
                                                                       const int64_t *indptr, const int32_t *indices, int64_t n, double *data)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t wb = w_indptr[i], we = w_indptr[i + 1];
    for (int64_t k = indptr[i] + lane; k < indptr[i + 1]; k += WAVE) {
        const int32_t col = indices[k];
        double v = 1.0;
        if ((int64_t)col < n) {
            int64_t lo = wb, hi = we;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (w_indices[mid] < col) lo = mid + 1; else hi = mid;
            }
            const double w = (lo < we && w_indices[lo] == col) ? w_data[lo] : 0.0;
            v = (col == (int32_t)i) ? 1.0 + w : w;
        }
        data[k] = v;
    }
}

int arcte_hip_features_from_result(arcte_hip_ctx *c, int with_base_block, arcte_hip_features **out)
{
    if (!c || !out) return fail(ARCTE_HIP_EINVAL, "bad argument");
    *out = nullptr;
    DevBuf<int64_t> indptr_d;
    DevBuf<uint32_t> cols;
    int64_t valid = 0;
    int64_t local_cols = 0;
    int rc = assemble_csr_device(c, with_base_block, indptr_d, cols, &valid, &local_cols);
    if (rc) { indptr_d.release(); release_cached(cols, c->device); return rc; }
    arcte_hip_features *f = new arcte_hip_features();
    f->device = c->device; f->n_rows = c->n; f->n_cols = (with_base_block ? c->n : 0) + local_cols; f->nnz = valid;
    f->indptr = indptr_d; indptr_d.p = nullptr;
    f->indices.p = reinterpret_cast<int32_t *>(cols.p); f->indices.count = (size_t)valid; f->indices.capacity = cols.capacity; cols.p = nullptr;
    rc = [&]() -> int {
        HIP_TRY(f->data.alloc(valid));
        int blocks = (int)((c->n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
        if (c->centrality_run && with_base_block)
            hipLaunchKernelGGL(k_feat_values_identity_plus_w, dim3(blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->indices.p, c->data.p,
                               f->indptr.p, f->indices.p, c->n, f->data.p);
        else
            hipLaunchKernelGGL(k_feat_values_of_result, dim3(blocks), dim3(BLOCK), 0, c->stream, c->indptr.p, c->indices.p, f->indptr.p, f->indices.p,
                               c->n, with_base_block, f->data.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }();
    if (rc) { std::string keep = g_err; arcte_hip_features_destroy(f); g_err = keep; return rc; }
    *out = f;
    return 0;
}

int arcte_hip_features_sizes(arcte_hip_features *f, int64_t *n_rows, int64_t *n_cols, int64_t *nnz)
{
    if (!f) return fail(ARCTE_HIP_EINVAL, "features is NULL");
    if (n_rows) *n_rows = f->n_rows;
    if (n_cols) *n_cols = f->n_cols;
    if (nnz) *nnz = f->nnz;
    return 0;
}

int arcte_hip_features_fetch(arcte_hip_features *f, int64_t *indptr, int32_t *indices, double *data)
{
    if (!f) return fail(ARCTE_HIP_EINVAL, "features is NULL");
    HIP_TRY(hipSetDevice(f->device));
    if (indptr) HIP_TRY(hipMemcpy(indptr, f->indptr.p, (f->n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (indices && f->nnz) HIP_TRY(hipMemcpy(indices, f->indices.p, f->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (data && f->nnz) HIP_TRY(hipMemcpy(data, f->data.p, f->nnz * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int arcte_hip_features_normalize_columns(arcte_hip_features *f)
{
    if (!f) return fail(ARCTE_HIP_EINVAL, "features is NULL");
    HIP_TRY(hipSetDevice(f->device));
    DevBuf<double> divisor;
    int rc = [&]() -> int {
        int r = column_counts(f);
        if (r) return r;
        HIP_TRY(divisor.alloc(f->n_cols));
        hipLaunchKernelGGL(k_feat_idf, dim3((unsigned)((f->n_cols + 255) / 256)), dim3(256), 0, 0, f->col_count.p, f->n_cols, divisor.p);
        if (f->nnz) hipLaunchKernelGGL(k_feat_divide_columns, dim3((unsigned)((f->nnz + 255) / 256)), dim3(256), 0, 0, f->indices.p, divisor.p, f->nnz, f->data.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        return 0;
    }();
    divisor.release();
    return rc;
}

int arcte_hip_features_normalize_rows(arcte_hip_features *f)
{
    if (!f) return fail(ARCTE_HIP_EINVAL, "features is NULL");
    HIP_TRY(hipSetDevice(f->device));
    if (f->n_rows)
        hipLaunchKernelGGL(k_feat_normalize_rows, dim3((unsigned)((f->n_rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)), dim3(BLOCK), 0, 0, f->indptr.p,
                           f->n_rows, f->data.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int arcte_hip_features_community_weighting(arcte_hip_features *f, const double *community_weights)
{
    if (!f || !community_weights) return fail(ARCTE_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(f->device));
    DevBuf<double> w_d, factor, data_out;
    DevBuf<int64_t> pos, indptr_out;
    DevBuf<int32_t> indices_out;
    DevBuf<char> temp;
    int rc = [&]() -> int {
        int r = column_counts(f);
        if (r) return r;
        HIP_TRY(w_d.alloc(f->n_cols));
        HIP_TRY(factor.alloc(f->n_cols));
        HIP_TRY(hipMemcpy(w_d.p, community_weights, f->n_cols * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_feat_reinforcement, dim3((unsigned)((f->n_cols + 255) / 256)), dim3(256), 0, 0, f->col_count.p, w_d.p, f->n_cols, factor.p);
        if (f->nnz) {
            hipLaunchKernelGGL(k_feat_multiply_columns, dim3((unsigned)((f->nnz + 255) / 256)), dim3(256), 0, 0, f->indices.p, factor.p, f->nnz, f->data.p);
            // eliminate_zeros() (:115-116)
            HIP_TRY(pos.alloc(f->nnz));
            hipLaunchKernelGGL(k_feat_nonzero_flags, dim3((unsigned)((f->nnz + 255) / 256)), dim3(256), 0, 0, f->data.p, f->nnz, pos.p);
            size_t tb = 0;
            HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tb, pos.p, pos.p, (size_t)f->nnz, 0));
            HIP_TRY(temp.alloc(tb));
            HIP_TRY(hipcub::DeviceScan::InclusiveSum(temp.p, tb, pos.p, pos.p, (size_t)f->nnz, 0));
            int64_t kept = 0;
            HIP_TRY(hipMemcpy(&kept, pos.p + (f->nnz - 1), sizeof(int64_t), hipMemcpyDeviceToHost));
            if (kept != f->nnz) {
                HIP_TRY(indices_out.alloc(kept));
                HIP_TRY(data_out.alloc(kept));
                HIP_TRY(indptr_out.alloc(f->n_rows + 1));
                hipLaunchKernelGGL(k_feat_compact, dim3((unsigned)((f->nnz + 255) / 256)), dim3(256), 0, 0, f->indices.p, f->data.p, pos.p, f->nnz,
                                   indices_out.p, data_out.p);
                hipLaunchKernelGGL(k_feat_compact_indptr, dim3((unsigned)((f->n_rows + 1 + 255) / 256)), dim3(256), 0, 0, f->indptr.p, pos.p, f->n_rows,
                                   indptr_out.p);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipDeviceSynchronize());
                f->indices.release(); f->data.release(); f->indptr.release();
                f->indices = indices_out; indices_out.p = nullptr;
                f->data = data_out; data_out.p = nullptr;
                f->indptr = indptr_out; indptr_out.p = nullptr;
                f->nnz = kept;
                f->col_count_valid = false;          // the pattern changed
            }
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }();
    w_d.release(); factor.release(); data_out.release(); pos.release(); indptr_out.release(); indices_out.release(); temp.release();
    if (rc) return rc;
    return arcte_hip_features_normalize_rows(f);        // :118-119
}

int arcte_hip_features_select_rows(arcte_hip_features *f, const int64_t *rows, int64_t nsel, arcte_hip_features **out)
{
    if (!f || !out || nsel < 0 || (nsel && !rows)) return fail(ARCTE_HIP_EINVAL, "bad argument");
    *out = nullptr;
    for (int64_t i = 0; i < nsel; i++)
        if (rows[i] < 0 || rows[i] >= f->n_rows) return fail(ARCTE_HIP_EINVAL, "row index out of range");
    HIP_TRY(hipSetDevice(f->device));
    arcte_hip_features *g = new arcte_hip_features();
    g->device = f->device; g->n_rows = nsel; g->n_cols = f->n_cols;
    DevBuf<int64_t> rows_d;
    DevBuf<char> temp;
    int rc = [&]() -> int {
        HIP_TRY(rows_d.alloc(nsel));
        HIP_TRY(g->indptr.alloc(nsel + 1));
        HIP_TRY(hipMemset(g->indptr.p, 0, sizeof(int64_t)));
        if (nsel) {
            HIP_TRY(hipMemcpy(rows_d.p, rows, nsel * sizeof(int64_t), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_feat_selected_lengths, dim3((unsigned)((nsel + 255) / 256)), dim3(256), 0, 0, f->indptr.p, rows_d.p, nsel, g->indptr.p + 1);
            size_t tb = 0;
            HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tb, g->indptr.p + 1, g->indptr.p + 1, (int)nsel, 0));
            HIP_TRY(temp.alloc(tb));
            HIP_TRY(hipcub::DeviceScan::InclusiveSum(temp.p, tb, g->indptr.p + 1, g->indptr.p + 1, (int)nsel, 0));
        }
        int64_t nnz = 0;
        HIP_TRY(hipMemcpy(&nnz, g->indptr.p + nsel, sizeof(int64_t), hipMemcpyDeviceToHost));
        g->nnz = nnz;
        HIP_TRY(g->indices.alloc(nnz));
        HIP_TRY(g->data.alloc(nnz));
        if (nsel) {
            int blocks = (int)((nsel + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
            hipLaunchKernelGGL(k_feat_copy_rows, dim3(blocks), dim3(BLOCK), 0, 0, f->indptr.p, f->indices.p, f->data.p, rows_d.p, nsel, g->indptr.p,
                               g->indices.p, g->data.p);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        return 0;
    }();
    rows_d.release(); temp.release();
    if (rc) { std::string keep = g_err; arcte_hip_features_destroy(g); g_err = keep; return rc; }
    *out = g;
    return 0;
}

int arcte_hip_features_chi2_psnr_weights(arcte_hip_features *f, const int64_t *y_indptr, const int32_t *y_indices, int64_t n_classes,
                                         double *contingency_out, double *weights_out)
{
    if (!f || !y_indptr || n_classes <= 0 || !weights_out) return fail(ARCTE_HIP_EINVAL, "bad argument");
    const int64_t ny = y_indptr[f->n_rows];
    if (ny && !y_indices) return fail(ARCTE_HIP_EINVAL, "y_indices is NULL");
    for (int64_t k = 0; k < ny; k++)
        if (y_indices[k] < 0 || y_indices[k] >= n_classes) return fail(ARCTE_HIP_EINVAL, "class index out of range");
    HIP_TRY(hipSetDevice(f->device));
    DevBuf<int64_t> yp;
    DevBuf<int32_t> yi;
    DevBuf<double> m, class_count, variance, weights;
    int rc = [&]() -> int {
        HIP_TRY(yp.alloc(f->n_rows + 1));
        HIP_TRY(yi.alloc(ny));
        HIP_TRY(m.alloc(n_classes * f->n_cols));
        HIP_TRY(class_count.alloc(n_classes));
        HIP_TRY(variance.alloc(n_classes));
        HIP_TRY(weights.alloc(f->n_cols));
        HIP_TRY(hipMemcpy(yp.p, y_indptr, (f->n_rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        if (ny) HIP_TRY(hipMemcpy(yi.p, y_indices, ny * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(m.p, 0, m.bytes()));
        HIP_TRY(hipMemset(class_count.p, 0, class_count.bytes()));
        int r = column_counts(f);               // feature_count = X.sum(axis=0) of the ones pattern (:29)
        if (r) return r;
        if (f->n_rows) {
            int blocks = (int)((f->n_rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
            hipLaunchKernelGGL(k_chi2_observed, dim3(blocks), dim3(BLOCK), 0, 0, f->indptr.p, f->indices.p, yp.p, yi.p, f->n_rows, f->n_cols, m.p,
                               class_count.p);
        }
        const int64_t cells = n_classes * f->n_cols;
        hipLaunchKernelGGL(k_chi2_statistic, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, 0, class_count.p, f->col_count.p, f->n_rows, n_classes,
                           f->n_cols, m.p);
        hipLaunchKernelGGL(k_psnr_row_variance, dim3((unsigned)n_classes), dim3(256), 0, 0, m.p, f->n_cols, variance.p);
        hipLaunchKernelGGL(k_psnr_weights, dim3((unsigned)((f->n_cols + 255) / 256)), dim3(256), 0, 0, m.p, n_classes, f->n_cols, variance.p, weights.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        if (contingency_out) HIP_TRY(hipMemcpy(contingency_out, m.p, cells * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(weights_out, weights.p, f->n_cols * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    }();
    yp.release(); yi.release(); m.release(); class_count.release(); variance.release(); weights.release();
    return rc;
}

int arcte_hip_peak_snr_weights(int device, int64_t n_classes, int64_t n_cols, const double *contingency, double *weights_out)
{
    if (n_classes <= 0 || n_cols <= 0 || !contingency || !weights_out) return fail(ARCTE_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(device));
    DevBuf<double> m, variance, weights;
    int rc = [&]() -> int {
        HIP_TRY(m.alloc(n_classes * n_cols));
        HIP_TRY(variance.alloc(n_classes));
        HIP_TRY(weights.alloc(n_cols));
        HIP_TRY(hipMemcpy(m.p, contingency, n_classes * n_cols * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_psnr_row_variance, dim3((unsigned)n_classes), dim3(256), 0, 0, m.p, n_cols, variance.p);
        hipLaunchKernelGGL(k_psnr_weights, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, 0, m.p, n_classes, n_cols, variance.p, weights.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(weights_out, weights.p, n_cols * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    }();
    m.release(); variance.release(); weights.release();
    return rc;
}

int arcte_hip_stream_bandwidth(int device, int64_t bytes, double *read_gbps, double *copy_gbps)
{
    if (bytes < (1 << 20) || !read_gbps || !copy_gbps) return fail(ARCTE_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const int64_t n16 = bytes / 16;
    DevBuf<uint4> a, b;
    DevBuf<unsigned long long> sink;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = [&]() -> int {
        HIP_TRY(a.alloc(n16));
        HIP_TRY(b.alloc(n16));
        HIP_TRY(sink.alloc(1));
        HIP_TRY(hipMemset(a.p, 1, a.bytes()));
        HIP_TRY(hipMemset(b.p, 0, b.bytes()));
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        const int blocks = prop.multiProcessorCount * 8;
        double best_r = 0, best_c = 0;
        for (int rep = 0; rep < 4; rep++) {
            float ms = 0;
            HIP_TRY(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, 0, a.p, n16, sink.p);
            HIP_TRY(hipEventRecord(e1, 0));
            HIP_TRY(hipEventSynchronize(e1));
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best_r = std::max(best_r, (double)n16 * 16 / (ms * 1e-3) / 1e9);
            HIP_TRY(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, 0, a.p, b.p, n16);
            HIP_TRY(hipEventRecord(e1, 0));
            HIP_TRY(hipEventSynchronize(e1));
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best_c = std::max(best_c, (double)n16 * 32 / (ms * 1e-3) / 1e9);
        }
        HIP_TRY(hipGetLastError());
        *read_gbps = best_r;
        *copy_gbps = best_c;
        return 0;
    }();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    a.release(); b.release(); sink.release();
    return rc;
}

int arcte_hip_trim(void)
{
    free_parked_buffers();
    std::lock_guard<std::mutex> lock(g_big_mutex);
    for (auto &e : g_big_cache) {
        (void)hipSetDevice(e.device);
        (void)hipFree(e.p);
    }
    g_big_cache.clear();
    return 0;
}

int arcte_hip_has_ab_builds(void)
{
#ifdef ARCTE_HIP_AB_BUILDS
    return 1;
#else
    return 0;
#endif
}

int arcte_hip_memory_info(int device, int64_t info[4])
{
    if (!info) return fail(ARCTE_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(device));
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    info[0] = (int64_t)parked_bytes_on(device);
    info[1] = (int64_t)cached_bytes_on(device);
    info[2] = (int64_t)free_b;
    info[3] = (int64_t)total_b;
    return 0;
}

int arcte_hip_set_float32(arcte_hip_ctx *c, int enable)
{
    if (!c) return fail(ARCTE_HIP_EINVAL, "ctx is NULL");
    c->float32 = enable ? 1 : 0;
    return 0;
}

int arcte_hip_launch_occupancy(arcte_hip_ctx *c, int *workgroups_per_cu)
{
    if (!c || !workgroups_per_cu) return fail(ARCTE_HIP_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->lines && !c->float32) {
        const size_t lds = (size_t)lines_hot_values(c) * sizeof(double) + c->l_M / 8;
        int per_cu = 0;
        if (c->pack && c->l_waves_per_cu > 12) {
            auto k4 = (c->l_MB > 0 && c->l_ind) ? k_arcte_lines<0, 0, 2, true, false, 1, 4, false, true>
                                                : (c->l_MB > 0 ? k_arcte_lines<0, 0, 2, true, false, 1, 4> : k_arcte_lines<0, 0, 2, false, false, 1, 4>);
            if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k4, WAVE, lds));
            *workgroups_per_cu = per_cu;
            return 0;
        }
        auto kernel = c->pack ? (c->l_MB > 0 ? k_arcte_lines<0, 0, 2, true, false, 1> : k_arcte_lines<0, 0, 2, false, false, 1>)
                      : c->narrow ? (c->l_MB > 0 ? k_arcte_lines<0, 0, 1, true, false, 1> : k_arcte_lines<0, 0, 1, false, false, 1>)
                                  : (c->l_MB > 0 ? k_arcte_lines<0, 0, 0, true, false, 1> : k_arcte_lines<0, 0, 0, false, false, 1>);
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE, lds));
        *workgroups_per_cu = per_cu;
        return 0;
    }
    const uint32_t k = hot_values_per_wave(c, sizeof(double));
    const size_t lds = (size_t)c->waves_per_block * k * sizeof(double);
    int per_cu = 0;
    if (k == 0) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (k_arcte_seeds<0, 0, double, 2, false>), c->waves_per_block * WAVE, lds));
    else {
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_arcte_seeds<0, 0, double, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (k_arcte_seeds<0, 0, double, 2, true>), c->waves_per_block * WAVE, lds));
    }
    *workgroups_per_cu = per_cu;
    return 0;
}

int arcte_hip_info(arcte_hip_ctx *c, int64_t info[10])
{
    if (!c || !info) return fail(ARCTE_HIP_EINVAL, "bad argument");
    if (c->lines && !c->float32) {
        info[0] = c->l_slots;
        info[1] = c->l_qcap;
        info[2] = (int64_t)c->device_bytes();
        info[3] = c->cus;
        info[4] = 1;
        info[5] = lines_hot_values(c);
        info[6] = (c->narrow && !c->pack && !c->l_ind && (c->tiles == 4 || c->tiles == 2)) ? c->tiles : 1;      // (what launch_lines_v launches)
        info[7] = c->l_waves_per_cu;
        info[8] = c->pack ? 2 : (c->narrow ? 1 : 0);
        info[9] = 0;
        return 0;
    }
    info[0] = c->slots;
    info[1] = c->qcap;
    info[2] = (int64_t)c->device_bytes();
    info[3] = c->cus;
    info[4] = c->waves_per_block;
    info[5] = hot_values_per_wave(c, c->float32 ? sizeof(float) : sizeof(double));
    info[6] = c->tiles;
    info[7] = c->waves_per_cu;
    info[8] = (c->narrow && !c->float32) ? 1 : 0;
    info[9] = c->warm_k2;
    return 0;
}

int arcte_hip_state_info(arcte_hip_ctx *c, int64_t info[14])
{
    if (!c || !info) return fail(ARCTE_HIP_EINVAL, "bad argument");
    const bool lines = c->lines && !c->float32;
    info[0] = lines ? 1 : 0;
    info[1] = lines ? (int64_t)c->l_M : 0;
    info[2] = lines ? (int64_t)c->l_pcap : 0;
    info[3] = lines ? (int64_t)c->l_scap : 0;
    info[4] = (int64_t)(lines ? c->slot_bytes_lines() : c->slot_bytes_dense());
    info[5] = lines ? (int64_t)(c->l_M / 8) : 0;
    info[6] = lines ? (int64_t)lines_lds_per_wave(c) : 0;
    info[7] = lines ? (int64_t)c->l_MB : 0;
    for (int i = 0; i < 4; i++) info[8 + i] = c->line_stats[i];
    info[12] = (lines && c->l_ind) ? 1 : 0;
    info[13] = (lines && c->l_ind) ? (int64_t)c->l_pool : 0;
    return 0;
}

int arcte_hip_placement_info(arcte_hip_ctx *c, int *kept, double *rates, int capacity, int *drawn)
{
    if (!c || !drawn || (capacity > 0 && !rates)) return fail(ARCTE_HIP_EINVAL, "bad argument");
    *drawn = (int)c->placement_probe.size();
    if (kept) *kept = c->placement_kept;
    for (int i = 0; i < capacity && i < *drawn; i++) rates[i] = c->placement_probe[(size_t)i];
    return 0;
}

}  // extern "C"
