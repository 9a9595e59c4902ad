// The per-seed propagation + extraction kernel of round 3 ("line state"), gfx950 / wave64.  Included by arcte_hip.hip
// after arcte_kernels.hpp (shares its helpers, PushParams, QEntry, SeedStatus).
//
// Same algorithm, same operation order per location as k_arcte_seeds (similarity.py:149-222, push.py:41-64,
// arcte.py:352-376) -- what changes is WHERE a seed's state lives, because that is what the kernel is bound by.
// Measured (tools/line_wall.hip, tools/line_study.py): the chip moves ~40-55 G random 64-byte requests per second whatever
// their size or direction; a read-modify-write of a cold entry costs two of them, and 80 % of a seed's cold updates are
// FIRST touches of their node -- the read fetches a stale line only to learn that the node is untouched.  A first touch
// that knows it is one writes blind; whole lines only (partial writes halve the rate), and the four lanes of a quad per
// line (one store instruction = whole lines: 41.7 G mixed updates/s against 22.3 with four stores per lane).
//
//   * Nodes are named by RANK (descending pattern in-count) inside the kernel.  The row streams carry ranks
//     (edge_rank), the per-node arrays are indexed by rank (rowspan, in_degree_r), results are translated back
//     (ranked_ids) when they are emitted.
//   * LDS level: ranks < K keep their one value (r == s until the node is pushed) on chip, as before.
//   * Line level: every other rank < 8 M has ONE float64 in a per-slot array, laid out STRIDED: value index =
//     (rank mod M) * 8 + rank div M, so the eight nodes of a 64-byte line are M ranks apart -- one frequently touched
//     node shares its line with seven rarely touched ones.  A bitmap of M bits in LDS (zeroed per seed: it IS the
//     reference's s[:] = 0; r[:] = 0, arcte.py:337-338) says which lines the seed has touched.  The first touch of a line
//     is a BLIND whole-line write (the deposit in its place, zeros in the seven others): no read, one full-line
//     request.  The arbiter is an LDS atomic OR, so two lanes that meet in an untouched line are ordered without a
//     memory round trip: one writes the line, the other reads it afterwards.  Only re-touches of a line are
//     read-modify-writes.  No epoch tags, no 32-byte entries: 8 bytes per node and slot.
//   * Pushed nodes (and the seed) need r and s apart: they move to a compact per-slot array PS[j] = {r, s}, j = order
//     of first push, and leave a NaN whose payload is j in their value's place (LDS or line level alike).  A seed
//     pushes ~190 distinct nodes: 3 KB that stay in the L2.
//   * Ranks beyond the LDS bitmap's reach (>= 8 M: "region B", template flag TAIL) have the same lines with their
//     touched-bits in a per-slot bitmap in global memory, claimed by a returning atomic one pipeline stage earlier.
//   * The row walk is a software pipeline of 64-edge steps (rows i+4, claims i+3, slot values i+2, pushed-state i+1,
//     add/store/enqueue i): every stage issues its loads unconditionally, and ONE vmcnt(0) at the top of a turn waits for
//     what the previous turn issued -- the counter is in order and counts stores, which the compiler cannot count under
//     a branch, so a wait anywhere else would drain the turn's own loads too.
//   * Occupancy decides: one tile per step keeps the kernel at 149 VGPRs = three wavefronts per SIMD (12 per CU).
//   * Rows (template parameter ROWS): 0 = wide (rank 4 + weight 8 + in_degree 8 bytes per edge), 1 = narrow (one weight per
//     row: rank 4 + float32 in_degree 4), 2 = PACKED (round 4): ONE 32-bit word per edge, the rank in its low rank_bits bits
//     and the target's in_degree -- an integer on graphs with narrow rows -- above; the all-ones code sends the lane to a
//     float32 table by rank instead (the few hundred highest-ranked nodes: L2-resident).  The kernel sits at the memory
//     system's request wall: half the row stream is 4 % fewer requests.
#pragma once

#include "arcte_kernels.hpp"

namespace {

constexpr int32_t ST_PUSHED_OVERFLOW = 6;   // more distinct pushed nodes than PS holds: re-run with a larger one
constexpr int32_t ST_SUP_OVERFLOW = 7;      // more candidates than the list holds
constexpr int32_t ST_POOL_OVERFLOW = 8;     // more touched lines of region B than its pool holds (indirect lines)

struct LineParams {
    const uint32_t *edge_rank;    // [nnz] rank of every stored edge's target (CSR order = FIFO order is kept); ROWS == 2: the packed words
    const float *in_degree_rf;    // [n] by rank, float32 (ROWS == 2: in_degrees the packed word has no room for)
    uint32_t rank_bits;           // ROWS == 2: bits of the packed word that hold the rank
    const uint32_t *node_rank;    // [n]
    const int32_t *ranked_ids;    // [n] rank -> node
    const int64_t *rowspan;       // [2n] by rank: first and one-past-last edge of the node's row
    const double *in_degree_r;    // [n] by rank
    double *vals;                 // region A: 8 M values per slot
    double *vals_b;               // region B: 8 MB values per slot, addressed with region A's index space (lines M ..)
    double2 *ps;                  // [slots][pcap]
    int32_t *sup;                 // [slots][scap] candidate list (ranks)
    uint32_t M, Mshift;           // region A: lines per slot whose touched-bits are in LDS (power of two), log2; ranks < 8 M
    uint32_t MB, MBshift;         // region B (TAIL): lines for the ranks >= 8 M, touched-bits in gbm (0: every rank is in A)
    uint32_t *gbm;                // [slots][MB / 32]
    // indirect region B (template flag IND): a line of region B lives in a line of a per-slot POOL (vals_b), its place is
    // kept in bidx[line] = seed generation << 32 | ~pool line; bgen[slot] is the last generation the slot used
    uint64_t *bidx;               // [slots][MB]
    uint32_t *bgen;               // [slots]
    uint32_t pool_cap;            // pool lines per slot
    int64_t bidx_stride;
    // distance between two slots' parts of the arrays above, in elements of each: a slot's often touched parts (region A,
    // ring, candidates, pushed state, region B's bits) lie in ONE block whose size is a power of two, region B's values in
    // another (arcte_hip.hip: lines_layout)
    int64_t vals_stride, valsb_stride, ps_stride, sup_stride, q_stride, gbm_stride;

    uint32_t pcap, scap;
    uint32_t K;                   // values of the LDS level
    unsigned long long *lstats;   // [0] LDS updates [1] blind line writes [2] read-modify-writes [3] updates of pushed nodes
    // debug entry arcte_hip_seed_state (one seed per launch): dense s[n] and r[n] by NODE id, gathered from every level of
    // the state when the FIFO has run dry -- what similarity.py:149-222 leaves with its caller.  nullptr: nothing is dumped
    double *dump_s, *dump_r;
};

__device__ __forceinline__ double moved_to(uint32_t j) { return __longlong_as_double((long long)(0x7FF8DEAD00000000ull | (uint64_t)j)); }
__device__ __forceinline__ bool moved_is(double x) { return (uint32_t)((uint64_t)__double_as_longlong(x) >> 32) == 0x7FF8DEADu; }
__device__ __forceinline__ uint32_t moved_index(double x) { return (uint32_t)(uint64_t)__double_as_longlong(x); }

// lane (quad base + C)'s value in all four lanes of every quad (DPP quad_perm: no LDS crossbar); every lane must be active
template <int C> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, C * 0x55, 0xF, 0xF, true);
}

// Blind write of whole lines, one store instruction per lane of the quad: in round C the four lanes of a quad write the
// four 16-byte chunks of the line that the quad's lane C owns (when it owns one), so every store instruction carries
// whole, contiguous 64-byte lines and the memory pipeline sees ONE full-line request per line -- 41.7 G mixed updates/s
// against 22.3 when each lane writes its own line with four 16-byte stores (tools/line_wall.hip, shapes 1 / 0).
template <int C>
__device__ __forceinline__ void blind_round(double *vals, double *vals_b, uint32_t RA, int lane, uint32_t own, uint32_t index, uint32_t plo, uint32_t phi)
{
    const uint32_t o = quad_bcast<C>(own);
    const uint32_t ix = quad_bcast<C>(index);          // value index of the owner's node: line = ix / 8, place = ix % 8
    const double p = __hiloint2double((int)quad_bcast<C>(phi), (int)quad_bcast<C>(plo));
    if (o) {
        const uint32_t ql = (uint32_t)lane & 3u, s_ = ix & 7u;
        const bool hit = ql == (s_ >> 1);
        reinterpret_cast<double2 *>((ix < RA ? vals : vals_b) + (size_t)(ix & ~7u))[ql] = make_double2((hit && !(s_ & 1)) ? p : 0.0, (hit && (s_ & 1)) ? p : 0.0);
    }
}

// registers of the pipeline stages, LT 64-edge tiles per step (narrow rows: the weight is the row's, the in_degree a float)
template <int LT, int ROWS> struct LRowT { bool a[LT]; uint32_t v[LT]; double w[LT], d[LT]; };
template <int LT> struct LRowT<LT, 1> { bool a[LT]; uint32_t v[LT]; float d[LT]; };
template <int LT> struct LRowT<LT, 2> { bool a[LT]; uint32_t v[LT]; };          // v: rank | in_degree code << rank_bits
template <int LT, bool PACK> struct LSlotT { double x[LT]; bool owner[LT]; uint32_t ix[LT]; };
template <int LT> struct LSlotT<LT, true> { double x[LT]; bool owner[LT]; uint32_t ix[LT]; float dq[LT]; };   // dq: in_degree from the table (escaped lanes)
template <int LT> struct LPushedT { double2 q[LT]; };
template <int LT> struct LClaimT { uint32_t old[LT]; uint64_t bi[LT]; uint32_t base; };   // IND: old = the lane's candidate, bi = the entry before its claim, base = first candidate of the step

// TAIL: the graph has more ranks than the LDS bitmap covers (8 M); the others' lines have their touched-bits in a
// per-slot bitmap in global memory (L2-resident for graphs of a few million nodes), claimed by a returning atomic OR one
// pipeline stage before the line is written or read.
// IND (with TAIL): region B's lines are INDIRECT.  Region B is as large as the graph (8 bytes per node and slot) but a seed
// touches a few thousand of its lines, so a line lives in a line of a small per-slot POOL and an 8-byte ENTRY per line says
// where: bidx[line] = generation << 32 | ~pool line, valid when it carries the generation of the seed in hand.
// Round 4 (profiles/r03/region_b_study.txt: 97 % of region B's updates are the first touch of their line): the CLAIM hands out
// the place.  Every lane that meets region B takes the next candidate pool line (a wavefront-private counter: ballot + mbcnt)
// and makes ONE returning atomic max of `generation << 32 | ~candidate` on the entry: an entry of an older generation loses
// (the lane owns the line and writes its candidate blind), an entry of this generation wins over every later candidate
// (candidates grow, their complements shrink: the entry keeps the FIRST claim's line, whichever lane's atomic arrives first)
// and tells the lane where the line is.  No touched-bit, no entry read ahead, no entry write, nothing to clear per seed; a
// non-owner's candidate stays unused (3 % of the claims).  Two lanes of ONE 64-edge step that meet in one line (rare) cannot
// know which atomic arrived first: whoever sees a place out of this step's candidates calls for the step's region-B lanes
// to read their entries again when the atomics are done, and the entry decides.  4 MB of entries per million nodes + the pool
// instead of 8 MB of values per million nodes; region A, the LDS level and the order of every floating-point operation are
// unchanged.
template <int MODE, int VAR, int ROWS, bool TAIL, bool PROF = false, int LT = 1, int WPE = 1, bool STAGE = false, bool IND = false>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(WPE))) void k_arcte_lines(PushParams P, LineParams L)
{
    static_assert(!IND || (TAIL && !STAGE), "indirect lines are region B's");
    static_assert(ROWS >= 0 && ROWS <= 2, "wide, narrow or packed rows");
    constexpr bool NARROW = ROWS != 0, PACK = ROWS == 2;
    typedef LRowT<LT, ROWS> LRow;
    typedef LSlotT<LT, PACK> LSlot;
    typedef LPushedT<LT> LPushed;
    typedef LClaimT<LT> LClaim;
    static_assert(MODE == 0 || MODE == 2, "worker or centrality");
    // STAGE (A/B of BASELINE.json's "rows staged through LDS", ARCTE_HIP_STAGE_ROWS=1): the row data of the steps in flight
    // (rank + float32 in_degree, 8 bytes per edge) go global -> LDS directly (global_load_lds), five 512-byte stages behind
    // the bitmap, and are read back when a stage needs them, instead of living in VGPRs across the turns
    static_assert(!STAGE || (ROWS == 1 && LT == 1), "row staging exists for one-tile steps of narrow rows");
    // PROF (ARCTE_HIP_PROFILE=1): s_memtime ticks per phase, the indices of PushParams::prof
    unsigned long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&]() -> unsigned long long { return PROF ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ULL; };
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x;
    const int64_t slot = blockIdx.x;
    const GraphDev &g = P.g;
    const uint32_t K = L.K;
    const uint32_t Mmask = L.M - 1, Mshift = L.Mshift;
    double *hot = reinterpret_cast<double *>(lds_raw);
    uint32_t *bm = reinterpret_cast<uint32_t *>(hot + K);
    uint32_t *ring = bm + (L.M >> 5);                 // STAGE: [5 stages][rank[64] | in_degree[64]]
    const uint32_t RA = L.M << 3;                    // first rank of region B
    const uint32_t MBmask = L.MB - 1, MBshift = L.MBshift;
    double *__restrict__ vals = L.vals + slot * L.vals_stride;
    double *__restrict__ vals_b = L.vals_b + slot * L.valsb_stride;
    uint32_t *__restrict__ gbm = L.gbm + slot * L.gbm_stride;
    uint64_t *__restrict__ bidx = IND ? L.bidx + slot * L.bidx_stride : nullptr;
    uint32_t gen = 0;                                // IND: generation of the seed in hand (bidx entries of other seeds are stale)
    if (IND) gen = L.bgen[slot];
    uint32_t npool = 0;                              // IND: pool lines taken by the seed in hand
    double2 *__restrict__ ps = L.ps + slot * L.ps_stride;
    int32_t *__restrict__ sup = L.sup + slot * L.sup_stride;
    QEntry *__restrict__ q = P.queue + slot * L.q_stride;
    const uint32_t qmask = P.qcap - 1;
    const double omr = P.one_minus_rho;

    auto next_work = [&]() -> unsigned long long {
        __builtin_amdgcn_wave_barrier();          // see k_arcte_seeds: keeps the draw out of jump threading's reach
        unsigned long long w = 0;
        if (lane == 0) w = atomicAdd(P.work_counter, 1ULL);
        return bcast_u64(w);
    };
    // packed rows: the rank is the low rank_bits bits of the word, the in_degree code the rest (all ones: see the table)
    const uint32_t rbits = PACK ? L.rank_bits : 0u, rmask = PACK ? ((1u << rbits) - 1u) : 0xFFFFFFFFu;
    const uint32_t desc = PACK ? (0xFFFFFFFFu >> rbits) : 0u;
    auto rk_of = [&](uint32_t v) -> uint32_t { if constexpr (PACK) return v & rmask; else return v; };
    auto in_b = [&](uint32_t rk) -> bool { return TAIL && rk >= RA; };
    auto value_index = [&](uint32_t rk) -> uint32_t {
        if (in_b(rk)) {
            const uint32_t rp = rk - RA;
            if constexpr (IND) return RA + (((0xFFFFFFFFu - (uint32_t)bidx[rp & MBmask]) << 3) | (rp >> MBshift));      // (of a line this seed has claimed)
            else return RA + (((rp & MBmask) << 3) | (rp >> MBshift));
        }
        return ((rk & Mmask) << 3) | (rk >> Mshift);
    };
    auto val_at = [&](uint32_t ix) -> double * { return (ix < RA ? vals : vals_b) + ix; };     // ix = line * 8 + place; region B's lines follow A's
    auto line_touched = [&](uint32_t rk) -> bool {
        if (in_b(rk)) {
            const uint32_t ln = (rk - RA) & MBmask;
            if constexpr (IND) return (uint32_t)(bidx[ln] >> 32) == gen;          // claimed by the seed in hand
            else return (gbm[ln >> 5] >> (ln & 31)) & 1u;
        }
        const uint32_t ln = rk & Mmask;
        return (bm[ln >> 5] >> (ln & 31)) & 1u;
    };
    // the value that stands in a node's place: on chip, in its line, or 0 when the line has not been touched
    auto raw_value = [&](uint32_t rk) -> double {
        if (rk < K) return hot[rk];
        return line_touched(rk) ? *val_at(value_index(rk)) : 0.0;
    };
    // a node that HAS a value (it was deposited to): no look at the bitmap
    auto live_value = [&](uint32_t rk) -> double { return rk < K ? hot[rk] : *val_at(value_index(rk)); };

    unsigned long long drawn = 0;
    unsigned long long c_lds = 0, c_blind = 0, c_rmw = 0, c_moved = 0;
    unsigned long long t_mark = tick();
    for (unsigned long long wk = next_work(); wk < (unsigned long long)P.nwork && drawn <= (unsigned long long)P.nwork;
         wk = next_work(), drawn++) {
        if (PROF) { const unsigned long long t = tick(); prof[9] += t - t_mark; t_mark = t; }
        const int32_t pos = P.work_pos ? P.work_pos[wk] : (int32_t)wk;
        const int32_t seed = P.seeds[pos];
        const uint32_t sr = L.node_rank[seed];
        const double eps = P.eps[pos];
        {
            unsigned long long cur = 0;
            if (lane == 0) cur = __hip_atomic_load(P.raw_cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cur = bcast_u64(cur);
            if (cur > P.rawcap) {          // the arena is full: the host drains it and runs this seed again
                if (lane == 0) {
                    P.status[pos] = ST_OUTPUT_OVERFLOW;
                    P.out_cnt[pos] = 0;
                    P.out_off[pos] = 0;
                    P.nop[pos] = 0;
                    atomicAdd(&P.stats[4], 1ULL);
                }
                continue;
            }
        }
        // s[:] = 0; r[:] = 0 (arcte.py:337-338): the on-chip values and the touched-line bitmap
        for (uint32_t i = lane; i < K; i += WAVE) hot[i] = 0.0;
        {
            uint64_t *bm64 = reinterpret_cast<uint64_t *>(bm);
            for (uint32_t i = lane; i < (L.M >> 6); i += WAVE) bm64[i] = 0;
            if (TAIL && !IND) {
                uint4 *g4 = reinterpret_cast<uint4 *>(gbm);
                for (uint32_t i = lane; i < (L.MB >> 7); i += WAVE) g4[i] = make_uint4(0, 0, 0, 0);
            }
        }

        if (IND) { gen++; npool = 0; }
        uint32_t head = 0, tail = 0;
        int32_t nsup = 0, nfirst = 0;
        double cand_thr = 0.0;
        int32_t npush = 0;
        uint32_t npushed = 0;          // entries of PS in use
        unsigned long long nedges = 0;
        bool ok = true, runaway = false;
        int32_t fail_status = ST_QUEUE_OVERFLOW;
        unsigned long long s_lds = 0, s_blind = 0, s_rmw = 0, s_moved = 0;   // this seed's updates by kind (counted when it completes)

        double last_rnew[LT];          // r of every target of the step processed last, as the push left it (one-step rows: forwarded to waiting entries)
#pragma unroll
        for (int t = 0; t < LT; t++) last_rnew[t] = 0.0;
        auto load_row = [&](int64_t base, int64_t re, double w_row, LRow &R) {
#pragma unroll
            for (int t = 0; t < LT; t++) {
                const int64_t k = base + t * WAVE + lane;
                R.a[t] = k < re;
                const int64_t kk = R.a[t] ? k : re - 1;
                R.v[t] = L.edge_rank[kk];
                if constexpr (PACK) {}
                else if constexpr (NARROW) R.d[t] = g.edge_in_degree_f[kk];
                else { R.w[t] = g.data[kk]; R.d[t] = g.edge_in_degree[kk]; }
            }
        };

        // ---- push.py:60-64 over the edges [rb, re) of u's row + the ordered enqueue of similarity.py:194-196 / :214-216
        auto walk = [&](uint32_t u, double c, double r_self, bool s_self_known, double s_self, int64_t rb, int64_t re, double w_row,
                        bool do_enqueue, bool use_pre, const LRow &pre) __attribute__((always_inline)) {
            // stage 2: claim untouched lines (LDS atomic OR: exactly one lane per line sees the bit clear), write them
            // blind -- 0 + p == p, so the line is complete at once -- and load the values of the touched ones
            // stage 1 (TAIL): region B's lines are claimed in the global bitmap; the answer is looked at one turn later
            auto claim = [&](const LRow &R, LClaim &C) {
#pragma unroll
                for (int t = 0; t < LT; t++) {
                    C.old[t] = 0;
                    C.bi[t] = 0;
                    const bool isb = TAIL && R.a[t] && rk_of(R.v[t]) >= RA;
                    if constexpr (IND) {
                        // the claim hands out the place: every lane offers the next candidate pool line, the entry keeps the
                        // first claim of this generation (see the kernel's header)
                        const uint64_t mb = __ballot(isb);
                        if (t == 0) C.base = npool;
                        uint32_t cand = npool + lane_below(mb);
                        npool += (uint32_t)__popcll(mb);
                        if (npool > L.pool_cap) { ok = false; fail_status = ST_POOL_OVERFLOW; }
                        if (cand >= L.pool_cap) cand = L.pool_cap - 1;          // (the seed fails; nothing may be written outside the pool)
                        C.old[t] = cand;
                        if (isb) {
                            const uint32_t ln = (rk_of(R.v[t]) - RA) & MBmask;
                            C.bi[t] = atomicMax(reinterpret_cast<unsigned long long *>(&bidx[ln]),
                                                ((unsigned long long)gen << 32) | (unsigned long long)(0xFFFFFFFFu - cand));
                        }
                    } else if (isb) {
                        const uint32_t ln = (rk_of(R.v[t]) - RA) & MBmask;
                        C.old[t] = atomicOr(&gbm[ln >> 5], 1u << (ln & 31));
                    }
                }
            };
            auto slots = [&](const LRow &R, LSlot &E, const LClaim &C, bool skip_if_none) {
                uint32_t ld_index[LT];
                bool any_load = false;
#pragma unroll
                for (int t = 0; t < LT; t++) {
                    const uint32_t rk = rk_of(R.v[t]);
                    const bool line_lvl = R.a[t] && rk >= K;
                    bool owner = false;
                    if (TAIL && line_lvl && rk >= RA) {
                        const uint32_t ln = (rk - RA) & MBmask;
                        if constexpr (IND) owner = (uint32_t)(C.bi[t] >> 32) != gen;      // the entry was an older seed's: the line is this lane's
                        else owner = !((C.old[t] >> (ln & 31)) & 1u);
                    } else if (line_lvl) {
                        const uint32_t ln = rk & Mmask;
                        const uint32_t bit = 1u << (ln & 31);
                        const uint32_t old = atomicOr(&bm[ln >> 5], bit);
                        owner = !(old & bit);
                    }
                    E.owner[t] = owner;
                    uint32_t index;
                    if constexpr (IND) {
                        const bool isb = line_lvl && rk >= RA;
                        const uint32_t rp = rk - RA, ln = rp & MBmask;
                        // owner: its own candidate; otherwise the place the entry named
                        uint32_t pl = owner ? C.old[t] : 0xFFFFFFFFu - (uint32_t)C.bi[t];
                        // A place out of THIS step's candidates: another lane of the step met the same line, and which of the two
                        // atomics arrived first is not ours to know -- the entry is (it keeps the smaller candidate): the step's
                        // region-B lanes read their entries again, now that the atomics are done, and the entry decides.
                        const bool again = isb && !owner && pl >= C.base;
                        if (__ballot(again)) {
                            __builtin_amdgcn_s_waitcnt(0x0F70);
                            if (isb) {
                                pl = 0xFFFFFFFFu - (uint32_t)bidx[ln];
                                owner = pl == C.old[t];
                            }
                            __builtin_amdgcn_s_waitcnt(0x0F70);
                            E.owner[t] = owner;
                        }
                        index = isb ? RA + ((pl << 3) | (rp >> MBshift)) : (((rk & Mmask) << 3) | (rk >> Mshift));
                        E.ix[t] = index;
                    } else {
                        index = value_index(rk);
                    }
                    {
                        double wt;
                        if constexpr (NARROW) wt = w_row; else wt = R.w[t];
                        const double p = c * wt;
                        const uint32_t own = owner ? 1u : 0u;
                        const uint32_t plo = (uint32_t)__double2loint(p), phi = (uint32_t)__double2hiint(p);
                        blind_round<0>(vals, vals_b, RA, lane, own, index, plo, phi);
                        blind_round<1>(vals, vals_b, RA, lane, own, index, plo, phi);
                        blind_round<2>(vals, vals_b, RA, lane, own, index, plo, phi);
                        blind_round<3>(vals, vals_b, RA, lane, own, index, plo, phi);
                    }
                    // (every lane issues the load: the others at a cached address of the slot)
                    ld_index[t] = (line_lvl && !owner) ? index : 0u;
                    any_load |= line_lvl && !owner;
                    E.x[t] = 0.0;
                    if constexpr (PACK) {
                        // an in_degree the packed word has no room for comes from the float32 table by rank (the highest
                        // ranks: a few KB, cached); the lookup rides with the value loads of this stage
                        E.dq[t] = 0.0f;
                        any_load |= R.a[t] && (R.v[t] >> rbits) == desc;
                    }
                }
                // (skip_if_none: a row of one step waits for these loads at once; when no lane has a value to read --
                //  first touches and on-chip nodes only -- there is nothing to wait for: a wave-uniform branch)
                if (skip_if_none && __ballot(any_load) == 0) return;
#pragma unroll
                for (int t = 0; t < LT; t++) {
                    E.x[t] = *val_at(ld_index[t]);
                    if constexpr (PACK) E.dq[t] = L.in_degree_rf[(R.a[t] && (R.v[t] >> rbits) == desc) ? rk_of(R.v[t]) : 0u];
                }
            };
            // stage 3: the value in the node's place; a pushed node's NaN points into PS
            // (skip_if_none: a row of one step has nothing to overlap the load with, so it is only issued when some
            //  target has been pushed -- a wave-uniform branch)
            auto pushed = [&](const LRow &R, LSlot &E, LPushed &Q, bool skip_if_none) {
                bool any = false;
#pragma unroll
                for (int t = 0; t < LT; t++) {
                    const uint32_t rk = rk_of(R.v[t]);
                    double x = 0.0;
                    if (R.a[t]) x = (rk < K) ? hot[rk] : (E.owner[t] ? 0.0 : E.x[t]);
                    E.x[t] = x;
                    any |= moved_is(x);
                    Q.q[t] = make_double2(0.0, 0.0);
                }
                if (skip_if_none && __ballot(any) == 0) return;
#pragma unroll
                for (int t = 0; t < LT; t++) Q.q[t] = ps[moved_is(E.x[t]) ? moved_index(E.x[t]) : 0u];
            };
            // stage 4
            auto process = [&](const LRow &R, const LSlot &E, const LPushed &Q) {
#pragma unroll
                for (int t = 0; t < LT; t++) {
                    const bool act = R.a[t];
                    const uint32_t rk = rk_of(R.v[t]);
                    double dv;
                    if constexpr (PACK) { const uint32_t dc = R.v[t] >> rbits; dv = (dc == desc) ? (double)E.dq[t] : (double)dc; }
                    else dv = (double)R.d[t];
                    const double x = E.x[t];
                    const bool mv = act && moved_is(x);
                    double wt;
                    if constexpr (NARROW) wt = w_row; else wt = R.w[t];
                    const double p = c * wt;                                       // push.py:62 / :17 / :38
                    const double r_old = mv ? ((rk != u) ? Q.q[t].x : r_self) : x;   // a self-loop sees r[u] as just set
                    const double s_old = mv ? ((rk == u && s_self_known) ? s_self : Q.q[t].y) : ((VAR == 0) ? x : 0.0);
                    const double r_new = r_old + p;                                // push.py:64
                    last_rnew[t] = r_new;
                    const double s_new = (VAR == 0) ? s_old + p : s_old;           // push.py:63 (ARCTE only)
                    if (act) {
                        if (mv) ps[moved_index(x)] = make_double2(r_new, s_new);
                        else if (rk < K) hot[rk] = r_new;
                        else if (!E.owner[t]) *val_at(IND ? E.ix[t] : value_index(rk)) = r_new;
                    }
                    s_moved += __popcll(__ballot(mv));
                    s_lds += __popcll(__ballot(act && !mv && rk < K));
                    s_blind += __popcll(__ballot(act && E.owner[t]));
                    s_rmw += __popcll(__ballot(act && !mv && rk >= K && !E.owner[t]));
                    if (VAR == 0) {
                        // candidate list (see k_arcte_seeds): nodes whose s/in_degree has reached a lower bound of the final threshold
                        const double bar = cand_thr * dv;
                        const bool cross = act && (s_new > 0.0 && s_new >= bar) && !(s_old > 0.0 && s_old >= bar);
                        const uint64_t mc = __ballot(cross);
                        if (mc) {
                            if ((uint32_t)nsup + (uint32_t)__popcll(mc) > L.scap) { ok = false; fail_status = ST_SUP_OVERFLOW; }
                            else {
                                if (cross) sup[nsup + lane_below(mc)] = (int32_t)rk;
                                nsup += __popcll(mc);
                            }
                        }
                        nfirst += __popcll(__ballot(act && s_old == 0.0 && s_new != 0.0));
                    }
                    if (!do_enqueue) continue;
                    const bool enq = act && (r_new / dv >= eps);                   // similarity.py:194/214
                    const uint64_t me = __ballot(enq);
                    const uint32_t cnt = __popcll(me);
                    if (cnt) {
                        if (tail - head + cnt > P.qcap) { ok = false; fail_status = ST_QUEUE_OVERFLOW; }
                        else {
                            if (enq) {
                                QEntry e;
                                e.v = (int32_t)rk; e.h = 0; e.d = dv;
                                q[(tail + lane_below(me)) & qmask] = e;
                            }
                            tail += cnt;
                        }
                    }
                }
            };
            constexpr int64_t STEP = LT * WAVE;
            if (STAGE && re - rb > STEP) {
                if constexpr (STAGE) {
                    // the row data of step j lives in ring stage j % 5 from the turn after it was requested
                    auto request = [&](int64_t base, int stage) {
                        const int64_t k = base + lane;
                        const int64_t kk = k < re ? k : re - 1;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(L.edge_rank + kk),
                                                         (__attribute__((address_space(3))) void *)(ring + stage * 128), 4, 0, 0);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g.edge_in_degree_f + kk),
                                                         (__attribute__((address_space(3))) void *)(ring + stage * 128 + 64), 4, 0, 0);
                    };
                    auto fetch = [&](int64_t base, int stage, LRow &R) {
                        R.a[0] = base + lane < re;
                        R.v[0] = ring[stage * 128 + lane];
                        R.d[0] = __uint_as_float(ring[stage * 128 + 64 + lane]);
                    };
                    LRow R0, R1, R2, R3;
                    LSlot E0, E1, E2;
                    LPushed Q0, Q1;
                    LClaim C0, C1, C2, C3;
                    for (int j = 0; j < 4; j++) request(rb + j * STEP, j);
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    fetch(rb, 0, R0); fetch(rb + STEP, 1, R1); fetch(rb + 2 * STEP, 2, R2);
                    claim(R0, C0); claim(R1, C1); claim(R2, C2);
                    slots(R0, E0, C0, false);
                    slots(R1, E1, C1, false);
                    pushed(R0, E0, Q0, false);
                    int st = 0;                                    // ring stage of the step being processed
                    for (int64_t base = rb; base < re; base += STEP) {
                        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the row data requested last turn is in LDS
                        fetch(base, st, R0);
                        fetch(base + STEP, (st + 1) % 5, R1);
                        fetch(base + 2 * STEP, (st + 2) % 5, R2);
                        fetch(base + 3 * STEP, (st + 3) % 5, R3);
                        pushed(R1, E1, Q1, false);
                        process(R0, E0, Q0);
                        slots(R2, E2, C2, false);
                        claim(R3, C3);
                        request(base + 4 * STEP, (st + 4) % 5);
                        if (!ok) break;
                        E0 = E1; E1 = E2; Q0 = Q1; C2 = C3;
                        st = (st + 1) % 5;
                    }
                }
            } else if (re - rb > STEP) {
                LRow R0, R1, R2, R3, R4;
                LSlot E0, E1, E2;
                LPushed Q0, Q1;
                LClaim C0, C1, C2, C3;
                load_row(rb, re, w_row, R0);
                load_row(rb + STEP, re, w_row, R1);
                load_row(rb + 2 * STEP, re, w_row, R2);
                if (TAIL) load_row(rb + 3 * STEP, re, w_row, R3);
                claim(R0, C0);
                claim(R1, C1);
                claim(R2, C2);
                slots(R0, E0, C0, false);
                slots(R1, E1, C1, false);
                pushed(R0, E0, Q0, false);
                for (int64_t base = rb; base < re; base += STEP) {
                    // Everything this turn consumes was issued in the previous one: wait for all of it HERE, once, before
                    // anything new is issued.  (The counter is in order and counts stores too; the compiler cannot count
                    // stores under a branch, so a wait placed after them would drain this turn's loads as well.)
                    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
                    pushed(R1, E1, Q1, false);
                    process(R0, E0, Q0);
                    slots(R2, E2, C2, false);
                    if (TAIL) {
                        claim(R3, C3);
                        load_row(base + 4 * STEP, re, w_row, R4);
                    } else {
                        load_row(base + 3 * STEP, re, w_row, R3);
                    }
                    if (!ok) break;
                    R0 = R1; R1 = R2; R2 = R3; E0 = E1; E1 = E2; Q0 = Q1;
                    if (TAIL) { R3 = R4; C2 = C3; }
                }
            } else if (re > rb) {
                LRow R0;
                LSlot E0;
                LPushed Q0;
                LClaim C0;
                if (use_pre) R0 = pre;               // fetched ahead, during the round trip of the previous push's re-test
                else load_row(rb, re, w_row, R0);
                claim(R0, C0);
                slots(R0, E0, C0, true);
                pushed(R0, E0, Q0, true);
                process(R0, E0, Q0);
            }
        };

        // ---- one push of node u (push.py:41-64).  `ru` is r[u] at pop time, `ju` its place in PS (-1: not pushed yet,
        //      its one value stands for r == s)
        auto push = [&](uint32_t u, int32_t ju, double ru, int64_t rb, int64_t re, bool do_enqueue, bool use_pre, const LRow &pre, double pre_w) {
            const unsigned long long t_push = tick();
            double c, r_self, s_self = 0.0;
            bool s_known = false;
            if (VAR == 0) { c = omr * ru; r_self = 0.0; }                                                  // push.py:56,59
            else if (VAR == 1) { c = omr * ru; r_self = 0.0; }                                             // push.py:11,15
            else { c = omr * (1 - P.lazy) * ru; r_self = omr * P.lazy * (ru); }                            // push.py:30-31
            const double A = P.rho * ru;                                                                   // push.py:10 / :29
            if (ju < 0) {
                // first push: the node moves to PS (s[u] == r[u] == ru for ARCTE, s[u] == 0 for the PageRank flavours)
                if (npushed >= L.pcap) { ok = false; fail_status = ST_PUSHED_OVERFLOW; return; }
                ju = (int32_t)npushed++;
                s_self = (VAR == 0) ? ru : 0.0 + A;                                                        // push.py:14 / :34
                s_known = true;
                if (lane == 0) {
                    ps[ju] = make_double2(r_self, s_self);
                    if (u < K) hot[u] = moved_to((uint32_t)ju);
                    else *val_at(value_index(u)) = moved_to((uint32_t)ju);
                }
                if (VAR != 0) {
                    const bool grew = s_self != 0.0;
                    if (grew) {
                        if ((uint32_t)nsup >= L.scap) { ok = false; fail_status = ST_SUP_OVERFLOW; return; }
                        if (lane == 0) sup[nsup] = (int32_t)u;       // s is non-zero exactly at pushed nodes
                        nsup++; nfirst++;
                    }
                }
            } else if (VAR == 0) {
                if (lane == 0) reinterpret_cast<double *>(ps + ju)[0] = 0.0;                               // push.py:59
            } else {
                const double s_old = ps[ju].y;
                s_self = s_old + A;                                                                        // push.py:14 / :34
                s_known = true;
                if (lane == 0) ps[ju] = make_double2(r_self, s_self);                                      // push.py:15 / :35
                if (s_old == 0.0 && s_self != 0.0) {
                    if ((uint32_t)nsup >= L.scap) { ok = false; fail_status = ST_SUP_OVERFLOW; return; }
                    if (lane == 0) sup[nsup] = (int32_t)u;
                    nsup++; nfirst++;
                }
            }
            const double w_row = use_pre ? pre_w : ((NARROW && re > rb) ? g.data[rb] : 0.0);
            walk(u, c, r_self, s_known, s_self, rb, re, w_row, do_enqueue, use_pre, pre);
            npush++;
            nedges += (unsigned long long)(re - rb);
            if (npush >= P.max_pushes) { ok = false; runaway = true; }
            if (PROF) {
                // (the time of a push is taken out of the phase it interrupts: t_mark moves forward by it)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                const unsigned long long dt = tick() - t_push;
                const bool is_long = re - rb > (int64_t)LT * WAVE;
                prof[is_long ? 3 : 2] += dt;
                prof[is_long ? 7 : 6] += 1;
                t_mark += dt;
            }
        };

        // ---- similarity.py:176-192: s[seed] = r[seed] = 1, one unconditional push.  The seed's state is PS[0] from the start.
        const int64_t seed_b = L.rowspan[2 * (int64_t)sr], seed_e = L.rowspan[2 * (int64_t)sr + 1];
        const double seed_d = L.in_degree_r[sr];
        if (lane == 0) {
            ps[0] = make_double2(1.0, (VAR == 0) ? 1.0 : 0.0);          // similarity.py:176-177 / :26 / :85
            if (sr < K) hot[sr] = moved_to(0);
            else {
                if (in_b(sr)) {
                    const uint32_t ln = (sr - RA) & MBmask;
                    if constexpr (IND) bidx[ln] = ((uint64_t)gen << 32) | 0xFFFFFFFFull;          // pool line 0
                    else gbm[ln >> 5] |= 1u << (ln & 31);
                }
                else { const uint32_t ln = sr & Mmask; bm[ln >> 5] |= 1u << (ln & 31); }
                const uint32_t ix = value_index(sr), sl = ix & 7u;
                double2 *line = reinterpret_cast<double2 *>(val_at(ix & ~7u));
                const double m0 = moved_to(0);
                for (uint32_t ch = 0; ch < 4; ch++)
                    line[ch] = (ch == (sl >> 1)) ? ((sl & 1) ? make_double2(0.0, m0) : make_double2(m0, 0.0)) : make_double2(0.0, 0.0);
            }
            if (VAR == 0) sup[0] = (int32_t)sr;
        }
        npushed = 1;
        if (IND && in_b(sr)) npool = 1;
        nsup = (VAR == 0) ? 1 : 0;
        nfirst = (VAR == 0) ? 1 : 0;
        if (MODE == 0 && VAR == 0) {
            // lower bound of the selection threshold (arcte.py:358-360), see k_arcte_seeds
            double lb = 1.0 / seed_d;
            const double c0 = omr * 1.0;
            for (int64_t k = seed_b + lane; k < seed_e; k += WAVE) {
                const double x = (c0 * g.data[k]) / g.edge_in_degree[k];
                lb = (x < lb) ? x : lb;
            }
            cand_thr = wave_min(lb) * cand_margin<double>();
        }
        if (PROF) { const unsigned long long t = tick(); prof[0] += t - t_mark; t_mark = t; }
        { LRow none; push(sr, 0, 1.0, seed_b, seed_e, true, false, none, 0.0); }
        if (VAR == 2) {
            // similarity.py:108-116: re-push the seed while it stays above the threshold, no enqueue
            while (ok) {
                const double ru2 = ps[0].x;
                if (!(ru2 / seed_d >= eps)) break;
                { LRow none; push(sr, 0, ru2, seed_b, seed_e, false, false, none, 0.0); }
            }
        }

        // ---- similarity.py:199-216: FIFO with duplicates, 64 entries per batch (see k_arcte_seeds)
        // r of node rk, and where it lives: its place in PS when it has been pushed
        auto read_r = [&](uint32_t rk, int32_t &j) -> double {
            const double x = live_value(rk);          // queued nodes were deposited to: they have a value
            j = -1;
            if (moved_is(x)) { j = (int32_t)moved_index(x); return ps[j].x; }
            return x;
        };
        while (ok && head != tail) {
            const uint32_t navail = tail - head;
            const uint32_t bn = navail < (uint32_t)WAVE ? navail : (uint32_t)WAVE;
            const bool valid = (uint32_t)lane < bn;
            uint32_t u_l = 0;
            int32_t j_l = -1;
            double r_l = 0.0, d_l = 1.0;
            int64_t rb_l = 0, re_l = 0;
            if (valid) {
                const QEntry e = q[(head + lane) & qmask];
                u_l = (uint32_t)e.v;
                d_l = e.d;
                r_l = read_r(u_l, j_l);
                const longlong2 sp = *reinterpret_cast<const longlong2 *>(L.rowspan + 2 * (int64_t)u_l);
                rb_l = sp.x;
                re_l = sp.y;
            }
            head += bn;
            int consumed = 0;
            bool pass = valid && (r_l / d_l >= eps);                                  // similarity.py:204
            if (PROF) {
                const unsigned long long any = __ballot(pass);        // (the ballot makes the wavefront wait for the loads of the batch)
                asm volatile("" ::"s"(any));
                const unsigned long long t = tick();
                prof[1] += t - t_mark; t_mark = t; prof[8] += 1;
            }
            bool fresh = valid;            // r_l is r as of now (no push since it was read has touched the node)
            LRow PF;                       // the row of entry pf_lane, fetched ahead (rows of one step only)
            double pf_w = 0.0;
            int pf_lane = -1;
            for (;;) {
                const uint64_t m = __ballot(pass && lane >= consumed);
                if (m == 0) break;
                const int i = __ffsll((unsigned long long)m) - 1;
                const uint32_t u = (uint32_t)__shfl((int)u_l, i, WAVE);
                const double du = shfl_f64(d_l, i);
                consumed = i + 1;
                // this entry's pop time is now (similarity.py:204): its r must be the current one -- the node may have been
                // pushed or deposited to since it was read (the queue holds duplicates)
                int32_t ju = __shfl(j_l, i, WAVE);
                double ru = shfl_f64(r_l, i);
                if (!__shfl((int)fresh, i, WAVE)) ru = read_r(u, ju);
                if (!(ru / du >= eps)) {
                    if (lane == i) pass = false;
                    continue;
                }
                const int64_t rb = shfl_i64(rb_l, i);
                const int64_t re = shfl_i64(re_l, i);
                if constexpr (NARROW) {
                    // A row of ONE step (most rows of a sparse graph; 458 pushes per seed on the 8M-node graph, nearly all of them
                    // such rows) is held here, not inside the walk: after the push its targets are compared with the waiting
                    // entries ON CHIP, so that only entries the push touched are read again.  Round 3 read every waiting entry
                    // again after every push: one more dependent round trip per push, 5.3 us per one-step push in all.
                    const bool one_step = re > rb && re - rb <= (int64_t)LT * WAVE;
                    if (one_step && pf_lane != i) {
                        pf_w = NARROW ? g.data[rb] : 0.0;
                        load_row(rb, re, pf_w, PF);
                        pf_lane = i;
                    }
                    const LRow CUR = PF;
                    const double cur_w = pf_w;
                    // the row of the entry that will probably be next (a waiting entry that passes today), fetched while this push
                    // runs: the graph does not change, so the row is right whether or not that entry is pushed in the end
                    // (wide rows -- weighted graphs -- are five registers per edge: their next row is fetched after the push, as in round 3,
                    //  or the kernel would lose a wavefront per SIMD)
                    LRow NX;
                    double nx_w = 0.0;
                    int nx_lane = -1;
                    auto fetch_next = [&]() {
                        const uint64_t mn = __ballot(pass && lane >= consumed);
                        const int nxt = mn ? __ffsll((unsigned long long)mn) - 1 : WAVE;
                        if (nxt < WAVE) {
                            const int64_t nb = shfl_i64(rb_l, nxt), ne = shfl_i64(re_l, nxt);
                            if (ne > nb && ne - nb <= (int64_t)LT * WAVE) {
                                nx_w = NARROW ? g.data[nb] : 0.0;
                                load_row(nb, ne, nx_w, NX);
                                nx_lane = nxt;
                            }
                        }
                    };
                    if constexpr (NARROW) fetch_next();
                    push(u, ju, ru, rb, re, true, one_step, CUR, cur_w);
                    if (VAR == 2) {
                        // similarity.py:136-144: re-push the same node while it stays above the threshold
                        while (ok) {
                            int32_t j2 = -1;
                            const double ru2 = read_r(u, j2);
                            if (!(ru2 / du >= eps)) break;
                            push(u, j2, ru2, rb, re, false, false, CUR, 0.0);          // (the row comes from the cache again: keeping it costs registers)
                        }
                    }
                    if (!ok) break;
                    // Which waiting entries did the push touch?  The pushed node itself (the queue holds duplicates) and the row's
                    // targets; a row of several steps is not held here: every entry counts as touched, as in round 3.
                    // A waiting entry that IS a target takes its new r from the lane that computed it (what the push stored: the value in
                    // the node's place, or PS[j].x of a pushed node) -- no memory round trip; only duplicates of the pushed node itself
                    // are read again.
                    bool touched = true;
                    if (one_step) {
                        touched = u_l == u;
                        bool hit = false;
    #pragma unroll
                        for (int t = 0; t < LT; t++) {
                            uint64_t ma = __ballot(CUR.a[t]);
                            const int lo_ = __double2loint(last_rnew[t]), hi_ = __double2hiint(last_rnew[t]);
                            while (ma) {
                                const int j = __ffsll((unsigned long long)ma) - 1;
                                ma &= ma - 1;
                                const uint32_t tr = rk_of((uint32_t)__builtin_amdgcn_readlane((int)CUR.v[t], j));
                                const int flo = __builtin_amdgcn_readlane(lo_, j), fhi = __builtin_amdgcn_readlane(hi_, j);
                                if (u_l == tr) {
                                    r_l = __hiloint2double(fhi, flo);
                                    hit = true;
                                }
                            }
                        }
                        // (an entry that was stale stays stale: its place in PS may have changed too, which only a read tells)
                        if (hit && valid) pass = r_l / d_l >= eps;
                    }
                    if (touched) fresh = false;
                    if constexpr (!NARROW) fetch_next();
                    PF = NX;
                    pf_w = nx_w;
                    pf_lane = nx_lane;
                    // One round trip reads again, when there is anything to read: (a) every touched waiting entry that did not pass
                    // -- the push may have lifted it over the threshold -- and (b) the passing entry whose turn comes next, if it
                    // was touched, so that its pop needs no second look (the other passing entries are read when their turn comes).
                    const uint64_t mp = __ballot(pass && lane >= consumed);
                    const int nxt = mp ? __ffsll((unsigned long long)mp) - 1 : WAVE;
                    const bool again = valid && lane >= consumed && !fresh && (!pass || lane == nxt);
                    if (__ballot(again)) {
                        if (again) {
                            r_l = read_r(u_l, j_l);
                            pass = r_l / d_l >= eps;
                            fresh = true;
                        }
                    }
                } else {
                    // (wide rows -- weighted graphs, five registers per edge of a row held here -- keep round 3's scheme, or the kernel
                    //  would lose a wavefront per SIMD: every waiting entry counts as touched by every push, and the next row is
                    //  fetched in the round trip that reads them again)
                    push(u, ju, ru, rb, re, true, pf_lane == i, PF, pf_w);
                    if (VAR == 2) {
                        while (ok) {
                            int32_t j2 = -1;
                            const double ru2 = read_r(u, j2);
                            if (!(ru2 / du >= eps)) break;
                            push(u, j2, ru2, rb, re, false, false, PF, 0.0);
                        }
                    }
                    if (!ok) break;
                    fresh = false;
                    const uint64_t mp = __ballot(pass && lane >= consumed);
                    const int nxt = mp ? __ffsll((unsigned long long)mp) - 1 : WAVE;
                    pf_lane = -1;
                    if (nxt < WAVE) {
                        const int64_t nb = shfl_i64(rb_l, nxt), ne = shfl_i64(re_l, nxt);
                        if (ne > nb && ne - nb <= (int64_t)LT * WAVE) {
                            pf_w = 0.0;
                            load_row(nb, ne, pf_w, PF);
                            pf_lane = nxt;
                        }
                    }
                    if (valid && lane >= consumed && (!pass || lane == nxt)) {
                        r_l = read_r(u_l, j_l);
                        pass = r_l / d_l >= eps;
                        fresh = true;
                    }
                }
            }
        }

        if (L.dump_s) {
            // arcte_hip_seed_state: on-chip values, lines of region A and B (0 where the seed touched nothing) and the
            // pushed-state array, as the dense vectors of the reference
            for (int64_t rk = lane; rk < g.n; rk += WAVE) {
                const double x = raw_value((uint32_t)rk);
                double rv = x, sv = (VAR == 0) ? x : 0.0;
                if (moved_is(x)) { const double2 e = ps[moved_index(x)]; rv = e.x; sv = e.y; }
                const int32_t v = L.ranked_ids[rk];
                L.dump_r[v] = rv;
                L.dump_s[v] = sv;
            }
        }
        if (PROF) { const unsigned long long t = tick(); prof[4] += t - t_mark; t_mark = t; }
        // ---- arcte.py:352-376: degree-normalise, threshold = min over the closed neighbourhood, select, emit
        auto s_of = [&](uint32_t rk) -> double {
            const double x = raw_value(rk);
            if (moved_is(x)) return ps[moved_index(x)].y;
            return (VAR == 0) ? x : 0.0;
        };
        int32_t sta = ok ? ST_OK : (runaway ? ST_RUNAWAY : fail_status);
        int32_t emitted = 0, support = 0;
        const int32_t ncand = nsup;
        unsigned long long off = 0;
        if (ok) {
            const double s_seed = ps[0].y;
            double thr = s_seed / seed_d;
            bool miss = s_seed == 0.0, selfloop = false;
            for (int64_t k = seed_b + lane; k < seed_e; k += WAVE) {
                const uint32_t rk = rk_of(L.edge_rank[k]);
                selfloop |= (rk == sr);
                const double sv = s_of(rk);
                miss |= (sv == 0.0);
                const double x = sv / g.edge_in_degree[k];
                thr = (x < thr) ? x : thr;
            }
            thr = wave_min(thr);
            const bool missing = __ballot(miss) != 0;
            const bool any_selfloop = __ballot(selfloop) != 0;
            unsigned long long coff = 0;
            bool contribute = false;
            if (MODE == 2) {
                if (lane == 0) coff = atomicAdd(P.contrib_cursor, (unsigned long long)nsup);
                coff = bcast_u64(coff);
                contribute = coff + (unsigned long long)nsup <= P.contrib_cap;
                if (!contribute) sta = ST_CONTRIB_OVERFLOW;
            }
            const int64_t base_size = (MODE == 2) ? (seed_e - seed_b) + (any_selfloop ? 0 : 1) : (seed_e - seed_b) + 1;
            if (MODE != 2 && VAR == 0 && missing) sta = ST_MISSING_BASE;
            else if (VAR != 0 && (missing || any_selfloop)) {
                support = nfirst;                    // arcte.py:129-133
            } else {
                int32_t cnt = 0;
                for (int32_t i0 = 0; i0 < nsup; i0 += WAVE) {
                    const int32_t i = i0 + lane;
                    bool sel = false;
                    int32_t rk = 0;
                    if (i < nsup) {
                        rk = sup[i];
                        const double sv = s_of((uint32_t)rk);
                        const double dv = L.in_degree_r[rk];
                        const double xn = sv / dv;
                        sel = xn >= thr;                                          // arcte.py:363-367
                        if (MODE == 2) {
                            sel = sel && !missing;
                            if (contribute) {
                                P.contrib_key[coff + i] = ((uint64_t)(uint32_t)L.ranked_ids[rk] << P.contrib_shift) | (uint64_t)(seed - P.contrib_seed_base);
                                P.contrib_val[coff + i] = xn;
                            }
                        }
                    }
                    const uint64_t ms = __ballot(sel);
                    if (sel) sup[cnt + lane_below(ms)] = rk;                  // in place: cnt <= i0
                    cnt += __popcll(ms);
                }
                support = nfirst;
                if ((int64_t)cnt > base_size && sta == ST_OK) {                   // arcte.py:370 / arcte.pyx:211
                    if (lane == 0) off = atomicAdd(P.raw_cursor, (unsigned long long)cnt);
                    off = bcast_u64(off);
                    if (off + (unsigned long long)cnt > P.rawcap) sta = ST_OUTPUT_OVERFLOW;
                    else {
                        for (int32_t i = lane; i < cnt; i += WAVE) P.raw[off + i] = L.ranked_ids[sup[i]];
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
                c_lds += s_lds; c_blind += s_blind; c_rmw += s_rmw; c_moved += s_moved;
                atomicAdd(&P.stats[0], (unsigned long long)npush);
                atomicAdd(&P.stats[1], nedges);
                atomicAdd(&P.stats[2], (unsigned long long)tail);
                atomicAdd(&P.stats[3], (unsigned long long)support);
                atomicAdd(&P.stats[5], (unsigned long long)ncand);
            } else {
                atomicAdd(&P.stats[4], 1ULL);
            }
        }
        if (PROF) { const unsigned long long t = tick(); prof[5] += t - t_mark; t_mark = t; }
    }
    if (PROF && lane == 0 && P.prof) {
#pragma unroll
        for (int k = 0; k < 10; k++) atomicAdd(P.prof + k, prof[k]);
    }
    if (IND && lane == 0) L.bgen[slot] = gen;
    if (lane == 0 && L.lstats) {
        atomicAdd(L.lstats + 0, c_lds);
        atomicAdd(L.lstats + 1, c_blind);
        atomicAdd(L.lstats + 2, c_rmw);
        atomicAdd(L.lstats + 3, c_moved);
    }
}

// ---- placement probe: how fast are random 8-byte read-modify-writes into the slots' often touched region? --------
// One wavefront per slot, `iters` updates per lane inside the first `span` bytes of the slot's block -- the access
// pattern of the push kernel's line level, a couple of milliseconds.  The rate depends on how the allocation is laid
// over the physical memory (arcte_hip.hip: lines_layout), so the library can compare candidates before it settles.
__global__ __launch_bounds__(WAVE) void k_probe_slots(char *base, int64_t stride_bytes, uint32_t span_values, int iters, unsigned long long *sink)
{
    double *v = reinterpret_cast<double *>(base + (int64_t)blockIdx.x * stride_bytes);
    uint64_t x = ((uint64_t)blockIdx.x * WAVE + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
    double acc = 0.0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
        double *p = v + (uint32_t)(x >> 20) % span_values;
        const double a = *p;
        *p = a + 1.0;
        acc += a;
    }
    if (acc == 12345.678) atomicAdd(sink, 1ULL);
}

// ---- rank space: the per-node arrays by rank, the rank of every edge's target ---------------------------------
__global__ void k_node_rank(const int32_t *ranked_ids, int64_t n, uint32_t *node_rank)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) node_rank[ranked_ids[i]] = (uint32_t)i;
}

__global__ void k_rank_space(const int32_t *ranked_ids, const int64_t *indptr, const double *in_degree, int64_t n, int64_t *rowspan,
                             double *in_degree_r)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = ranked_ids[i];
    rowspan[2 * i] = indptr[v];
    rowspan[2 * i + 1] = indptr[v + 1];
    in_degree_r[i] = in_degree[v];
}

// ---- packed rows (ROWS == 2): rank | in_degree code << rank_bits per edge; the all-ones code = "look the in_degree up"
__device__ __forceinline__ uint32_t degree_code(double d, uint32_t rank_bits)
{
    const uint32_t esc = 0xFFFFFFFFu >> rank_bits;
    return (d >= 0.0 && d < (double)esc && d == floor(d)) ? (uint32_t)d : esc;
}

// float32 in_degree by rank + which ranks the packed word cannot carry: stats[0] their number, stats[1] one past the last
__global__ void k_in_degree_rf(const double *in_degree_r, int64_t n, uint32_t rank_bits, float *out, unsigned long long *stats)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = in_degree_r[i];
    out[i] = (float)d;
    if (degree_code(d, rank_bits) == (0xFFFFFFFFu >> rank_bits)) {
        atomicAdd(stats + 0, 1ULL);
        atomicMax(stats + 1, (unsigned long long)(i + 1));
    }
}

__global__ void k_edge_pack(const int32_t *indices, const uint32_t *node_rank, const double *in_degree_r, uint32_t rank_bits,
                            uint32_t *edge_pack, int64_t nnz)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const uint32_t rk = node_rank[indices[k]];
    edge_pack[k] = rk | (degree_code(in_degree_r[rk], rank_bits) << rank_bits);
}

__global__ void k_edge_rank(const int32_t *indices, const uint32_t *node_rank, uint32_t *edge_rank, int64_t nnz)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) edge_rank[k] = node_rank[indices[k]];
}

}  // namespace
