// Microbenchmark, round 2: what bounds random entry updates on MI355X, by entry size, write shape and footprint.
//
//   rmw_wall2 ENTRY MODE TABLE_MB WAVES_PER_CU PRIVATE STREAM [ITERS]
//     ENTRY   16 | 32 | 64 | 128 bytes per entry, entries aligned to their size
//     MODE    0 read the whole entry
//             1 write the whole entry (no read)
//             2 read the whole entry, write its first 16 bytes          (the push kernel's per-edge pattern)
//             3 read the whole entry, write the whole entry
//             4 read the first 16 bytes only
//     TABLE_MB   total footprint
//     PRIVATE 0: every lane draws from the whole table; 1: every wavefront draws from its own TABLE/waves slice
//     STREAM  bytes per access of coalesced streaming reads issued beside the random accesses (0, 16, 32): the CSR rows
//
// Prints G accesses/s.  Run under rocprofv3 --pmc for TCC_EA0_RDREQ/WRREQ and TCC_HIT/MISS.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int ENTRY, int MODE, int STREAM>
__global__ __launch_bounds__(256) void k(char *tab, uint64_t entries_per_domain, int private_domains, const double2 *stream,
                                         uint64_t stream_elems, int iters, double *sink)
{
    const uint64_t id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t wave = id >> 6;
    char *base = tab + (private_domains ? wave * entries_per_domain * ENTRY : 0);
    double acc = 0;
    uint64_t spos = id;
    constexpr int Q = ENTRY / 16;
    for (int it = 0; it < iters; it++) {
        const uint64_t j = mix(id * 1315423911ULL + it) % entries_per_domain;
        double2 *e = reinterpret_cast<double2 *>(base + j * ENTRY);
        double2 v[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) v[q] = make_double2(1.0, 2.0);
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int q = 0; q < Q; q++) v[q] = e[q];
        }
        if (MODE == 4) v[0] = e[0];
        if (STREAM) {
#pragma unroll
            for (int q = 0; q < STREAM / 16; q++) {
                const double2 t = stream[spos % stream_elems];
                spos += (uint64_t)gridDim.x * blockDim.x;
                acc += t.x;
            }
        }
#pragma unroll
        for (int q = 0; q < Q; q++) acc += v[q].x;
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int q = 0; q < Q; q++) e[q] = make_double2(v[q].x + 1.0, v[q].y + (double)it);
        }
        if (MODE == 2) e[0] = make_double2(v[0].x + 1.0, v[0].y + v[Q - 1].x);
    }
    if (acc == 12345.678) sink[0] = acc;
}

template <int ENTRY, int MODE>
void launch_s(int stream_b, dim3 g, char *tab, uint64_t epd, int priv, const double2 *st, uint64_t se, int iters, double *sink)
{
    if (stream_b == 0) hipLaunchKernelGGL((k<ENTRY, MODE, 0>), g, dim3(256), 0, 0, tab, epd, priv, st, se, iters, sink);
    else if (stream_b == 16) hipLaunchKernelGGL((k<ENTRY, MODE, 16>), g, dim3(256), 0, 0, tab, epd, priv, st, se, iters, sink);
    else hipLaunchKernelGGL((k<ENTRY, MODE, 32>), g, dim3(256), 0, 0, tab, epd, priv, st, se, iters, sink);
}

template <int ENTRY>
void launch_m(int mode, int stream_b, dim3 g, char *tab, uint64_t epd, int priv, const double2 *st, uint64_t se, int iters, double *sink)
{
    switch (mode) {
    case 0: launch_s<ENTRY, 0>(stream_b, g, tab, epd, priv, st, se, iters, sink); break;
    case 1: launch_s<ENTRY, 1>(stream_b, g, tab, epd, priv, st, se, iters, sink); break;
    case 2: launch_s<ENTRY, 2>(stream_b, g, tab, epd, priv, st, se, iters, sink); break;
    case 3: launch_s<ENTRY, 3>(stream_b, g, tab, epd, priv, st, se, iters, sink); break;
    default: launch_s<ENTRY, 4>(stream_b, g, tab, epd, priv, st, se, iters, sink); break;
    }
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: rmw_wall2 ENTRY MODE TABLE_MB WAVES_PER_CU PRIVATE STREAM [ITERS]\n"); return 2; }
    const int entry = atoi(argv[1]), mode = atoi(argv[2]);
    const size_t table_mb = (size_t)atoll(argv[3]);
    const int wpc = atoi(argv[4]), priv = atoi(argv[5]), stream_b = atoi(argv[6]);
    const int iters = argc > 7 ? atoi(argv[7]) : 256;
    const int blocks = 256 * wpc / 4;
    const uint64_t waves = (uint64_t)blocks * 4;
    const size_t bytes = table_mb << 20;
    uint64_t epd = bytes / entry;
    if (priv) epd = bytes / waves / entry;
    if (epd == 0) { fprintf(stderr, "table too small\n"); return 2; }
    char *tab;
    double *sink;
    double2 *st;
    const uint64_t stream_elems = ((size_t)4 << 30) / 16;      // 4 GB of rows to stream
    // RMW_ALLOC=contiguous: physically contiguous table (hipDeviceMallocContiguous) -- does the random-access rate depend
    // on how the allocation is backed?
    const char *how = getenv("RMW_ALLOC");
    hipError_t te = hipSuccess;
    if (how && how[0] == 'v') {
        // RMW_ALLOC=vmm[:ALIGN_MB[:CHUNK_MB]]: a virtual range of chosen alignment, backed by physical chunks of chosen size
        size_t align_mb = 1024, chunk_mb = 1024;
        sscanf(how, "vmm:%zu:%zu", &align_mb, &chunk_mb);
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        te = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        const size_t chunk = (chunk_mb << 20) / gran * gran;
        const size_t total = (bytes + chunk - 1) / chunk * chunk;
        void *va = nullptr;
        if (te == hipSuccess) te = hipMemAddressReserve(&va, total, align_mb << 20, nullptr, 0);
        for (size_t off = 0; te == hipSuccess && off < total; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            te = hipMemCreate(&h, chunk, &prop, 0);
            if (te == hipSuccess) te = hipMemMap((char *)va + off, chunk, 0, h, 0);
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (te == hipSuccess) te = hipMemSetAccess(va, total, &acc, 1);
        tab = (char *)va;
        fprintf(stderr, "vmm: granularity %zu, chunk %zu MB, va %p\n", gran, chunk >> 20, va);
    } else if (how && how[0] == 'c') te = hipExtMallocWithFlags((void **)&tab, bytes, hipDeviceMallocContiguous);
    else te = hipMalloc(&tab, bytes);
    if (te != hipSuccess) { fprintf(stderr, "table allocation failed: %s\n", hipGetErrorString(te)); return 1; }
    fprintf(stderr, "table at %p\n", (void *)tab);
    if (hipMalloc(&sink, 8) != hipSuccess || hipMalloc(&st, stream_elems * 16) != hipSuccess) {
        fprintf(stderr, "alloc failed\n");
        return 1;
    }
    hipMemset(tab, 0, bytes);
    hipMemset(st, 0, stream_elems * 16);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        if (entry == 16) launch_m<16>(mode, stream_b, dim3(blocks), tab, epd, priv, st, stream_elems, iters, sink);
        else if (entry == 32) launch_m<32>(mode, stream_b, dim3(blocks), tab, epd, priv, st, stream_elems, iters, sink);
        else if (entry == 64) launch_m<64>(mode, stream_b, dim3(blocks), tab, epd, priv, st, stream_elems, iters, sink);
        else launch_m<128>(mode, stream_b, dim3(blocks), tab, epd, priv, st, stream_elems, iters, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    const double ops = (double)waves * 64 * iters;
    static const char *names[] = {"read entry", "write entry", "read entry + write 16 B", "read entry + write entry", "read 16 B"};
    printf("entry %3d B  %-26s table %6zu MB  %2d waves/CU  %s  stream %2d B/access : %7.2f G accesses/s  (%.2f ms)\n", entry,
           names[mode], table_mb, wpc, priv ? "private slices" : "shared table  ", stream_b, ops / best / 1e6, best);
    return 0;
}
