// Is the random-update rate a property of WHERE in HBM a buffer lands?  Draws N physical chunks (hipMemCreate), measures
// the random read-modify-write rate of every chunk by itself, then composes three tables of K chunks each -- the K
// fastest, the K slowest, the first K drawn -- in fresh virtual ranges and measures those.
//   chunk_probe [N chunks] [chunk MB] [K chunks per table]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// one private slice per wavefront, 32-byte entries: read the entry, write its first 16 bytes (the push kernel's pattern)
__global__ __launch_bounds__(64) void k(char *tab, uint64_t entries_per_wave, int iters, double *sink)
{
    const uint64_t id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    char *base = tab + blockIdx.x * entries_per_wave * 32;
    double acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint64_t j = mix(id * 1315423911ULL + it) % entries_per_wave;
        double2 *e = reinterpret_cast<double2 *>(base + j * 32);
        const double2 a = e[0], b = e[1];
        acc += a.x + b.x;
        e[0] = make_double2(a.x + 1.0, a.y + b.y);
    }
    if (acc == 12345.678) sink[0] = acc;
}

static double rate(char *tab, size_t bytes, double *sink)
{
    const int waves = 1536, iters = 512;
    const uint64_t epw = bytes / waves / 32;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, tab, epw, iters, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0) best = std::min(best, ms);
    }
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return (double)waves * 64 * iters / best / 1e6;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 40;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 4096) << 20;
    const int kk = argc > 3 ? atoi(argv[3]) : 14;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    double *sink;
    CHECK(hipMalloc(&sink, 8));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    std::vector<double> r(n);
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, chunk, 0, nullptr, 0));
    for (int i = 0; i < n; i++) {
        CHECK(hipMemCreate(&h[i], chunk, &prop, 0));
        CHECK(hipMemMap(va, chunk, 0, h[i], 0));
        CHECK(hipMemSetAccess(va, chunk, &acc, 1));
        CHECK(hipMemset(va, 0, chunk));
        r[i] = rate((char *)va, chunk, sink);
        CHECK(hipMemUnmap(va, chunk));
        printf("chunk %2d: %6.2f G updates/s\n", i, r[i]);
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return r[a] > r[b]; });
    auto compose = [&](const char *what, std::vector<int> pick) {
        void *t = nullptr;
        CHECK(hipMemAddressReserve(&t, chunk * pick.size(), 0, nullptr, 0));
        for (size_t i = 0; i < pick.size(); i++) CHECK(hipMemMap((char *)t + i * chunk, chunk, 0, h[pick[i]], 0));
        CHECK(hipMemSetAccess(t, chunk * pick.size(), &acc, 1));
        const double x = rate((char *)t, chunk * pick.size(), sink);
        printf("table of the %-12s %zu chunks (%zu GB): %6.2f G updates/s\n", what, pick.size(), chunk * pick.size() >> 30, x);
        CHECK(hipMemUnmap(t, chunk * pick.size()));
        CHECK(hipMemAddressFree(t, chunk * pick.size()));
    };
    compose("fastest", std::vector<int>(order.begin(), order.begin() + kk));
    compose("slowest", std::vector<int>(order.end() - kk, order.end()));
    std::vector<int> first(kk);
    for (int i = 0; i < kk; i++) first[i] = i;
    compose("first drawn", first);
    compose("fastest", std::vector<int>(order.begin(), order.begin() + kk));
    for (int i = 0; i < n; i++) CHECK(hipMemRelease(h[i]));
    // the same table from plain hipMalloc, and from chunks of other sizes
    for (int rep = 0; rep < 3; rep++) {
        char *t;
        CHECK(hipMalloc(&t, chunk * kk));
        CHECK(hipMemset(t, 0, chunk * kk));
        printf("hipMalloc table (%zu GB): %6.2f G updates/s\n", chunk * kk >> 30, rate(t, chunk * kk, sink));
        CHECK(hipFree(t));
    }
    for (size_t cm : {2, 64, 1024, 4096}) {
        const size_t c2 = cm << 20, total = chunk * kk / c2 * c2;
        void *t = nullptr;
        CHECK(hipMemAddressReserve(&t, total, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hh(total / c2);
        for (size_t i = 0; i < hh.size(); i++) {
            CHECK(hipMemCreate(&hh[i], c2, &prop, 0));
            CHECK(hipMemMap((char *)t + i * c2, c2, 0, hh[i], 0));
        }
        CHECK(hipMemSetAccess(t, total, &acc, 1));
        CHECK(hipMemset(t, 0, total));
        printf("table of %zu MB chunks (%zu GB): %6.2f G updates/s\n", cm, total >> 30, rate((char *)t, total, sink));
        CHECK(hipMemUnmap(t, total));
        for (auto &x : hh) CHECK(hipMemRelease(x));
        CHECK(hipMemAddressFree(t, total));
    }
    return 0;
}
