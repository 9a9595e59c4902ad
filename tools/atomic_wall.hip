// Microbenchmark, round 4: is a memory-side float64 atomic add (one request, the old value returned) a cheaper re-touch of a
// state line than the load + store read-modify-write the push kernel does (a read request, later a write-back)?
//
//   atomic_wall MODE BLIND_PCT SLICE_MB WAVES_PER_CU [ITERS]
//     MODE       0 load + store (today)   1 returning atomic add   2 atomic add without return
//     BLIND_PCT  share of the updates that are blind quad-cooperative whole-line writes (the push kernel: 65)
// Private slice per wavefront, results consumed one iteration later (software pipeline).  Prints G updates/s.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(64) void k(double *tab, uint64_t values_per_wave, uint32_t blind_cut, int iters, double *sink)
{
    const uint64_t wave = blockIdx.x;
    const int lane = threadIdx.x;
    double *base = tab + wave * values_per_wave;
    double acc = 0, pending = 0;
    double *pending_at = base + lane;
    bool pending_valid = false;
    for (int it = 0; it < iters; it++) {
        const uint64_t h = mix((wave * 64 + lane) * 1315423911ULL + it);
        const uint64_t j = (h >> 16) % values_per_wave;
        const bool blind = (uint32_t)(h & 0xFFFF) < blind_cut;
        double *at = base + j;
        const int ql = lane & 3;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int src = (lane & ~3) | c;
            const unsigned long long jj = __shfl((unsigned long long)j, src, 64);
            const int bl = __shfl((int)blind, src, 64);
            if (bl) {
                double2 *line = reinterpret_cast<double2 *>(base + (jj & ~7ull));
                const int chunk = (int)((jj & 7) >> 1);
                const double p = (double)it;
                line[ql] = (ql == chunk) ? ((jj & 1) ? make_double2(0.0, p) : make_double2(p, 0.0)) : make_double2(0.0, 0.0);
            }
        }
        double got = 0;
        if (MODE == 0) {
            got = *(blind ? base + lane : at);
            if (pending_valid) *pending_at = pending + 1.0;
            pending = got;
            pending_at = at;
            pending_valid = !blind;
        } else if (MODE == 1) {
            acc += pending;                                    // last iteration's returned value is consumed here
            if (!blind) pending = __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)at, 1.0);
        } else {
            if (!blind) __hip_atomic_fetch_add(at, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (MODE == 0 && pending_valid) *pending_at = pending + 1.0;
    acc += pending;
    if (acc == 12345.678) sink[0] = acc;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: atomic_wall MODE BLIND_PCT SLICE_MB WAVES_PER_CU [ITERS]\n"); return 2; }
    const int mode = atoi(argv[1]), blind_pct = atoi(argv[2]);
    const double slice_mb = atof(argv[3]);
    const int wpc = atoi(argv[4]);
    const int iters = argc > 5 ? atoi(argv[5]) : 512;
    const int waves = 256 * wpc;
    const uint64_t vpw = (uint64_t)(slice_mb * 1048576.0 / 8.0) & ~7ull;
    const size_t bytes = (size_t)waves * vpw * 8;
    double *tab, *sink;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    hipMemset(tab, 0, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    const uint32_t cut = (uint32_t)(65536.0 * blind_pct / 100.0);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, tab, vpw, cut, iters, sink);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, tab, vpw, cut, iters, sink);
        else hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, tab, vpw, cut, iters, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    static const char *names[3] = {"load + store", "returning atomic add", "atomic add, no return"};
    printf("%-22s blind %3d%%  slice %6.2f MB (%6.1f GB in all)  %2d waves/CU : %7.2f G updates/s  (%.2f ms)\n", names[mode], blind_pct, slice_mb,
           bytes / 1073741824.0, wpc, (double)waves * 64 * iters / best / 1e6, best);
    return 0;
}
