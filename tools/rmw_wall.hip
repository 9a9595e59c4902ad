// Microbenchmark: rate of random 32-byte-entry read-modify-writes vs footprint (what bounds the eps-push kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
struct __attribute__((aligned(32))) E { double a, b, c; uint64_t t; };
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int MODE>   // 0: read 32B + write 16B (RMW)   1: read only 32B   2: read 16B + write 16B   3: write-only 32B   4: 43% reads + write 32B
__global__ void k(E *tab, uint64_t n, int iters, double *sink)
{
    uint64_t id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    double acc = 0;
    for (int it = 0; it < iters; it++) {
        uint64_t j = mix(id * 1315423911ULL + it) % n;
        double2 lo = make_double2(1, 2);
        double2 hi = make_double2(0, 0);
        const bool rd = MODE < 3 || (MODE == 4 && (mix(j) % 100) < 43);
        if (rd) lo = *reinterpret_cast<double2 *>(tab + j);
        if (rd && MODE != 2) hi = *(reinterpret_cast<double2 *>(tab + j) + 1);
        acc += lo.x + hi.x;
        if (MODE != 1) *reinterpret_cast<double2 *>(tab + j) = make_double2(lo.x + 1.0, lo.y + hi.x);
        if (MODE >= 3) *(reinterpret_cast<double2 *>(tab + j) + 1) = make_double2(lo.y, (double)it);
    }
    if (acc == 12345.678) sink[0] = acc;
}
int main(int argc, char **argv)
{
    double *sink; hipMalloc(&sink, 8);
    size_t sizes_mb[] = {128, 16384, 131072};
    for (size_t smb : sizes_mb) {
        uint64_t n = smb * 1024ull * 1024ull / 32;
        E *tab; if (hipMalloc(&tab, n * 32) != hipSuccess) { printf("alloc fail %zu MB\n", smb); continue; }
        hipMemset(tab, 0, n * 32);
        for (int mode = 0; mode < 5; mode++) {
            for (int waves_per_cu : {4, 16}) {
                int blocks = 256 * waves_per_cu / 4, iters = 64;
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(a);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks * 8), dim3(256), 0, 0, tab, n, iters, sink);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks * 8), dim3(256), 0, 0, tab, n, iters, sink);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks * 8), dim3(256), 0, 0, tab, n, iters, sink);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks * 8), dim3(256), 0, 0, tab, n, iters, sink);
                    if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks * 8), dim3(256), 0, 0, tab, n, iters, sink);
                    hipEventRecord(b); hipEventSynchronize(b);
                }
                float ms; hipEventElapsedTime(&ms, a, b);
                double ops = (double)blocks * 8 * 256 * iters;
                printf("footprint %6zu MB mode %d (%s) grid %5d blocks: %.2f G accesses/s\n", smb, mode,
                       mode == 0 ? "rd32+wr16" : mode == 1 ? "rd32" : mode == 2 ? "rd16+wr16" : mode == 3 ? "wr32 only" : "43% rd32 + wr32", blocks * 8, ops / ms / 1e6);
                fflush(stdout);
            }
        }
        hipFree(tab);
    }
    return 0;
}
