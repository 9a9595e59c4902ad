// Calibration of FETCH_SIZE / WRITE_SIZE for THIS access pattern (MI355X_MICROARCH.md: "calibrate on a known byte
// count in your own access pattern"): N random 32-byte-entry gathers (two 16-B loads per lane), optionally with a
// 16-B store back, over a 16 GB table, one kernel launch.  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((aligned(32))) E { double a, b, c; uint64_t t; };
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int WRITE>
__global__ void k_calib(E *tab, uint64_t n, int iters, double *sink)
{
    uint64_t id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    double acc = 0;
    for (int it = 0; it < iters; it++) {
        uint64_t j = mix(id * 1315423911ULL + it) % n;
        double2 lo = *reinterpret_cast<double2 *>(tab + j);
        double2 hi = *(reinterpret_cast<double2 *>(tab + j) + 1);
        acc += lo.x + hi.x;
        if (WRITE) *reinterpret_cast<double2 *>(tab + j) = make_double2(lo.x + 1.0, lo.y + hi.x);
    }
    if (acc == 12345.678) sink[0] = acc;
}
int main()
{
    double *sink; hipMalloc(&sink, 8);
    uint64_t n = 16ull * 1024 * 1024 * 1024 / 32;
    E *tab; hipMalloc(&tab, n * 32); hipMemset(tab, 0, n * 32);
    const int blocks = 16384, iters = 64;
    hipLaunchKernelGGL(k_calib<0>, dim3(blocks), dim3(256), 0, 0, tab, n, iters, sink);
    hipLaunchKernelGGL(k_calib<1>, dim3(blocks), dim3(256), 0, 0, tab, n, iters, sink);
    hipDeviceSynchronize();
    printf("accesses per launch: %llu (each: 32 B read%s)\n", (unsigned long long)blocks * 256 * iters, ", second kernel also 16 B write");
    return 0;
}
