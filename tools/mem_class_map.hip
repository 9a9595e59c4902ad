// Round 3: WHERE in the device's memory are the two classes the placement probe sees?  Allocates N chunks one after another
// (all alive), runs the library's probe pattern on each (one wavefront per 4 MB block, random 8-byte read-modify-writes inside
// the block's first 2 MB) and prints the rate next to the chunk's virtual address.
//   mem_class_map [N chunks] [chunk MB] [block KB] [span KB: the part of a block that is touched] [1: every chunk physically
//                 contiguous (hipExtMallocWithFlags, hipDeviceMallocContiguous); 2: every other chunk] [parts: also probe
//                 every 1/parts of a chunk on its own, with as many wavefronts as the whole chunk gets] [use MB: probe only
//                 the first so many MB of every chunk]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_probe(char *base, int64_t stride_bytes, uint32_t span_values, int iters, unsigned long long *sink, uint32_t wrap = 0xFFFFFFFFu)
{
    double *v = reinterpret_cast<double *>(base + (int64_t)(blockIdx.x % wrap) * stride_bytes);
    uint64_t x = ((uint64_t)blockIdx.x * 64 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
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

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 18;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 12288) << 20;
    const size_t block = (size_t)(argc > 3 ? atoi(argv[3]) : 4096) << 10;
    const size_t span = (size_t)(argc > 4 ? atoi(argv[4]) : 2048) << 10;
    const int contiguous = argc > 5 ? atoi(argv[5]) : 0;
    const int parts = argc > 6 ? atoi(argv[6]) : 0;
    const size_t use = argc > 7 ? (size_t)atoll(argv[7]) << 20 : chunk;
    const int slots = (int)(std::min(use, chunk) / block);
    unsigned long long *sink;
    CHECK(hipMalloc(&sink, 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<char *> p;
    for (int i = 0; i < n; i++) {
        char *q = nullptr;
        const bool cont = contiguous == 1 || (contiguous == 2 && (i & 1));
        if ((cont ? hipExtMallocWithFlags((void **)&q, chunk, hipDeviceMallocContiguous) : hipMalloc(&q, chunk)) != hipSuccess) { (void)hipGetLastError(); printf("chunk %2d: no memory left\n", i); break; }
        p.push_back(q);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(k_probe, dim3(slots), dim3(64), 0, 0, q, (int64_t)block, (uint32_t)(std::min<size_t>(block, span) / 8), 128, sink);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            CHECK(hipEventElapsedTime(&ms, a, b));
        }
        printf("chunk %2d%s at %p (%6.1f GB into the first chunk's address): %6.2f G updates/s\n", i, cont ? " (contiguous)" : "", (void *)q,
               ((intptr_t)q - (intptr_t)p[0]) / 1073741824.0, (double)slots * 64 * 128 / (ms * 1e-3) / 1e9);
        if (parts > 1) {
            printf("          by part:");
            const int per = slots / parts;
            for (int s = 0; s < parts; s++) {
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(a));
                    hipLaunchKernelGGL(k_probe, dim3(slots), dim3(64), 0, 0, q + (size_t)s * per * block, (int64_t)block, (uint32_t)(std::min<size_t>(block, span) / 8), 128, sink, (uint32_t)per);
                    CHECK(hipEventRecord(b));
                    CHECK(hipEventSynchronize(b));
                    CHECK(hipEventElapsedTime(&ms, a, b));
                }
                printf(" %5.2f", (double)slots * 64 * 128 / (ms * 1e-3) / 1e9);
            }
            printf("\n");
        }
    }
    return 0;
}
