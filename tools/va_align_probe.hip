// Round 4: are the "levels" of device memory (profiles/r03/mem_class_*.txt: the same random read-modify-write pattern runs at
// 20 / 22 / 24 / 26 G updates/s depending on the ALLOCATION it touches) a matter of how virtual and physical addresses are
// aligned to each other?  The GPU's page tables describe a range by fragments: a translation covers 2^k pages when the virtual
// AND the physical address of the range are both aligned to that size, so (virtual - physical) mod 2^k decides the largest
// fragment an allocation can have -- a number hipMalloc leaves to chance.  Here ONE physical allocation (hipMemCreate) is mapped
// at base + j * 2 MB for a series of j (the physical memory, and where every wavefront's block lies in it, stay the same) and
// probed each time with the library's placement probe.
//   va_align_probe [physical allocations] [GB each] [block KB per wavefront] [span KB touched per block]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_probe(char *base, int64_t stride_bytes, uint32_t span_values, int iters, unsigned long long *sink)
{
    double *v = reinterpret_cast<double *>(base + (int64_t)blockIdx.x * stride_bytes);
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
    const int nphys = argc > 1 ? atoi(argv[1]) : 3;
    const size_t size = (size_t)(argc > 2 ? atoi(argv[2]) : 12) << 30;
    const size_t block = (size_t)(argc > 3 ? atoi(argv[3]) : 4096) << 10;
    const size_t span = (size_t)(argc > 4 ? atoi(argv[4]) : 2048) << 10;
    const int slots = (int)(size / block);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t gran = 0;
    CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("allocation granularity %zu KB; %d wavefronts, one per %zu KB block, %zu KB touched in each\n", gran >> 10, slots, block >> 10, span >> 10);
    unsigned long long *sink;
    CHECK(hipMalloc(&sink, 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const size_t slack = (size_t)1 << 30;
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, size + slack, (size_t)1 << 30, nullptr, 0));
    printf("virtual range at %p (aligned to 1 GB: %s)\n", va, ((uintptr_t)va & (((uintptr_t)1 << 30) - 1)) == 0 ? "yes" : "no");
    const int js[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 511, 0};
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (int ph = 0; ph < nphys; ph++) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, size, &prop, 0) != hipSuccess) { (void)hipGetLastError(); printf("physical allocation %d: no memory\n", ph); break; }
        hs.push_back(h);          // (kept alive: the next one is other memory)
        printf("physical allocation %d, mapped at base + j * 2 MB:\n", ph);
        for (int j : js) {
            char *at = (char *)va + (size_t)j * ((size_t)2 << 20);
            CHECK(hipMemMap(at, size, 0, h, 0));
            CHECK(hipMemSetAccess(at, size, &acc, 1));
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(k_probe, dim3(slots), dim3(64), 0, 0, at, (int64_t)block, (uint32_t)(span / 8), 128, sink);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                CHECK(hipEventElapsedTime(&ms, a, b));
            }
            printf("   j = %3d  %6.2f G updates/s\n", j, (double)slots * 64 * 128 / (ms * 1e-3) / 1e9);
            fflush(stdout);
            CHECK(hipMemUnmap(at, size));
        }
    }
    // the same pattern on plain hipMalloc buffers, for the levels of this box
    for (int i = 0; i < 4; i++) {
        char *q = nullptr;
        if (hipMalloc(&q, size) != hipSuccess) { (void)hipGetLastError(); break; }
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(k_probe, dim3(slots), dim3(64), 0, 0, q, (int64_t)block, (uint32_t)(span / 8), 128, sink);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            CHECK(hipEventElapsedTime(&ms, a, b));
        }
        printf("hipMalloc buffer %d at %p: %6.2f G updates/s\n", i, (void *)q, (double)slots * 64 * 128 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
