// How long do large device allocations take?  (context creation allocates tens of GB of slot state)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv)
{
    hipFree(0);
    for (int rep = 0; rep < 2; rep++)
        for (int a = 1; a < argc; a++) {
            const size_t gb = (size_t)atoll(argv[a]);
            void *p = nullptr;
            auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipMalloc(&p, gb << 30);
            auto t1 = std::chrono::steady_clock::now();
            if (e != hipSuccess) { printf("%zu GB: alloc failed\n", gb); continue; }
            hipMemset(p, 0, gb << 30);
            hipDeviceSynchronize();
            auto t2 = std::chrono::steady_clock::now();
            hipFree(p);
            auto t3 = std::chrono::steady_clock::now();
            printf("rep %d  %3zu GB: hipMalloc %.3f s, memset %.3f s, hipFree %.3f s\n", rep, gb,
                   std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(),
                   std::chrono::duration<double>(t3 - t2).count());
            fflush(stdout);
        }
    return 0;
}
