// (round 4) What does giving back the losers of the slot-memory draw cost, and who pays?
// Allocates CANDS buffers of GB gigabytes, writes each once (as the draw's probe does), frees all but the last, then times the next
// allocations of several sizes.  Variants: (a) frees on the calling thread; (b) a small allocation first; (c) frees on a second thread
// while this one launches kernels on the kept buffer and times them.
// usage: free_cost_probe [GB=60] [CANDS=4]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_touch(char *p, size_t bytes, size_t stride)
{
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (i < bytes) p[i] = 1;
}

__global__ void k_spin(unsigned long long *out, int iters)
{
    unsigned long long a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a * 6364136223846793005ull + 1442695040888963407ull;
    if (a == 42) *out = a;
}

static void touch(char *p, size_t bytes)
{
    const size_t stride = 4096;
    const size_t n = bytes / stride;
    hipLaunchKernelGGL(k_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, bytes, stride);
    CK(hipDeviceSynchronize());
}

static int g_touch = 1;          // 1: every page written; 2: 2 MB of every 16 MB (what the draw's probe writes); 0: nothing

static std::vector<char *> make(int cands, size_t bytes)
{
    std::vector<char *> c;
    for (int i = 0; i < cands; i++) {
        char *p = nullptr;
        double t = now();
        CK(hipMalloc((void **)&p, bytes));
        double t1 = now();
        if (g_touch == 1) touch(p, bytes);
        if (g_touch == 2) for (size_t o = 0; o + ((size_t)2 << 20) <= bytes; o += (size_t)16 << 20) {
            hipLaunchKernelGGL(k_touch, dim3(2), dim3(256), 0, 0, p + o, (size_t)2 << 20, (size_t)4096);
        }
        CK(hipDeviceSynchronize());
        printf("   candidate %d: hipMalloc %.3f s, first touch %.3f s\n", i, t1 - t, now() - t1);
        c.push_back(p);
    }
    return c;
}

static void time_alloc(const char *what, size_t bytes)
{
    char *p = nullptr;
    double t = now();
    CK(hipMalloc((void **)&p, bytes));
    double t1 = now();
    CK(hipFree(p));
    printf("   %-40s hipMalloc of %6.2f GB: %.3f s (its hipFree %.3f s)\n", what, bytes / 1e9, t1 - t, now() - t1);
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)((argc > 1 ? atof(argv[1]) : 60.0) * 1e9);
    const int cands = argc > 2 ? atoi(argv[2]) : 4;
    CK(hipSetDevice(0));
    time_alloc("fresh process:", (size_t)35e8);

    if (argc > 3) {
        // the later variants only
        for (int v = 0; v < 3; v++) {
            g_touch = v == 0 ? 0 : (v == 1 ? 2 : 1);
            printf("(%c) %d candidates, %s; losers freed%s\n", "def"[v], cands, v == 0 ? "nothing written" : (v == 1 ? "2 MB of every 16 MB written" : "every page written"),
                   v == 2 ? ", then 8 s of sleep" : "");
            auto c = make(cands, bytes);
            for (int i = 0; i + 1 < cands; i++) { double t = now(); CK(hipFree(c[i])); printf("   hipFree of loser %d: %.3f s\n", i, now() - t); }
            if (v == 2) std::this_thread::sleep_for(std::chrono::seconds(8));
            time_alloc("first allocation after the frees:", (size_t)35e8);
            time_alloc("second, 40 GB:", (size_t)40e9);
            time_alloc("third, 14 GB:", (size_t)14e9);
            CK(hipFree(c.back()));
            std::this_thread::sleep_for(std::chrono::seconds(3));
            time_alloc("after the kept one went too (+3 s):", (size_t)35e8);
        }
        return 0;
    }
    printf("(a) %d candidates, losers freed on this thread\n", cands);
    {
        auto c = make(cands, bytes);
        for (int i = 0; i + 1 < cands; i++) { double t = now(); CK(hipFree(c[i])); printf("   hipFree of loser %d: %.3f s\n", i, now() - t); }
        time_alloc("first allocation after the frees:", (size_t)35e8);
        time_alloc("second:", (size_t)35e8);
        time_alloc("third, 14 GB:", (size_t)14e9);
        CK(hipFree(c.back()));
        time_alloc("after the kept one went too:", (size_t)35e8);
    }
    printf("(b) the same, a 1 MB allocation first\n");
    {
        auto c = make(cands, bytes);
        for (int i = 0; i + 1 < cands; i++) CK(hipFree(c[i]));
        time_alloc("1 MB after the frees:", (size_t)1 << 20);
        time_alloc("then 3.5 GB:", (size_t)35e8);
        CK(hipFree(c.back()));
    }
    printf("(c) losers freed by a second thread, which then allocates 1 MB; this thread launches kernels meanwhile\n");
    {
        auto c = make(cands, bytes);
        unsigned long long *out;
        CK(hipMalloc((void **)&out, 8));
        std::atomic<int> done{0};
        double t0 = now();
        std::thread reaper([&]() {
            hipSetDevice(0);
            for (int i = 0; i + 1 < cands; i++) hipFree(c[i]);
            double t = now();
            char *q = nullptr;
            hipMalloc((void **)&q, (size_t)1 << 20);
            hipFree(q);
            printf("   reaper: frees %.3f s, its 1 MB allocation %.3f s\n", t - t0, now() - t);
            t = now();
            hipMalloc((void **)&q, (size_t)35e8);
            hipFree(q);
            printf("   reaper: its 3.5 GB allocation %.3f s\n", now() - t);
            done = 1;
        });
        double worst = 0, sum = 0;
        int n = 0;
        while (!done) {
            double t = now();
            hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, 0, out, 20000);
            CK(hipDeviceSynchronize());
            double d = now() - t;
            worst = d > worst ? d : worst;
            sum += d;
            n++;
        }
        reaper.join();
        printf("   %d kernels meanwhile: mean %.3f ms, worst %.3f ms\n", n, sum / n * 1e3, worst * 1e3);
        time_alloc("this thread afterwards:", (size_t)35e8);
        time_alloc("14 GB:", (size_t)14e9);
        CK(hipFree(c.back()));
    }
    return 0;
}
