// How fast can a large result leave the GPU for a PAGEABLE host buffer (a fresh numpy array)?
//   (a) one hipMemcpy into untouched pages      (b) the same into pages touched before
//   (c) hipMemcpy into pinned memory (the DMA rate)
//   (d) T threads, each streaming its chunks through two pinned staging buffers and copying them out
// usage: d2h_rate [GB] [threads] [chunk MB]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void pipelined(char *dst, const char *src, size_t bytes, int threads, size_t chunk)
{
    std::vector<std::thread> pool;
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([=]() {
            hipSetDevice(0);
            hipStream_t s;
            hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            char *pin[2];
            hipEvent_t ev[2];
            for (int i = 0; i < 2; i++) { hipHostMalloc((void **)&pin[i], chunk, hipHostMallocDefault); hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); }
            int cur = 0;
            size_t pending = (size_t)-1;
            for (size_t c = t; c < nchunks + threads; c += threads) {
                if (c < nchunks) {
                    const size_t off = c * chunk, len = std::min(chunk, bytes - off);
                    hipMemcpyAsync(pin[cur], src + off, len, hipMemcpyDeviceToHost, s);
                    hipEventRecord(ev[cur], s);
                }
                if (pending != (size_t)-1) {
                    const size_t off = pending * chunk, len = std::min(chunk, bytes - off);
                    hipEventSynchronize(ev[cur ^ 1]);
                    memcpy(dst + off, pin[cur ^ 1], len);
                }
                pending = c < nchunks ? c : (size_t)-1;
                cur ^= 1;
            }
            for (int i = 0; i < 2; i++) { hipHostFree(pin[i]); hipEventDestroy(ev[i]); }
            hipStreamDestroy(s);
        });
    for (auto &th : pool) th.join();
}

int main(int argc, char **argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 3.5;
    const int threads = argc > 2 ? atoi(argv[2]) : 4;
    const size_t chunk = (size_t)(argc > 3 ? atoi(argv[3]) : 16) << 20;
    const size_t bytes = (size_t)(gb * (1 << 30));
    char *dev;
    if (hipMalloc((void **)&dev, bytes) != hipSuccess) return 1;
    hipMemset(dev, 0x5a, bytes);
    hipDeviceSynchronize();
    double t;
    char *h = (char *)malloc(bytes);
    t = now(); hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost); printf("(a) hipMemcpy, untouched pageable pages   %6.1f GB/s\n", bytes / (now() - t) / 1e9);
    t = now(); hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost); printf("(b) hipMemcpy, touched pageable pages     %6.1f GB/s\n", bytes / (now() - t) / 1e9);
    free(h);
    char *p;
    hipHostMalloc((void **)&p, bytes, hipHostMallocDefault);
    t = now(); hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost); printf("(c) hipMemcpy into pinned memory          %6.1f GB/s\n", bytes / (now() - t) / 1e9);
    hipHostFree(p);
    for (int th : {1, 2, threads, 2 * threads}) {
        h = (char *)malloc(bytes);
        t = now(); pipelined(h, dev, bytes, th, chunk);
        const double dt = now() - t;
        bool ok = h[0] == 0x5a && h[bytes - 1] == 0x5a && h[bytes / 2] == 0x5a;
        printf("(d) %2d threads x 2 pinned buffers of %zu MB, untouched pages %6.1f GB/s %s\n", th, chunk >> 20, bytes / dt / 1e9, ok ? "" : "CORRUPT");
        free(h);
    }
    return 0;
}
