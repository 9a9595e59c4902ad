// Microbenchmark, round 3: what does a state update cost when FIRST touches of a 64-byte line are blind whole-line
// writes (no read) and only re-touches are 8-byte read-modify-writes?
//
//   line_wall BLIND_PCT SLICE_MB WAVES_PER_CU STREAM_BYTES [ITERS]
//     BLIND_PCT     share of the updates that are blind whole-line writes (0 = every update a read-modify-write: today)
//     SLICE_MB      private dense float64 array per wavefront (8 MB = one value per node at n = 1M)
//     STREAM_BYTES  coalesced bytes streamed per update beside the random accesses (the CSR row: 8)
//
// Loads of iteration i are consumed in iteration i + 1 (software pipeline, as in the push kernel).  Prints G updates/s.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int STREAM, int SHAPE>
__global__ __launch_bounds__(64) void k(double *tab, uint64_t values_per_wave, uint32_t blind_cut, const double *stream,
                                        uint64_t stream_elems, int iters, double *sink, int noload)
{
    const uint64_t wave = blockIdx.x;
    const int lane = threadIdx.x;
    double *base = tab + wave * values_per_wave;
    double acc = 0;
    uint64_t spos = wave * 64 + lane;
    double pending = 0;
    double *pending_at = base + lane;
    bool pending_valid = false;
    for (int it = 0; it < iters; it++) {
        const uint64_t h = mix((wave * 64 + lane) * 1315423911ULL + it);
        const uint64_t j = (h >> 16) % values_per_wave;
        const bool blind = (uint32_t)(h & 0xFFFF) < blind_cut;
        double *at = base + j;
        // this iteration's accesses: blind lanes write their whole line now, the others load their value
        double loaded = 0;
        if (SHAPE == 0) {
            if (blind) {
                double2 *line = reinterpret_cast<double2 *>(base + (j & ~7ull));
                const int chunk = (int)((j & 7) >> 1);
                const double p = (double)it;
#pragma unroll
                for (int c = 0; c < 4; c++) line[c] = (c == chunk) ? ((j & 1) ? make_double2(0.0, p) : make_double2(p, 0.0)) : make_double2(0.0, 0.0);
            }
        } else {
            // quad-cooperative: in instruction c the four lanes of a quad write the four 16-byte chunks of the line of the
            // quad's c-th lane (when that lane is blind): every store instruction carries whole 64-byte lines
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
        }
        // (every lane issues the load: a blind lane at a cached dummy address); NOLOAD: pure writes, nothing waits
        if (!noload) loaded = *(blind ? stream + lane : at);
        if (STREAM) {
#pragma unroll
            for (int q = 0; q < STREAM / 8; q++) {
                acc += stream[spos % stream_elems];
                spos += (uint64_t)gridDim.x * 64;
            }
        }
        // the previous iteration's read-modify-writes complete
        if (pending_valid) *pending_at = pending + 1.0;
        pending = loaded;
        pending_at = at;
        pending_valid = !blind;
    }
    if (pending_valid) *pending_at = pending + 1.0;
    if (acc == 12345.678) sink[0] = acc;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: line_wall BLIND_PCT SLICE_MB WAVES_PER_CU STREAM_BYTES [ITERS [SHAPE [NOLOAD]]]\n"); return 2; }
    const int blind_pct = atoi(argv[1]);
    const double slice_mb = atof(argv[2]);
    const int wpc = atoi(argv[3]), stream_b = atoi(argv[4]);
    const int iters = argc > 5 ? atoi(argv[5]) : 512;
    const int shape = argc > 6 ? atoi(argv[6]) : 0;
    const int noload = argc > 7 ? atoi(argv[7]) : 0;
    const int waves = 256 * wpc;
    const uint64_t vpw = (uint64_t)(slice_mb * 1048576.0 / 8.0) & ~7ull;
    const size_t bytes = (size_t)waves * vpw * 8;
    double *tab, *sink, *st;
    const uint64_t stream_elems = ((size_t)2 << 30) / 8;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess || hipMalloc(&st, stream_elems * 8) != hipSuccess) {
        fprintf(stderr, "alloc failed\n");
        return 1;
    }
    hipMemset(tab, 0, bytes);
    hipMemset(st, 0, stream_elems * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    const uint32_t cut = (uint32_t)(65536.0 * blind_pct / 100.0);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        if (shape == 0) {
            if (stream_b == 0) hipLaunchKernelGGL((k<0, 0>), dim3(waves), dim3(64), 0, 0, tab, vpw, cut, st, stream_elems, iters, sink, noload);
            else hipLaunchKernelGGL((k<8, 0>), dim3(waves), dim3(64), 0, 0, tab, vpw, cut, st, stream_elems, iters, sink, noload);
        } else {
            if (stream_b == 0) hipLaunchKernelGGL((k<0, 1>), dim3(waves), dim3(64), 0, 0, tab, vpw, cut, st, stream_elems, iters, sink, noload);
            else hipLaunchKernelGGL((k<8, 1>), dim3(waves), dim3(64), 0, 0, tab, vpw, cut, st, stream_elems, iters, sink, noload);
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    const double ops = (double)waves * 64 * iters;
    printf("shape %d noload %d  blind %3d%%  slice %6.2f MB (%6.1f GB in all)  %2d waves/CU  stream %2d B/update : %7.2f G updates/s  (%.2f ms)\n", blind_pct,
           slice_mb, bytes / 1073741824.0, wpc, stream_b, ops / best / 1e6, best);
    return 0;
}
