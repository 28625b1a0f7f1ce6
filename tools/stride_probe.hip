// Probe: what HBM bandwidth does the access pattern of a strided transform pass reach, compute aside?
// A workgroup of 512 threads moves a tile of 512 rows x SEG bytes, `stride` bytes apart (read from A, written to B at the
// same offsets) -- the shape of the z pass of the interleaved grid (512 rows x 256 bytes, 32 KB apart).  Variants: the row
// pitch padded by one segment; loads of the whole tile first (a column lives in registers, as in the transform), in
// batches of 8 rows, or row by row (streaming); 128-, 256- and 1024-byte segments.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stride_probe tools/stride_probe.hip && gpurun_out/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// tile t: segment = t % tiles_per_row, slab = t / tiles_per_row; row r of the tile at slab*slab_bytes + r*stride + seg*SEG.
// Each thread moves 16 bytes per step; 512 threads cover 8 KB = 8192 / SEG rows per step; STEPS = 512 rows / that.
template <int SEG, int BATCH /* rows loaded before they are stored; 0: one at a time */>
__global__ void __launch_bounds__(512)
k_probe(const float4* __restrict__ a, float4* __restrict__ b, long stride, int tiles_per_row, long slab_bytes) {
    constexpr int LPS = SEG / 16, RPS = 512 / LPS, STEPS = 512 / RPS;
    const int t = blockIdx.x, seg = t % tiles_per_row, slab = t / tiles_per_row;
    const int lane_in = threadIdx.x % LPS, row0 = threadIdx.x / LPS;
    const long base = (long)slab * slab_bytes + (long)seg * SEG + lane_in * 16;
    if (BATCH == 0) {
        for (int s = 0; s < STEPS; ++s) {
            const long o = (base + (long)(row0 + s * RPS) * stride) / 16;
            b[o] = a[o];
        }
    } else {
        constexpr int NB = BATCH == 0 ? 1 : (BATCH < STEPS ? BATCH : STEPS);
        float4 v[NB];
        for (int s0 = 0; s0 < STEPS; s0 += NB) {
#pragma unroll
            for (int s = 0; s < NB; ++s) v[s] = a[(base + (long)(row0 + (s0 + s) * RPS) * stride) / 16];
#pragma unroll
            for (int s = 0; s < NB; ++s) b[(base + (long)(row0 + (s0 + s) * RPS) * stride) / 16] = v[s];
        }
    }
}

template <int SEG, int BATCH>
static void run(const char* name, const float4* a, float4* b, long pad, hipEvent_t e0, hipEvent_t e1) {
    const long row_bytes = 32768;                      // 512 x-points x 8 coils x 8 bytes
    const int nslab = 256;
    const long stride = row_bytes + pad;
    const int tiles_per_row = (int)(row_bytes / SEG);
    const long slab_bytes = stride * 512;
    const int blocks = tiles_per_row * nslab;
    const double bytes = 2.0 * (double)blocks * 512 * SEG;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_probe<SEG, BATCH>), dim3(blocks), dim3(512), 0, 0, a, b, stride, tiles_per_row, slab_bytes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("%-62s %7.3f ms  %6.2f TB/s (%.2f GB moved)\n", name, best, bytes / best / 1e9, bytes / 1e9);
}

int main() {
    float4 *a, *b;
    const size_t cap = (size_t)256 * 512 * (32768 + 4096) + (1 << 20);
    CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap));
    CK(hipMemset(a, 1, cap)); CK(hipMemset(b, 0, cap));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run<256, 64>("256-B segments, 32 KB stride, whole tile loaded first", a, b, 0, e0, e1);
    run<256, 64>("256-B segments, 32 KB + 256 B stride, whole tile first", a, b, 256, e0, e1);
    run<128, 64>("128-B segments, 32 KB stride, whole tile first", a, b, 0, e0, e1);
    run<128, 64>("128-B segments, 32 KB + 128 B stride, whole tile first", a, b, 128, e0, e1);
    run<1024, 64>("1-KB segments, 32 KB stride, whole tile first", a, b, 0, e0, e1);
    run<256, 8>("256-B segments, 32 KB stride, batches of 8 rows", a, b, 0, e0, e1);
    run<256, 4>("256-B segments, 32 KB stride, batches of 4 rows", a, b, 0, e0, e1);
    run<1024, 8>("1-KB segments, 32 KB stride, batches of 8 rows", a, b, 0, e0, e1);
    run<128, 0>("128-B segments, 32 KB stride, row by row", a, b, 0, e0, e1);
    run<256, 0>("256-B segments, 32 KB stride, row by row", a, b, 0, e0, e1);
    run<256, 0>("256-B segments, 32 KB + 256 B stride, row by row", a, b, 256, e0, e1);
    run<1024, 0>("1-KB segments, 32 KB stride, row by row", a, b, 0, e0, e1);
    return 0;
}
