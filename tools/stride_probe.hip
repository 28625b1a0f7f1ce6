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


// Round 5: the z passes of the headline move ALL rows on one side and a flagged 16 % of them on the other (the k-space support
// table).  k_sparse: a tile reads rows [0, 256) and writes the rows of [0, 512) a hash flags (PAD: the zero-padded pass), or
// reads the flagged rows and writes rows [0, 256) (the cropped pass).  RUN = length of the runs of flagged rows (1, 4).
__device__ inline bool flagged(int tile, int row, int run, int percent) {
    unsigned h = (unsigned)tile * 2654435761u + (unsigned)(row / run) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (int)(h % 100u) < percent;
}
template <bool PAD>
__global__ void __launch_bounds__(512)
k_sparse(const float4* __restrict__ a, float4* __restrict__ b, long stride, int tiles_per_row, long slab_bytes, int run, int percent, int delay) {
    extern __shared__ float4 pad_lds[];              // only to limit the workgroups per CU (74 KB: two, 50 KB: three)
    constexpr int SEG = 256, LPS = SEG / 16, RPS = 512 / LPS, STEPS = 512 / RPS;      // 32 rows per step, 16 steps
    const int t = blockIdx.x, seg = t % tiles_per_row, slab = t / tiles_per_row;
    const int lane_in = threadIdx.x % LPS, row0 = threadIdx.x / LPS;
    const long base = (long)slab * slab_bytes + (long)seg * SEG + lane_in * 16;
    float4 v[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int row = row0 + s * RPS;
        const bool rd = PAD ? row < 256 : flagged(t, row, run, percent);
        v[s] = rd ? a[(base + (long)row * stride) / 16] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // `delay` dependent multiply-adds per value stand in for the transform's arithmetic (the loads must have arrived first)
    for (int i = 0; i < delay; ++i) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) { v[s].x = fmaf(v[s].x, 1.0000001f, v[(s + 1) % STEPS].y); v[s].y = fmaf(v[s].y, 0.9999999f, v[s].x); }
    }
    if (delay) __syncthreads();
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int row = row0 + s * RPS;
        const bool wr = PAD ? flagged(t, row, run, percent) : row < 256;
        if (wr) b[(base + (long)row * stride) / 16] = v[(s + 3) % STEPS];
    }
}

template <bool PAD>
static void run_sparse(const char* name, const float4* a, float4* b, int run, int percent, hipEvent_t e0, hipEvent_t e1, int lds = 0, int delay = 0) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sparse<PAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const long stride = 32768;
    const int nslab = 256, tiles_per_row = 128;
    const long slab_bytes = stride * 512;
    const int blocks = tiles_per_row * nslab;
    const double bytes = (double)blocks * 256.0 * (256.0 + 512.0 * percent / 100.0);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_sparse<PAD>), dim3(blocks), dim3(512), lds, 0, a, b, stride, tiles_per_row, slab_bytes, run, percent, delay);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("%-62s lds %3d KB delay %3d %7.3f ms  %6.2f TB/s (%.2f GB moved, %d tiles: %.1f ns per tile)\n", name, lds / 1024, delay, best, bytes / best / 1e9, bytes / 1e9, blocks, best * 1e6 / blocks);
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
    run_sparse<true>("pad z: read rows 0..255, write 16 % of 512 rows (single rows)", a, b, 1, 16, e0, e1);
    run_sparse<true>("pad z: read rows 0..255, write 16 % of 512 rows (runs of 4)", a, b, 4, 16, e0, e1);
    run_sparse<true>("pad z: read rows 0..255, write 47 % of 512 rows (single rows)", a, b, 1, 47, e0, e1);
    run_sparse<true>("pad z: read rows 0..255, write 100 % of 512 rows", a, b, 1, 100, e0, e1);
    run_sparse<false>("crop z: read 16 % of 512 rows (single rows), write rows 0..255", a, b, 1, 16, e0, e1);
    run_sparse<false>("crop z: read 16 % of 512 rows (runs of 4), write rows 0..255", a, b, 4, 16, e0, e1);
    run_sparse<false>("crop z: read 47 % of 512 rows (single rows), write rows 0..255", a, b, 1, 47, e0, e1);
    run_sparse<false>("crop z: read 100 % of 512 rows, write rows 0..255", a, b, 1, 100, e0, e1);
    for (int lds : {0, 50 * 1024, 74 * 1024})
        for (int delay : {0, 8, 16, 32}) {
            run_sparse<false>("crop z 16 % (runs of 4)", a, b, 4, 16, e0, e1, lds, delay);
            run_sparse<true>("pad z 16 % (runs of 4)", a, b, 4, 16, e0, e1, lds, delay);
        }
    return 0;
}
