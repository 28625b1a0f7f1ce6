// Probe: what HBM bandwidth does the access pattern of a strided transform pass reach, compute aside?
// A workgroup of 512 threads moves a tile of `rows` segments of SEG bytes each, `stride` bytes apart (read from A, written
// to B at the same offsets) -- the shape of the z pass of the interleaved grid (512 rows x 256 bytes, 32 KB apart) -- for
// a power-of-two stride and for the same stride plus one segment (a padded row pitch).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stride_probe tools/stride_probe.hip && gpurun_out/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// tile t: segment index within a row = t % tiles_per_row, slab = t / tiles_per_row; row r of the tile at
// slab * slab_bytes + r * stride + seg * SEG.  Each thread moves 16 bytes; 512 threads x 16 B = 8 KB per step = 8 KB / SEG rows.
template <int SEG>
__global__ void __launch_bounds__(512)
k_probe(const float4* __restrict__ a, float4* __restrict__ b, int rows, long stride, int tiles_per_row, long slab_bytes, int hold) {
    const int t = blockIdx.x, seg = t % tiles_per_row, slab = t / tiles_per_row;
    constexpr int LPS = SEG / 16;                      // lanes per segment
    const int lane_in = threadIdx.x % LPS, row0 = threadIdx.x / LPS, rows_per_step = 512 / LPS;
    const long base = (long)slab * slab_bytes + (long)seg * SEG + lane_in * 16;
    float4 v[32];
    const int steps = rows / rows_per_step;            // <= 32
    // like the transform pass: all loads first (a column lives in registers), then all stores
    if (hold) {
#pragma unroll
        for (int s = 0; s < 32; ++s) if (s < steps) v[s] = a[(base + (long)(row0 + s * rows_per_step) * stride) / 16];
#pragma unroll
        for (int s = 0; s < 32; ++s) if (s < steps) b[(base + (long)(row0 + s * rows_per_step) * stride) / 16] = v[s];
    } else {
        for (int s = 0; s < steps; ++s) {
            const long o = (base + (long)(row0 + s * rows_per_step) * stride) / 16;
            b[o] = a[o];
        }
    }
}

int main() {
    const int rows = 512;
    const long row_bytes = 32768;                      // 512 x-points x 8 coils x 8 bytes
    const int nslab = 256;                             // y
    float4 *a, *b;
    const size_t cap = (size_t)nslab * rows * (row_bytes + 4096) + (1 << 20);
    CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap));
    CK(hipMemset(a, 1, cap)); CK(hipMemset(b, 0, cap));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int seg; long pad; int hold; };
    const Case cases[] = {
        {"256-B segments, 32 KB stride, loads first", 256, 0, 1},
        {"256-B segments, 32 KB + 256 B stride, loads first", 256, 256, 1},
        {"256-B segments, 32 KB + 1 KB stride, loads first", 256, 1024, 1},
        {"128-B segments, 32 KB stride, loads first", 128, 0, 1},
        {"128-B segments, 32 KB + 128 B stride, loads first", 128, 128, 1},
        {"1-KB segments, 32 KB stride, loads first", 1024, 0, 1},
        {"256-B segments, 32 KB stride, streaming", 256, 0, 0},
        {"256-B segments, 32 KB + 256 B stride, streaming", 256, 256, 0},
    };
    for (const Case& c : cases) {
        const long stride = row_bytes + c.pad;
        const int tiles_per_row = (int)(row_bytes / c.seg);
        const long slab_bytes = stride * rows;
        const int blocks = tiles_per_row * nslab;
        const double bytes = 2.0 * (double)blocks * rows * c.seg;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (c.seg == 256) hipLaunchKernelGGL(k_probe<256>, dim3(blocks), dim3(512), 0, 0, a, b, rows, stride, tiles_per_row, slab_bytes, c.hold);
            else if (c.seg == 128) hipLaunchKernelGGL(k_probe<128>, dim3(blocks), dim3(512), 0, 0, a, b, rows, stride, tiles_per_row, slab_bytes, c.hold);
            else hipLaunchKernelGGL(k_probe<1024>, dim3(blocks), dim3(512), 0, 0, a, b, rows, stride, tiles_per_row, slab_bytes, c.hold);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-55s %7.3f ms  %6.2f TB/s (%.2f GB moved)\n", c.name, best, bytes / best / 1e9, bytes / 1e9);
    }
    return 0;
}
