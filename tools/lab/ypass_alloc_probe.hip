// LAB (round 5): does the speed of the y-pass traffic pattern depend on WHICH allocation it runs on?  The zero-padded y pass reads
// 256 rows of 256 bytes 32 KB apart (the compact intermediate) and writes 512 rows 16 MB apart (the grid); two processes on one box
// differ by 5 % in exactly these passes.  K pairs of buffers from hipMalloc, all kept alive; the same kernel on each.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ypass_probe tools/lab/ypass_alloc_probe.hip && gpurun_out/ypass_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// tile (tr, z): reads in[z * 8 MB + ky * 32 KB + tr * 256 + lane], ky < 256; writes out[ky * 16 MB + z * 32 KB + tr * 256 + lane], ky < 512
__global__ void __launch_bounds__(512)
k_ypass(const float4* __restrict__ in, float4* __restrict__ out) {
    const int tr = blockIdx.x, z = blockIdx.y;
    const int lane = threadIdx.x % 16, row0 = threadIdx.x / 16;           // 32 rows per step
    float4 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        v[s] = in[((size_t)z * (8u << 20) + (size_t)(row0 + 32 * s) * 32768 + tr * 256 + lane * 16) / 16];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s)
        out[((size_t)(row0 + 32 * s) * (16u << 20) + (size_t)z * 32768 + tr * 256 + lane * 16) / 16] = v[s & 7];
}

int main() {
    const int K = 6;
    const size_t in_bytes = (size_t)256 * (8u << 20), out_bytes = (size_t)512 * (16u << 20);
    float4 *in[K], *out[K];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < K; ++k) {
        CK(hipMalloc(&in[k], in_bytes)); CK(hipMalloc(&out[k], out_bytes));
        CK(hipMemset(in[k], 1, in_bytes));
    }
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < K; ++k) {
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_ypass, dim3(128, 256), dim3(512), 0, 0, in[k], out[k]);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r && ms < best) best = ms;
            }
            const double bytes = 128.0 * 256 * 256 * (256 + 512);
            printf("pair %d  in %p out %p  %7.3f ms  %5.2f TB/s\n", k, (void*)in[k], (void*)out[k], best, bytes / best / 1e9);
        }
    // the same pattern with the input taken from ANOTHER pair (relative placement of the two streams)
    for (int k = 0; k < K; ++k) {
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_ypass, dim3(128, 256), dim3(512), 0, 0, in[(k + 1) % K], out[k]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        printf("in %d -> out %d  %7.3f ms\n", (k + 1) % K, k, best);
    }
    return 0;
}
