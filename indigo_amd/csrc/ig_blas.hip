// BLAS-1 glue kernels: axpby, scale, dotc, nrm2^2, elementwise max.
// All are pure HBM streams: 16-byte (two complex64) accesses per lane,
// grid-stride over a grid capped at 8 workgroups per CU.
//
// Reference semantics: indigo/backends/np.py:53-74,141-145 (numpy oracle),
// indigo/backends/cuda.py:239-302 (cuBLAS two-pass axpby that this fuses
// into one pass), indigo/backends/_customgpu.cu:7-13 (cu_max).
#include "ig_common.h"

namespace {

constexpr int BLK = 256;

inline int grid_for(const ig_ctx* ctx, int64_t work_items) {
    int64_t g = (work_items + BLK - 1) / BLK;
    int64_t cap = (int64_t)ctx->num_cu * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// MODE 0: y = a*x            (beta == 0, y not read)
// MODE 1: y = y + a*x        (beta == 1)
// MODE 2: y = b*y + a*x      (general)
// MODE 3: y = b*y            (alpha == 0)
// d_b / d_a (optional): device-resident real factors, b *= *d_b, a *= *d_a -- the scalars of a solver iteration
// (CG's alpha = rr / <p, Ap>, beta = r2 / rr) never visit the host.
template <int MODE>
__global__ void __launch_bounds__(BLK)
k_caxpby(int64_t n, float2 b, float2* __restrict__ y, float2 a, const float2* __restrict__ x, int vec_ok,
         const double* __restrict__ d_b = nullptr, const double* __restrict__ d_a = nullptr) {
    if (d_b) { const float f = (float)*d_b; b.x *= f; b.y *= f; }
    if (d_a) { const float f = (float)*d_a; a.x *= f; a.y *= f; }
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * BLK;
    if (vec_ok) {
        const int64_t n2 = n >> 1;
        float4* y4 = reinterpret_cast<float4*>(y);
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (int64_t i = tid; i < n2; i += nth) {
            float4 xv = (MODE != 3) ? x4[i] : make_float4(0, 0, 0, 0);
            float4 yv = (MODE != 0) ? y4[i] : make_float4(0, 0, 0, 0);
            float2 r0, r1;
            float2 x0 = make_float2(xv.x, xv.y), x1 = make_float2(xv.z, xv.w);
            float2 y0 = make_float2(yv.x, yv.y), y1 = make_float2(yv.z, yv.w);
            if (MODE == 0)      { r0 = cmul(a, x0); r1 = cmul(a, x1); }
            else if (MODE == 1) { r0 = y0; cfma(r0, a, x0); r1 = y1; cfma(r1, a, x1); }
            else if (MODE == 2) { r0 = cmul(b, y0); cfma(r0, a, x0); r1 = cmul(b, y1); cfma(r1, a, x1); }
            else                { r0 = cmul(b, y0); r1 = cmul(b, y1); }
            y4[i] = make_float4(r0.x, r0.y, r1.x, r1.y);
        }
        // odd tail element
        if ((n & 1) && tid == 0) {
            const int64_t i = n - 1;
            float2 xv = (MODE != 3) ? x[i] : make_float2(0, 0);
            float2 yv = (MODE != 0) ? y[i] : make_float2(0, 0);
            float2 r;
            if (MODE == 0)      r = cmul(a, xv);
            else if (MODE == 1) { r = yv; cfma(r, a, xv); }
            else if (MODE == 2) { r = cmul(b, yv); cfma(r, a, xv); }
            else                r = cmul(b, yv);
            y[i] = r;
        }
    } else {
        for (int64_t i = tid; i < n; i += nth) {
            float2 xv = (MODE != 3) ? x[i] : make_float2(0, 0);
            float2 yv = (MODE != 0) ? y[i] : make_float2(0, 0);
            float2 r;
            if (MODE == 0)      r = cmul(a, xv);
            else if (MODE == 1) { r = yv; cfma(r, a, xv); }
            else if (MODE == 2) { r = cmul(b, yv); cfma(r, a, xv); }
            else                r = cmul(b, yv);
            y[i] = r;
        }
    }
}

__global__ void __launch_bounds__(BLK)
k_smax(int64_t n, float val, float* __restrict__ a, int vec_ok) {
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * BLK;
    if (vec_ok) {
        float4* a4 = reinterpret_cast<float4*>(a);
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nth) {
            float4 v = a4[i];
            v.x = fmaxf(v.x, val); v.y = fmaxf(v.y, val); v.z = fmaxf(v.z, val); v.w = fmaxf(v.w, val);
            a4[i] = v;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nth) a[i] = fmaxf(a[i], val);
    } else {
        for (int64_t i = tid; i < n; i += nth) a[i] = fmaxf(a[i], val);
    }
}

// y[k] = beta*y[k] + alpha * sum_j X[k + j*ld]   (coil combination: the VStack adjoint's accumulation in one pass)
template <bool BETA0>
__global__ void __launch_bounds__(BLK)
k_csum_cols(int64_t rows, int64_t ncols, const float2* __restrict__ X, int64_t ld, float2 alpha, float2 beta,
            float2* __restrict__ y) {
    for (int64_t k = (int64_t)blockIdx.x * BLK + threadIdx.x; k < rows; k += (int64_t)gridDim.x * BLK) {
        float2 acc = make_float2(0.f, 0.f);
        for (int64_t j = 0; j < ncols; ++j) acc = cadd(acc, X[k + j * ld]);
        float2 out = cmul(alpha, acc);
        if (!BETA0) cfma(out, beta, y[k]);
        y[k] = out;
    }
}

// y[k] = alpha * sum_j X[k*ncols + j] + beta*y[k]: the panel's rows are contiguous (coil-interleaved layout)
template <bool BETA0, bool VEC>
__global__ void __launch_bounds__(BLK)
k_csum_il(int64_t rows, int64_t ncols, const float2* __restrict__ X, float2 alpha, float2 beta, float2* __restrict__ y) {
    for (int64_t k = (int64_t)blockIdx.x * BLK + threadIdx.x; k < rows; k += (int64_t)gridDim.x * BLK) {
        float2 acc = make_float2(0.f, 0.f);
        if (VEC) {              // even ncols, 16-byte aligned rows
            const float4* __restrict__ p = reinterpret_cast<const float4*>(X + k * ncols);
            for (int64_t h = 0; h < ncols / 2; ++h) {
                const float4 t = p[h];
                acc.x += t.x + t.z; acc.y += t.y + t.w;
            }
        } else {
            for (int64_t j = 0; j < ncols; ++j) acc = cadd(acc, X[k * ncols + j]);
        }
        float2 out = cmul(alpha, acc);
        if (!BETA0) cfma(out, beta, y[k]);
        y[k] = out;
    }
}

// the same for ncols = 2*PARTS in {2, 4, 8, 16, 32}: PARTS lanes share a row, each loads 16 bytes, so a wave reads
// 1 KB of consecutive addresses per instruction; the partial sums meet by lane shuffles
template <bool BETA0, int PARTS>
__global__ void __launch_bounds__(BLK)
k_csum_il_parts(int64_t rows, const float2* __restrict__ X, float2 alpha, float2 beta, float2* __restrict__ y) {
    const int64_t total = rows * PARTS;
    const int64_t span = (int64_t)gridDim.x * BLK;
    // every lane of a wave runs the same number of trips (the shuffles need the whole wave)
    for (int64_t e0 = (int64_t)blockIdx.x * BLK; e0 < total; e0 += span) {
        const int64_t e = e0 + threadIdx.x;
        float2 acc = make_float2(0.f, 0.f);
        if (e < total) {
            const float4 t = reinterpret_cast<const float4*>(X)[e];
            acc.x = t.x + t.z; acc.y = t.y + t.w;
        }
#pragma unroll
        for (int off = 1; off < PARTS; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off, 64);
            acc.y += __shfl_xor(acc.y, off, 64);
        }
        if (e < total && (threadIdx.x % PARTS) == 0) {
            const int64_t k = e / PARTS;
            float2 out = cmul(alpha, acc);
            if (!BETA0) cfma(out, beta, y[k]);
            y[k] = out;
        }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// per-thread float accumulation over a strided slice, then double from the
// wave reduction upwards; block partials land in `partials` (2 per block).
// DOT: sum conj(x)*y (re, im);  !DOT: sum |x|^2 (re only).
template <bool DOT>
__global__ void __launch_bounds__(BLK)
k_reduce(int64_t n, const float2* __restrict__ x, const float2* __restrict__ y, double* __restrict__ partials) {
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * BLK;
    double re = 0.0, im = 0.0;
    // accumulate short runs in float, flush to double to bound the error growth
    int64_t i = tid;
    while (i < n) {
        float fr = 0.f, fi = 0.f;
#pragma unroll 4
        for (int k = 0; k < 16 && i < n; ++k, i += nth) {
            float2 a = x[i];
            if (DOT) {
                float2 b = y[i];
                fr = fmaf(a.x, b.x, fr); fr = fmaf(a.y, b.y, fr);
                fi = fmaf(a.x, b.y, fi); fi = fmaf(-a.y, b.x, fi);
            } else {
                fr = fmaf(a.x, a.x, fr); fr = fmaf(a.y, a.y, fr);
            }
        }
        re += (double)fr; im += (double)fi;
    }
    re = wave_sum(re);
    if (DOT) im = wave_sum(im);
    __shared__ double s_re[BLK / 64], s_im[BLK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { s_re[wid] = re; s_im[wid] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0, m = 0;
        for (int w = 0; w < BLK / 64; ++w) { r += s_re[w]; m += s_im[w]; }
        partials[2 * blockIdx.x] = r;
        partials[2 * blockIdx.x + 1] = m;
    }
}

__global__ void __launch_bounds__(BLK)
k_reduce_final(int nblocks, const double* __restrict__ partials, double* __restrict__ out) {
    double re = 0, im = 0;
    for (int i = threadIdx.x; i < nblocks; i += BLK) { re += partials[2 * i]; im += partials[2 * i + 1]; }
    re = wave_sum(re); im = wave_sum(im);
    __shared__ double s_re[BLK / 64], s_im[BLK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { s_re[wid] = re; s_im[wid] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0, m = 0;
        for (int w = 0; w < BLK / 64; ++w) { r += s_re[w]; m += s_im[w]; }
        out[0] = r; out[1] = m;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- conjugate gradients: an iteration's vector work in THREE passes (reference loop: indigo/backends/backend.py:666-686) ----
//   k_cg_dot    Ap += lamda p (if lamda != 0);  block partials of Re<p, Ap>                          2 reads (+1 write)
//   k_cg_step_r alpha = rr / <p, Ap> (gated);   r -= alpha Ap;  block partials of ||r||^2             2 reads + 1 write
//   k_cg_step_xp beta = r2 / rr;                x += alpha p;   p = r + beta p                        3 reads + 2 writes
// instead of five vector passes and two two-kernel reductions with one-thread scalar kernels between them (12 launches, 9
// reads + 3 writes).  There is no "final" reduction kernel and no atomic: EVERY block of the consuming kernel sums the
// producer's block partials itself (<= 2048 doubles out of the L2, the same order in every block, hence the same bits) --
// kernel boundaries are the only synchronisation.  Block 0 of the consumer records the scalars (alpha; rr for the next
// iteration into the OTHER of two slots, since the blocks of this launch still read the current one; the history entry).
__device__ __forceinline__ double block_sum_partials(const double* __restrict__ partials, int nparts) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nparts; i += BLK) v += partials[i];
    v = wave_sum(v);
    __shared__ double s_part[BLK / 64];
    __shared__ double s_total;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) s_part[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLK / 64; ++w) t += s_part[w];
        s_total = t;
    }
    __syncthreads();
    return s_total;
}
__device__ __forceinline__ void block_store_partial(double v, double* __restrict__ partials) {
    v = wave_sum(v);
    __shared__ double s_out[BLK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) s_out[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLK / 64; ++w) t += s_out[w];
        partials[blockIdx.x] = t;
    }
}

// (VEC: both vectors 16-byte aligned -> two complex64 per lane and access; the odd last element goes to thread 0)
template <bool VEC, bool LAMDA>
__global__ void __launch_bounds__(BLK)
k_cg_dot(int64_t n, const float2* __restrict__ p, float2* __restrict__ Ap, float lamda, double* __restrict__ partials) {
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x, nth = (int64_t)gridDim.x * BLK;
    double acc = 0.0;
    auto one = [&](float2 pv, float2& av) -> float {
        if (LAMDA) { av.x = fmaf(lamda, pv.x, av.x); av.y = fmaf(lamda, pv.y, av.y); }
        return fmaf(pv.x, av.x, pv.y * av.y);                       // Re(conj(p) * Ap)
    };
    if (VEC) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        float4* a4 = reinterpret_cast<float4*>(Ap);
        const int64_t n2 = n >> 1;
        int64_t i = tid;
        while (i < n2) {
            float f = 0.f;
#pragma unroll 4
            for (int k = 0; k < 8 && i < n2; ++k, i += nth) {      // short float runs, flushed to double
                const float4 pv = p4[i];
                float4 av = a4[i];
                float2 a0 = make_float2(av.x, av.y), a1 = make_float2(av.z, av.w);
                f += one(make_float2(pv.x, pv.y), a0);
                f += one(make_float2(pv.z, pv.w), a1);
                if (LAMDA) a4[i] = make_float4(a0.x, a0.y, a1.x, a1.y);
            }
            acc += (double)f;
        }
        if ((n & 1) && tid == 0) { float2 av = Ap[n - 1]; acc += (double)one(p[n - 1], av); if (LAMDA) Ap[n - 1] = av; }
    } else {
        for (int64_t i = tid; i < n; i += nth) { float2 av = Ap[i]; acc += (double)one(p[i], av); if (LAMDA) Ap[i] = av; }
    }
    block_store_partial(acc, partials);
}

template <bool VEC>
__global__ void __launch_bounds__(BLK)
k_cg_step_r(int64_t n, float2* __restrict__ r, const float2* __restrict__ Ap, const double* __restrict__ pap_partials, int nparts,
            const double* __restrict__ d_rr, const double* __restrict__ d_r0, double tol2, double* __restrict__ d_alpha,
            double* __restrict__ rr_partials) {
    const double pap = block_sum_partials(pap_partials, nparts);
    const double rr = d_rr[0];
    const bool stop = !(rr >= tol2 * d_r0[0]);                     // the reference has left its loop by now (backend.py:683-685)
    const double alpha_d = (pap == 0.0 || stop) ? 0.0 : rr / pap; // 0/0 of an exactly converged system: a zero step, not NaN
    if (blockIdx.x == 0 && threadIdx.x == 0) d_alpha[0] = alpha_d;
    const float na = -(float)alpha_d;
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x, nth = (int64_t)gridDim.x * BLK;
    double acc = 0.0;
    auto one = [&](float2& rv, float2 av) -> float {
        rv.x = fmaf(na, av.x, rv.x); rv.y = fmaf(na, av.y, rv.y);
        return fmaf(rv.x, rv.x, rv.y * rv.y);
    };
    if (VEC) {
        float4* r4 = reinterpret_cast<float4*>(r);
        const float4* a4 = reinterpret_cast<const float4*>(Ap);
        const int64_t n2 = n >> 1;
        int64_t i = tid;
        while (i < n2) {
            float f = 0.f;
#pragma unroll 4
            for (int k = 0; k < 8 && i < n2; ++k, i += nth) {
                const float4 av = a4[i];
                const float4 rv = r4[i];
                float2 r0 = make_float2(rv.x, rv.y), r1 = make_float2(rv.z, rv.w);
                f += one(r0, make_float2(av.x, av.y));
                f += one(r1, make_float2(av.z, av.w));
                r4[i] = make_float4(r0.x, r0.y, r1.x, r1.y);
            }
            acc += (double)f;
        }
        if ((n & 1) && tid == 0) { float2 rv = r[n - 1]; acc += (double)one(rv, Ap[n - 1]); r[n - 1] = rv; }
    } else {
        for (int64_t i = tid; i < n; i += nth) { float2 rv = r[i]; acc += (double)one(rv, Ap[i]); r[i] = rv; }
    }
    block_store_partial(acc, rr_partials);
}

template <bool VEC>
__global__ void __launch_bounds__(BLK)
k_cg_step_xp(int64_t n, float2* __restrict__ x, float2* __restrict__ p, const float2* __restrict__ r,
             const double* __restrict__ rr_partials, int nparts, const double* __restrict__ d_alpha, const double* __restrict__ d_rr,
             double* __restrict__ d_rr_next, const double* __restrict__ d_r0, double* __restrict__ d_hist) {
    const double r2 = block_sum_partials(rr_partials, nparts);
    const double rr = d_rr[0];
    const float beta = (float)(rr == 0.0 ? 0.0 : r2 / rr);
    const float alpha = (float)d_alpha[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d_rr_next[0] = r2;
        const double r0 = d_r0[0];
        if (d_hist) d_hist[0] = r0 == 0.0 ? 0.0 : r2 / r0;        // (relative residual)^2 of this iteration
    }
    const int64_t tid = (int64_t)blockIdx.x * BLK + threadIdx.x, nth = (int64_t)gridDim.x * BLK;
    if (VEC) {
        float4* x4 = reinterpret_cast<float4*>(x);
        float4* p4 = reinterpret_cast<float4*>(p);
        const float4* r4 = reinterpret_cast<const float4*>(r);
        const int64_t n2 = n >> 1;
        for (int64_t i = tid; i < n2; i += nth) {
            const float4 pv = p4[i], rv = r4[i];
            float4 xv = x4[i];
            xv.x = fmaf(alpha, pv.x, xv.x); xv.y = fmaf(alpha, pv.y, xv.y); xv.z = fmaf(alpha, pv.z, xv.z); xv.w = fmaf(alpha, pv.w, xv.w);
            x4[i] = xv;
            p4[i] = make_float4(fmaf(beta, pv.x, rv.x), fmaf(beta, pv.y, rv.y), fmaf(beta, pv.z, rv.z), fmaf(beta, pv.w, rv.w));
        }
        if ((n & 1) && tid == 0) {
            const int64_t i = n - 1;
            const float2 pv = p[i], rv = r[i];
            float2 xv = x[i];
            xv.x = fmaf(alpha, pv.x, xv.x); xv.y = fmaf(alpha, pv.y, xv.y);
            x[i] = xv;
            p[i] = make_float2(fmaf(beta, pv.x, rv.x), fmaf(beta, pv.y, rv.y));
        }
    } else {
        for (int64_t i = tid; i < n; i += nth) {
            const float2 pv = p[i], rv = r[i];
            float2 xv = x[i];
            xv.x = fmaf(alpha, pv.x, xv.x); xv.y = fmaf(alpha, pv.y, xv.y);
            x[i] = xv;
            p[i] = make_float2(fmaf(beta, pv.x, rv.x), fmaf(beta, pv.y, rv.y));
        }
    }
}

inline int cg_grid(const ig_ctx* ctx, int64_t n) {
    int g = grid_for(ctx, (n + 1) / 2);
    return g > IG_MAX_RED_BLOCKS ? IG_MAX_RED_BLOCKS : g;
}

// tiny scalar programs on device-resident doubles (one thread): the glue between a solver's reductions and its updates
__global__ void k_scalar_ratio(double* __restrict__ out, const double* __restrict__ num, const double* __restrict__ den, double scale) {
    const double d = den[0];
    out[0] = d == 0.0 ? 0.0 : scale * num[0] / d;         // 0/0 of an exactly converged solver: a zero step, not NaN
}
// the same, forced to zero once gate_num < gate_tol * gate_den (CG: the step length after the residual passed the tolerance,
// so that iterations enqueued beyond convergence leave x and r alone)
__global__ void k_scalar_ratio_gated(double* __restrict__ out, const double* __restrict__ num, const double* __restrict__ den, double scale,
                                     const double* __restrict__ gate_num, const double* __restrict__ gate_den, double gate_tol) {
    const double d = den[0];
    const bool stop = !(gate_num[0] >= gate_tol * gate_den[0]);
    out[0] = (d == 0.0 || stop) ? 0.0 : scale * num[0] / d;
}
__global__ void k_scalar_copy(double* __restrict__ dst, const double* __restrict__ src, int count) {
    for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
}

// the same two-kernel deterministic reduction, result left on the device (d_out[0..1]); no host synchronisation
int reduce_dev(ig_ctx* ctx, bool dot, int64_t n, const void* x, const void* y, double* d_out) {
    if (int rc = ig_set_device(ctx)) return rc;
    if (n == 0) { IG_HIP(ctx, hipMemsetAsync(d_out, 0, 2 * sizeof(double), ctx->stream)); return IG_OK; }
    int g = grid_for(ctx, n);
    if (g > IG_MAX_RED_BLOCKS) g = IG_MAX_RED_BLOCKS;
    ig_prof_scope prof(ctx, dot ? "cdotc" : "scnrm2", (double)n * 8.0 * (dot ? 2 : 1));
    if (dot)
        hipLaunchKernelGGL(k_reduce<true>, dim3(g), dim3(BLK), 0, ctx->stream, n, (const float2*)x, (const float2*)y, ctx->d_partials);
    else
        hipLaunchKernelGGL(k_reduce<false>, dim3(g), dim3(BLK), 0, ctx->stream, n, (const float2*)x, (const float2*)x, ctx->d_partials);
    IG_LAUNCH_CHECK(ctx, "k_reduce");
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(BLK), 0, ctx->stream, g, ctx->d_partials, d_out);
    IG_LAUNCH_CHECK(ctx, "k_reduce_final");
    return IG_OK;
}

int reduce_common(ig_ctx* ctx, bool dot, int64_t n, const void* x, const void* y, double out[2]) {
    if (int rc = ig_set_device(ctx)) return rc;
    if (n == 0) { out[0] = out[1] = 0.0; return IG_OK; }
    int g = grid_for(ctx, n);
    if (g > IG_MAX_RED_BLOCKS) g = IG_MAX_RED_BLOCKS;
    ig_prof_scope prof(ctx, dot ? "cdotc" : "scnrm2", (double)n * 8.0 * (dot ? 2 : 1));
    if (dot)
        hipLaunchKernelGGL(k_reduce<true>, dim3(g), dim3(BLK), 0, ctx->stream, n,
                           (const float2*)x, (const float2*)y, ctx->d_partials);
    else
        hipLaunchKernelGGL(k_reduce<false>, dim3(g), dim3(BLK), 0, ctx->stream, n,
                           (const float2*)x, (const float2*)x, ctx->d_partials);
    IG_LAUNCH_CHECK(ctx, "k_reduce");
    // the final (re, im) pair lives in the extra slot behind the per-block partials
    double* d_out = ctx->d_partials + 2 * IG_MAX_RED_BLOCKS;
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(BLK), 0, ctx->stream, g, ctx->d_partials, d_out);
    IG_LAUNCH_CHECK(ctx, "k_reduce_final");
    IG_HIP(ctx, hipMemcpyAsync(ctx->h_result, d_out, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    IG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out[0] = ctx->h_result[0];
    out[1] = ctx->h_result[1];
    return IG_OK;
}

}  // namespace

extern "C" {

int ig_caxpby(ig_ctx* ctx, int64_t n, float br, float bi, void* y, float ar, float ai, const void* x) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_caxpby: ctx is NULL");
    IG_REQUIRE(ctx, n >= 0, "ig_caxpby: negative length");
    if (n == 0) return IG_OK;
    IG_REQUIRE(ctx, y != nullptr, "ig_caxpby: y is NULL");
    const bool a0 = (ar == 0.f && ai == 0.f);
    const bool b0 = (br == 0.f && bi == 0.f);
    const bool b1 = (br == 1.f && bi == 0.f);
    IG_REQUIRE(ctx, a0 || x != nullptr, "ig_caxpby: x is NULL");
    if (a0 && b1) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    if (a0 && b0) {
        IG_HIP(ctx, hipMemsetAsync(y, 0, (size_t)n * 8, ctx->stream));
        return IG_OK;
    }
    const int vec_ok = aligned16(y) && (a0 || aligned16(x));
    const int g = grid_for(ctx, vec_ok ? (n + 1) / 2 : n);
    const float2 a = make_float2(ar, ai), b = make_float2(br, bi);
    ig_prof_scope prof(ctx, "caxpby", (double)n * 8.0 * ((a0 ? 0 : 1) + (b0 ? 1 : 2)));
    float2* yp = (float2*)y;
    const float2* xp = (const float2*)x;
    if (a0)       hipLaunchKernelGGL(k_caxpby<3>, dim3(g), dim3(BLK), 0, ctx->stream, n, b, yp, a, xp, vec_ok);
    else if (b0)  hipLaunchKernelGGL(k_caxpby<0>, dim3(g), dim3(BLK), 0, ctx->stream, n, b, yp, a, xp, vec_ok);
    else if (b1)  hipLaunchKernelGGL(k_caxpby<1>, dim3(g), dim3(BLK), 0, ctx->stream, n, b, yp, a, xp, vec_ok);
    else          hipLaunchKernelGGL(k_caxpby<2>, dim3(g), dim3(BLK), 0, ctx->stream, n, b, yp, a, xp, vec_ok);
    IG_LAUNCH_CHECK(ctx, "k_caxpby");
    return IG_OK;
}

int ig_scalars(ig_ctx* ctx, double** d_slots, int* nslots) {
    IG_REQUIRE(ctx, ctx && d_slots, "ig_scalars: bad arguments");
    if (int rc = ig_set_device(ctx)) return rc;
    if (!ctx->d_scalars) {
        IG_HIP(ctx, hipMalloc((void**)&ctx->d_scalars, sizeof(double) * IG_NUM_SCALARS));
        ctx->scalars_bytes = sizeof(double) * IG_NUM_SCALARS;
        IG_HIP(ctx, hipMemsetAsync(ctx->d_scalars, 0, sizeof(double) * IG_NUM_SCALARS, ctx->stream));
    }
    *d_slots = ctx->d_scalars;
    if (nslots) *nslots = IG_NUM_SCALARS;
    return IG_OK;
}

int ig_caxpby_dev(ig_ctx* ctx, int64_t n, const double* d_beta, float beta_scale, void* y,
                  const double* d_alpha, float alpha_scale, const void* x) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_caxpby_dev: ctx is NULL");
    IG_REQUIRE(ctx, n >= 0, "ig_caxpby_dev: negative length");
    if (n == 0) return IG_OK;
    IG_REQUIRE(ctx, y != nullptr && x != nullptr, "ig_caxpby_dev: NULL vector");
    if (int rc = ig_set_device(ctx)) return rc;
    const int vec_ok = aligned16(y) && aligned16(x);
    const int g = grid_for(ctx, vec_ok ? (n + 1) / 2 : n);
    ig_prof_scope prof(ctx, "caxpby", (double)n * 8.0 * 3);
    const float2 a = make_float2(alpha_scale, 0.f), b = make_float2(beta_scale, 0.f);
    if (!d_beta && beta_scale == 1.f)
        hipLaunchKernelGGL(k_caxpby<1>, dim3(g), dim3(BLK), 0, ctx->stream, n, b, (float2*)y, a, (const float2*)x, vec_ok, d_beta, d_alpha);
    else
        hipLaunchKernelGGL(k_caxpby<2>, dim3(g), dim3(BLK), 0, ctx->stream, n, b, (float2*)y, a, (const float2*)x, vec_ok, d_beta, d_alpha);
    IG_LAUNCH_CHECK(ctx, "k_caxpby(dev)");
    return IG_OK;
}

int ig_cdotc_dev(ig_ctx* ctx, int64_t n, const void* x, const void* y, double* d_out) {
    IG_REQUIRE(ctx, ctx && d_out, "ig_cdotc_dev: bad arguments");
    IG_REQUIRE(ctx, n >= 0 && (n == 0 || (x && y)), "ig_cdotc_dev: bad vector arguments");
    return reduce_dev(ctx, true, n, x, y, d_out);
}

int ig_scnrm2sq_dev(ig_ctx* ctx, int64_t n, const void* x, double* d_out) {
    IG_REQUIRE(ctx, ctx && d_out, "ig_scnrm2sq_dev: bad arguments");
    IG_REQUIRE(ctx, n >= 0 && (n == 0 || x), "ig_scnrm2sq_dev: bad vector argument");
    return reduce_dev(ctx, false, n, x, nullptr, d_out);
}

int ig_scalar_ratio(ig_ctx* ctx, double* d_out, const double* d_num, const double* d_den, double scale) {
    IG_REQUIRE(ctx, ctx && d_out && d_num && d_den, "ig_scalar_ratio: bad arguments");
    if (int rc = ig_set_device(ctx)) return rc;
    hipLaunchKernelGGL(k_scalar_ratio, dim3(1), dim3(1), 0, ctx->stream, d_out, d_num, d_den, scale);
    IG_LAUNCH_CHECK(ctx, "k_scalar_ratio");
    return IG_OK;
}

int ig_scalar_ratio_gated(ig_ctx* ctx, double* d_out, const double* d_num, const double* d_den, double scale,
                          const double* d_gate_num, const double* d_gate_den, double gate_tol) {
    IG_REQUIRE(ctx, ctx && d_out && d_num && d_den && d_gate_num && d_gate_den, "ig_scalar_ratio_gated: bad arguments");
    if (int rc = ig_set_device(ctx)) return rc;
    hipLaunchKernelGGL(k_scalar_ratio_gated, dim3(1), dim3(1), 0, ctx->stream, d_out, d_num, d_den, scale, d_gate_num, d_gate_den, gate_tol);
    IG_LAUNCH_CHECK(ctx, "k_scalar_ratio_gated");
    return IG_OK;
}

int ig_scalar_copy(ig_ctx* ctx, double* d_dst, const double* d_src, int64_t count) {
    IG_REQUIRE(ctx, ctx && d_dst && d_src && count >= 0 && count <= 4096, "ig_scalar_copy: bad arguments");
    if (count == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    hipLaunchKernelGGL(k_scalar_copy, dim3(1), dim3(64), 0, ctx->stream, d_dst, d_src, (int)count);
    IG_LAUNCH_CHECK(ctx, "k_scalar_copy");
    return IG_OK;
}

// ---- the fused CG iteration (see k_cg_dot / k_cg_step_r / k_cg_step_xp).  The three calls of one iteration share the block
// partials in the context's reduction scratch: dot partials in its first half, ||r||^2 partials in its second; they must be
// issued in this order, with the same n, on this context.
int ig_cg_dot(ig_ctx* ctx, int64_t n, const void* p, void* Ap, float lamda) {
    IG_REQUIRE(ctx, ctx && n > 0 && p && Ap, "ig_cg_dot: bad arguments");
    if (int rc = ig_set_device(ctx)) return rc;
    const int g = cg_grid(ctx, n);
    const bool vec = aligned16(p) && aligned16(Ap);
    ig_prof_scope prof(ctx, "cg_dot", (double)n * 8.0 * (lamda != 0.f ? 3 : 2));
    double* parts = ctx->d_partials;
#define IG_CG_DOT(V_, L_) hipLaunchKernelGGL((k_cg_dot<V_, L_>), dim3(g), dim3(BLK), 0, ctx->stream, n, (const float2*)p, (float2*)Ap, lamda, parts)
    if (vec) { if (lamda != 0.f) IG_CG_DOT(true, true); else IG_CG_DOT(true, false); }
    else     { if (lamda != 0.f) IG_CG_DOT(false, true); else IG_CG_DOT(false, false); }
#undef IG_CG_DOT
    IG_LAUNCH_CHECK(ctx, "k_cg_dot");
    return IG_OK;
}

int ig_cg_step_r(ig_ctx* ctx, int64_t n, void* r, const void* Ap, const double* d_rr, const double* d_r0, double tol2, double* d_alpha) {
    IG_REQUIRE(ctx, ctx && n > 0 && r && Ap && d_rr && d_r0 && d_alpha, "ig_cg_step_r: bad arguments");
    if (int rc = ig_set_device(ctx)) return rc;
    const int g = cg_grid(ctx, n);
    const bool vec = aligned16(r) && aligned16(Ap);
    ig_prof_scope prof(ctx, "cg_step_r", (double)n * 8.0 * 3);
    double* parts = ctx->d_partials;
    if (vec) hipLaunchKernelGGL(k_cg_step_r<true>, dim3(g), dim3(BLK), 0, ctx->stream, n, (float2*)r, (const float2*)Ap, parts, g, d_rr, d_r0, tol2, d_alpha, parts + IG_MAX_RED_BLOCKS);
    else     hipLaunchKernelGGL(k_cg_step_r<false>, dim3(g), dim3(BLK), 0, ctx->stream, n, (float2*)r, (const float2*)Ap, parts, g, d_rr, d_r0, tol2, d_alpha, parts + IG_MAX_RED_BLOCKS);
    IG_LAUNCH_CHECK(ctx, "k_cg_step_r");
    return IG_OK;
}

int ig_cg_step_xp(ig_ctx* ctx, int64_t n, void* x, void* p, const void* r, const double* d_alpha, const double* d_rr, double* d_rr_next,
                  const double* d_r0, double* d_hist) {
    IG_REQUIRE(ctx, ctx && n > 0 && x && p && r && d_alpha && d_rr && d_rr_next && d_r0 && d_rr != d_rr_next, "ig_cg_step_xp: bad arguments (rr and rr_next must be two slots)");
    if (int rc = ig_set_device(ctx)) return rc;
    const int g = cg_grid(ctx, n);
    const bool vec = aligned16(x) && aligned16(p) && aligned16(r);
    ig_prof_scope prof(ctx, "cg_step_xp", (double)n * 8.0 * 5);
    const double* parts = ctx->d_partials + IG_MAX_RED_BLOCKS;
    if (vec) hipLaunchKernelGGL(k_cg_step_xp<true>, dim3(g), dim3(BLK), 0, ctx->stream, n, (float2*)x, (float2*)p, (const float2*)r, parts, g, d_alpha, d_rr, d_rr_next, d_r0, d_hist);
    else     hipLaunchKernelGGL(k_cg_step_xp<false>, dim3(g), dim3(BLK), 0, ctx->stream, n, (float2*)x, (float2*)p, (const float2*)r, parts, g, d_alpha, d_rr, d_rr_next, d_r0, d_hist);
    IG_LAUNCH_CHECK(ctx, "k_cg_step_xp");
    return IG_OK;
}

int ig_scalar_read(ig_ctx* ctx, const double* d_src, int64_t count, double* host) {
    IG_REQUIRE(ctx, ctx && d_src && host && count >= 0, "ig_scalar_read: bad arguments");
    if (count == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    IG_HIP(ctx, hipMemcpyAsync(host, d_src, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    IG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IG_OK;
}

int ig_cscal(ig_ctx* ctx, int64_t n, float ar, float ai, void* x) {
    // x = alpha * x  ==  axpby with beta := alpha, alpha := 0
    return ig_caxpby(ctx, n, ar, ai, x, 0.f, 0.f, nullptr);
}

int ig_cdotc(ig_ctx* ctx, int64_t n, const void* x, const void* y, double out[2]) {
    IG_REQUIRE(ctx, ctx && out, "ig_cdotc: bad arguments");
    IG_REQUIRE(ctx, n >= 0 && (n == 0 || (x && y)), "ig_cdotc: bad vector arguments");
    return reduce_common(ctx, true, n, x, y, out);
}

int ig_scnrm2sq(ig_ctx* ctx, int64_t n, const void* x, double* out) {
    IG_REQUIRE(ctx, ctx && out, "ig_scnrm2sq: bad arguments");
    IG_REQUIRE(ctx, n >= 0 && (n == 0 || x), "ig_scnrm2sq: bad vector argument");
    double r[2];
    int rc = reduce_common(ctx, false, n, x, nullptr, r);
    if (rc == IG_OK) *out = r[0];
    return rc;
}

int ig_csum_cols(ig_ctx* ctx, int64_t rows, int64_t ncols, const void* X, int64_t ldx,
                 float ar, float ai, float br, float bi, void* y) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_csum_cols: ctx is NULL");
    IG_REQUIRE(ctx, rows >= 0 && ncols >= 0, "ig_csum_cols: negative dimension");
    if (rows == 0) return IG_OK;
    IG_REQUIRE(ctx, y && (ncols == 0 || X), "ig_csum_cols: NULL pointer");
    IG_REQUIRE(ctx, ncols <= 1 || ldx >= rows, "ig_csum_cols: ldx smaller than rows");
    if (int rc = ig_set_device(ctx)) return rc;
    const bool b0 = (br == 0.f && bi == 0.f);
    ig_prof_scope prof(ctx, "csum_cols", (double)rows * 8.0 * (ncols + (b0 ? 1 : 2)));
    const int g = grid_for(ctx, rows);
    if (b0) hipLaunchKernelGGL(k_csum_cols<true>, dim3(g), dim3(BLK), 0, ctx->stream, rows, ncols, (const float2*)X, ldx,
                               make_float2(ar, ai), make_float2(br, bi), (float2*)y);
    else    hipLaunchKernelGGL(k_csum_cols<false>, dim3(g), dim3(BLK), 0, ctx->stream, rows, ncols, (const float2*)X, ldx,
                               make_float2(ar, ai), make_float2(br, bi), (float2*)y);
    IG_LAUNCH_CHECK(ctx, "k_csum_cols");
    return IG_OK;
}

int ig_csum_il(ig_ctx* ctx, int64_t rows, int64_t ncols, const void* X_il,
               float ar, float ai, float br, float bi, void* y) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_csum_il: ctx is NULL");
    IG_REQUIRE(ctx, rows >= 0 && ncols >= 0, "ig_csum_il: negative dimension");
    if (rows == 0) return IG_OK;
    IG_REQUIRE(ctx, y && (ncols == 0 || X_il), "ig_csum_il: NULL pointer");
    if (int rc = ig_set_device(ctx)) return rc;
    const bool b0 = (br == 0.f && bi == 0.f);
    const bool vec = ncols % 2 == 0 && aligned16(X_il);
    ig_prof_scope prof(ctx, "csum_cols", (double)rows * 8.0 * (ncols + (b0 ? 1 : 2)));
    const float2 a = make_float2(ar, ai), b = make_float2(br, bi);
    if (vec && (ncols == 2 || ncols == 4 || ncols == 8 || ncols == 16 || ncols == 32)) {
        const int gp = grid_for(ctx, rows * (ncols / 2));
#define IG_CSUM_P(P_) do { if (b0) hipLaunchKernelGGL((k_csum_il_parts<true, P_>), dim3(gp), dim3(BLK), 0, ctx->stream, rows, (const float2*)X_il, a, b, (float2*)y); \
                           else hipLaunchKernelGGL((k_csum_il_parts<false, P_>), dim3(gp), dim3(BLK), 0, ctx->stream, rows, (const float2*)X_il, a, b, (float2*)y); } while (0)
        switch (ncols) { case 2: IG_CSUM_P(1); break; case 4: IG_CSUM_P(2); break; case 8: IG_CSUM_P(4); break;
                         case 16: IG_CSUM_P(8); break; default: IG_CSUM_P(16); break; }
#undef IG_CSUM_P
        IG_LAUNCH_CHECK(ctx, "k_csum_il_parts");
        return IG_OK;
    }
    const int g = grid_for(ctx, rows);
    if (b0 && vec)       hipLaunchKernelGGL((k_csum_il<true, true>),   dim3(g), dim3(BLK), 0, ctx->stream, rows, ncols, (const float2*)X_il, a, b, (float2*)y);
    else if (b0)         hipLaunchKernelGGL((k_csum_il<true, false>),  dim3(g), dim3(BLK), 0, ctx->stream, rows, ncols, (const float2*)X_il, a, b, (float2*)y);
    else if (vec)        hipLaunchKernelGGL((k_csum_il<false, true>),  dim3(g), dim3(BLK), 0, ctx->stream, rows, ncols, (const float2*)X_il, a, b, (float2*)y);
    else                 hipLaunchKernelGGL((k_csum_il<false, false>), dim3(g), dim3(BLK), 0, ctx->stream, rows, ncols, (const float2*)X_il, a, b, (float2*)y);
    IG_LAUNCH_CHECK(ctx, "k_csum_il");
    return IG_OK;
}

int ig_cmax(ig_ctx* ctx, int64_t nfloats, float val, void* arr) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_cmax: ctx is NULL");
    IG_REQUIRE(ctx, nfloats >= 0, "ig_cmax: negative length");
    if (nfloats == 0) return IG_OK;
    IG_REQUIRE(ctx, arr != nullptr, "ig_cmax: NULL pointer");
    if (int rc = ig_set_device(ctx)) return rc;
    const int vec_ok = aligned16(arr);
    const int g = grid_for(ctx, vec_ok ? (nfloats + 3) / 4 : nfloats);
    hipLaunchKernelGGL(k_smax, dim3(g), dim3(BLK), 0, ctx->stream, nfloats, val, (float*)arr, vec_ok);
    IG_LAUNCH_CHECK(ctx, "k_smax");
    return IG_OK;
}

}  // extern "C"
