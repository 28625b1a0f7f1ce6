// CSR x dense-panel SpMM for complex64 on gfx950.
//
//   gather  (forward, and adjoint via a transposed copy):
//       Y[row, :] = alpha * sum_p op(val[p]) * X[col[p], :] + beta * Y[row, :]
//   scatter (adjoint straight from A's CSR):
//       Y[col[p], :] += conj(val[p]) * alpha * X[row, :]      after  Y *= beta
//
// Reference contract: Backend.ccsrmm (indigo/backends/backend.py:514-519),
// numpy oracle np.py:120-127, native algorithms _customcpu.c:14-114 and
// _customgpu.cu:49-81.  This is a new design for 64-wide wavefronts:
//
// A wavefront is cut into RPW "row slots" of NL x CL lanes
//     lane = c + CL * (i + NL * r),   c: panel-column lane, i: nonzero lane, r: row slot
// so one wave works on RPW rows at a time; inside a slot the NL nonzero-lanes
// stride over the row's nonzeros and the CL column-lanes cover the panel
// columns (for N > 64 the slot loops over column chunks).  Partial sums are
// folded across the nonzero-lanes with __shfl_xor (no LDS).  CL is the
// smallest power of two >= N (<= 64) and NL the power of two nearest to the
// mean row length that still fits, chosen on the host per call: wide panels
// get one row per wave, diagonal-like matrices with one column get a row per
// lane, and in both cases loads of val/col are contiguous across the wave.
//
// Everything here is bandwidth/latency bound integer+fp32 work: no MFMA.
#include "ig_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <vector>
#include <cstring>
#include <cstdlib>

namespace {

constexpr int BLK = 256;
constexpr int WAVES_PER_BLOCK = BLK / 64;

// Blocks are dealt round-robin to the 8 XCDs (each with a private L2).  Give
// each XCD a contiguous range of row blocks so neighbouring rows -- which in
// gridding matrices touch neighbouring panel rows -- share one L2.  Speed
// only; the map is a bijection for every grid size.
__device__ __forceinline__ int64_t xcd_block(int64_t b, int64_t nb) {
    const int64_t q = nb >> 3, rem = nb & 7;
    const int64_t xcd = b & 7, idx = b >> 3;
    return (xcd < rem) ? xcd * (q + 1) + idx : rem * (q + 1) + (xcd - rem) * q + idx;
}

// BMODE: 0 => beta == 0 (Y not read), 1 => general beta
template <int CL, int NL, bool CONJ, int BMODE>
__global__ void __launch_bounds__(BLK)
k_csrmm_gather(int64_t M, int64_t N,
               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
               const float2* __restrict__ vals,
               const float2* __restrict__ X, int64_t ldx,
               float2* __restrict__ Y, int64_t ldy,
               float2 alpha, float2 beta, int xcd_remap) {
    constexpr int RPW = 64 / (CL * NL);
    const int lane = threadIdx.x & 63;
    const int c = lane % CL;
    const int i = (lane / CL) % NL;
    const int r = lane / (CL * NL);
    const int64_t blk = xcd_remap ? xcd_block(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
    const int64_t wave = blk * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + r;
    const bool row_ok = row < M;

    int32_t p0 = 0, p1 = 0;
    if (row_ok) { p0 = rowptr[row]; p1 = rowptr[row + 1]; }

    for (int64_t jb = 0; jb < N; jb += CL) {
        const int64_t j = jb + c;
        const bool col_ok = j < N;
        const float2* __restrict__ xcol = X + (col_ok ? j : 0) * ldx;
        float2 acc = make_float2(0.f, 0.f);
        int32_t p = p0 + i;
        // two nonzeros per trip keep two panel gathers in flight per lane
        for (; p + NL < p1; p += 2 * NL) {
            const int32_t k0 = colind[p], k1 = colind[p + NL];
            const float2 v0 = vals[p], v1 = vals[p + NL];
            const float2 x0 = xcol[k0], x1 = xcol[k1];
            if (CONJ) { acc = cadd(acc, cmulc(v0, x0)); acc = cadd(acc, cmulc(v1, x1)); }
            else      { cfma(acc, v0, x0); cfma(acc, v1, x1); }
        }
        if (p < p1) {
            const int32_t k0 = colind[p];
            const float2 v0 = vals[p];
            const float2 x0 = xcol[k0];
            if (CONJ) acc = cadd(acc, cmulc(v0, x0));
            else      cfma(acc, v0, x0);
        }
#pragma unroll
        for (int off = CL * NL / 2; off >= CL; off >>= 1) {
            acc.x += __shfl_xor(acc.x, off, 64);
            acc.y += __shfl_xor(acc.y, off, 64);
        }
        if (i == 0 && row_ok && col_ok) {
            float2* yp = Y + j * ldy + row;
            float2 out = cmul(alpha, acc);
            if (BMODE == 1) cfma(out, beta, *yp);
            *yp = out;
        }
    }
}

// Adjoint as a scatter over A's rows.  ATOMIC=false requires every column of
// A to hold at most one nonzero (exwrite), so no two lanes ever update the
// same element.
template <int CL, int NL, bool ATOMIC>
__global__ void __launch_bounds__(BLK)
k_csrmm_scatter(int64_t M, int64_t N,
                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                const float2* __restrict__ vals,
                const float2* __restrict__ X, int64_t ldx,
                float2* __restrict__ Y, int64_t ldy,
                float2 alpha, int xcd_remap) {
    constexpr int RPW = 64 / (CL * NL);
    const int lane = threadIdx.x & 63;
    const int c = lane % CL;
    const int i = (lane / CL) % NL;
    const int r = lane / (CL * NL);
    const int64_t blk = xcd_remap ? xcd_block(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
    const int64_t wave = blk * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + r;
    if (row >= M) return;
    const int32_t p0 = rowptr[row], p1 = rowptr[row + 1];
    if (p0 == p1) return;

    for (int64_t jb = 0; jb < N; jb += CL) {
        const int64_t j = jb + c;
        if (j >= N) continue;
        const float2 ax = cmul(alpha, X[j * ldx + row]);
        float2* __restrict__ ycol = Y + j * ldy;
        for (int32_t p = p0 + i; p < p1; p += NL) {
            const int32_t k = colind[p];
            const float2 t = cmulc(vals[p], ax);   // conj(val) * alpha * x
            if (ATOMIC) {
                unsafeAtomicAdd(&ycol[k].x, t.x);
                unsafeAtomicAdd(&ycol[k].y, t.y);
            } else {
                float2 y = ycol[k];
                y.x += t.x; y.y += t.y;
                ycol[k] = y;
            }
        }
    }
}

// Y(rows x N, leading dim ld) *= beta   (beta == 0 writes zeros without reading)
template <bool ZERO>
__global__ void __launch_bounds__(BLK)
k_panel_scale(int64_t rows, int64_t N, float2* __restrict__ Y, int64_t ld, float2 beta) {
    const int64_t j = blockIdx.y;
    float2* __restrict__ col = Y + j * ld;
    for (int64_t k = (int64_t)blockIdx.x * BLK + threadIdx.x; k < rows; k += (int64_t)gridDim.x * BLK) {
        if (ZERO) col[k] = make_float2(0.f, 0.f);
        else      col[k] = cmul(beta, col[k]);
    }
}

inline int pow2_ceil(int64_t v, int cap) {
    int p = 1;
    while (p < v && p < cap) p <<= 1;
    return p;
}

struct Shape { int CL, NL; };

inline Shape pick_shape(int64_t rows, int64_t N, int64_t nnz) {
    Shape s;
    s.CL = pow2_ceil(N, 64);
    const int64_t mean = rows > 0 ? (nnz + rows - 1) / rows : 1;
    s.NL = pow2_ceil(mean > 0 ? mean : 1, 64 / s.CL);
    return s;
}

inline bool env_flag(const char* name, bool dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    return e[0] != '0';
}

template <bool CONJ>
int launch_gather(ig_ctx* ctx, int64_t rows, int64_t N, int64_t nnz,
                  const int32_t* rowptr, const int32_t* colind, const float2* vals,
                  const float2* X, int64_t ldx, float2* Y, int64_t ldy, float2 alpha, float2 beta) {
    const Shape s = pick_shape(rows, N, nnz);
    const int rpw = 64 / (s.CL * s.NL);
    const int64_t waves = (rows + rpw - 1) / rpw;
    const int64_t blocks = (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    IG_REQUIRE(ctx, blocks <= 0x7fffffffLL, "csrmm: matrix too large for one launch (%lld blocks)", (long long)blocks);
    const int xcd = env_flag("INDIGO_HIP_SPMM_XCD", true) ? 1 : 0;
    const bool b0 = (beta.x == 0.f && beta.y == 0.f);
    ig_prof_scope prof(ctx, CONJ ? "csrmm_gather_conj" : "csrmm_gather");
#define IG_GATHER(CL_, NL_)                                                                        \
    do {                                                                                           \
        if (b0) hipLaunchKernelGGL((k_csrmm_gather<CL_, NL_, CONJ, 0>), dim3((unsigned)blocks),    \
                    dim3(BLK), 0, ctx->stream, rows, N, rowptr, colind, vals, X, ldx, Y, ldy,      \
                    alpha, beta, xcd);                                                             \
        else    hipLaunchKernelGGL((k_csrmm_gather<CL_, NL_, CONJ, 1>), dim3((unsigned)blocks),    \
                    dim3(BLK), 0, ctx->stream, rows, N, rowptr, colind, vals, X, ldx, Y, ldy,      \
                    alpha, beta, xcd);                                                             \
    } while (0)
#define IG_NL_SWITCH(CL_, MACRO)                                                                   \
    switch (s.NL) {                                                                                \
        case 1:  MACRO(CL_, 1); break;                                                             \
        case 2:  if constexpr (CL_ * 2  <= 64) { MACRO(CL_, 2);  } break;                          \
        case 4:  if constexpr (CL_ * 4  <= 64) { MACRO(CL_, 4);  } break;                          \
        case 8:  if constexpr (CL_ * 8  <= 64) { MACRO(CL_, 8);  } break;                          \
        case 16: if constexpr (CL_ * 16 <= 64) { MACRO(CL_, 16); } break;                          \
        case 32: if constexpr (CL_ * 32 <= 64) { MACRO(CL_, 32); } break;                          \
        case 64: if constexpr (CL_ * 64 <= 64) { MACRO(CL_, 64); } break;                          \
    }
#define IG_CL_SWITCH(MACRO)                                                                        \
    switch (s.CL) {                                                                                \
        case 1:  IG_NL_SWITCH(1, MACRO)  break;                                                    \
        case 2:  IG_NL_SWITCH(2, MACRO)  break;                                                    \
        case 4:  IG_NL_SWITCH(4, MACRO)  break;                                                    \
        case 8:  IG_NL_SWITCH(8, MACRO)  break;                                                    \
        case 16: IG_NL_SWITCH(16, MACRO) break;                                                    \
        case 32: IG_NL_SWITCH(32, MACRO) break;                                                    \
        case 64: IG_NL_SWITCH(64, MACRO) break;                                                    \
    }
    IG_CL_SWITCH(IG_GATHER)
#undef IG_GATHER
    IG_LAUNCH_CHECK(ctx, "k_csrmm_gather");
    return IG_OK;
}

int launch_panel_scale(ig_ctx* ctx, int64_t rows, int64_t N, float2* Y, int64_t ld, float2 beta) {
    if (beta.x == 1.f && beta.y == 0.f) return IG_OK;
    if (rows == 0 || N == 0) return IG_OK;
    IG_REQUIRE(ctx, N <= 65535, "csrmm: panel has too many columns (%lld) for the beta pre-pass", (long long)N);
    int64_t gx = (rows + BLK - 1) / BLK;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (gx > cap) gx = cap;
    const bool zero = (beta.x == 0.f && beta.y == 0.f);
    ig_prof_scope prof(ctx, "panel_scale", (double)rows * (double)N * 8.0 * (zero ? 1 : 2));
    if (zero && ld == rows) {
        IG_HIP(ctx, hipMemsetAsync(Y, 0, (size_t)rows * (size_t)N * 8, ctx->stream));
        return IG_OK;
    }
    if (zero) hipLaunchKernelGGL(k_panel_scale<true>,  dim3((unsigned)gx, (unsigned)N), dim3(BLK), 0, ctx->stream, rows, N, Y, ld, beta);
    else      hipLaunchKernelGGL(k_panel_scale<false>, dim3((unsigned)gx, (unsigned)N), dim3(BLK), 0, ctx->stream, rows, N, Y, ld, beta);
    IG_LAUNCH_CHECK(ctx, "k_panel_scale");
    return IG_OK;
}

template <bool ATOMIC>
int launch_scatter(ig_ctx* ctx, int64_t M, int64_t N, int64_t nnz,
                   const int32_t* rowptr, const int32_t* colind, const float2* vals,
                   const float2* X, int64_t ldx, float2* Y, int64_t ldy, float2 alpha) {
    const Shape s = pick_shape(M, N, nnz);
    const int rpw = 64 / (s.CL * s.NL);
    const int64_t waves = (M + rpw - 1) / rpw;
    const int64_t blocks = (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    IG_REQUIRE(ctx, blocks <= 0x7fffffffLL, "csrmm: matrix too large for one launch (%lld blocks)", (long long)blocks);
    const int xcd = env_flag("INDIGO_HIP_SPMM_XCD", true) ? 1 : 0;
    ig_prof_scope prof(ctx, ATOMIC ? "csrmm_scatter_atomic" : "csrmm_scatter_exwrite");
#define IG_SCATTER(CL_, NL_)                                                                       \
    hipLaunchKernelGGL((k_csrmm_scatter<CL_, NL_, ATOMIC>), dim3((unsigned)blocks), dim3(BLK), 0,  \
                       ctx->stream, M, N, rowptr, colind, vals, X, ldx, Y, ldy, alpha, xcd)
    IG_CL_SWITCH(IG_SCATTER)
#undef IG_SCATTER
    IG_LAUNCH_CHECK(ctx, "k_csrmm_scatter");
    return IG_OK;
}

int check_panel_args(ig_ctx* ctx, const char* who, int64_t xrows, int64_t yrows, int64_t N, int64_t nnz,
                     const void* vals, const int32_t* colind, const int32_t* rowptr,
                     const void* X, int64_t ldx, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, xrows >= 0 && yrows >= 0 && N >= 0 && nnz >= 0, "%s: negative dimension", who);
    IG_REQUIRE(ctx, nnz <= 0x7fffffffLL, "%s: nnz %lld exceeds int32 row pointers", who, (long long)nnz);
    IG_REQUIRE(ctx, xrows <= 0x7fffffffLL && yrows <= 0x7fffffffLL, "%s: dimension exceeds int32 column indices", who);
    IG_REQUIRE(ctx, rowptr != nullptr, "%s: rowptr is NULL", who);
    IG_REQUIRE(ctx, nnz == 0 || (vals && colind), "%s: vals/colind NULL with nnz > 0", who);
    IG_REQUIRE(ctx, N == 0 || yrows == 0 || Y != nullptr, "%s: Y is NULL", who);
    IG_REQUIRE(ctx, N == 0 || xrows == 0 || X != nullptr, "%s: X is NULL", who);
    IG_REQUIRE(ctx, N <= 1 || ldx >= xrows, "%s: ldx (%lld) smaller than X rows (%lld)", who, (long long)ldx, (long long)xrows);
    IG_REQUIRE(ctx, N <= 1 || ldy >= yrows, "%s: ldy (%lld) smaller than Y rows (%lld)", who, (long long)ldy, (long long)yrows);
    return IG_OK;
}

}  // namespace

extern "C" {

int ig_ccsrmm(ig_ctx* ctx, int adjoint, int exwrite,
              int64_t M, int64_t K, int64_t N, int64_t nnz,
              float ar, float ai, const void* vals, const int32_t* colind, const int32_t* rowptr,
              const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm: ctx is NULL");
    const int64_t xrows = adjoint ? M : K, yrows = adjoint ? K : M;
    if (int rc = check_panel_args(ctx, "ig_ccsrmm", xrows, yrows, N, nnz, vals, colind, rowptr, X, ldx, Y, ldy)) return rc;
    if (N == 0 || yrows == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const float2 alpha = make_float2(ar, ai), beta = make_float2(br, bi);
    if (!adjoint) {
        return launch_gather<false>(ctx, M, N, nnz, rowptr, colind, (const float2*)vals,
                                    (const float2*)X, ldx, (float2*)Y, ldy, alpha, beta);
    }
    if (int rc = launch_panel_scale(ctx, K, N, (float2*)Y, ldy, beta)) return rc;
    if (nnz == 0 || M == 0 || (ar == 0.f && ai == 0.f)) return IG_OK;
    if (exwrite)
        return launch_scatter<false>(ctx, M, N, nnz, rowptr, colind, (const float2*)vals,
                                     (const float2*)X, ldx, (float2*)Y, ldy, alpha);
    return launch_scatter<true>(ctx, M, N, nnz, rowptr, colind, (const float2*)vals,
                                (const float2*)X, ldx, (float2*)Y, ldy, alpha);
}

int ig_ccsrmm_t(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                float ar, float ai, const void* vals_t, const int32_t* colind_t, const int32_t* rowptr_t,
                const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_t: ctx is NULL");
    if (int rc = check_panel_args(ctx, "ig_ccsrmm_t", M, K, N, nnz, vals_t, colind_t, rowptr_t, X, ldx, Y, ldy)) return rc;
    if (N == 0 || K == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    return launch_gather<true>(ctx, K, N, nnz, rowptr_t, colind_t, (const float2*)vals_t,
                               (const float2*)X, ldx, (float2*)Y, ldy,
                               make_float2(ar, ai), make_float2(br, bi));
}

// ---- host-side structure analysis -------------------------------------------

int ig_csr_inspect(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K,
                   int64_t* nzrow, int64_t* nzcol, int* exwrite) {
    if (!rowptr || !nzrow || !nzcol || !exwrite || M < 0 || K < 0)
        return ig_fail(nullptr, IG_ERR_ARG, "ig_csr_inspect: bad arguments");
    std::vector<uint8_t> seen((size_t)K, 0);   // 0, 1, or 2 (= "more than one")
    int64_t rows = 0;
    for (int64_t m = 0; m < M; ++m) {
        const int32_t b = rowptr[m], e = rowptr[m + 1];
        if (e < b) return ig_fail(nullptr, IG_ERR_ARG, "ig_csr_inspect: rowptr not monotone at row %lld", (long long)m);
        if (e > b) ++rows;
        for (int32_t p = b; p < e; ++p) {
            const int32_t k = colind[p];
            if (k < 0 || k >= K)
                return ig_fail(nullptr, IG_ERR_ARG, "ig_csr_inspect: column index %d out of range at nnz %d", k, p);
            if (seen[k] < 2) ++seen[k];
        }
    }
    int64_t cols = 0;
    int exw = 1;
    for (int64_t k = 0; k < K; ++k) {
        if (seen[k]) ++cols;
        if (seen[k] > 1) exw = 0;
    }
    *nzrow = rows; *nzcol = cols; *exwrite = exw;
    return IG_OK;
}

int ig_csr_transpose(int64_t M, int64_t K, int64_t nnz,
                     const int32_t* rowptr, const int32_t* colind, const void* vals,
                     int32_t* rowptr_t, int32_t* colind_t, void* vals_t) {
    if (M < 0 || K < 0 || nnz < 0 || !rowptr || !rowptr_t || (nnz > 0 && (!colind || !vals || !colind_t || !vals_t)))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_csr_transpose: bad arguments");
    if (nnz > 0x7fffffffLL || M > 0x7fffffffLL || K > 0x7fffffffLL)
        return ig_fail(nullptr, IG_ERR_ARG, "ig_csr_transpose: dimensions exceed int32 indexing");
    if (rowptr[M] - rowptr[0] != nnz)
        return ig_fail(nullptr, IG_ERR_ARG, "ig_csr_transpose: rowptr[M]-rowptr[0] != nnz");
    const uint64_t* v = (const uint64_t*)vals;   // complex64 moved as 8-byte words
    uint64_t* vt = (uint64_t*)vals_t;
    std::memset(rowptr_t, 0, sizeof(int32_t) * (size_t)(K + 1));
    const int32_t base = rowptr[0];
    for (int64_t p = 0; p < nnz; ++p) {
        const int32_t k = colind[base + p];
        if (k < 0 || k >= K) return ig_fail(nullptr, IG_ERR_ARG, "ig_csr_transpose: column index out of range");
        ++rowptr_t[k + 1];
    }
    for (int64_t k = 0; k < K; ++k) rowptr_t[k + 1] += rowptr_t[k];
    std::vector<int32_t> cursor(rowptr_t, rowptr_t + K);
    for (int64_t m = 0; m < M; ++m) {
        for (int32_t p = rowptr[m]; p < rowptr[m + 1]; ++p) {
            const int32_t k = colind[p];
            const int32_t q = cursor[k]++;
            colind_t[q] = (int32_t)m;
            vt[q] = v[p];
        }
    }
    return IG_OK;
}

}  // extern "C"
