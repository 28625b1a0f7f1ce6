// The remaining leaves of the Backend contract, outside the SENSE tree but part of the plugin surface
// (SURVEY 8f rank 3): matrix of ones, diagonal-storage (DIA) sparse matrices, dense matrices.
//
//   onemm   Y = beta*Y + alpha * ones(M, K) * X        indigo/backends/backend.py:528-533, np.py:94-97,
//                                                      CUDA kernel _customgpu.cu:15-47
//   cdiamm  Y = beta*Y + alpha * op(A) * X, A in DIA   backend.py:521-526, np.py:129-136, _customgpu.cu:83-143
//   cgemm   Y = beta*Y + alpha * op(M) * X, M dense    backend.py:481-485, np.py:76-87, cuBLAS cgemm cuda.py:314-360
//           (csymm = the same product with a real symmetric M, from the left or from the right: np.py:89-90)
//
// All are HBM-bound for the shapes the reference uses them on (tall panels, few columns), so: one pass over the
// panels, coalesced along the rows, beta == 0 never reads Y.  The dense product is an LDS-tiled fp32 kernel -- these
// are small factors (temporal bases, phase-space kernels: tens to hundreds of columns); no MFMA.
#include "ig_common.h"

namespace {

constexpr int BLK = 256;

// ---- ones ---------------------------------------------------------------------------------------------
// one workgroup per panel column: column sum in double (a tall column of positive values loses digits in float),
// then the broadcast.  K rows in, M rows out.
template <bool B0>
__global__ void __launch_bounds__(BLK)
k_onemm(int64_t M, int64_t K, float2 alpha, const float2* __restrict__ X, int64_t ldx, float2 beta,
        float2* __restrict__ Y, int64_t ldy) {
    __shared__ double red[2][BLK / 64];
    const int tid = threadIdx.x;
    const int64_t n = blockIdx.x;
    const float2* __restrict__ x = X + n * ldx;
    double sr = 0.0, si = 0.0;
    for (int64_t k = tid; k < K; k += BLK) { const float2 v = x[k]; sr += v.x; si += v.y; }
    for (int off = 32; off >= 1; off >>= 1) { sr += __shfl_xor(sr, off, 64); si += __shfl_xor(si, off, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = sr; red[1][tid >> 6] = si; }
    __syncthreads();
    sr = 0.0; si = 0.0;
    for (int w = 0; w < BLK / 64; ++w) { sr += red[0][w]; si += red[1][w]; }
    const float2 s = cmul(alpha, make_float2((float)sr, (float)si));
    float2* __restrict__ y = Y + n * ldy;
    for (int64_t m = tid; m < M; m += BLK) {
        float2 out = s;
        if (!B0) cfma(out, beta, y[m]);
        y[m] = out;
    }
}

// ---- DIA ----------------------------------------------------------------------------------------------
// A is M x K with nd stored diagonals; data is ld_d x nd column-major, data[j + d*ld_d] = A[j - off_d, j] (scipy's
// dia_matrix.data transposed, as the reference uploads it: backend.py:609).
//   forward : Y[m, :] = beta*Y[m, :] + alpha * sum_d data[m + off_d, d]       * X[m + off_d, :]     rows m of A
//   adjoint : Y[k, :] = beta*Y[k, :] + alpha * sum_d conj(data[k, d])         * X[k - off_d, :]     rows k of A^H
// A lane owns an output row and NC panel columns in registers; consecutive lanes read consecutive addresses of every
// array (data, X, Y): all accesses coalesced.
template <int NC, bool ADJ, bool B0>
__global__ void __launch_bounds__(BLK)
k_cdiamm(int64_t rows, int64_t inner, int64_t N, int nd, const int32_t* __restrict__ offsets,
         const float2* __restrict__ data, int64_t ld_d, float2 alpha,
         const float2* __restrict__ X, int64_t ldx, float2 beta, float2* __restrict__ Y, int64_t ldy) {
    const int64_t r = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (r >= rows) return;
    for (int64_t jb = 0; jb < N; jb += NC) {
        float2 acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = make_float2(0.f, 0.f);
        for (int d = 0; d < nd; ++d) {
            const int64_t off = offsets[d];
            const int64_t src = ADJ ? r - off : r + off;            // row of X
            if (src < 0 || src >= inner) continue;
            const int64_t j = ADJ ? r : src;                        // column of A the stored value belongs to
            if (j >= ld_d) continue;                                // beyond the stored part of the diagonal: zero
            float2 v = data[j + (int64_t)d * ld_d];
            if (ADJ) v.y = -v.y;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (jb + c < N) cfma(acc[c], v, X[src + (jb + c) * ldx]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (jb + c < N) {
                float2* yp = Y + r + (jb + c) * ldy;
                float2 out = cmul(alpha, acc[c]);
                if (!B0) cfma(out, beta, *yp);
                *yp = out;
            }
        }
    }
}

// ---- dense --------------------------------------------------------------------------------------------
// C(m x n) = alpha * A(m x k) * B(k x n) + beta * C with generic element strides, A optionally conjugated:
//   A(i, l) = a[i*sai + l*sal],  B(l, j) = b[l*sbl + j*sbj],  C(i, j) = c[i*sci + j*scj]
// which covers op(M) * X (forward / adjoint, left) and X * M (right) without copies.  64 x 64 tiles of C per
// workgroup, 16 x 16 threads, 4 x 4 outputs per thread, k in steps of 16 through LDS.
template <bool CONJA, bool B0>
__global__ void __launch_bounds__(256)
k_cgemm(int64_t m, int64_t n, int64_t k, float2 alpha,
        const float2* __restrict__ a, int64_t sai, int64_t sal,
        const float2* __restrict__ b, int64_t sbl, int64_t sbj,
        float2 beta, float2* __restrict__ c, int64_t sci, int64_t scj) {
    __shared__ float2 As[16][65], Bs[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
    float2 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = make_float2(0.f, 0.f);
    for (int64_t l0 = 0; l0 < k; l0 += 16) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            // lanes run along whichever index of A (B) is contiguous in memory
            const int li = sai <= sal ? e / 64 : e % 16, ii = sai <= sal ? e % 64 : e / 16;
            const int64_t i = i0 + ii, l = l0 + li;
            float2 v = (i < m && l < k) ? a[i * sai + l * sal] : make_float2(0.f, 0.f);
            if (CONJA) v.y = -v.y;
            As[li][ii] = v;
            const int lj = sbl <= sbj ? e % 16 : e / 64, jj = sbl <= sbj ? e / 16 : e % 64;
            const int64_t j = j0 + jj, l2 = l0 + lj;
            Bs[lj][jj] = (j < n && l2 < k) ? b[l2 * sbl + j * sbj] : make_float2(0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            float2 av[4], bv[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) av[p] = As[l][tx + 16 * p];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = Bs[l][ty + 16 * q];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) cfma(acc[p][q], av[p], bv[q]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t i = i0 + tx + 16 * p, j = j0 + ty + 16 * q;
            if (i < m && j < n) {
                float2* cp = c + i * sci + j * scj;
                float2 out = cmul(alpha, acc[p][q]);
                if (!B0) cfma(out, beta, *cp);
                *cp = out;
            }
        }
}

}  // namespace

extern "C" {

int ig_conemm(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, float ar, float ai, const void* X, int64_t ldx,
              float br, float bi, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_conemm: ctx is NULL");
    IG_REQUIRE(ctx, M >= 0 && K >= 0 && N >= 0, "ig_conemm: negative dimension");
    IG_REQUIRE(ctx, N == 0 || ((K == 0 || X) && (M == 0 || Y)), "ig_conemm: NULL panel");
    IG_REQUIRE(ctx, N <= 1 || (ldx >= K && ldy >= M), "ig_conemm: leading dimension smaller than the panel");
    IG_REQUIRE(ctx, N <= 0x7fffffffLL, "ig_conemm: too many columns");
    if (N == 0 || M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    ig_prof_scope prof(ctx, "onemm", (double)(K + M * ((br == 0.f && bi == 0.f) ? 1 : 2)) * N * 8.0);
    const float2 alpha = make_float2(ar, ai), beta = make_float2(br, bi);
    if (br == 0.f && bi == 0.f)
        hipLaunchKernelGGL(k_onemm<true>, dim3((unsigned)N), dim3(BLK), 0, ctx->stream, M, K, alpha, (const float2*)X, ldx, beta, (float2*)Y, ldy);
    else
        hipLaunchKernelGGL(k_onemm<false>, dim3((unsigned)N), dim3(BLK), 0, ctx->stream, M, K, alpha, (const float2*)X, ldx, beta, (float2*)Y, ldy);
    IG_LAUNCH_CHECK(ctx, "k_onemm");
    return IG_OK;
}

int ig_cdiamm(ig_ctx* ctx, int adjoint, int64_t M, int64_t K, int64_t N, int64_t ndiag, const int32_t* offsets,
              const void* data, int64_t ld_data, float ar, float ai, const void* X, int64_t ldx,
              float br, float bi, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_cdiamm: ctx is NULL");
    IG_REQUIRE(ctx, M >= 0 && K >= 0 && N >= 0 && ndiag >= 0 && ndiag <= 0x7fffffffLL, "ig_cdiamm: bad dimension");
    IG_REQUIRE(ctx, ndiag == 0 || (offsets && data && ld_data >= 1), "ig_cdiamm: NULL diagonals");
    const int64_t rows = adjoint ? K : M, inner = adjoint ? M : K;      // rows of Y, rows of X
    IG_REQUIRE(ctx, N == 0 || ((inner == 0 || X) && (rows == 0 || Y)), "ig_cdiamm: NULL panel");
    IG_REQUIRE(ctx, N <= 1 || (ldx >= inner && ldy >= rows), "ig_cdiamm: leading dimension smaller than the panel");
    if (N == 0 || rows == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const bool b0 = br == 0.f && bi == 0.f;
    ig_prof_scope prof(ctx, adjoint ? "cdiamm_adj" : "cdiamm", (double)ndiag * ld_data * 8.0 + (double)(inner + rows * (b0 ? 1 : 2)) * N * 8.0);
    const float2 alpha = make_float2(ar, ai), beta = make_float2(br, bi);
    const int64_t blocks = (rows + BLK - 1) / BLK;
    IG_REQUIRE(ctx, blocks <= 0x7fffffffLL, "ig_cdiamm: matrix too large for one launch");
#define IG_DIA(NC_, ADJ_, B0_) hipLaunchKernelGGL((k_cdiamm<NC_, ADJ_, B0_>), dim3((unsigned)blocks), dim3(BLK), 0, ctx->stream, \
        rows, inner, N, (int)ndiag, offsets, (const float2*)data, ld_data, alpha, (const float2*)X, ldx, beta, (float2*)Y, ldy)
#define IG_DIA_NC(NC_) do { if (adjoint) { if (b0) IG_DIA(NC_, true, true); else IG_DIA(NC_, true, false); }   \
                            else { if (b0) IG_DIA(NC_, false, true); else IG_DIA(NC_, false, false); } } while (0)
    if (N >= 8) IG_DIA_NC(8); else if (N >= 4) IG_DIA_NC(4); else if (N >= 2) IG_DIA_NC(2); else IG_DIA_NC(1);
#undef IG_DIA_NC
#undef IG_DIA
    IG_LAUNCH_CHECK(ctx, "k_cdiamm");
    return IG_OK;
}

int ig_cgemm(ig_ctx* ctx, int adjoint, int right, int64_t rows_m, int64_t cols_m, int64_t p,
             float ar, float ai, const void* Mv, int64_t ldm, const void* X, int64_t ldx,
             float br, float bi, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_cgemm: ctx is NULL");
    IG_REQUIRE(ctx, rows_m >= 0 && cols_m >= 0 && p >= 0, "ig_cgemm: negative dimension");
    IG_REQUIRE(ctx, ldm >= rows_m || cols_m <= 1, "ig_cgemm: ldm smaller than the matrix");
    // op(M): r x c
    const int64_t r = adjoint ? cols_m : rows_m, c = adjoint ? rows_m : cols_m;
    // left : Y(r x p) = alpha * op(M)(r x c) * X(c x p) + beta*Y        right: Y(p x c) = alpha * X(p x r) * op(M)(r x c) + beta*Y
    const int64_t m = right ? p : r, n = right ? c : p, k = right ? r : c;
    IG_REQUIRE(ctx, (m == 0 || n == 0) || (Y && (k == 0 || (Mv && X))), "ig_cgemm: NULL array");
    IG_REQUIRE(ctx, ldx >= (right ? p : c) || (right ? r : p) <= 1, "ig_cgemm: ldx smaller than X");
    IG_REQUIRE(ctx, ldy >= m || n <= 1, "ig_cgemm: ldy smaller than Y");
    if (m == 0 || n == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const bool b0 = br == 0.f && bi == 0.f;
    ig_prof_scope prof(ctx, "cgemm", ((double)rows_m * cols_m + (double)k * (right ? m : n) + (double)m * n * (b0 ? 1 : 2)) * 8.0);
    const float2 alpha = make_float2(ar, ai), beta = make_float2(br, bi);
    const float2* Mp = (const float2*)Mv;
    const float2* Xp = (const float2*)X;
    // element strides of op(M)(i, l): M is column-major with leading dimension ldm
    const int64_t s_row = adjoint ? ldm : 1, s_col = adjoint ? 1 : ldm;        // op(M)(i, l) = M[i*s_row + l*s_col] (conjugated if adjoint)
    const dim3 grid((unsigned)((m + 63) / 64), (unsigned)((n + 63) / 64));
    IG_REQUIRE(ctx, (m + 63) / 64 <= 0x7fffffffLL && (n + 63) / 64 <= 65535, "ig_cgemm: panel too large for one launch");
    if (!right) {
        // C = op(M) * X : A = op(M) (conjugated when adjoint), B = X
#define IG_GEMM(CJ_, B0_) hipLaunchKernelGGL((k_cgemm<CJ_, B0_>), grid, dim3(256), 0, ctx->stream, m, n, k, alpha, \
            Mp, s_row, s_col, Xp, (int64_t)1, ldx, beta, (float2*)Y, (int64_t)1, ldy)
        if (adjoint) { if (b0) IG_GEMM(true, true); else IG_GEMM(true, false); }
        else         { if (b0) IG_GEMM(false, true); else IG_GEMM(false, false); }
#undef IG_GEMM
    } else {
        // C = X * op(M) = (op(M)^T * X^T)^T : computed directly with A = X (p x r), B = op(M) (r x c).  The kernel conjugates
        // its A operand only, so the adjoint case runs on conj: C = conj( conj(X) * M^T ) is avoided by swapping roles:
        // C^T(c x p) = op(M)^T(c x r) * X^T(r x p); op(M)^T(i, l) = op(M)(l, i) = M[l*s_row + i*s_col] (conjugated if adjoint)
#define IG_GEMM(CJ_, B0_) hipLaunchKernelGGL((k_cgemm<CJ_, B0_>), dim3((unsigned)((n + 63) / 64), (unsigned)((m + 63) / 64)), dim3(256), 0, ctx->stream, \
            n, m, k, alpha, Mp, s_col, s_row, Xp, ldx, (int64_t)1, beta, (float2*)Y, ldy, (int64_t)1)
        IG_REQUIRE(ctx, (m + 63) / 64 <= 65535, "ig_cgemm: panel too large for one launch");
        if (adjoint) { if (b0) IG_GEMM(true, true); else IG_GEMM(true, false); }
        else         { if (b0) IG_GEMM(false, true); else IG_GEMM(false, false); }
#undef IG_GEMM
    }
    IG_LAUNCH_CHECK(ctx, "k_cgemm");
    return IG_OK;
}

}  // extern "C"
