/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Plain-C restatement of the reference's CSR x dense-panel product
 * (indigo/backends/_customcpu.c:14-114, contract backend.py:514-519):
 *
 *   forward :  C(MxN) = alpha *  A   * B(KxN) + beta * C
 *   adjoint :  C(KxN) = alpha * A^H * B(MxN) + beta * C
 *
 * A is M x K CSR (int32 rowptr/colind, complex float values), B and C are
 * column-major with leading dimensions ldb / ldc.  Arithmetic is complex
 * float in row-sequential order, like the reference's loops.  Pointer-only
 * signatures so ctypes can call it (the reference passes `complex float` by
 * value).  beta == 0 does not read C (BLAS rule).
 *
 * Also restates `inspect` (_customcpu.c:179-215).
 *
 * Parity status: pinned -- tests/test_oracle.py checks these functions against
 * the golden vectors captured from the reference and against the reference's
 * own compiled C (oracle/_ref) where that has been built.
 */
#include <complex.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_ccsrmm(int adjoint, int64_t M, int64_t N, int64_t K,
                   const float* alpha_ri, const float complex* val, const int32_t* col,
                   const int32_t* rowptr, const float complex* B, int64_t ldb,
                   const float* beta_ri, float complex* C, int64_t ldc)
{
    const float complex alpha = alpha_ri[0] + I * alpha_ri[1];
    const float complex beta = beta_ri[0] + I * beta_ri[1];
    const int beta0 = (beta_ri[0] == 0.0f && beta_ri[1] == 0.0f);

    if (!adjoint) {
        /* _customcpu.c:80-112 without the 8-row blocking: one accumulator per (row, column) */
        #pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; m++) {
            for (int64_t n = 0; n < N; n++) {
                float complex acc = 0.0f;
                for (int32_t i = rowptr[m]; i < rowptr[m + 1]; i++)
                    acc += val[i] * B[col[i] + n * ldb];
                float complex* c = &C[m + n * ldc];
                *c = beta0 ? alpha * acc : alpha * acc + beta * (*c);
            }
        }
    } else {
        /* _customcpu.c:20-42 / 49-79: pre-scale C by beta, then scatter conj(val)*alpha*B.
           Parallel over panel columns (disjoint outputs) instead of omp atomics. */
        #pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; n++) {
            float complex* c = &C[n * ldc];
            for (int64_t k = 0; k < K; k++)
                c[k] = beta0 ? 0.0f : beta * c[k];
            for (int64_t m = 0; m < M; m++) {
                const float complex b = B[m + n * ldb];
                for (int32_t i = rowptr[m]; i < rowptr[m + 1]; i++)
                    c[col[i]] += (alpha * conjf(val[i])) * b;
            }
        }
    }
}

/* nonzero rows, nonzero columns, exwrite = every column has at most one nonzero */
void oracle_inspect(int64_t M, int64_t K, const int32_t* col, const int32_t* rowptr,
                    int64_t* nzrows, int64_t* nzcols, int* exwrite)
{
    int32_t* cnt = calloc((size_t)(K > 0 ? K : 1), sizeof(int32_t));
    int64_t rows = 0, cols = 0;
    int exw = 1;
    for (int64_t m = 0; m < M; m++) {
        if (rowptr[m + 1] > rowptr[m]) rows++;
        for (int32_t i = rowptr[m]; i < rowptr[m + 1]; i++) cnt[col[i]]++;
    }
    for (int64_t k = 0; k < K; k++) {
        if (cnt[k] > 0) cols++;
        if (cnt[k] > 1) exw = 0;
    }
    free(cnt);
    *nzrows = rows; *nzcols = cols; *exwrite = exw;
}
