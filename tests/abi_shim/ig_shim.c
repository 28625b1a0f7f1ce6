/* HOST-ONLY stand-in for libindigo_hip.so -- TEST INFRASTRUCTURE, never shipped, never loaded by indigo_amd.
 *
 * It defines the entry points of include/indigo_hip.h that the literal `Backend` contract needs (INTEGRATION.md section 2:
 * context, memory, axpby / scale / dot / norm2 / max, ccsrmm, inspect, fftn / ifftn, onemm, cdiamm, cgemm) as plain CPU loops over
 * host memory, so that the reference-side binding (integration/hip_backend_for_indigo.py) can be RUN under the reference's own
 * Backend base class and backend tests in the build container, which has the reference but no GPU.  Because this file includes the
 * real header, a prototype that drifts from include/indigo_hip.h does not compile (tests/test_reference_binding.py builds it).
 * "Device" pointers are host pointers; everything is synchronous.  The arithmetic follows the reference's numpy backend
 * (indigo/backends/np.py:53-145) in float32 / float64 as the HIP kernels do. */
#define _GNU_SOURCE
#include "indigo_hip.h"
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct ig_ctx { char err[256]; };
struct ig_fft { int rank; int64_t dims[3]; int64_t batch; };
static _Thread_local char tls_err[256];

static int fail(ig_ctx* ctx, int code, const char* msg) {
    snprintf(ctx ? ctx->err : tls_err, 256, "%s", msg);
    return code;
}

int ig_abi_version(void) { return IG_ABI_VERSION; }
int ig_device_count(int* count) { if (count) *count = 1; return IG_OK; }
int ig_init(int device_id, ig_ctx** out) {
    if (!out || device_id != 0) return fail(NULL, IG_ERR_ARG, "ig_init: bad arguments");
    *out = (ig_ctx*)calloc(1, sizeof(ig_ctx));
    return *out ? IG_OK : IG_ERR_NOMEM;
}
void ig_destroy(ig_ctx* ctx) { free(ctx); }
const char* ig_last_error(ig_ctx* ctx) { return ctx ? ctx->err : tls_err; }
int ig_sync(ig_ctx* ctx) { return ctx ? IG_OK : IG_ERR_ARG; }

int ig_malloc(ig_ctx* ctx, size_t nbytes, void** dptr) {
    if (!ctx || !dptr) return fail(ctx, IG_ERR_ARG, "ig_malloc: bad arguments");
    *dptr = NULL;
    if (nbytes == 0) nbytes = 1;
    if (posix_memalign(dptr, 256, (nbytes + 255) & ~(size_t)255)) return fail(ctx, IG_ERR_NOMEM, "ig_malloc: out of memory");
    return IG_OK;
}
int ig_free(ig_ctx* ctx, void* dptr) { (void)ctx; free(dptr); return IG_OK; }
int ig_memset0(ig_ctx* ctx, void* dptr, size_t nbytes) {
    if (nbytes && !dptr) return fail(ctx, IG_ERR_ARG, "ig_memset0: NULL pointer");
    if (nbytes) memset(dptr, 0, nbytes);
    return IG_OK;
}
int ig_copy2d(ig_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t height, int kind) {
    if (width_bytes == 0 || height == 0) return IG_OK;
    if (!dst || !src) return fail(ctx, IG_ERR_ARG, "ig_copy2d: NULL pointer");
    if (kind < IG_H2D || kind > IG_D2D) return fail(ctx, IG_ERR_ARG, "ig_copy2d: unknown kind");
    if (height > 1 && (dpitch < width_bytes || spitch < width_bytes)) return fail(ctx, IG_ERR_ARG, "ig_copy2d: pitch smaller than row width");
    for (size_t r = 0; r < height; ++r) memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width_bytes);
    return IG_OK;
}

typedef float complex c64;

int ig_caxpby(ig_ctx* ctx, int64_t n, float br, float bi, void* y, float ar, float ai, const void* x) {
    (void)ctx;
    c64* yy = (c64*)y; const c64* xx = (const c64*)x;
    const c64 a = ar + ai * I, b = br + bi * I;
    for (int64_t i = 0; i < n; ++i) yy[i] = ((br == 0.f && bi == 0.f) ? 0 : b * yy[i]) + a * xx[i];
    return IG_OK;
}
int ig_cscal(ig_ctx* ctx, int64_t n, float ar, float ai, void* x) {
    (void)ctx;
    c64* xx = (c64*)x; const c64 a = ar + ai * I;
    for (int64_t i = 0; i < n; ++i) xx[i] *= a;
    return IG_OK;
}
int ig_cdotc(ig_ctx* ctx, int64_t n, const void* x, const void* y, double out[2]) {
    (void)ctx;
    const c64* xx = (const c64*)x; const c64* yy = (const c64*)y;
    double re = 0, im = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double complex p = conj((double complex)xx[i]) * (double complex)yy[i];
        re += creal(p); im += cimag(p);
    }
    out[0] = re; out[1] = im;
    return IG_OK;
}
int ig_scnrm2sq(ig_ctx* ctx, int64_t n, const void* x, double* out) {
    (void)ctx;
    const float* f = (const float*)x;
    double s = 0;
    for (int64_t i = 0; i < 2 * n; ++i) s += (double)f[i] * f[i];
    *out = s;
    return IG_OK;
}
int ig_cmax(ig_ctx* ctx, int64_t nfloats, float val, void* arr) {
    (void)ctx;
    float* f = (float*)arr;
    for (int64_t i = 0; i < nfloats; ++i) if (f[i] < val) f[i] = val;
    return IG_OK;
}

int ig_ccsrmm(ig_ctx* ctx, int adjoint, int exwrite, int64_t M, int64_t K, int64_t N, int64_t nnz,
              float ar, float ai, const void* vals, const int32_t* colind, const int32_t* rowptr,
              const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy) {
    (void)exwrite; (void)nnz;
    if (!ctx || !rowptr) return fail(ctx, IG_ERR_ARG, "ig_ccsrmm: bad arguments");
    const c64* v = (const c64*)vals; const c64* x = (const c64*)X; c64* y = (c64*)Y;
    const c64 a = ar + ai * I, b = br + bi * I;
    const int64_t yrows = adjoint ? K : M;
    const int bzero = br == 0.f && bi == 0.f;
    for (int64_t j = 0; j < N; ++j) {
        for (int64_t r = 0; r < yrows; ++r) y[j * ldy + r] = bzero ? 0 : b * y[j * ldy + r];
        for (int64_t r = 0; r < M; ++r)
            for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
                if (!adjoint) y[j * ldy + r] += a * v[p] * x[j * ldx + colind[p]];
                else          y[j * ldy + colind[p]] += a * conjf(v[p]) * x[j * ldx + r];
            }
    }
    return IG_OK;
}
int ig_csr_inspect(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K, int64_t* nzrow, int64_t* nzcol, int* exwrite) {
    if (!rowptr || M < 0 || K < 0) return fail(NULL, IG_ERR_ARG, "ig_csr_inspect: bad arguments");
    int32_t* cnt = (int32_t*)calloc((size_t)(K > 0 ? K : 1), sizeof(int32_t));
    int64_t nr = 0, nc = 0; int ex = 1;
    for (int64_t r = 0; r < M; ++r) {
        if (rowptr[r + 1] > rowptr[r]) ++nr;
        for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) ++cnt[colind[p]];
    }
    for (int64_t c = 0; c < K; ++c) { if (cnt[c]) ++nc; if (cnt[c] > 1) ex = 0; }
    free(cnt);
    if (nzrow) *nzrow = nr;
    if (nzcol) *nzcol = nc;
    if (exwrite) *exwrite = ex;
    return IG_OK;
}

/* unnormalised DFT, both directions, over the first `rank` axes of a Fortran-ordered batch (np.py:102-115): one axis at a time, O(n^2) */
int ig_fft_plan(ig_ctx* ctx, int rank, const int64_t* dims, int64_t batch, ig_fft** plan, size_t* workspace_bytes) {
    if (!ctx || !dims || !plan || rank < 1 || rank > 3 || batch < 1) return fail(ctx, IG_ERR_ARG, "ig_fft_plan: bad arguments");
    ig_fft* p = (ig_fft*)calloc(1, sizeof(ig_fft));
    p->rank = rank; p->batch = batch;
    for (int a = 0; a < 3; ++a) p->dims[a] = a < rank ? dims[a] : 1;
    *plan = p;
    if (workspace_bytes) *workspace_bytes = 0;
    return IG_OK;
}
int ig_fft_destroy(ig_fft* plan) { free(plan); return IG_OK; }
int ig_fft_exec(ig_fft* plan, const void* x, void* y, int direction, void* workspace) {
    (void)workspace;
    if (!plan || !x || !y || (direction != -1 && direction != 1)) return fail(NULL, IG_ERR_ARG, "ig_fft_exec: bad arguments");
    const int64_t n0 = plan->dims[0], n1 = plan->dims[1], n2 = plan->dims[2], vol = n0 * n1 * n2;
    double complex* a = (double complex*)malloc(sizeof(double complex) * (size_t)vol);
    double complex* line = (double complex*)malloc(sizeof(double complex) * (size_t)(n0 > n1 ? (n0 > n2 ? n0 : n2) : (n1 > n2 ? n1 : n2)));
    const int64_t n[3] = {n0, n1, n2}, stride[3] = {1, n0, n0 * n1};
    for (int64_t b = 0; b < plan->batch; ++b) {
        const c64* xi = (const c64*)x + b * vol; c64* yo = (c64*)y + b * vol;
        for (int64_t i = 0; i < vol; ++i) a[i] = xi[i];
        for (int ax = 0; ax < 3; ++ax) {
            if (n[ax] == 1) continue;
            for (int64_t base = 0; base < vol; ++base) {
                if ((base / stride[ax]) % n[ax]) continue;                       /* first element of a line along `ax` */
                for (int64_t k = 0; k < n[ax]; ++k) {
                    double complex s = 0;
                    for (int64_t j = 0; j < n[ax]; ++j)
                        s += a[base + j * stride[ax]] * cexp(direction * 2.0 * M_PI * I * (double)((j * k) % n[ax]) / (double)n[ax]);
                    line[k] = s;
                }
                for (int64_t k = 0; k < n[ax]; ++k) a[base + k * stride[ax]] = line[k];
            }
        }
        for (int64_t i = 0; i < vol; ++i) yo[i] = (c64)a[i];
    }
    free(a); free(line);
    return IG_OK;
}

int ig_conemm(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, float ar, float ai, const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy) {
    (void)ctx;
    const c64* x = (const c64*)X; c64* y = (c64*)Y;
    const c64 a = ar + ai * I, b = br + bi * I;
    for (int64_t j = 0; j < N; ++j) {
        double complex s = 0;
        for (int64_t k = 0; k < K; ++k) s += x[j * ldx + k];
        for (int64_t r = 0; r < M; ++r) y[j * ldy + r] = ((br == 0.f && bi == 0.f) ? 0 : b * y[j * ldy + r]) + a * (c64)s;
    }
    return IG_OK;
}
int ig_cdiamm(ig_ctx* ctx, int adjoint, int64_t M, int64_t K, int64_t N, int64_t ndiag, const int32_t* offsets, const void* data, int64_t ldd,
              float ar, float ai, const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy) {
    (void)ctx;
    /* data(:, d) is scipy's dia_matrix.data row d: entry (i, i + off) is stored at data[i + off] of that diagonal */
    const c64* dd = (const c64*)data; const c64* x = (const c64*)X; c64* y = (c64*)Y;
    const c64 a = ar + ai * I, b = br + bi * I;
    const int64_t yrows = adjoint ? K : M;
    for (int64_t j = 0; j < N; ++j) {
        for (int64_t r = 0; r < yrows; ++r) y[j * ldy + r] = (br == 0.f && bi == 0.f) ? 0 : b * y[j * ldy + r];
        for (int64_t d = 0; d < ndiag; ++d)
            for (int64_t i = 0; i < M; ++i) {
                const int64_t c = i + offsets[d];
                if (c < 0 || c >= K) continue;
                const c64 v = dd[d * ldd + c];
                if (!adjoint) y[j * ldy + i] += a * v * x[j * ldx + c];
                else          y[j * ldy + c] += a * conjf(v) * x[j * ldx + i];
            }
    }
    return IG_OK;
}
int ig_cgemm(ig_ctx* ctx, int adjoint, int right, int64_t rows_m, int64_t cols_m, int64_t p, float ar, float ai,
             const void* Mm, int64_t ldm, const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy) {
    (void)ctx;
    /* left:  Y (r x p) = beta Y + alpha op(M) X (c x p);   right: Y (p x c) = beta Y + alpha X (p x r) op(M);   op(M): r x c */
    const c64* m = (const c64*)Mm; const c64* x = (const c64*)X; c64* y = (c64*)Y;
    const c64 a = ar + ai * I, b = br + bi * I;
    const int64_t r = adjoint ? cols_m : rows_m, c = adjoint ? rows_m : cols_m;
    const int bz = br == 0.f && bi == 0.f;
#define OPM(i, k) (adjoint ? conjf(m[(i) * ldm + (k)]) : m[(k) * ldm + (i)])
    if (!right) {
        for (int64_t j = 0; j < p; ++j)
            for (int64_t i = 0; i < r; ++i) {
                double complex s = 0;
                for (int64_t k = 0; k < c; ++k) s += (double complex)OPM(i, k) * x[j * ldx + k];
                y[j * ldy + i] = (bz ? 0 : b * y[j * ldy + i]) + a * (c64)s;
            }
    } else {
        for (int64_t j = 0; j < c; ++j)
            for (int64_t i = 0; i < p; ++i) {
                double complex s = 0;
                for (int64_t k = 0; k < r; ++k) s += (double complex)x[k * ldx + i] * OPM(k, j);
                y[j * ldy + i] = (bz ? 0 : b * y[j * ldy + i]) + a * (c64)s;
            }
    }
#undef OPM
    return IG_OK;
}
