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
// Besides this row-slot gather (k_csrmm_gather, and k_csrmm_gather_v with 16-byte loads over row-major panels) the
// file holds the kernels for matrices with mostly empty rows (transposed gridding matrices): k_csrmm_dense64 (64
// nonzeros per trip, one per lane, LDS segmented sums), k_csrmm_rowlane (a row per lane), the deferred-row kernels for
// the few very long rows, the panel repacking kernels, and the scatter form of the adjoint.
//
// Everything here is bandwidth/latency bound integer+fp32 work: no MFMA.
#include "ig_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <vector>
#include <thread>
#include <atomic>
#include <functional>
#include <algorithm>
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

// ---- deferred rows ---------------------------------------------------------------
// Gridding transposes have a few rows that are orders of magnitude longer than the
// mean (the k-space centre).  The row-slot kernel therefore only computes rows of
// at most `thr_mid` nonzeros inline; longer rows are appended to one of two device
// work lists (one atomic per deferred row) and finished by
//   k_csrmm_rows_wave   one wavefront per listed row   (thr_mid < nnz <= thr_long)
//   k_csrmm_rows_block  one 1024-thread workgroup per listed row (nnz > thr_long)
// both of which read the list length from device memory, so the host never syncs.
// If a list is full the row is simply computed inline (slow, still correct).
// Optional support of a 3-D grid of output rows, row = kx + n0*(km + nm*ks): a row is only computed and written
// if its 16-row segment is flagged: bit (km >> 4) of bits[(ks*(n0/16) + kx/16)*16 + (km & 15)] (the third part of
// the support table of ig_fft_exec_padded).  bits == nullptr: every row.
struct GridMask {
    const uint32_t* bits;
    int64_t n0, nm;
};

// Deferred-row lists are split into WL_SUB sub-lists with their own counters: a wave appends all its long rows
// with ONE atomic on the counter its id hashes to.  (One atomic per row on a single counter serialised in L2:
// 150k long rows of the 134M-row transposed gridding matrix cost 0.9 ms of a 3.6 ms launch.)
constexpr int WL_SUB = 32;
struct WorkLists {
    int32_t*  rows[2];     // [0] wave-per-row lists, [1] workgroup-per-row lists; sub-list s starts at s*cap
    uint32_t* count;       // count[which*WL_SUB + s]
    uint32_t  cap;         // per sub-list
    const int32_t* yperm;  // optional: result row r is stored at Y[yperm[r]] (rows of A were reordered for locality)
};

// All lanes of the wave call this; lanes with `want` get a slot in sub-list (which, sub).  Returns true if the
// lane's row was appended (false: list full, compute inline).
__device__ __forceinline__ bool wl_append(const WorkLists& wl, int which, int sub, bool want, int32_t row) {
    const uint64_t bal = __ballot(want);
    if (!bal) return false;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)bal) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&wl.count[which * WL_SUB + sub], (uint32_t)__popcll(bal));
    base = __shfl(base, leader, 64);
    if (!want) return false;
    const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (idx >= wl.cap) return false;
    wl.rows[which][(size_t)sub * wl.cap + idx] = row;
    return true;
}

__device__ __forceinline__ int64_t out_row(const int32_t* __restrict__ perm, int64_t row) {
    return perm ? (int64_t)perm[row] : row;
}

template <bool CONJ>
__device__ __forceinline__ void acc_nz(float2& acc, float2 v, float2 x) {
    if (CONJ) acc = cadd(acc, cmulc(v, x));
    else      cfma(acc, v, x);
}

// BMODE: 0 => beta == 0 (Y not read), 1 => general beta
template <int CL, int NL, bool CONJ, int BMODE>
__global__ void __launch_bounds__(BLK)
k_csrmm_gather(int64_t M, int64_t N,
               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
               const float2* __restrict__ vals,
               const float2* __restrict__ X, int64_t ldx, int64_t sxr,
               float2* __restrict__ Y, int64_t ldy,
               float2 alpha, float2 beta, int xcd_remap,
               WorkLists wl, int32_t thr_mid, int32_t thr_long) {
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

    // defer long rows: the slot's first lane claims a list entry, the slot learns the outcome by shuffle
    int deferred = 0;
    const int32_t len = p1 - p0;
    {
        const int sub = (int)(wave & (WL_SUB - 1));
        const bool head = c == 0 && i == 0;
        if (wl_append(wl, 0, sub, head && len > thr_mid && len <= thr_long, (int32_t)row)) deferred = 1;
        if (wl_append(wl, 1, sub, head && len > thr_long, (int32_t)row)) deferred = 1;
    }
    deferred = __shfl(deferred, r * CL * NL, 64);
    if (deferred) p1 = p0;          // nothing to do here, and no store below

    for (int64_t jb = 0; jb < N; jb += CL) {
        const int64_t j = jb + c;
        const bool col_ok = j < N;
        const float2* __restrict__ xcol = X + (col_ok ? j : 0) * ldx;
        float2 acc = make_float2(0.f, 0.f);
        // Four nonzeros per trip and lane, predicated (index clamped to the row's last nonzero, value zeroed):
        // a gridding row (27 nonzeros on 8 nonzero-lanes) is then ONE trip, i.e. a dependent chain of three
        // memory round trips (rowptr -> index/value -> panel) instead of five.  This kernel is latency bound.
        for (int32_t p = p0 + i; p < p1; p += 4 * NL) {
            int32_t k4[4];
            float2 v4[4], x4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t pu = p + u * NL;
                const bool ok = pu < p1;
                const int32_t q = ok ? pu : p1 - 1;
                k4[u] = colind[q];
                const float2 vv = vals[q];
                v4[u] = ok ? vv : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) x4[u] = xcol[k4[u] * sxr];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc_nz<CONJ>(acc, v4[u], x4[u]);
        }
#pragma unroll
        for (int off = CL * NL / 2; off >= CL; off >>= 1) {
            acc.x += __shfl_xor(acc.x, off, 64);
            acc.y += __shfl_xor(acc.y, off, 64);
        }
        if (i == 0 && row_ok && col_ok && !deferred) {
            float2* yp = Y + j * ldy + out_row(wl.yperm, row);
            float2 out = cmul(alpha, acc);
            if (BMODE == 1) cfma(out, beta, *yp);
            *yp = out;
        }
    }
}

// Row-slot gather over a ROW-MAJOR panel (packed or coil-interleaved: the N values of a panel row contiguous).  A lane
// owns VW consecutive columns and fetches them with 16-byte loads, so a row needs only (N/VW) x NL lanes and a wave
// covers 64 / ((N/VW)*NL) rows with the same dependent chain as one row per wave: the kernel is latency bound, more
// rows per wave is more rows per unit time.
// REALW: `vals` points to FLOATS, the real parts of a matrix whose weights are all real (4 bytes per nonzero instead of 8).
template <int VW, int CLV, int NL, bool CONJ, int BMODE, bool REALW = false>
__global__ void __launch_bounds__(BLK)
k_csrmm_gather_v(int64_t M, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                 const float2* __restrict__ vals, const float2* __restrict__ X, int64_t sxr,
                 float2* __restrict__ Y, int64_t ldy, float2 alpha, float2 beta, int xcd_remap,
                 WorkLists wl, int32_t thr_mid, int32_t thr_long) {
    static_assert(VW % 2 == 0 && CLV * NL <= 64, "16-byte loads; a row's lanes fit a wave");
    constexpr int LPR = CLV * NL, RPW = 64 / LPR, NV = VW / 2;
    const int lane = threadIdx.x & 63;
    const int c = lane % CLV, i = (lane / CLV) % NL, r = lane / LPR;
    const int64_t blk = xcd_remap ? xcd_block(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
    const int64_t wave = blk * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + r;
    const bool row_ok = row < M;
    int32_t p0 = 0, p1 = 0;
    if (row_ok) { p0 = rowptr[row]; p1 = rowptr[row + 1]; }
    int deferred = 0;
    const int32_t len = p1 - p0;
    {
        const int sub = (int)(wave & (WL_SUB - 1));
        const bool head = c == 0 && i == 0;
        if (wl_append(wl, 0, sub, head && len > thr_mid && len <= thr_long, (int32_t)row)) deferred = 1;
        if (wl_append(wl, 1, sub, head && len > thr_long, (int32_t)row)) deferred = 1;
    }
    deferred = __shfl(deferred, r * LPR, 64);
    if (deferred) p1 = p0;
    float2 acc[VW];
#pragma unroll
    for (int u = 0; u < VW; ++u) acc[u] = make_float2(0.f, 0.f);
    const float4* __restrict__ X4 = reinterpret_cast<const float4*>(X + c * VW);
    for (int32_t p = p0 + i; p < p1; p += 4 * NL) {
        int32_t k4[4];
        float2 v4[4];
        float4 x4[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t pu = p + u * NL;
            const bool ok = pu < p1;
            const int32_t q = ok ? pu : p1 - 1;
            k4[u] = colind[q];
            const float2 vv = REALW ? make_float2(reinterpret_cast<const float*>(vals)[q], 0.f) : vals[q];
            v4[u] = ok ? vv : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < NV; ++h) x4[u][h] = X4[((int64_t)k4[u] * sxr) / 2 + h];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < NV; ++h) {
                if (REALW) {            // (conj of a real weight is itself)
                    acc[2 * h].x = fmaf(v4[u].x, x4[u][h].x, acc[2 * h].x);         acc[2 * h].y = fmaf(v4[u].x, x4[u][h].y, acc[2 * h].y);
                    acc[2 * h + 1].x = fmaf(v4[u].x, x4[u][h].z, acc[2 * h + 1].x); acc[2 * h + 1].y = fmaf(v4[u].x, x4[u][h].w, acc[2 * h + 1].y);
                } else {
                    acc_nz<CONJ>(acc[2 * h], v4[u], make_float2(x4[u][h].x, x4[u][h].y));
                    acc_nz<CONJ>(acc[2 * h + 1], v4[u], make_float2(x4[u][h].z, x4[u][h].w));
                }
            }
    }
#pragma unroll
    for (int off = LPR / 2; off >= CLV; off >>= 1)
#pragma unroll
        for (int u = 0; u < VW; ++u) {
            acc[u].x += __shfl_xor(acc[u].x, off, 64);
            acc[u].y += __shfl_xor(acc[u].y, off, 64);
        }
    if (i == 0 && row_ok && !deferred) {
        const int64_t orow = out_row(wl.yperm, row);
#pragma unroll
        for (int u = 0; u < VW; ++u) {
            float2* yp = Y + (int64_t)(c * VW + u) * ldy + orow;
            float2 out = cmul(alpha, acc[u]);
            if (BMODE == 1) cfma(out, beta, *yp);
            *yp = out;
        }
    }
}

// 64-column row-major panel (packed), rows of some tens of nonzeros (a gridding matrix: 27): a workgroup computes a TILE of 64
// consecutive rows, wave w the rows w, w + 4, w + 8, ... one after the other.  What the one-row-per-wave form above pays three
// dependent round trips for (row pointers -> indices / values -> panel rows), this one overlaps: a wave's 17 pairs of row
// pointers are one load, the indices and values of its next row are requested before the panel rows of the current one are
// waited for.  The four waves work on four CONSECUTIVE rows at any time -- consecutive samples of a trajectory, half a grid
// cell apart, gather mostly the same panel rows -- so their requests meet in the CU's L1 (a first version gave every wave 16
// consecutive rows of its own: 14.3 GB from HBM by the PMC counters instead of 6.4, the reuse a row apart in time instead of
// side by side).  Lane (c, i): panel columns 4c .. 4c+3 (two 16-byte loads), nonzero lane i of 4; a pass covers 28 nonzeros of
// a row.  The 64 x 64 results go through an LDS tile and leave as full 128-byte lines of the column-major result (lanes = 16
// rows x 4 columns).  Rows beyond thr_long nonzeros go to the workgroup-per-row list as everywhere else.
template <bool CONJ, int BMODE, int NW /* waves per workgroup: 4 or 16 */>
__global__ void __launch_bounds__(64 * NW)
k_csrmm_gather_tile64(int64_t M, int64_t nnz, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                      const float2* __restrict__ vals, const float2* __restrict__ X /* [row][64] */,
                      float2* __restrict__ Y, int64_t ldy, float2 alpha, float2 beta, int xcd_remap,
                      WorkLists wl, int32_t thr_long) {
    constexpr int TLD = 65, U = 7, RPW = 64 / NW;     // rows per wave          // 28 nonzeros per pass: a 27-tap gridding row in one, four idle slots fewer than at 8
    __shared__ float2 tile[64 * TLD];
    const int lane = threadIdx.x & 63, c = lane & 15, i = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t blk = xcd_remap ? xcd_block(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
    const int64_t row0 = blk * 64;
    if (row0 >= M) return;
    // lanes 2j / 2j + 1 hold the row pointers p0 / p1 of this wave's j-th row, row0 + wv + NW j
    int32_t myp;
    {
        int64_t r = row0 + wv + NW * (lane >> 1) + (lane & 1);
        if (lane >= 2 * RPW || r > M) r = M;
        myp = rowptr[r];
    }
    const float4* __restrict__ X4 = reinterpret_cast<const float4*>(X) + c * 2;
    const int32_t last = (int32_t)(nnz - 1);
    // The indices and values of a pass: ONE coalesced load each (lane l: nonzero a + l, l < 28), spread to the (column group,
    // nonzero lane) roles through the LDS crossbar afterwards.  (Every lane loading its own copy -- 16 lanes the same address --
    // cost 14 memory instructions per row beside the 14 of the panel rows, and the address unit takes its 16 clocks per wave
    // instruction whatever the addresses are: it was busy 85 % of the kernel, TA_TA_BUSY.)
    auto load_raw = [&](int32_t a, int32_t b, int32_t& rk, float2& rv) {
        rk = 0; rv = make_float2(0.f, 0.f);
        if (lane < 4 * U) {
            const int32_t pu = a + lane;
            const bool ok = pu < b;
            int32_t q = ok ? pu : b - 1;                 // a slot past the row's end re-reads its last nonzero with value 0
            q = q < 0 ? 0 : (q > last ? last : q);
            rk = colind[q];
            const float2 t = vals[q];
            rv = ok ? t : make_float2(0.f, 0.f);
        }
    };
    auto spread = [&](int32_t rk, float2 rv, int32_t (&kk)[U], float2 (&vv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int src = (i + 4 * u) * 4;             // lane (c, i) takes nonzero i + 4u of the pass
            kk[u] = __builtin_amdgcn_ds_bpermute(src, rk);
            vv[u].x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(rv.x)));
            vv[u].y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(rv.y)));
        }
    };
    auto load_idx = [&](int32_t a, int32_t b, int32_t (&kk)[U], float2 (&vv)[U]) {
        int32_t rk; float2 rv;
        load_raw(a, b, rk, rv);
        spread(rk, rv, kk, vv);
    };
    auto gather = [&](const int32_t (&kk)[U], const float2 (&vv)[U], float2 (&acc)[4]) {
        float4 x4[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x4[u][0] = X4[(int64_t)kk[u] * 32];
            x4[u][1] = X4[(int64_t)kk[u] * 32 + 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc_nz<CONJ>(acc[0], vv[u], make_float2(x4[u][0].x, x4[u][0].y));
            acc_nz<CONJ>(acc[1], vv[u], make_float2(x4[u][0].z, x4[u][0].w));
            acc_nz<CONJ>(acc[2], vv[u], make_float2(x4[u][1].x, x4[u][1].y));
            acc_nz<CONJ>(acc[3], vv[u], make_float2(x4[u][1].z, x4[u][1].w));
        }
    };
    unsigned skip = 0;                                   // bit j: this wave's j-th row is someone else's (deferred), or lies past M
    int32_t p0 = __builtin_amdgcn_readlane(myp, 0), p1 = __builtin_amdgcn_readlane(myp, 1);
    int32_t k[U];
    float2 v[U];
    load_idx(p0, p1, k, v);
#pragma unroll 1
    for (int j = 0; j < RPW; ++j) {
        const int jn = j + 1 < RPW ? j + 1 : RPW - 1;
        int32_t q0 = __builtin_amdgcn_readlane(myp, 2 * jn), q1 = __builtin_amdgcn_readlane(myp, 2 * jn + 1);
        if (j == RPW - 1) q1 = q0;                       // nothing after the last row
        const int64_t row = row0 + wv + NW * j;
        bool deferred = false;
        if (p1 - p0 > thr_long && row < M) {
            const bool mine = wl_append(wl, 1, (int)((blk * NW + wv) & (WL_SUB - 1)), lane == 0, (int32_t)row);
            deferred = __builtin_amdgcn_readfirstlane((int)mine) != 0;
        }
        if (deferred || row >= M) skip |= 1u << j;
        float2 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = make_float2(0.f, 0.f);
        int32_t rkn;
        float2 rvn;
        if (!deferred) {
            // (program order = issue order: the panel rows of this row, then the indices of the wave's next row, and the
            // multiply-adds below only wait for the former -- loads return in order)
            float4 x4[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x4[u][0] = X4[(int64_t)k[u] * 32];
                x4[u][1] = X4[(int64_t)k[u] * 32 + 1];
            }
            load_raw(q0, q1, rkn, rvn);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc_nz<CONJ>(acc[0], v[u], make_float2(x4[u][0].x, x4[u][0].y));
                acc_nz<CONJ>(acc[1], v[u], make_float2(x4[u][0].z, x4[u][0].w));
                acc_nz<CONJ>(acc[2], v[u], make_float2(x4[u][1].x, x4[u][1].y));
                acc_nz<CONJ>(acc[3], v[u], make_float2(x4[u][1].z, x4[u][1].w));
            }
            for (int32_t pp = p0 + 4 * U; pp < p1; pp += 4 * U) {        // rows of more than 28 nonzeros: further passes
                int32_t k2[U];
                float2 v2[U];
                load_idx(pp, p1, k2, v2);
                gather(k2, v2, acc);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u].x += __shfl_xor(acc[u].x, 16, 64); acc[u].y += __shfl_xor(acc[u].y, 16, 64);
                acc[u].x += __shfl_xor(acc[u].x, 32, 64); acc[u].y += __shfl_xor(acc[u].y, 32, 64);
            }
        } else {
            load_raw(q0, q1, rkn, rvn);
        }
        if (i == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) tile[(wv + NW * j) * TLD + 4 * c + u] = acc[u];
        }
        spread(rkn, rvn, k, v);
        p0 = q0; p1 = q1;
    }
    // which of the tile's 64 rows are to be stored: wave w's bit j is row w + NW j
    __shared__ unsigned skip_all[NW];
    if (lane == 0) skip_all[wv] = skip;
    __syncthreads();
    // the tile leaves as full lines of the column-major result: lane = row of the tile (64 consecutive rows = 512 bytes of one
    // column per wave store), wave w the columns w, w + NW, ...
    const int tr = lane;                                     // computed by wave tr % NW as its (tr / NW)-th row
    const bool keep = !((skip_all[tr % NW] >> (tr / NW)) & 1u);
#pragma unroll 4
    for (int col = wv; col < 64; col += NW) {
        float2 out = cmul(alpha, tile[tr * TLD + col]);
        float2* dst = Y + (int64_t)col * ldy + row0 + tr;
        if (keep) {
            if (BMODE == 1) cfma(out, beta, *dst);
            *dst = out;
        }
    }
}

// ---- forward product over a 64-column row-major panel, the panel rows of a RUN of 16 matrix rows loaded ONCE, results in REGISTERS ----
// k_csrmm_gather_tile64 loads the panel row of every nonzero: 5e7 x 512 bytes through the vector memory path for a gridding matrix
// whose 16 consecutive samples (half a grid cell apart along a spoke) touch ~100 DISTINCT panel rows with their 432 nonzeros, and
// that path -- ~25-30 bytes per clock and CU for 16-byte-per-lane gathers -- is what bounds it (profiles/r04_cfg3_forward_notes.txt:
// neither whole-line loads, nor software pipelining, nor folding the panel into the L1 moved it).  Here the host groups the
// nonzeros of a run of 16 rows by panel row (ig_csr_runs_build): a wave walks the run's distinct panel rows -- lane = panel column,
// ONE 512-byte load per panel row, eight rows in flight in a ring of registers -- and adds w * row into the result rows of the
// samples that use it.  A first version kept those 16 result rows in LDS (read-add-write per entry): 7.5-10 instructions per entry,
// issue-bound at the old kernel's time.  The results now sit in REGISTERS (lane = column: 16 rows x (re, im) = v[128..159]) addressed
// through the VGPR index mode -- an entry's row is wave-uniform, so its multiply-adds name v128 / v129 with M0 = {destination and
// third source relative, 2 row}: 2 v_readlane + s_mov m0 + 2 v_fma per real-weight entry, no LDS traffic, no read-modify-write
// chain through memory (the technique of k_bricks_wide64r).  The compiler never sees those registers: it is capped at 96 VGPRs
// (amdgpu_num_vgpr), the kernel claims 160, and every access to v96..v159 is an assembly block with literal register numbers.
//   dcols[j]   : panel row (compact column of the matrix) | (number of its entries - 1) << 27, the run's distinct rows in turn
//   entries[e] : {row of the run 0..15, re, im}, grouped by distinct panel row in the order of dcols; a run's entries sit at
//                rowptr[16 run] .. rowptr[16 (run + 1)) like its nonzeros in the CSR
constexpr int RUN_ROWS = 16;
struct RunEntry { uint32_t row; float re, im; };

template <int I, int N, typename F>
__device__ __forceinline__ void static_for_runs(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_runs<I + 1, N>(f);
    }
}
template <int K, int END>
__device__ __forceinline__ void runs_acc_zero() {
    if constexpr (K < END) {
        asm volatile("v_mov_b64 v[%0:%1], 0" :: "i"(96 + 2 * K), "i"(97 + 2 * K));
        runs_acc_zero<K + 1, END>();
    }
}
template <int K, int END>
__device__ __forceinline__ void runs_acc_store(float2* __restrict__ dst /* + K * TLD */, int tld) {
    if constexpr (K < END) {
        float re, im;
        asm volatile("v_mov_b32 %0, v[%2]\n\tv_mov_b32 %1, v[%3]" : "=v"(re), "=v"(im) : "i"(96 + 2 * K), "i"(97 + 2 * K));
        dst[K * tld] = make_float2(re, im);
        runs_acc_store<K + 1, END>(dst, tld);
    }
}

template <int BMODE, bool REALW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(56)))
k_csrmm_runs64r(int64_t M, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ run_dptr, const uint32_t* __restrict__ dcols,
                const RunEntry* __restrict__ entries, const int32_t* __restrict__ run_order /* optional: the runs in the order to process them */,
                const float2* __restrict__ X /* [row][64] */, float2* __restrict__ Y, int64_t ldy, float2 alpha, float2 beta) {
    constexpr int TLD = 65;
    __shared__ float2 tile[64 * TLD];
    asm volatile("" ::: "v127");                          // the kernel owns 128 VGPRs: v0..v55 the compiler's, v56..v59 the prefetched
                                                          // windows, v64..v95 a ring of 16 panel rows, v96..v127 the run's 16 result rows
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // What bounds the kernel is the bytes its panel rows cost in HBM (the loads alone take 1.40 of its 1.58 ms), and a panel row is
    // wanted by ~2.3 runs: runs that are neighbours in SPACE should run close in time behind one L2.  The host may hand over an
    // order of the runs (by the grid brick they start in); every XCD walks a contiguous range of it (blocks are dealt round-robin
    // to the XCDs otherwise).  A run's 16 result rows are whole 128-byte lines of the column-major result, so any order stores
    // full lines.
    const int64_t blk = xcd_block(blockIdx.x, gridDim.x);
    const int64_t nruns = (M + RUN_ROWS - 1) / RUN_ROWS;
    const int64_t slot = blk * 4 + wv;
    const int64_t run = slot < nruns ? (run_order ? (int64_t)run_order[slot] : slot) : nruns;
    asm volatile("s_set_gpr_idx_on %0, 0x0" :: "s"(0));   // index mode on for the whole kernel; M0 = 0: nothing relative
    runs_acc_zero<0, RUN_ROWS>();
    if (run < nruns) {
        const int32_t d0 = run_dptr[run], nd = run_dptr[run + 1] - d0;
        const int64_t rlo = run * RUN_ROWS, rhi = rlo + RUN_ROWS < M ? rlo + RUN_ROWS : M;
        const int32_t e0 = rowptr[rlo], ne = rowptr[rhi] - e0;
        const rsrc_t r_d = make_rsrc(dcols + d0), r_e = make_rsrc(entries + e0);
        // the repacked panel may exceed 2 GB (config 3: 2.57 GB): a descriptor over the whole 4 GB offset range.  Rows past the run's
        // end are requested all the same (row 0: the load counts, so the waits below stay static)
        const rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(X), 0, -1, 0x00020000);
        const unsigned x_voff = (unsigned)lane * 8u;
        // Windows of 64 distinct rows / 64 entries, one per lane, handed out with v_readlane (wave-uniform control).  Inside the loop
        // EVERY load is an assembly block: one load the compiler knows about makes its wait-count pass drain the whole ring in front
        // of each use of the window registers (s_waitcnt vmcnt(0) at the head of every row: 3.5 ms).  The next windows are
        // prefetched into v112..v114 (entries) and v115 (distinct rows) and moved over when the current ones are used up.
        int32_t dbase = 0, ebase = 0;
        uint32_t dv = (uint32_t)buf_ld_i32(r_d, lane < nd ? (unsigned)lane * 4u : IG_OOB);
        struct Win { uint32_t m0w; float re, im; };
        auto cook = [&](u3 raw) { Win wn; wn.m0w = ((raw.x & 15u) * 2u) | 0xC000u; wn.re = __uint_as_float(raw.y); wn.im = __uint_as_float(raw.z); return wn; };
        Win ev = cook(buf_ld_u3(r_e, lane < ne ? (unsigned)lane * 12u : IG_OOB));
        auto prefetch_entries = [&](int32_t base) __attribute__((always_inline)) {     // entries base .. base + 63 -> v112..v114
            const unsigned o = base + lane < ne ? (unsigned)(base + lane) * 12u : IG_OOB;
            const rsrc_t re_ = r_e;
            asm volatile("buffer_load_dwordx3 v[56:58], %0, %1, 0 offen" :: "v"(o), "s"(re_) : "memory");
        };
        auto prefetch_rows = [&](int32_t base) __attribute__((always_inline)) {        // distinct rows base .. base + 63 -> v115
            const unsigned o = base + lane < nd ? (unsigned)(base + lane) * 4u : IG_OOB;
            const rsrc_t rd_ = r_d;
            asm volatile("buffer_load_dword v59, %0, %1, 0 offen" :: "v"(o), "s"(rd_) : "memory");
        };
        auto take_entries = [&]() __attribute__((always_inline)) -> Win {
            u3 raw;
            asm volatile("v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58" : "=v"(raw.x), "=v"(raw.y), "=v"(raw.z));
            return cook(raw);
        };
        prefetch_entries(64);
        prefetch_rows(64);
        int32_t ei = 0;                                   // next entry inside the window `ev`
        int32_t since = 0;                                // panel rows requested since the entry window in flight was asked for
        // panel row j of the run -> ring slot KF (v[64 + 2 KF], v[65 + 2 KF]); returns the row's word (column | (entries - 1) << 27)
        auto request = [&](auto kf, int32_t j) __attribute__((always_inline)) -> uint32_t {
            constexpr int KF = decltype(kf)::value;
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)dv, (j - dbase) & 63);
            const uint32_t o = j < nd ? (w & 0x7ffffffu) * 512u : 0u;       // (past the run's end: panel row 0 once more -- a hit in the L1)
            const rsrc_t rx = r_x;
            const unsigned vo = x_voff;
            asm volatile("s_nop 4\n\tbuffer_load_dwordx2 v[%0:%1], %2, %3, %4 offen" :: "i"(64 + 2 * KF), "i"(65 + 2 * KF), "v"(vo), "s"(rx), "s"(o) : "memory");
            ++since;
            return w;
        };
        // acc[row] += w * x for ONE entry: M0 = {third source and destination relative, 2 row}
#define IG_RUNS_MAC_R(M, WR)      "s_mov_b32 m0, " M "\n\ts_nop 0\n\tv_fma_f32 v96, " WR ", v[%[xr]], v96\n\tv_fma_f32 v97, " WR ", v[%[xi]], v97\n\t"
#define IG_RUNS_MAC_C(M, WR, WI)  IG_RUNS_MAC_R(M, WR) "v_fma_f32 v96, -" WI ", v[%[xi]], v96\n\tv_fma_f32 v97, " WI ", v[%[xr]], v97\n\t"
        auto consume = [&](auto kf, uint32_t w, int32_t j) __attribute__((always_inline)) {
            constexpr int KF = decltype(kf)::value;
            if (j >= nd) return;
            int cnt = (int)(w >> 27) + 1;
            // this row has arrived: fifteen younger panel-row loads may still be out (anything else in flight is younger than they are
            // or older than this row; loads return in order)
            asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            while (cnt > 0) {
                if (ei >= 64) {                           // the window of entries is used up: the prefetched one takes over
                    // it was asked for a window ago: once sixteen panel rows have been requested since, the vmcnt(15) above has
                    // covered it (in-order return); a run of very dense rows gets there sooner and waits for everything
                    if (since < 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    ev = take_entries(); ebase += 64; ei = 0;
                    prefetch_entries(ebase + 64);
                    since = 0;
                    // (that load is younger than every panel row in flight: the static waits stay valid, only stricter)
                }
                const int avail = 64 - ei;
                if (cnt >= 4 && avail >= 4) {
                    uint32_t m[4]; float a[4], bi[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        m[q] = (uint32_t)__builtin_amdgcn_readlane((int)ev.m0w, ei + q);
                        a[q] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ev.re), ei + q));
                        if (!REALW) bi[q] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ev.im), ei + q));
                    }
                    if (REALW)
                        asm volatile(IG_RUNS_MAC_R("%0", "%4") IG_RUNS_MAC_R("%1", "%5") IG_RUNS_MAC_R("%2", "%6") IG_RUNS_MAC_R("%3", "%7") "s_mov_b32 m0, 0"
                                     :: "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "s"(a[0]), "s"(a[1]), "s"(a[2]), "s"(a[3]),
                                        [xr] "i"(64 + 2 * KF), [xi] "i"(65 + 2 * KF) : "memory");
                    else
                        asm volatile(IG_RUNS_MAC_C("%0", "%4", "%8") IG_RUNS_MAC_C("%1", "%5", "%9") IG_RUNS_MAC_C("%2", "%6", "%10") IG_RUNS_MAC_C("%3", "%7", "%11") "s_mov_b32 m0, 0"
                                     :: "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "s"(a[0]), "s"(a[1]), "s"(a[2]), "s"(a[3]),
                                        "s"(bi[0]), "s"(bi[1]), "s"(bi[2]), "s"(bi[3]), [xr] "i"(64 + 2 * KF), [xi] "i"(65 + 2 * KF) : "memory");
                    ei += 4; cnt -= 4;
                } else {
                    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)ev.m0w, ei);
                    const float a = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ev.re), ei));
                    if (REALW) {
                        asm volatile(IG_RUNS_MAC_R("%0", "%1") "s_mov_b32 m0, 0" :: "s"(m), "s"(a), [xr] "i"(64 + 2 * KF), [xi] "i"(65 + 2 * KF) : "memory");
                    } else {
                        const float bi = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ev.im), ei));
                        asm volatile(IG_RUNS_MAC_C("%0", "%1", "%2") "s_mov_b32 m0, 0" :: "s"(m), "s"(a), "s"(bi), [xr] "i"(64 + 2 * KF), [xi] "i"(65 + 2 * KF) : "memory");
                    }
                    ei += 1; cnt -= 1;
                }
            }
        };
#undef IG_RUNS_MAC_C
#undef IG_RUNS_MAC_R
        // ring of sixteen: rows d .. d + 15 in flight while row d is consumed; slot KF is refilled (row d + 16) right after its row is
        // used up.  (What bounds the kernel is bytes in flight: with eight rows per wave and three waves per SIMD -- 48 KB per CU --
        // it fetched its 5.9 GB at 3.2 TB/s, 1.87 ms, the per-nonzero gather's time.)
        uint32_t wr_[16];
        auto fill = [&](auto k) __attribute__((always_inline)) { wr_[decltype(k)::value] = request(k, decltype(k)::value); };
        static_for_runs<0, 16>(fill);
        for (int32_t d = 0; d < nd; d += 16) {
            const int32_t dn = d + 16;
            const bool move = dn < nd && ((dn - dbase) & 63) == 0;
            auto step = [&](auto k) __attribute__((always_inline)) {
                constexpr int KF = decltype(k)::value;
                consume(k, wr_[KF], d + KF);
                if (KF == 0 && move) {                    // (the next window was asked for 64 rows ago: the vmcnt(15) above covers it)
                    asm volatile("v_mov_b32 %0, v59" : "=v"(dv));
                    dbase = dn;
                    prefetch_rows(dn + 64);
                }
                wr_[KF] = request(k, dn + KF);
            };
            static_for_runs<0, 16>(step);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the out-of-range tail requests write the ring registers too)
    }
    asm volatile("s_set_gpr_idx_off");
    if (run >= nruns) return;
    // the run's results: registers -> the wave's quarter of the tile -> full 128-byte lines of the column-major result (lanes = 16
    // rows x 4 columns).  Only this wave touches its quarter: LDS operations of a wave execute in order, the fence pins that for
    // the compiler.
    float2* __restrict__ mine = tile + (wv * RUN_ROWS) * TLD;
    runs_acc_store<0, RUN_ROWS>(mine + lane, TLD);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int r16 = lane & 15, cg = lane >> 4;
    const int64_t orow = run * RUN_ROWS + r16;
    const bool keep = orow < M;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int col = cg + 4 * i;
        float2 out = cmul(alpha, mine[r16 * TLD + col]);
        float2* dst = Y + (int64_t)col * ldy + orow;
        if (keep) {
            if (BMODE == 1) cfma(out, beta, *dst);
            *dst = out;
        }
    }
}

// Row-per-lane variant for matrices whose rows are mostly empty or very short (mean <= 1 nonzero per
// row, e.g. the transposed gridding matrix: 89 % empty rows).  A lane owns a row and keeps NC panel
// columns in registers, so a wave covers 64 rows, every store instruction writes 512 contiguous bytes
// of one panel column, and each nonzero issues NC independent gathers.  With one row per 8 lanes the
// same matrix needs 8x more waves, each a short dependent chain: that version is latency bound.
// BUF: index/value/panel loads go through buffer descriptors (all three arrays < 2 GB): the lanes of a trip that
// are past their row's end get an out-of-range offset, i.e. no memory access at all, without a branch.
template <int NC, bool CONJ, int BMODE, bool PACKED, int U, bool BUF>
__global__ void __launch_bounds__(BLK)
k_csrmm_rowlane(int64_t M, int64_t N,
                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                const float2* __restrict__ vals,
                const float2* __restrict__ X, int64_t ldx, int64_t sxr,
                float2* __restrict__ Y, int64_t ldy,
                float2 alpha, float2 beta, int xcd_remap,
                WorkLists wl, int32_t thr_mid, int32_t thr_long, GridMask mask) {
    const int64_t blk = xcd_remap ? xcd_block(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
    const int64_t row = blk * BLK + threadIdx.x;
    bool live = row < M;
    if (live && mask.bits) {
        // segments (16 rows) outside the gridding support hold no nonzero and nobody reads them: no load, no store.
        // 32-bit index arithmetic: a grid has < 2^31 rows.
        const uint32_t r32 = (uint32_t)row, n0 = (uint32_t)mask.n0, nm = (uint32_t)mask.nm;
        const uint32_t rest = r32 / n0, kx = r32 - rest * n0;
        const uint32_t ks = rest / nm, km = rest - ks * nm;
        const uint32_t wbits = mask.bits[((size_t)ks * (n0 >> 4) + (kx >> 4)) * 16 + (km & 15)];
        live = (wbits >> (km >> 4)) & 1u;
    }
    if (!__ballot(live)) return;
    int32_t p0 = 0, p1 = 0;
    if (live) { p0 = rowptr[row]; p1 = rowptr[row + 1]; }
    const int32_t len = p1 - p0;
    {
        const int sub = (int)((blk * WAVES_PER_BLOCK + (threadIdx.x >> 6)) & (WL_SUB - 1));
        if (wl_append(wl, 0, sub, len > thr_mid && len <= thr_long, (int32_t)row)) live = false;
        if (wl_append(wl, 1, sub, len > thr_long, (int32_t)row)) live = false;
    }
    if (!live) return;
    for (int64_t jb = 0; jb < N; jb += NC) {
        float2 acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = make_float2(0.f, 0.f);
        // U nonzeros per trip, predicated (index clamped to the row's last nonzero, value zeroed): the kernel is
        // bound by the latency of its dependent chain (rowptr -> index/value -> panel row) times the few waves a
        // CU holds, so U independent panel-row gathers in flight per lane shorten it almost U-fold.
        if (BUF) {
            const rsrc_t rc = make_rsrc(colind), rv = make_rsrc(vals), rx = make_rsrc(X);
            for (int32_t p = p0; p < p1; p += U) {
                int32_t k[U];
                float2 v[U];
                float4 t4[U][NC / 2 > 0 ? NC / 2 : 1];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = p + u < p1;
                    k[u] = buf_ld_i32(rc, ok ? (unsigned)(p + u) * 4u : IG_OOB);
                    v[u] = buf_ld<false>(rv, ok ? (unsigned)(p + u) * 8u : IG_OOB, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = p + u < p1;
                    const unsigned xo = ((unsigned)k[u] * (unsigned)sxr + (unsigned)jb) * 8u;
#pragma unroll
                    for (int h = 0; h < NC / 2; ++h) t4[u][h] = buf_ld_f4(rx, ok ? xo + 16u * h : IG_OOB);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int h = 0; h < NC / 2; ++h) {
                        acc_nz<CONJ>(acc[2 * h], v[u], make_float2(t4[u][h].x, t4[u][h].y));
                        acc_nz<CONJ>(acc[2 * h + 1], v[u], make_float2(t4[u][h].z, t4[u][h].w));
                    }
            }
        } else
        for (int32_t p = p0; p < p1; p += U) {
            int32_t k[U];
            float2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = p + u < p1;
                const int32_t q = ok ? p + u : p1 - 1;
                k[u] = colind[q];
                const float2 vv = vals[q];
                v[u] = ok ? vv : make_float2(0.f, 0.f);
            }
            if (PACKED && NC >= 2) {
                // packed panel: row k holds its columns contiguously (sxr elements per row); this chunk of NC
                // columns is NC*8 contiguous, 16-byte aligned bytes
                float4 t4[U][NC / 2 > 0 ? NC / 2 : 1];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4* __restrict__ q = reinterpret_cast<const float4*>(X + (int64_t)k[u] * sxr + jb);
#pragma unroll
                    for (int h = 0; h < NC / 2; ++h) t4[u][h] = q[h];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int h = 0; h < NC / 2; ++h) {
                        acc_nz<CONJ>(acc[2 * h], v[u], make_float2(t4[u][h].x, t4[u][h].y));
                        acc_nz<CONJ>(acc[2 * h + 1], v[u], make_float2(t4[u][h].z, t4[u][h].w));
                    }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (jb + c < N)
                            acc_nz<CONJ>(acc[c], v[u], X[PACKED ? (int64_t)k[u] * sxr + jb + c : (jb + c) * ldx + k[u]]);
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (jb + c < N) {
                float2* yp = Y + (jb + c) * ldy + out_row(wl.yperm, row);
                float2 out = cmul(alpha, acc[c]);
                if (BMODE == 1) cfma(out, beta, *yp);
                *yp = out;
            }
        }
    }
}

// Dense-lane variant of the row-per-lane kernel for matrices with mostly empty rows (transposed gridding: 89 %
// empty, the rest 3.4 nonzeros on average).  With a lane walking its own row only ~14 % of the lanes of a gather
// instruction are active, and the texture-address unit spends the same cycles on it as on a full one: that, not
// latency or HBM, bounded the row-per-lane kernel (2.4 ms; unrolling the row loop changed nothing).  Here the
// nonzeros of a wave's 64 rows -- one contiguous range of the CSR arrays -- are processed 64 at a time, one per lane,
// all lanes active: coalesced index/value loads, one gathered panel row per lane, products parked in LDS; then each
// row's lane sums its own run of products (LDS reads, no atomics) and stores as before, 512 contiguous bytes per
// panel column.  Rows beyond thr_mid go to the deferred-row lists and are skipped here.
template <int NC, bool CONJ, bool YIL>
__global__ void __launch_bounds__(BLK, 1)     // (74 VGPRs, 6 waves/SIMD.  Measured: a 72-VGPR cap for 7 waves is 4 % SLOWER
                                                              // (1.68 vs 1.61 ms), a 64-VGPR cap for 8 waves spills and takes 2.69 ms: not occupancy bound)
k_csrmm_dense64(int64_t M, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                const float2* __restrict__ vals, const float2* __restrict__ Xp,
                float2* __restrict__ Y, int64_t ldy, float2 alpha,
                WorkLists wl, int32_t thr_mid, int32_t thr_long, GridMask mask, int coalesce) {
    __shared__ float4 prod[WAVES_PER_BLOCK][64][NC / 2 > 0 ? NC / 2 : 1];     // NC == 1 uses the .xy half of a slot
    static_assert(NC == 1 || NC % 2 == 0, "one column or an even number of them");
    static_assert(NC > 1 || !YIL, "a single column has no interleaved form");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // A wave strides over the 64-row tasks (the launcher normally gives every task its own wave; with a capped grid
    // the next task's row pointers and bitmap word are fetched while the current one is processed).
    const int64_t ntasks = (M + 63) / 64;
    const int64_t stride = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    const rsrc_t rc = make_rsrc(colind), rv = make_rsrc(vals), rx = make_rsrc(Xp);

    auto fetch = [&](int64_t task, int32_t& R, int32_t& R1, uint32_t& wbits, uint32_t& km) {
        const int64_t row = task * 64 + lane;
        const bool inb = task < ntasks && row < M;
        R = rowptr[inb ? row : M];
        R1 = rowptr[inb ? row + 1 : M];
        wbits = inb ? 0xffffffffu : 0u;
        km = 0;
        if (inb && mask.bits) {
            // 32-bit index arithmetic: a grid has < 2^31 rows
            const uint32_t r32 = (uint32_t)row, n0 = (uint32_t)mask.n0, nm = (uint32_t)mask.nm;
            const uint32_t rest = r32 / n0, kx = r32 - rest * n0;
            const uint32_t ks = rest / nm;
            km = rest - ks * nm;
            wbits = mask.bits[((size_t)ks * (n0 >> 4) + (kx >> 4)) * 16 + (km & 15)];
        }
    };

    int64_t task = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wv;
    int32_t R, R1;
    uint32_t wbits, km;
    fetch(task, R, R1, wbits, km);
    for (; task < ntasks; task += stride) {
        int32_t Rn, R1n;
        uint32_t wbn, kmn;
        fetch(task + stride, Rn, R1n, wbn, kmn);
        const int64_t row = task * 64 + lane;
        const bool live = (wbits >> (km >> 4)) & 1u;      // segments (16 rows) outside the support: no load, no store
        if (__ballot(live)) {
            const int32_t len = R1 - R;
            bool mine = live;                                 // this lane computes and stores its row here
            {
                const int sub = (int)(task & (WL_SUB - 1));
                if (wl_append(wl, 0, sub, live && len > thr_mid && len <= thr_long, (int32_t)row)) mine = false;
                if (wl_append(wl, 1, sub, live && len > thr_long, (int32_t)row)) mine = false;
            }
            // compact numbering of the wave's inline nonzeros: row r owns [pe - n_r, pe)
            const int32_t n_r = mine ? len : 0;
            int32_t pe = n_r;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int32_t t = __shfl_up(pe, d, 64);
                if (lane >= d) pe += t;
            }
            const int32_t ps = pe - n_r;
            const int32_t total = __shfl(pe, 63, 64);

            float2 acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = make_float2(0.f, 0.f);
            for (int32_t base = 0; base < total; base += 64) {
                const int32_t q = base + lane;
                const bool ok = q < total;
                // owner of compact nonzero q: the first row whose inclusive prefix exceeds q
                int lo = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int32_t t = __shfl(pe, lo + step - 1, 64);
                    if (t <= q) lo += step;
                }
                lo &= 63;
                const int32_t p = __shfl(R, lo, 64) + (q - __shfl(ps, lo, 64));
                const int32_t k = buf_ld_i32(rc, ok ? (unsigned)p * 4u : IG_OOB);
                const float2 v = buf_ld<false>(rv, ok ? (unsigned)p * 8u : IG_OOB, 0);
                if constexpr (NC == 1) {
                    const float2 x1 = buf_ld<false>(rx, ok ? (unsigned)k * 8u : IG_OOB, 0);
                    float2 a = make_float2(0.f, 0.f);
                    acc_nz<CONJ>(a, v, x1);
                    prod[wv][lane][0] = make_float4(a.x, a.y, 0.f, 0.f);
                } else {
                    float4 x4[NC / 2 > 0 ? NC / 2 : 1];
#pragma unroll
                    for (int h = 0; h < NC / 2; ++h) x4[h] = buf_ld_f4(rx, ok ? (unsigned)k * (NC * 8u) + 16u * h : IG_OOB);
#pragma unroll
                    for (int h = 0; h < NC / 2; ++h) {
                        float2 a = make_float2(0.f, 0.f), b = a;
                        acc_nz<CONJ>(a, v, make_float2(x4[h].x, x4[h].y));
                        acc_nz<CONJ>(b, v, make_float2(x4[h].z, x4[h].w));
                        prod[wv][lane][h] = make_float4(a.x, a.y, b.x, b.y);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int32_t j0 = (ps > base ? ps : base) - base, j1 = (pe < base + 64 ? pe : base + 64) - base;
                for (int32_t j = j0; j < j1; ++j) {
                    if constexpr (NC == 1) {
                        const float4 t = prod[wv][j][0];
                        acc[0].x += t.x; acc[0].y += t.y;
                    } else {
#pragma unroll
                        for (int h = 0; h < NC / 2; ++h) {
                            const float4 t = prod[wv][j][h];
                            acc[2 * h].x += t.x; acc[2 * h].y += t.y;
                            acc[2 * h + 1].x += t.z; acc[2 * h + 1].y += t.w;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if constexpr (NC == 1) {
                if (mine) Y[out_row(wl.yperm, row)] = cmul(alpha, acc[0]);
            } else
            if (YIL && !wl.yperm && coalesce) {
                // Row-major panel: the wave's 64 rows are NC*512 contiguous bytes.  The rows pass through LDS so that
                // every store instruction writes 1 KB of consecutive addresses (a lane storing its own row would put
                // 16 bytes into each of 64 different 64-byte rows per instruction).
                const uint64_t mmask = __ballot(mine);
#pragma unroll
                for (int h = 0; h < NC / 2; ++h) {
                    const float2 a = cmul(alpha, acc[2 * h]), b = cmul(alpha, acc[2 * h + 1]);
                    prod[wv][lane][h] = make_float4(a.x, a.y, b.x, b.y);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float4* lin = &prod[wv][0][0];
                float4* yp = reinterpret_cast<float4*>(Y + task * 64 * NC);
#pragma unroll
                for (int h = 0; h < NC / 2; ++h) {
                    const int e = h * 64 + lane;                 // 16-byte element of the wave's block; row e / (NC/2)
                    if ((mmask >> (e / (NC / 2))) & 1ull) yp[e] = lin[e];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else if (mine) {
                const int64_t orow = out_row(wl.yperm, row);
                if (YIL) {
                    float4* yp = reinterpret_cast<float4*>(Y + orow * NC);
#pragma unroll
                    for (int h = 0; h < NC / 2; ++h) {
                        const float2 a = cmul(alpha, acc[2 * h]), b = cmul(alpha, acc[2 * h + 1]);
                        yp[h] = make_float4(a.x, a.y, b.x, b.y);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) Y[c * ldy + orow] = cmul(alpha, acc[c]);
                }
            }
        }
        R = Rn; R1 = R1n; wbits = wbn; km = kmn;
    }
}

// Wide row-major panel (exactly 64 columns, packed), short rows: the adjoint of a gridding matrix applied to a
// 64-column panel (BASELINE config 3).  A lane per row (k_csrmm_rowlane) gathers 64-byte pieces of 64 different panel
// rows per instruction and walks its row once per column chunk; here the lanes run along the 64 COLUMNS, so one
// nonzero is one coalesced 512-byte read, a wave sums 16 consecutive rows one after the other (their nonzeros -- one
// contiguous CSR range -- are fetched 64 at a time, one per lane, and broadcast), and the workgroup's 64 x 64 result
// tile goes through LDS so that the stores are 512 contiguous bytes per panel column.
template <bool CONJ, int BMODE>
__global__ void __launch_bounds__(BLK)
k_csrmm_rowtile64(int64_t M, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                  const float2* __restrict__ vals, const float2* __restrict__ Xp,
                  float2* __restrict__ Y, int64_t ldy, float2 alpha, float2 beta,
                  WorkLists wl, int32_t thr_mid, int32_t thr_long) {
    __shared__ float2 tile[64][65];
    __shared__ unsigned char skip[64];                   // row was handed to the deferred-row kernels
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 64 + wv * 16;      // this wave's 16 rows
    // row pointers of the 16 rows (+1) in lanes 0..16
    const int64_t rr = row0 + (lane < 17 ? lane : 16);
    const int32_t rp = rowptr[rr < M ? rr : M];
    const int32_t rpn = __shfl_down(rp, 1, 64);
    const int32_t len = (lane < 16 && row0 + lane < M) ? rpn - rp : 0;
    bool deferred = false;
    {
        const int sub = (int)((blockIdx.x * WAVES_PER_BLOCK + wv) & (WL_SUB - 1));
        if (wl_append(wl, 0, sub, len > thr_mid && len <= thr_long, (int32_t)(row0 + lane))) deferred = true;
        if (wl_append(wl, 1, sub, len > thr_long, (int32_t)(row0 + lane))) deferred = true;
    }
    if (lane < 16) skip[wv * 16 + lane] = deferred ? 1 : 0;
    const uint64_t defmask = __ballot(deferred);
    for (int r = 0; r < 16; ++r) tile[wv * 16 + r][lane] = make_float2(0.f, 0.f);
    const int32_t P0 = __shfl(rp, 0, 64), P1 = __shfl(rp, 16, 64);
    float2 acc = make_float2(0.f, 0.f);
    int cur = 0;                                          // row (0..15) the accumulator belongs to
    for (int32_t base = P0; base < P1; base += 64) {
        const int32_t p = base + lane;
        const bool ok = p < P1;
        const int32_t k_i = ok ? colind[p] : 0;
        const float2 v_i = ok ? vals[p] : make_float2(0.f, 0.f);
        // owner row of nonzero p: the last of the 16 rows whose pointer is <= p
        int r_i = 0;
#pragma unroll
        for (int step = 8; step >= 1; step >>= 1) {
            const int32_t t = __shfl(rp, r_i + step, 64);
            if (t <= p) r_i += step;
        }
        const int nb = (P1 - base) < 64 ? (P1 - base) : 64;
        constexpr int UT = 16;                             // independent 512-byte panel-row reads in flight per wave
        for (int j0 = 0; j0 < nb; j0 += UT) {
            int32_t kk[UT]; float2 vv[UT], xx[UT]; int rj[UT];
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                const int j = j0 + u < nb ? j0 + u : nb - 1;
                kk[u] = __shfl(k_i, j, 64);
                vv[u] = make_float2(__shfl(v_i.x, j, 64), __shfl(v_i.y, j, 64));
                rj[u] = __shfl(r_i, j, 64);
                if (j0 + u >= nb) vv[u] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UT; ++u) xx[u] = Xp[(int64_t)kk[u] * 64 + lane];
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                if (rj[u] != cur) {                       // wave-uniform: rows change in CSR order
                    tile[wv * 16 + cur][lane] = acc;
                    acc = make_float2(0.f, 0.f);
                    cur = rj[u];
                }
                if (!((defmask >> rj[u]) & 1ull)) acc_nz<CONJ>(acc, vv[u], xx[u]);
            }
        }
    }
    tile[wv * 16 + cur][lane] = acc;
    __syncthreads();
    // column-wise stores: wave wv writes columns wv*16 .. +15, lanes = the workgroup's 64 rows
    const int64_t orow = (int64_t)blockIdx.x * 64 + lane;
    if (orow < M && !skip[lane]) {
#pragma unroll 4
        for (int c = wv * 16; c < wv * 16 + 16; ++c) {
            float2* yp = Y + (int64_t)c * ldy + out_row(wl.yperm, orow);
            float2 out = cmul(alpha, tile[lane][c]);
            if (BMODE == 1) cfma(out, beta, *yp);
            *yp = out;
        }
    }
}

// (A cooperative variant -- the 64 rows' nonzeros compacted into an LDS list, four lanes per nonzero,
// per-row LDS accumulators with ds_add_f32 -- was built and measured on the 134M-row transposed gridding
// matrix: 9.2 ms against 4.3 ms for the kernel above; 3.6 ms of that were the LDS float atomics on
// same-row addresses and the list/scan bookkeeping cost more than the scattered loads it removed.)

// one wavefront per listed row: 64/CL nonzero-lanes x CL column-lanes, four gathers in flight per lane
template <int CL, bool CONJ, int BMODE>
__global__ void __launch_bounds__(BLK)
k_csrmm_rows_wave(int64_t N, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                  const float2* __restrict__ vals, const float2* __restrict__ X, int64_t ldx, int64_t sxr,
                  float2* __restrict__ Y, int64_t ldy, int64_t syr, float2 alpha, float2 beta,
                  const int32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t cap,
                  const int32_t* __restrict__ yperm) {
    constexpr int NLW = 64 / CL;
    const int lane = threadIdx.x & 63;
    const int c = lane % CL, i = lane / CL;
    // wave g serves sub-list g % WL_SUB (the grid has a multiple of WL_SUB waves)
    const uint32_t gw = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const uint32_t sub = gw % WL_SUB, nwaves = gridDim.x * WAVES_PER_BLOCK / WL_SUB;
    uint32_t n = count[sub];
    if (n > cap) n = cap;
    list += (size_t)sub * cap;
    for (uint32_t e = gw / WL_SUB; e < n; e += nwaves) {
        const int64_t row = list[e];
        const int32_t p0 = rowptr[row], p1 = rowptr[row + 1];
        for (int64_t jb = 0; jb < N; jb += CL) {
            const int64_t j = jb + c;
            const bool col_ok = j < N;
            const float2* __restrict__ xcol = X + (col_ok ? j : 0) * ldx;
            float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            int32_t p = p0 + i;
            for (; p + 3 * NLW < p1; p += 4 * NLW) {
                const int32_t k0 = colind[p], k1 = colind[p + NLW], k2 = colind[p + 2 * NLW], k3 = colind[p + 3 * NLW];
                const float2 v0 = vals[p], v1 = vals[p + NLW], v2 = vals[p + 2 * NLW], v3 = vals[p + 3 * NLW];
                const float2 x0 = xcol[k0 * sxr], x1 = xcol[k1 * sxr], x2 = xcol[k2 * sxr], x3 = xcol[k3 * sxr];
                acc_nz<CONJ>(a0, v0, x0); acc_nz<CONJ>(a1, v1, x1);
                acc_nz<CONJ>(a2, v2, x2); acc_nz<CONJ>(a3, v3, x3);
            }
            for (; p < p1; p += NLW) acc_nz<CONJ>(a0, vals[p], xcol[colind[p] * sxr]);
            float2 acc = cadd(cadd(a0, a1), cadd(a2, a3));
#pragma unroll
            for (int off = 32; off >= CL; off >>= 1) {
                acc.x += __shfl_xor(acc.x, off, 64);
                acc.y += __shfl_xor(acc.y, off, 64);
            }
            if (i == 0 && col_ok) {
                float2* yp = Y + j * ldy + out_row(yperm, row) * syr;
                float2 out = cmul(alpha, acc);
                if (BMODE == 1) cfma(out, beta, *yp);
                *yp = out;
            }
        }
    }
}

// one 1024-thread workgroup per listed row; partial sums meet in LDS in a fixed order (deterministic)
template <int CL, bool CONJ, int BMODE>
__global__ void __launch_bounds__(1024)
k_csrmm_rows_block(int64_t N, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                   const float2* __restrict__ vals, const float2* __restrict__ X, int64_t ldx, int64_t sxr,
                   float2* __restrict__ Y, int64_t ldy, int64_t syr, float2 alpha, float2 beta,
                   const int32_t* __restrict__ list, const uint32_t* __restrict__ count, uint32_t cap,
                   const int32_t* __restrict__ yperm) {
    constexpr int NLB = 1024 / CL;
    __shared__ float2 part[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = tid % CL, i = tid / CL;
    const uint32_t sub = blockIdx.x % WL_SUB, nblk = gridDim.x / WL_SUB;     // grid is a multiple of WL_SUB blocks
    uint32_t n = count[sub];
    if (n > cap) n = cap;
    list += (size_t)sub * cap;
    for (uint32_t e = blockIdx.x / WL_SUB; e < n; e += nblk) {
        const int64_t row = list[e];
        const int32_t p0 = rowptr[row], p1 = rowptr[row + 1];
        for (int64_t jb = 0; jb < N; jb += CL) {
            const int64_t j = jb + c;
            const bool col_ok = j < N;
            const float2* __restrict__ xcol = X + (col_ok ? j : 0) * ldx;
            float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            int32_t p = p0 + i;
            for (; p + 3 * NLB < p1; p += 4 * NLB) {
                const int32_t k0 = colind[p], k1 = colind[p + NLB], k2 = colind[p + 2 * NLB], k3 = colind[p + 3 * NLB];
                const float2 v0 = vals[p], v1 = vals[p + NLB], v2 = vals[p + 2 * NLB], v3 = vals[p + 3 * NLB];
                const float2 x0 = xcol[k0 * sxr], x1 = xcol[k1 * sxr], x2 = xcol[k2 * sxr], x3 = xcol[k3 * sxr];
                acc_nz<CONJ>(a0, v0, x0); acc_nz<CONJ>(a1, v1, x1);
                acc_nz<CONJ>(a2, v2, x2); acc_nz<CONJ>(a3, v3, x3);
            }
            for (; p < p1; p += NLB) acc_nz<CONJ>(a0, vals[p], xcol[colind[p] * sxr]);
            float2 acc = cadd(cadd(a0, a1), cadd(a2, a3));
            // fold the nonzero-lanes that share a wave (lanes with equal c), then across the 16 waves
#pragma unroll
            for (int off = 32; off >= CL; off >>= 1) {
                acc.x += __shfl_xor(acc.x, off, 64);
                acc.y += __shfl_xor(acc.y, off, 64);
            }
            part[wid][lane] = acc;
            __syncthreads();
            if (tid < CL) {
                float2 sum = make_float2(0.f, 0.f);
#pragma unroll
                for (int w = 0; w < 16; ++w) sum = cadd(sum, part[w][tid]);
                if (col_ok) {
                    float2* yp = Y + j * ldy + out_row(yperm, row) * syr;
                    float2 out = cmul(alpha, sum);
                    if (BMODE == 1) cfma(out, beta, *yp);
                    *yp = out;
                }
            }
            __syncthreads();
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

// X(rows x N, column-major, leading dim ld)  ->  Xp[rows][NP] with the NP (= pow2 >= N) columns of a
// row contiguous; pad columns are zero.  Used when every panel row is gathered many times.
template <int NP>
__global__ void __launch_bounds__(BLK)
k_pack_panel(int64_t rows, int64_t N, const float2* __restrict__ X, int64_t ld, float2* __restrict__ Xp,
             const int32_t* __restrict__ xperm) {
    // xperm (optional): packed row k is panel row xperm[k] (the matrix's columns were renumbered for locality)
    for (int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x; e < rows * NP; e += (int64_t)gridDim.x * BLK) {
        const int64_t k = e / NP;
        const int c = (int)(e % NP);
        const int64_t src = xperm ? (int64_t)xperm[k] : k;
        Xp[e] = c < N ? X[c * ld + src] : make_float2(0.f, 0.f);
    }
}

// The same repacking for wide panels (NP = 16, 32, 64) as an LDS-tiled transpose: a workgroup moves 64 panel
// rows x NP columns, reading 512 contiguous bytes per column and writing NP*8 contiguous bytes per row.
template <int NP>
__global__ void __launch_bounds__(BLK)
k_pack_panel_tiled(int64_t rows, int64_t N, const float2* __restrict__ X, int64_t ld, float2* __restrict__ Xp,
                   const int32_t* __restrict__ xperm) {
    __shared__ float2 tile[NP][65];
    const int tid = threadIdx.x;
    const int kk = tid & 63, c0 = tid >> 6;
    constexpr int PER = NP / (BLK / 64);                 // columns per thread
    // software-pipelined over the workgroup's tiles: the loads of the next tile are in flight while this one is written out
    float2 v[PER];
    auto load = [&](int64_t k0) {
        const int64_t k = k0 + kk;
        const int64_t src = k < rows ? (xperm ? (int64_t)xperm[k] : k) : 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = c0 + u * (BLK / 64);
            v[u] = (c < N && k < rows) ? X[c * ld + src] : make_float2(0.f, 0.f);
        }
    };
    int64_t k0 = (int64_t)blockIdx.x * 64;
    if (k0 < rows) load(k0);
    for (; k0 < rows; k0 += (int64_t)gridDim.x * 64) {
#pragma unroll
        for (int u = 0; u < PER; ++u) tile[c0 + u * (BLK / 64)][kk] = v[u];
        __syncthreads();
        const int64_t kn = k0 + (int64_t)gridDim.x * 64;
        if (kn < rows) load(kn);
        for (int e = tid; e < 64 * NP; e += BLK) {
            const int r = e / NP, c = e % NP;
            if (k0 + r < rows) Xp[(k0 + r) * NP + c] = tile[c][r];
        }
        __syncthreads();
    }
}

// ---- brick-binned adjoint gridding ------------------------------------------------------------------------------
// Y_il = alpha * G^H X for a gridding matrix G (rows = k-space samples, ~27 taps each, columns = grid points numbered
// kx + n0*(km + nm*ks)) -- the SCATTER view, made race-free by binning.  The grid is cut into bricks of 16 x BM x BS cells.
// On the host the nonzeros are sorted by the brick of their column (stable: inside a brick they stay in sample order)
// into 12-byte entries {cell inside the brick, value}; the entries one sample has in one brick are padded to a multiple of
// 64/NC -- a ROUND: one wave instruction of the accumulation, 64/NC entries x NC coils, only ever holds entries of one
// sample, i.e. distinct cells -- and the sample of every round goes into a separate list (round_rows).  A WAVE owns a run of
// consecutive non-empty bricks: it streams their entries (coalesced, no pointer chasing, nothing depends on a previous
// load), gathers X[sample, :] and accumulates conj(value) * X into ONE LDS image of the current brick with plain
// read-add-write (no other wave touches that image, no two lanes of an instruction share a cell, LDS operations of a wave
// execute in order), and at each brick boundary stores the flagged 16-row segments (512-byte wave stores) and clears the
// image.  Bricks near the k-space centre hold 10^4..10^5 nonzeros: their entry ranges are cut into pieces, each piece adds
// its image into the (pre-zeroed) grid with float atomics.
//
// Against the gather over the transpose (k_csrmm_dense64): no 134-M-entry row-pointer array, no per-row segmented sums,
// no deferred long rows.  Measured on the way (BASELINE config 4, 8 coils; the gather took 1.82 ms): per-brick SAMPLE lists
// walked through pairs -> rowptr -> colind/vals (6.7 ms: dependent loads at 8 waves per CU); a workgroup per brick with LDS
// float atomics (5.4 ms: ds_add_f32 retires about one lane every four clocks); a wave per brick, every lane loading its own
// copy of the entry and the X value (1.60 ms); runs of bricks (1.31 ms); the form below (0.91 ms).
struct BrickTask { int32_t lo, hi, bt, nb_flags; };     // entries [lo, hi) = bricks table[bt .. bt + (nb_flags & 0xffff)); bit 16: shared
struct BrickRef { int32_t brick, end; };                  // a non-empty brick and where its entries end
struct BrickEntry { uint32_t cell; float re, im; };       // 12 bytes; cell == 0xffffffff: padding

// One wave per task.  A task is a run of consecutive non-empty bricks (a few thousand entries, at most 64 bricks and
// 512 segments) or a piece of one heavy brick (shared).  The wave keeps ONE brick image in LDS and walks the run in
// super-trips of 64 entries.  Loading and accumulating use different lane roles:
//   * loading: lane e fetches entry e of the super-trip (ONE 12-byte load per lane: 768 useful bytes per instruction), and
//     lane (r, coil) fetches X[sample of round r, coil] (one load: the 64/TPR panel rows of the super-trip), the samples
//     coming from `round_rows` a super-trip earlier.  A super-trip in flight costs 6 registers, so entries are requested
//     three super-trips (192 entries) ahead and panel rows two;
//   * accumulating: in round r lane (t, coil) takes entry r*TPR + t and X[round r, coil] from the loading lanes with
//     ds_bpermute and adds conj(v) * x into the image with a plain read-add-write.
// An earlier form gave every lane of a round its own copy of the entry and the panel value straight from memory: eight
// times the load instructions (address-unit time) and registers, which capped the prefetch depth at 64 entries.
// At a brick boundary (wave-uniform test, once per round -- a brick holds at least one round) the image is stored and
// cleared; with NSEG > 0 the segment loop is unrolled.  Brick ends, grid offsets and segment flags of the run sit one
// brick per lane (v_readlane).
// PAIR: the support table has 4 kx points per entry (16 / 4 x bm x bs = 2 NSEG segments per brick), and the flush handles them two
// at a time -- segments 2 p and 2 p + 1 are the two halves of the same 8 cells x NC values a wave stores with one instruction, so the
// loop keeps the shape of the 8-point table's (NSEG unrolled stores) and a half-wave is predicated by its own flag.  (Unrolled 16
// times the flush spilled; the generic loop cost 0.86 against 0.76 ms and ate what the finer table saves the z passes.)
// REALW: the entries are 8 bytes {cell, re} -- every weight real (a gridding matrix times the +-1 modulation of a centred transform on
// an even grid): a third less of the format to read, four ds_bpermute per round instead of five, two multiply-adds instead of four.
template <int NC, int NSEG /* bm * bs, or 0: any */, bool PAIR = false, bool REALW = false>
__global__ void __launch_bounds__(BLK)
k_grid_bricks(const BrickTask* __restrict__ tasks, int ntasks, const BrickRef* __restrict__ btab,
              const BrickEntry* __restrict__ entries /* REALW: {cell, re} pairs */, const uint32_t* __restrict__ round_rows,
              const float2* __restrict__ Xp /* packed rows: [t][NC] */,
              float2* __restrict__ Y, float2 alpha, const uint32_t* __restrict__ bits,
              int n0, int nm, int bm_log2, int bs_log2, int nbx, int nbm, int st_log2 /* log2(cells per segment) */, int zw /* words per entry of the support bitmaps */) {
    extern __shared__ float2 acc_all[];                  // per wave: [cells][NC]
    constexpr int TPR = 64 / NC;                         // entries per round (one wave instruction of the accumulation)
    constexpr int RS = 64 / TPR;                         // rounds per super-trip
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int task = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (task >= ntasks) return;                          // (no workgroup barrier below: waves are independent)
    const BrickTask tk = tasks[task];
    const int nb = tk.nb_flags & 0xffff;
    const bool shared = (tk.nb_flags >> 16) & 1;
    const int32_t nent = tk.hi - tk.lo, nround = nent / TPR;
    // a segment = 2^st_log2 consecutive cells of one grid row (the support table's granularity: 16, 8 or 4 kx points);
    // segment index inside a brick: s = xs + XS*(im + BM*is) with XS = 16 >> st_log2 segments per brick row
    const int xs_log2 = 4 - st_log2, seg_log2 = xs_log2 + bm_log2 + bs_log2;
    const int BM = 1 << bm_log2, nseg = NSEG ? (PAIR ? 2 * NSEG : NSEG) : 1 << seg_log2, ncell = 16 << (bm_log2 + bs_log2);
    float2* __restrict__ acc = acc_all + (size_t)wv * ncell * NC;
    const int coil = lane % NC, tsub = lane / NC, xround = (lane / NC) % RS;
    const rsrc_t r_en = REALW ? make_rsrc(reinterpret_cast<const float2*>(entries) + tk.lo) : make_rsrc(entries + tk.lo);
    const rsrc_t r_rr = make_rsrc(round_rows + tk.lo / TPR), r_x = make_rsrc(Xp);

    struct Set { u3 en; uint32_t rid; float2 x; };
    Set s0, s1, s2, s3;
    // In-order return: a wait for one load is a wait for every older one, so what is needed soonest is requested first --
    // samples five super-trips ahead, entries three, panel rows two.
    auto request_samples = [&](Set& s, int st) {
        const int32_t q = st * RS + xround;
        s.rid = (uint32_t)buf_ld_i32(r_rr, q < nround ? (unsigned)q * 4u : IG_OOB);
    };
    auto request_entries = [&](Set& s, int st) {         // (past the end of the task: nothing is fetched)
        const int32_t idx = st * 64 + lane;
        if (REALW) {
            const float2 t = buf_ld<false>(r_en, idx < nent ? (unsigned)idx * 8u : IG_OOB, 0);
            s.en.x = __float_as_uint(t.x); s.en.y = __float_as_uint(t.y); s.en.z = 0u;
        } else {
            s.en = buf_ld_u3(r_en, idx < nent ? (unsigned)idx * 12u : IG_OOB);
        }
    };
    auto request_rows = [&](Set& s, int st) {            // panel rows of super-trip st (its samples have arrived)
        const int32_t q = st * RS + xround;
        s.x = buf_ld<false>(r_x, q < nround ? (s.rid * (unsigned)NC + (unsigned)coil) * 8u : IG_OOB, 0);
    };
    request_samples(s0, 0);
    request_samples(s1, 1);
    // the run's bricks, one per lane
    const float2 my_ref = buf_ld<false>(make_rsrc(btab + tk.bt), lane < nb ? (unsigned)lane * 8u : IG_OOB, 0);
    request_entries(s0, 0);
    request_samples(s2, 2);
    request_samples(s3, 3);
    request_entries(s1, 1);
    request_entries(s2, 2);
    request_rows(s0, 0);
    request_samples(s0, 4);
    request_rows(s1, 1);
    int my_end = 0x7fffffff, my_pt = 0, my_bx = 0, my_m0 = 0, my_s0 = 0;
    if (lane < nb) {
        const int brick = (int)__float_as_uint(my_ref.x);
        if (!shared) my_end = (int)__float_as_uint(my_ref.y) - tk.lo;
        my_bx = brick % nbx;
        my_m0 = ((brick / nbx) % nbm) << bm_log2;
        my_s0 = (brick / (nbx * nbm)) << bs_log2;
        my_pt = my_bx * 16 + n0 * (my_m0 + nm * my_s0);
    }
    // ... and which of their segments are flagged: 64 (brick, segment) pairs per pass, eight passes in flight
    uint32_t my_mask = 0xffffffffu;
    if (bits) {
        const int nt = n0 >> st_log2;                    // support entries per grid row
        const rsrc_t r_bits = make_rsrc(bits);
        uint32_t w[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int pair = p * 64 + lane, j = pair >> seg_log2, seg = pair & (nseg - 1);
            const int jbx = __shfl(my_bx, j & 63), jm0 = __shfl(my_m0, j & 63), js0 = __shfl(my_s0, j & 63);
            const int xs = seg & ((1 << xs_log2) - 1), im = (seg >> xs_log2) & (BM - 1), is = seg >> (xs_log2 + bm_log2);
            const int km = jm0 + im, ks = js0 + is;
            const int kmq = km / zw, kmr = km - kmq * zw;            // bit km / zw of word km % zw
            w[p] = (uint32_t)buf_ld_i32(r_bits, j < nb ? (unsigned)((ks * nt + (jbx << xs_log2) + xs) * zw + kmr) * 4u : IG_OOB) >> kmq;
        }
        my_mask = 0u;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint64_t bal = __ballot(w[p] & 1u);
            const int first = (p * 64) >> seg_log2, per = 64 >> seg_log2;                  // bricks of this pass
            if (lane >= first && lane < first + per)
                my_mask = (uint32_t)(bal >> ((lane - first) << seg_log2)) & (nseg == 32 ? 0xffffffffu : (1u << nseg) - 1u);
        }
    }
    for (int e = lane; e < ncell * NC; e += 64) acc[e] = make_float2(0.f, 0.f);

    int cur = 0;
    int32_t cur_end = __builtin_amdgcn_readlane(my_end, 0);
    // store the image of brick `cur` (flagged segments: 16 cells x NC coils = NC*128 bytes each) and clear it
    auto flush_segment = [&](int pt, uint32_t mask, int seg) {
        // PAIR: `seg` numbers PAIRS of 4-cell segments = the 8-cell pieces of the 8-point table's geometry; lane half h keeps its
        // values iff segment 2 seg + h is flagged
        const uint32_t flags = PAIR ? (mask >> (2 * seg)) & 3u : (mask >> seg) & 1u;
        if (!flags) return;
        const bool mine = PAIR ? ((flags >> (lane >> 5)) & 1u) != 0 : true;
        const int f_log2 = PAIR ? 3 : st_log2, fx_log2 = 4 - f_log2;
        const int xs = seg & ((1 << fx_log2) - 1), im = (seg >> fx_log2) & (BM - 1), is = seg >> (fx_log2 + bm_log2);
        const int cell0 = (xs << f_log2) + 16 * (im + BM * is);           // first cell of the segment in the image
        float2* src = acc + (size_t)cell0 * NC;
        float2* dst = Y + ((int64_t)pt + (xs << f_log2) + (int64_t)n0 * (im + (int64_t)nm * is)) * NC;
        const int nel = NC << f_log2;                                     // float2 values of the segment (128 for 16 cells x 8 coils)
        for (int e = lane; e < nel; e += 64) {
            const float2 o = cmul(alpha, src[e]);
            // Stores and atomics as asm statements: the compiler's wait-count bookkeeping does not see them.  With ordinary
            // stores in this (inner) loop it drains every outstanding load before each super-trip (s_waitcnt vmcnt(0):
            // it cannot bound the number of stores between a load and its use), which undoes the prefetching; not
            // counting them only makes its counted waits for loads somewhat earlier than necessary.  An 8-byte store
            // reads its data registers at issue (no write-after-read hazard), and nothing here reads Y back.
            if (mine) {
                if (shared) {
                    asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %2, off offset:4"
                                 :: "v"(dst + e), "v"(o.x), "v"(o.y) : "memory");
                } else {
                    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst + e), "v"(o) : "memory");
                }
            }
            src[e] = make_float2(0.f, 0.f);
        }
    };
    auto flush = [&]() {
        const int pt = __builtin_amdgcn_readlane(my_pt, cur);
        const uint32_t mask = (uint32_t)__builtin_amdgcn_readlane((int)my_mask, cur);
        if (NSEG) {
#pragma unroll
            for (int seg = 0; seg < (NSEG ? NSEG : 1); ++seg) flush_segment(pt, mask, seg);
        } else {
            for (int seg = 0; seg < nseg; ++seg) flush_segment(pt, mask, seg);
        }
        ++cur;
        cur_end = __builtin_amdgcn_readlane(my_end, cur & 63);
    };
    // super-trip st: request panel rows two, entries three and samples five super-trips ahead, then accumulate the set `s`
    auto super_trip = [&](const Set& s, Set& next, Set& rows_ahead, Set& entries_ahead, int st) {
        const uint32_t e_cell = s.en.x, e_re = s.en.y, e_im = s.en.z;
        const float x_re = s.x.x, x_im = s.x.y;
        request_rows(rows_ahead, st + 2);
        request_entries(entries_ahead, st + 3);
        request_samples(next, st + 5);                                    // (the samples it held have served their purpose)
        for (int r = 0; r < RS; ++r) {
            const int32_t start = st * 64 + r * TPR;                      // wave-uniform; tasks begin and end on multiples of TPR
            if (start >= nent) break;
            if (start >= cur_end) flush();                                // (every brick of the table holds at least one round)
            const int from_e = (r * TPR + tsub) * 4, from_x = (r * NC + coil) * 4;
            const uint32_t cell = (uint32_t)__builtin_amdgcn_ds_bpermute(from_e, (int)e_cell);
            const float vr = __int_as_float(__builtin_amdgcn_ds_bpermute(from_e, (int)e_re));
            const float vi = REALW ? 0.f : __int_as_float(__builtin_amdgcn_ds_bpermute(from_e, (int)e_im));
            const float xr = __int_as_float(__builtin_amdgcn_ds_bpermute(from_x, __float_as_int(x_re)));
            const float xi = __int_as_float(__builtin_amdgcn_ds_bpermute(from_x, __float_as_int(x_im)));
            if (cell != 0xffffffffu) {
                float2* a = acc + (int)cell * NC + coil;
                float2 v = *a;                                            // plain read-add-write: the entries of a round belong to
                if (REALW) {                                              // one sample (distinct cells), the image is this wave's,
                    v.x = fmaf(vr, xr, v.x);                              // and a wave's LDS operations execute in order.  conj(v) * x
                    v.y = fmaf(vr, xi, v.y);
                } else {
                    v.x += fmaf(vr, xr, vi * xi);
                    v.y += fmaf(vr, xi, -vi * xr);
                }
                *a = v;
            }
        }
    };
    const int nst = (nent + 63) / 64;
    for (int st = 0; st < nst; st += 4) {
        super_trip(s0, s1, s2, s3, st);                   // (super-trips past the end request and accumulate nothing: no early
        super_trip(s1, s2, s3, s0, st + 1);               // exits, which would give the loop header a predecessor the
        super_trip(s2, s3, s0, s1, st + 2);               // compiler's wait-count bookkeeping knows nothing about)
        super_trip(s3, s0, s1, s2, st + 3);
    }
    flush();
}

// ---- the brick scatter for panels of 1, 2 or 4 columns: SLOTS instead of rounds ------------------------------------------
// With NC columns a round holds 64 / NC entries of ONE sample, and a sample leaves ~6 entries in a 16 x 2 x 2 brick: at 4 columns
// the padded format is 2.3x the nonzeros, at 2 and 1 it would be 5x and 11x -- which is why the ranks of a coil-sharded run
// that hold one or two coils used to gather over the transposed matrix instead (0.77 - 0.83 ms of their 2.2 - 3.1 ms).
// Here a lane is an ENTRY and loops over the NC columns itself.  What makes a wave instruction race-free is the order of
// the entries: the host sorts every brick's entries by (occurrence of their cell, cell) -- the k-th nonzero that falls on a
// cell goes into group k -- so a group never holds a cell twice; a SLOT is a group (or a piece of at most 64 entries of one).
// No padding at all: 16-byte entries {cell, re, im, sample} and one int32 offset per slot.  The lanes of a slot are often few
// (a brick a single trajectory crosses has many cells hit twice or thrice: 5 slots of ~18 entries), but a slot is ~30
// instructions and there are few of them.  Entries are requested two slots ahead, the panel rows of their samples one.
struct SlotEntry { uint32_t cell; float re, im; uint32_t row; };

// REALW: 12-byte entries {cell, re, sample} of a matrix whose weights are all real.
template <int NC, bool REALW = false>
__global__ void __launch_bounds__(BLK)
k_grid_slots(const BrickTask* __restrict__ tasks, int ntasks, const BrickRef* __restrict__ btab,
             const int32_t* __restrict__ slot_ptr, const SlotEntry* __restrict__ entries,
             const float2* __restrict__ Xp /* packed rows: [t][NC] */, float2* __restrict__ Y, float2 alpha,
             const uint32_t* __restrict__ bits, int n0, int nm, int bm_log2, int bs_log2, int nbx, int nbm, int st_log2, int zw /* words per entry of the support bitmaps */) {
    extern __shared__ float2 acc_all[];                  // per wave: [cells][NC]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int task = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (task >= ntasks) return;
    const BrickTask tk = tasks[task];
    const int nb = tk.nb_flags & 0xffff;
    const bool shared = (tk.nb_flags >> 16) & 1;
    const int32_t nslot = tk.hi - tk.lo;
    const int xs_log2 = 4 - st_log2, seg_log2 = xs_log2 + bm_log2 + bs_log2;
    const int BM = 1 << bm_log2, nseg = 1 << seg_log2, ncell = 16 << (bm_log2 + bs_log2);
    float2* __restrict__ acc = acc_all + (size_t)wv * ncell * NC;
    // the run's bricks, one per lane: where each ends (in slots of this task), its first grid point, its flagged segments
    const float2 my_ref = buf_ld<false>(make_rsrc(btab + tk.bt), lane < nb ? (unsigned)lane * 8u : IG_OOB, 0);
    int my_end = 0x7fffffff, my_pt = 0, my_bx = 0, my_m0 = 0, my_s0 = 0;
    if (lane < nb) {
        const int brick = (int)__float_as_uint(my_ref.x);
        if (!shared) my_end = (int)__float_as_uint(my_ref.y) - tk.lo;
        my_bx = brick % nbx;
        my_m0 = ((brick / nbx) % nbm) << bm_log2;
        my_s0 = (brick / (nbx * nbm)) << bs_log2;
        my_pt = my_bx * 16 + n0 * (my_m0 + nm * my_s0);
    }
    uint32_t my_mask = 0xffffffffu;
    if (bits) {
        const int nt = n0 >> st_log2;
        const rsrc_t r_bits = make_rsrc(bits);
        uint32_t w[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int pair = p * 64 + lane, j = pair >> seg_log2, seg = pair & (nseg - 1);
            const int jbx = __shfl(my_bx, j & 63), jm0 = __shfl(my_m0, j & 63), js0 = __shfl(my_s0, j & 63);
            const int xs = seg & ((1 << xs_log2) - 1), im = (seg >> xs_log2) & (BM - 1), is = seg >> (xs_log2 + bm_log2);
            const int km = jm0 + im, ks = js0 + is;
            const int kmq = km / zw, kmr = km - kmq * zw;            // bit km / zw of word km % zw
            w[p] = (uint32_t)buf_ld_i32(r_bits, j < nb ? (unsigned)((ks * nt + (jbx << xs_log2) + xs) * zw + kmr) * 4u : IG_OOB) >> kmq;
        }
        my_mask = 0u;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint64_t bal = __ballot(w[p] & 1u);
            const int first = (p * 64) >> seg_log2, per = 64 >> seg_log2;
            if (lane >= first && lane < first + per)
                my_mask = (uint32_t)(bal >> ((lane - first) << seg_log2)) & (nseg == 32 ? 0xffffffffu : (1u << nseg) - 1u);
        }
    }
    for (int e = lane; e < ncell * NC; e += 64) acc[e] = make_float2(0.f, 0.f);

    int cur = 0;
    int32_t cur_end = __builtin_amdgcn_readlane(my_end, 0);
    auto flush = [&]() {
        // The lanes that read a cell below are not the lanes that accumulated into it.  A wave's LDS operations execute in
        // program order, so the hardware needs nothing here; the fence pair pins that order for the COMPILER too (it may not
        // move the reads of the image above the read-add-writes of the slot before, whatever it learns about the pointers).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int pt = __builtin_amdgcn_readlane(my_pt, cur);
        const uint32_t mask = (uint32_t)__builtin_amdgcn_readlane((int)my_mask, cur);
        for (int seg = 0; seg < nseg; ++seg) {
            if (!((mask >> seg) & 1u)) continue;
            const int xs = seg & ((1 << xs_log2) - 1), im = (seg >> xs_log2) & (BM - 1), is = seg >> (xs_log2 + bm_log2);
            const int cell0 = (xs << st_log2) + 16 * (im + BM * is);
            float2* src = acc + (size_t)cell0 * NC;
            float2* dst = Y + ((int64_t)pt + (xs << st_log2) + (int64_t)n0 * (im + (int64_t)nm * is)) * NC;
            const int nel = NC << st_log2;
            for (int e = lane; e < nel; e += 64) {
                const float2 o = cmul(alpha, src[e]);
                if (shared) {
                    asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %2, off offset:4"
                                 :: "v"(dst + e), "v"(o.x), "v"(o.y) : "memory");
                } else {
                    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst + e), "v"(o) : "memory");
                }
                src[e] = make_float2(0.f, 0.f);
            }
        }
        ++cur;
        cur_end = __builtin_amdgcn_readlane(my_end, cur & 63);
    };

    // slot offsets: lane l holds slot_ptr[tk.lo + base + l] for a window of 64 slots (63 usable: a slot needs its end too)
    const rsrc_t r_sp = make_rsrc(slot_ptr + tk.lo), r_en = make_rsrc(entries), r_x = make_rsrc(Xp);
    struct XV { float2 v[NC]; };
    auto load_entry = [&](int32_t off, int32_t n) -> SlotEntry {
        SlotEntry e;
        if (REALW) {
            const u3 q = buf_ld_u3(r_en, lane < n ? (unsigned)(off + lane) * 12u : IG_OOB);
            e.cell = lane < n ? q.x : 0xffffffffu; e.re = __uint_as_float(q.y); e.im = 0.f; e.row = q.z;
        } else {
            const float4 q = buf_ld_f4(r_en, lane < n ? (unsigned)(off + lane) * 16u : IG_OOB);
            e.cell = lane < n ? __float_as_uint(q.x) : 0xffffffffu; e.re = q.y; e.im = q.z; e.row = __float_as_uint(q.w);
        }
        return e;
    };
    auto load_x = [&](const SlotEntry& e) -> XV {
        XV x;
        const unsigned o = e.cell != 0xffffffffu ? e.row * (unsigned)(NC * 8) : IG_OOB;
        if (NC == 1) x.v[0] = buf_ld<false>(r_x, o, 0);
        else {
#pragma unroll
            for (int c = 0; c < NC; c += 2) {
                const float4 q = buf_ld_f4(r_x, o == IG_OOB ? IG_OOB : o + (unsigned)c * 8u);
                x.v[c] = make_float2(q.x, q.y); x.v[c + 1] = make_float2(q.z, q.w);
            }
        }
        return x;
    };
    for (int32_t base = 0; base < nslot; base += 63) {
        const int32_t left = nslot - base, cnt = left < 63 ? left : 63;           // slots of this window
        const int32_t sp = buf_ld_i32(r_sp, lane <= cnt ? (unsigned)(base + lane) * 4u : IG_OOB);
        auto slot_off = [&](int i) { return __builtin_amdgcn_readlane(sp, i); };
        SlotEntry e0 = load_entry(slot_off(0), slot_off(1) - slot_off(0));
        SlotEntry e1 = cnt > 1 ? load_entry(slot_off(1), slot_off(2) - slot_off(1)) : SlotEntry{0xffffffffu, 0.f, 0.f, 0u};
        XV x0 = load_x(e0);
#pragma unroll 1
        for (int i = 0; i < cnt; ++i) {
            SlotEntry e2{0xffffffffu, 0.f, 0.f, 0u};
            if (i + 2 < cnt) e2 = load_entry(slot_off(i + 2), slot_off(i + 3) - slot_off(i + 2));
            const XV x1 = load_x(e1);                                            // (an empty slot: every lane out of range)
            if (base + i >= cur_end) flush();                                   // this slot belongs to the next brick
            if (e0.cell != 0xffffffffu) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float2* a = acc + (int)e0.cell * NC + c;
                    float2 t = *a;                                               // the cells of a slot are distinct
                    if (REALW) {
                        t.x = fmaf(e0.re, x0.v[c].x, t.x);
                        t.y = fmaf(e0.re, x0.v[c].y, t.y);
                    } else {
                        t.x += fmaf(e0.re, x0.v[c].x, e0.im * x0.v[c].y);        // conj(v) * x
                        t.y += fmaf(e0.re, x0.v[c].y, -e0.im * x0.v[c].x);
                    }
                    *a = t;
                }
            }
            e0 = e1; e1 = e2; x0 = x1;
        }
    }
    flush();
}

// ---- the same scatter for a 64-column panel at the reference boundary (column-major Y) -----------------------------
// Y(K x 64, column-major) = alpha * A^H X for any CSR whose K is a multiple of 16: bricks of 16 consecutive rows of Y,
// 12-byte entries {row of Y inside the brick, re, im} in brick order plus the row of X of every entry (no padding: a round
// is ONE entry, the 64 lanes are the 64 columns).  Entries and rows arrive through the scalar cache (wave-uniform
// addresses: s_load; rows three groups of four ahead, payloads one), the packed X row of an entry is one 512-byte wave
// load at a scalar offset (two groups ahead), the brick image [16][64 (+1)] sits in LDS, and a finished brick goes out as
// full 128-byte lines (lanes = 16 rows x 4 columns).  Y is zeroed first (beta == 0: rows no nonzero touches are zero);
// pieces of heavy bricks add with float atomics.
constexpr int WIDE_LD = 65;                 // image row stride in float2: conflict-free both as [cell][lane] and transposed

__global__ void __launch_bounds__(BLK)
k_bricks_wide64(const BrickTask* __restrict__ tasks, int ntasks, const BrickRef* __restrict__ btab,
                const BrickEntry* __restrict__ entries, const uint32_t* __restrict__ entry_rows,
                const float2* __restrict__ Xp /* [row][64] */,
                float2* __restrict__ Y, int64_t ldy, float2 alpha) {
    __shared__ float2 acc_all[WAVES_PER_BLOCK * 16 * WIDE_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = (int)blockIdx.x;
    const int task = blk * WAVES_PER_BLOCK + wv;
    if (blk >= (int)gridDim.x || task >= ntasks) return;
    const BrickTask tk = tasks[task];
    const int nb = tk.nb_flags & 0xffff;
    const bool shared = (tk.nb_flags >> 16) & 1;
    const int32_t nent = tk.hi - tk.lo;
    float2* __restrict__ acc = acc_all + wv * 16 * WIDE_LD;
    const BrickEntry* __restrict__ en = entries + tk.lo;
    const uint32_t* __restrict__ er = entry_rows + tk.lo;
    const rsrc_t r_x = make_rsrc(Xp);

    struct Pay { BrickEntry e[4]; };
    struct Rows { uint32_t r[4]; };
    Pay pa, pb, pc;
    Rows ra, rb, rc;
    float2 xa[4], xb[4], xc[4];
    auto fetch_rows = [&](Rows& g, int gi) {             // wave-uniform addresses: scalar loads
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int32_t idx = gi * 4 + i; g.r[i] = er[idx < nent ? idx : nent - 1]; }
    };
    auto fetch_pay = [&](Pay& g, int gi) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int32_t idx = gi * 4 + i; g.e[i] = en[idx < nent ? idx : nent - 1]; }
    };
    auto request = [&](float2* x, const Rows& g, int gi) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            x[i] = buf_ld<false>(r_x, gi * 4 + i < nent ? (unsigned)lane * 8u : IG_OOB, g.r[i] * 512u);
    };
    fetch_rows(ra, 0);
    fetch_rows(rb, 1);
    fetch_rows(rc, 2);
    fetch_pay(pa, 0);
    const float2 my_ref = buf_ld<false>(make_rsrc(btab + tk.bt), lane < nb ? (unsigned)lane * 8u : IG_OOB, 0);
    for (int e = lane; e < 16 * WIDE_LD; e += 64) acc[e] = make_float2(0.f, 0.f);
    request(xa, ra, 0);
    request(xb, rb, 1);
    int my_end = 0x7fffffff;
    if (lane < nb && !shared) my_end = (int)__float_as_uint(my_ref.y) - tk.lo;
    const int my_row0 = (int)__float_as_uint(my_ref.x) * 16;

    int cur = 0;
    int32_t cur_end = __builtin_amdgcn_readlane(my_end, 0);
    auto flush = [&]() {
        const int row0 = __builtin_amdgcn_readlane(my_row0, cur);
        const int cell = lane & 15, cg = lane >> 4;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int col = cg + 4 * i;
            float2* a = acc + cell * WIDE_LD + col;
            const float2 o = cmul(alpha, *a);
            float2* dst = Y + (int64_t)col * ldy + row0 + cell;           // 16 lanes = one 128-byte line of column `col`
            if (shared) {
                asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %2, off offset:4"
                             :: "v"(dst), "v"(o.x), "v"(o.y) : "memory");
            } else {
                asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(o) : "memory");
            }
            *a = make_float2(0.f, 0.f);
        }
        ++cur;
        cur_end = __builtin_amdgcn_readlane(my_end, cur & 63);
    };
    // group gi: request the X rows of group gi + 2, fetch the rows of gi + 3 and the payloads of gi + 1, accumulate gi
    auto body = [&](const Pay& p, Pay& pnext, const float2* x, float2* x2, const Rows& r2, Rows& r3, int gi) {
        request(x2, r2, gi + 2);
        fetch_rows(r3, gi + 3);
        fetch_pay(pnext, gi + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int32_t idx = gi * 4 + i;
            if (idx < nent) {
                if (idx >= cur_end) flush();
                const float vr = p.e[i].re, vi = p.e[i].im;
                float2* a = acc + (int)p.e[i].cell * WIDE_LD + lane;
                float2 t = *a;
                t.x += fmaf(vr, x[i].x, vi * x[i].y);                     // conj(v) * x
                t.y += fmaf(vr, x[i].y, -vi * x[i].x);
                *a = t;
            }
        }
    };
    const int ngroup = (nent + 3) / 4;
    for (int gi = 0; gi < ngroup; gi += 3) {             // (groups past the end do nothing: no early exit, see k_grid_bricks)
        body(pa, pb, xa, xc, rc, ra, gi);
        body(pb, pc, xb, xa, ra, rb, gi + 1);
        body(pc, pa, xc, xb, rb, rc, gi + 2);
    }
    flush();
}

// The same scatter with the brick image in REGISTERS (round 3).  The 16-row bricks above fetch a row of X once per ENTRY
// (the L1 serves the repeats: 16 tag look-ups per 512-byte wave load, 1.0 G of them per launch) and once per (sample, brick)
// group from HBM -- ten groups per sample for a 27-tap gridding kernel, 9.7 GB of the launch's 15.6.  Bigger bricks need fewer
// groups (16 x 2 x 2 grid points: 4.5 per sample) but 33 KB of LDS per wave.  Here
//   * lane c keeps column c of the whole brick in 128 VGPRs (v[128 + 2 cell] = re, v[129 + 2 cell] = im), and because the cell
//     of an entry is wave-uniform the accumulation addresses them through the VGPR index mode: M0 = {third source and
//     destination relative, index 2 cell}, four multiply-adds per entry, no LDS traffic, no read-after-write through memory;
//   * entries come in QUADS: a sample's share of a brick is padded to a multiple of four (ig_grid_bricks_fill, unit 4; the
//     padding repeats a real cell with weight zero) and ONE panel row is loaded per quad -- 13 M wave loads instead of 50 M;
//     brick boundaries fall between quads.
// How it is written:
//   * The compiler never sees the image or the panel rows: the kernel caps its own allocation at v0..v71 (amdgpu_num_vgpr),
//     v72..v79 hold two trips of entries in flight, v80..v103 twelve panel rows, and every access to v72..v255 is an
//     assembly block with literal register numbers.  (Handing the image to the compiler as four 32-register operands made it
//     copy whole tuples at every join of the control flow; leaving the loads to it kept one group in flight.)
//   * Loads return in order, so "at most 11 younger loads outstanding" means the row about to be used has arrived, whatever
//     else -- the trips' loads, the stores of a flush -- is in flight.
//   * The index mode is switched on ONCE, with no operand relative (M0[15:12] = 0: the compiler's code runs unchanged; it never
//     writes M0 in this kernel); a block that addresses the image writes M0 itself and clears it again.
//   * Entries do not come through the scalar cache (at the eight waves per CU that 256 registers allow, every miss stalled a
//     wave for a trip to HBM): lane e of the wave loads entry e of a TRIP of 48 (12 quads), two trips ahead; wave-uniform
//     values are read out with v_readlane (constant lane numbers) when their turn comes.  The rows sit in a second stream
//     shifted by the look-ahead of 11 quads, so quad k of a trip requests with lane k.
//   * A wave issues one instruction per four clocks whatever its kind, and two waves per SIMD hide little: the loop is
//     written for instruction count.
// The flush goes tile by tile (16 rows of Y) through the [16][65] LDS image of the kernel above, full lines out.
template <int K, int END>
__device__ __forceinline__ void wide_img_zero() {
    if constexpr (K < END) {
        asm volatile("v_mov_b64 v[%0:%1], 0" :: "i"(128 + 2 * K), "i"(129 + 2 * K));
        wide_img_zero<K + 1, END>();
    }
}

// REALW: 8-byte entries {cell, re} of a matrix whose weights are all real (a plain gridding matrix): a third less of the format to read,
// two v_readlane and two multiply-adds less per entry.
template <int NT /* 16-row tiles per brick: bm * bs = 2, 4 */, bool REALW = false>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_num_vgpr(72)))
k_bricks_wide64r(const BrickTask* __restrict__ tasks, int ntasks, const BrickRef* __restrict__ btab,
                 const BrickEntry* __restrict__ entries, const uint32_t* __restrict__ quad_rows,
                 const float2* __restrict__ Xp /* [row][64] */, float2* __restrict__ Y, int64_t ldy, float2 alpha,
                 int n0, int nm, int bm_log2, int bs_log2, int nbx, int nbm) {
    __shared__ float2 acc_all[WAVES_PER_BLOCK * 16 * WIDE_LD];
    asm volatile("" ::: "v255");                          // the kernel owns 256 VGPRs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = (int)blockIdx.x;
    const int task = blk * WAVES_PER_BLOCK + wv;
    if (blk >= (int)gridDim.x || task >= ntasks) return;
    const BrickTask tk = tasks[task];
    const int nb = tk.nb_flags & 0xffff;
    const bool shared = (tk.nb_flags >> 16) & 1;
    const int32_t nent = tk.hi - tk.lo;                   // a multiple of 4
    const int32_t nquad = nent >> 2;
    float2* __restrict__ acc = acc_all + wv * 16 * WIDE_LD;
    asm volatile("s_set_gpr_idx_on %0, 0x0" :: "s"(0));
    rsrc_t r_en = REALW ? make_rsrc(reinterpret_cast<const float2*>(entries) + tk.lo) : make_rsrc(entries + tk.lo);   // (not const: operands of assembly blocks inside generic lambdas)
    rsrc_t r_qr = make_rsrc(quad_rows + (tk.lo >> 2));
    rsrc_t r_x = make_rsrc(Xp);

    constexpr int TRIP = 48, TQ = TRIP / 4;               // entries / quads per trip
    constexpr int AHEAD = TQ - 1;                         // a panel row is requested 11 quads before its entries are used
    struct Trip { uint32_t m0w; float re, im; uint32_t rowoff; };             // lane e: entry e; lane q: the row of quad q + AHEAD
    // raw trips in flight: slot A = v[72:75], slot B = v[76:79]
    auto load_raw = [&](auto slot, int ti) __attribute__((always_inline)) {
        constexpr int R = 72 + 4 * decltype(slot)::value;
        const int32_t e = ti * TRIP + lane, q = ti * TQ + lane + AHEAD;
        const unsigned oe = (lane < TRIP && e < nent) ? (unsigned)e * (REALW ? 8u : 12u) : IG_OOB;
        const unsigned oq = (lane < TQ && q < nquad) ? (unsigned)q * 4u : IG_OOB;           // (past the end: row 0, never used)
        const rsrc_t ren = r_en, rqr = r_qr;             // (copies: a generic lambda does not capture a variable only an asm operand names)
        if (REALW)
            asm volatile("buffer_load_dwordx2 v[%0:%1], %3, %4, 0 offen\n\t"
                         "buffer_load_dword v[%2], %5, %6, 0 offen"
                         :: "i"(R), "i"(R + 1), "i"(R + 3), "v"(oe), "s"(ren), "v"(oq), "s"(rqr) : "memory");
        else
            asm volatile("buffer_load_dwordx3 v[%0:%1], %3, %4, 0 offen\n\t"
                         "buffer_load_dword v[%2], %5, %6, 0 offen"
                         :: "i"(R), "i"(R + 2), "i"(R + 3), "v"(oe), "s"(ren), "v"(oq), "s"(rqr) : "memory");
    };
    auto cook_a = [&]() __attribute__((always_inline)) {                        // slot A -> the words the loop reads with v_readlane
        uint32_t c, re, im, row;
        asm volatile("v_mov_b32 %0, v72\n\tv_mov_b32 %1, v73\n\tv_mov_b32 %2, v74\n\tv_mov_b32 %3, v75" : "=v"(c), "=v"(re), "=v"(im), "=v"(row));
        Trip t;
        t.m0w = ((c & (16u * NT - 1u)) * 2u) | 0xC000u;                       // M0: third source and destination relative, index 2 cell
        t.re = __uint_as_float(re); t.im = __uint_as_float(im);
        t.rowoff = row * 512u;
        return t;
    };
    load_raw(std::integral_constant<int, 0>{}, 0);
    load_raw(std::integral_constant<int, 1>{}, 1);
    const float2 my_ref = buf_ld<false>(make_rsrc(btab + tk.bt), lane < nb ? (unsigned)lane * 8u : IG_OOB, 0);
    wide_img_zero<0, 16 * NT>();
    const unsigned x_voff = (unsigned)lane * 8u;
    // the panel row whose byte offset sits in lane `ql` of `src` -> buffer KF (v[80 + 2 KF], v[81 + 2 KF])
    auto request = [&](auto kf, auto ql, uint32_t src) __attribute__((always_inline)) {
        constexpr int KF = decltype(kf)::value;
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)src, decltype(ql)::value);
        const rsrc_t rx = r_x;
        const unsigned vo = x_voff;
        // (s_nop: five wait states between a VALU writing an SGPR and a memory instruction reading it)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx2 v[%0:%1], %2, %3, %4 offen" :: "i"(80 + 2 * KF), "i"(81 + 2 * KF), "v"(vo), "s"(rx), "s"(o) : "memory");
    };
    // quads 0 .. 10: their rows are not in the shifted stream
    {
        const uint32_t first = (uint32_t)buf_ld_i32(r_qr, (lane < AHEAD && lane < nquad) ? (unsigned)lane * 4u : IG_OOB) * 512u;
        auto pro = [&](auto k) __attribute__((always_inline)) { request(k, k, first); };
        pro(std::integral_constant<int, 0>{}); pro(std::integral_constant<int, 1>{}); pro(std::integral_constant<int, 2>{});
        pro(std::integral_constant<int, 3>{}); pro(std::integral_constant<int, 4>{}); pro(std::integral_constant<int, 5>{});
        pro(std::integral_constant<int, 6>{}); pro(std::integral_constant<int, 7>{}); pro(std::integral_constant<int, 8>{});
        pro(std::integral_constant<int, 9>{}); pro(std::integral_constant<int, 10>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Trip tc = cook_a();
    asm volatile("v_mov_b32 v72, v76\n\tv_mov_b32 v73, v77\n\tv_mov_b32 v74, v78\n\tv_mov_b32 v75, v79" ::: "memory");
    Trip tn = cook_a();
    load_raw(std::integral_constant<int, 0>{}, 2);
    load_raw(std::integral_constant<int, 1>{}, 3);
    int my_end = 0x7fffffff;
    if (lane < nb && !shared) my_end = (int)__float_as_uint(my_ref.y) - tk.lo;
    int my_row0;                                          // first row of Y of the brick's tile (0, 0)
    {
        const int b = (int)__float_as_uint(my_ref.x);
        const int bx = b % nbx, bmi = (b / nbx) % nbm, bsi = b / (nbx * nbm);
        my_row0 = bx * 16 + n0 * ((bmi << bm_log2) + nm * (bsi << bs_log2));
    }
    const int tile_cell = lane & 15, tile_cg = lane >> 4;

    int cur = 0;
    int32_t cur_end = __builtin_amdgcn_readlane(my_end, 0);
    auto tile_out = [&](int q, int row00) __attribute__((always_inline)) {       // the LDS image (tile q of the brick) -> Y
        const int row0 = row00 + n0 * ((q & ((1 << bm_log2) - 1)) + nm * (q >> bm_log2));
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int col = tile_cg + 4 * i;
            const float2 o = cmul(alpha, acc[tile_cell * WIDE_LD + col]);
            float2* dst = Y + (int64_t)col * ldy + row0 + tile_cell;        // 16 lanes = one 128-byte line of column `col`
            if (shared) {
                asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %2, off offset:4"
                             :: "v"(dst), "v"(o.x), "v"(o.y) : "memory");
            } else {
                asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(o) : "memory");
            }
        }
    };
    auto flush = [&]() __attribute__((always_inline)) {
        const int row00 = __builtin_amdgcn_readlane(my_row0, cur);
#pragma unroll 1
        for (int q = 0; q < NT; ++q) {                   // a run-time loop: the kernel holds one flush site per quad of a trip
            // tile q of the image (registers v[128 + 32 q ...]) -> LDS, read through the index mode (first source relative)
            uint32_t any = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float t[16];
                asm volatile("s_mov_b32 m0, %16\n\ts_nop 0\n\t"
                             "v_mov_b32 %0, v128\n\tv_mov_b32 %1, v129\n\tv_mov_b32 %2, v130\n\tv_mov_b32 %3, v131\n\t"
                             "v_mov_b32 %4, v132\n\tv_mov_b32 %5, v133\n\tv_mov_b32 %6, v134\n\tv_mov_b32 %7, v135\n\t"
                             "v_mov_b32 %8, v136\n\tv_mov_b32 %9, v137\n\tv_mov_b32 %10, v138\n\tv_mov_b32 %11, v139\n\t"
                             "v_mov_b32 %12, v140\n\tv_mov_b32 %13, v141\n\tv_mov_b32 %14, v142\n\tv_mov_b32 %15, v143\n\t"
                             "s_mov_b32 m0, 0"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
                               "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]), "=&v"(t[15])
                             : "s"((32 * q + 16 * h) | 0x1000));
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[(8 * h + c) * WIDE_LD + lane] = make_float2(t[2 * c], t[2 * c + 1]);
                if (shared) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) any |= __float_as_uint(t[c]) << 1;       // (-0.0f counts as nothing)
                }
            }
            if (shared && !__builtin_amdgcn_ballot_w64(any != 0)) continue;               // a piece that never touched this tile
            tile_out(q, row00);
        }
        wide_img_zero<0, 16 * NT>();
        ++cur;
        cur_end = __builtin_amdgcn_readlane(my_end, cur & 63);
    };
    // quad k of the current trip (entries in lanes 4k .. 4k + 3, panel row in buffer k): request the row of the quad 11 ahead
    // into the buffer the previous quad has just released, flush if a new brick starts here, accumulate.
    auto body = [&](auto k, int32_t base) __attribute__((always_inline)) {
        constexpr int KQ = decltype(k)::value;
        request(std::integral_constant<int, (KQ + TQ - 1) % TQ>{}, k, tc.rowoff);
        if (base >= nent) return;                                            // (trips are whole; the run is not)
        if (base >= cur_end) flush();                                        // bricks are not empty: one flush
        uint32_t mw[4]; float wr[4], wi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mw[i] = (uint32_t)__builtin_amdgcn_readlane((int)tc.m0w, 4 * KQ + i);
            wr[i] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tc.re), 4 * KQ + i));
            if (!REALW) wi[i] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tc.im), 4 * KQ + i));
        }
        if (REALW) {
            // image[cell] += w * x, w real
#define IG_WIDE_MAC_R(M, WR) \
                     "s_mov_b32 m0, " M "\n\ts_nop 0\n\t" \
                     "v_fma_f32 v128, " WR ", v[%8], v128\n\t" \
                     "v_fma_f32 v129, " WR ", v[%9], v129\n\t"
            asm volatile("s_waitcnt vmcnt(11)\n\t"                            // this quad's row has arrived
                         IG_WIDE_MAC_R("%0", "%4") IG_WIDE_MAC_R("%1", "%5") IG_WIDE_MAC_R("%2", "%6") IG_WIDE_MAC_R("%3", "%7")
                         "s_mov_b32 m0, 0"
                         :: "s"(mw[0]), "s"(mw[1]), "s"(mw[2]), "s"(mw[3]), "s"(wr[0]), "s"(wr[1]), "s"(wr[2]), "s"(wr[3]),
                            "i"(80 + 2 * KQ), "i"(81 + 2 * KQ) : "memory");
#undef IG_WIDE_MAC_R
            return;
        }
        // image[cell] += conj(v) * x for the quad's four entries (one panel row)
#define IG_WIDE_MAC(M, WR, WI) \
                     "s_mov_b32 m0, " M "\n\ts_nop 0\n\t" \
                     "v_fma_f32 v128, " WR ", v[%12], v128\n\t" \
                     "v_fma_f32 v129, " WR ", v[%13], v129\n\t" \
                     "v_fma_f32 v128, " WI ", v[%13], v128\n\t" \
                     "v_fma_f32 v129, -" WI ", v[%12], v129\n\t"
        asm volatile("s_waitcnt vmcnt(11)\n\t"                                // this quad's row has arrived
                     IG_WIDE_MAC("%0", "%4", "%5") IG_WIDE_MAC("%1", "%6", "%7") IG_WIDE_MAC("%2", "%8", "%9") IG_WIDE_MAC("%3", "%10", "%11")
                     "s_mov_b32 m0, 0"
                     :: "s"(mw[0]), "s"(mw[1]), "s"(mw[2]), "s"(mw[3]),
                        "s"(wr[0]), "s"(wi[0]), "s"(wr[1]), "s"(wi[1]), "s"(wr[2]), "s"(wi[2]), "s"(wr[3]), "s"(wi[3]),
                        "i"(80 + 2 * KQ), "i"(81 + 2 * KQ) : "memory");
#undef IG_WIDE_MAC
    };
    int32_t base = 0;
    for (int t = 0; t * TRIP < nent; ++t, base += TRIP) {
        body(std::integral_constant<int, 0>{}, base);      body(std::integral_constant<int, 1>{}, base + 4);
        body(std::integral_constant<int, 2>{}, base + 8);  body(std::integral_constant<int, 3>{}, base + 12);
        body(std::integral_constant<int, 4>{}, base + 16); body(std::integral_constant<int, 5>{}, base + 20);
        body(std::integral_constant<int, 6>{}, base + 24); body(std::integral_constant<int, 7>{}, base + 28);
        body(std::integral_constant<int, 8>{}, base + 32); body(std::integral_constant<int, 9>{}, base + 36);
        body(std::integral_constant<int, 10>{}, base + 40); body(std::integral_constant<int, 11>{}, base + 44);
        // (at most 11 loads are outstanding here, all of them panel rows requested after slot B's trip was)
        asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        tc = tn;
        tn = cook_a();
        asm volatile("v_mov_b32 v72, v76\n\tv_mov_b32 v73, v77\n\tv_mov_b32 v74, v78\n\tv_mov_b32 v75, v79" ::: "memory");
        load_raw(std::integral_constant<int, 1>{}, t + 4);
    }
    flush();
    asm volatile("s_set_gpr_idx_off");
}

// Zero the 16-row tiles of a column-major K x 64 result that no single task owns (bit clear in `owned`: tiles no nonzero
// touches, which beta == 0 defines as zero, and the heavy tiles several tasks add into): the owning task of every other tile
// stores all of its 16 x 64 values itself, so nothing is written twice.  A workgroup takes 4096 rows of every column.
__global__ void __launch_bounds__(BLK)
k_wide_zero_unowned(const uint32_t* __restrict__ owned, float2* __restrict__ Y, int64_t ldy, int64_t K) {
    const int64_t r0 = (int64_t)blockIdx.x * 4096;
    const int tid = threadIdx.x;
    uint32_t own[8];          // this thread's eight 16-byte pieces q = tid + 256 u cover rows 2 q, 2 q + 1: tile q / 8
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t tile = (r0 >> 4) + ((tid + 256 * u) >> 3);
        own[u] = (tile << 4) < K ? ((owned[tile >> 5] >> (tile & 31)) & 1u) : 1u;
    }
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    const v4f_t z = {0.f, 0.f, 0.f, 0.f};
    for (int col = 0; col < 64; ++col) {
        v4f_t* __restrict__ dst = reinterpret_cast<v4f_t*>(Y + (int64_t)col * ldy + r0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (!own[u]) __builtin_nontemporal_store(z, dst + tid + 256 * u);
    }
}

// zero the flagged segments of the bricks that several tasks add into
template <int NC>
__global__ void __launch_bounds__(BLK)
k_grid_bricks_zero(const int32_t* __restrict__ shared_bricks, float2* __restrict__ Y, const uint32_t* __restrict__ bits,
                   int n0, int nm, int bm_log2, int bs_log2, int nbx, int nbm, int st_log2, int zw) {
    const int brick = shared_bricks[blockIdx.x];
    const int xs_log2 = 4 - st_log2;
    const int BM = 1 << bm_log2, nseg = 1 << (xs_log2 + bm_log2 + bs_log2);
    const int bx = brick % nbx, bmi = (brick / nbx) % nbm, bsi = brick / (nbx * nbm);
    const int x0 = bx * 16, m0 = bmi << bm_log2, s0 = bsi << bs_log2;
    const int nt = n0 >> st_log2, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int seg = wv; seg < nseg; seg += WAVES_PER_BLOCK) {
        const int xs = seg & ((1 << xs_log2) - 1), im = (seg >> xs_log2) & (BM - 1), is = seg >> (xs_log2 + bm_log2);
        const int km = m0 + im, ks = s0 + is;
        if (bits && !((bits[((size_t)ks * nt + (bx << xs_log2) + xs) * zw + (km % zw)] >> (km / zw)) & 1u)) continue;
        float2* dst = Y + ((int64_t)x0 + (xs << st_log2) + (int64_t)n0 * (km + (int64_t)nm * ks)) * NC;
        for (int e = lane; e < (NC << st_log2); e += 64) dst[e] = make_float2(0.f, 0.f);
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
    // a lane takes four nonzeros per trip, so 8 nonzero-lanes cover a 32-nonzero row in one trip: more lanes per row
    // only mean fewer rows per wave (the kernel is latency bound)
    constexpr int nl_cap = 8;
    int cap = 64 / s.CL;
    if (nl_cap > 0 && mean <= 4 * nl_cap && cap > nl_cap) cap = nl_cap;
    s.NL = pow2_ceil(mean > 0 ? mean : 1, cap);
    return s;
}

constexpr uint32_t WL_CAP = 1u << 20;

int ensure_worklists(ig_ctx* ctx) {
    if (ctx->d_worklist) return IG_OK;
    const size_t bytes = sizeof(int32_t) * 2 * WL_CAP + 2 * WL_SUB * sizeof(uint32_t);
    IG_HIP(ctx, hipMalloc((void**)&ctx->d_worklist, bytes));
    ctx->worklist_bytes = bytes;
    return IG_OK;
}

template <bool CONJ>
int launch_gather(ig_ctx* ctx, int64_t rows, int64_t xrows, int64_t N, int64_t nnz,
                  const int32_t* rowptr, const int32_t* colind, const float2* vals,
                  const float2* X, int64_t ldx, float2* Y, int64_t ldy, float2 alpha, float2 beta,
                  GridMask mask = GridMask{nullptr, 0, 0}, const int32_t* yperm = nullptr,
                  const int32_t* xperm = nullptr, bool x_il = false, bool y_il = false, const float* rvals = nullptr) {
    // rvals: the real parts of `vals` as floats, for a matrix whose weights are all real; the several-rows-per-wave gather over an
    // interleaved panel of 2, 4 or 8 columns reads them instead (rows it defers to the long-row kernels still take `vals`)
    // x_il / y_il: the panel is stored row-major (its N values of a row contiguous, row stride N) instead of
    // column-major with a leading dimension -- the coil-interleaved grid of the fused transform's layout 2
    const Shape s = pick_shape(rows, N, nnz);
    const int64_t syr = y_il ? N : 1;
    if (y_il) ldy = 1;
    // Panel rows that are gathered many times each (nnz >> xrows) from a small panel: repack the panel
    // once so that one gathered row is one contiguous 16..64-byte access instead of N scattered ones.
    int64_t sxc = ldx, sxr = 1;          // element (k, j) of X lives at X[j*sxc + k*sxr]
    bool packed = false;
    if (x_il) { sxc = 1; sxr = N; packed = (N & (N - 1)) == 0 && N >= 1; }   // already "packed" when N is a power of two
    if (!x_il && N == 1 && !xperm) { sxc = ldx; sxr = 1; packed = true; }      // so is a single column
    // Packing costs one read + one write of the panel (16 B per element) and turns nnz*N scattered 8-byte gathers
    // (each pulling a 32..64-byte sector) into nnz contiguous N*8-byte ones: worth it once every panel row is
    // gathered at least about once (nnz >= xrows); always for small hot panels.
    if (!x_il && ((N >= 2 && N <= 64 && nnz >= xrows) || (xperm && N <= 64))) {
        const int np = s.CL;             // pow2 >= N, <= 64
        const size_t need = (size_t)xrows * np * 8;
        // The repack buffer is a per-context scratch that only grows (reported by ig_mem_info, not by the caller's own
        // accounting): take it only while it is at most a quarter of the free device memory, otherwise -- or if the
        // allocation fails -- run the product unpacked (slower, same result).
        bool can_pack = need > 0 && need <= ((size_t)16 << 30);
        if (can_pack && ctx->xpack_bytes < need) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            if (need > (free_b + ctx->xpack_bytes) / 4) can_pack = false;
            else {
                if (ctx->d_xpack) { IG_HIP(ctx, hipStreamSynchronize(ctx->stream)); IG_HIP(ctx, hipFree(ctx->d_xpack)); ctx->d_xpack = nullptr; ctx->xpack_bytes = 0; }
                if (hipMalloc((void**)&ctx->d_xpack, need) != hipSuccess) { (void)hipGetLastError(); ctx->d_xpack = nullptr; can_pack = false; }
                else ctx->xpack_bytes = need;
            }
        }
        if (can_pack) {
            ig_prof_scope prof(ctx, "pack_panel", (double)xrows * N * 8.0 + (double)need);
            int64_t g = (xrows * np + BLK - 1) / BLK;
            const int64_t cap = (int64_t)ctx->num_cu * 16;
            if (g > cap) g = cap;
            float2* xp = (float2*)ctx->d_xpack;
            if (np == 1)      hipLaunchKernelGGL(k_pack_panel<1>, dim3((unsigned)g), dim3(BLK), 0, ctx->stream, xrows, N, X, ldx, xp, xperm);
            else if (np == 2) hipLaunchKernelGGL(k_pack_panel<2>, dim3((unsigned)g), dim3(BLK), 0, ctx->stream, xrows, N, X, ldx, xp, xperm);
            else if (np == 4) hipLaunchKernelGGL(k_pack_panel<4>, dim3((unsigned)g), dim3(BLK), 0, ctx->stream, xrows, N, X, ldx, xp, xperm);
            else if (np == 8) hipLaunchKernelGGL(k_pack_panel<8>, dim3((unsigned)g), dim3(BLK), 0, ctx->stream, xrows, N, X, ldx, xp, xperm);
            else {
                int64_t gt = (xrows + 63) / 64;
                if (gt > cap) gt = cap;
                if (np == 16)      hipLaunchKernelGGL(k_pack_panel_tiled<16>, dim3((unsigned)gt), dim3(BLK), 0, ctx->stream, xrows, N, X, ldx, xp, xperm);
                else if (np == 32) hipLaunchKernelGGL(k_pack_panel_tiled<32>, dim3((unsigned)gt), dim3(BLK), 0, ctx->stream, xrows, N, X, ldx, xp, xperm);
                else               hipLaunchKernelGGL(k_pack_panel_tiled<64>, dim3((unsigned)gt), dim3(BLK), 0, ctx->stream, xrows, N, X, ldx, xp, xperm);
            }
            IG_LAUNCH_CHECK(ctx, "k_pack_panel");
            X = xp; sxc = 1; sxr = np; packed = true;
        }
    }
    IG_REQUIRE(ctx, !xperm || packed, "csrmm: a panel-row list needs the packed path (2..64 columns and room for the repacked panel: a quarter of the free device memory, at most 16 GiB)");
    const int rpw = 64 / (s.CL * s.NL);
    const int64_t waves = (rows + rpw - 1) / rpw;
    const int64_t blocks = (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    IG_REQUIRE(ctx, blocks <= 0x7fffffffLL, "csrmm: matrix too large for one launch (%lld blocks)", (long long)blocks);
    const int xcd = 1;          // every XCD gets a contiguous range of row blocks (xcd_block)
    const bool b0 = (beta.x == 0.f && beta.y == 0.f);

    // row deferral thresholds (nonzeros): a wave-per-row pass only pays when the slot is narrower than a wave
    const bool defer = true;
    const int nlw = 64 / s.CL;
    const int32_t thr_long = defer ? 256 * nlw : 0x7fffffff;
    const int32_t thr_mid = !defer ? 0x7fffffff : (nlw > s.NL ? 16 * s.NL : thr_long);
    WorkLists wl{};
    wl.yperm = yperm;
    if (defer) {
        if (int rc = ensure_worklists(ctx)) return rc;
        wl.rows[0] = ctx->d_worklist;
        wl.rows[1] = ctx->d_worklist + WL_CAP;
        wl.count = reinterpret_cast<uint32_t*>(ctx->d_worklist + 2 * WL_CAP);
        wl.cap = WL_CAP / WL_SUB;
        IG_HIP(ctx, hipMemsetAsync(wl.count, 0, 2 * WL_SUB * sizeof(uint32_t), ctx->stream));
    }
    // row-per-lane only pays for (mostly) empty / very short rows; NL == 1 alone also happens for wide panels
    // ... and for wide packed panels with short rows, where a row-slot's 8-byte stores would each hit a different
    // line of Y (config 3's transpose, 64 columns, 3 nonzeros/row: 15.7 ms row-per-lane vs 56.6 ms row-slot; its
    // forward, 27 nonzeros/row, is the other way round: 4.8 vs 17.1 ms)
    // wide row-major panels (16, 32, 64 columns) with rows of 8+ nonzeros: the vector row-slot kernel below.  (For the
    // short rows of a transposed gridding matrix it was measured no better than the row-per-lane kernel: 13.0 + 2.1 ms
    // against 14.9 + 0.7 ms at 64 columns -- that product is bound by its 32-byte result stores.)
    constexpr int vw = 4;
    const bool wide_v = vw > 0 && packed && sxc == 1 && N == sxr && (N == 64 || N == 32 || N == 16) && !y_il &&
                        nnz >= 8 * rows && (reinterpret_cast<uintptr_t>(X) & 15u) == 0;
    const bool rowlane = !wide_v && (nnz <= 2 * rows || (packed && N >= 16 && nnz <= 8 * rows));
    if (rowlane) {
        ig_prof_scope prof(ctx, CONJ ? "csrmm_rowlane_conj" : "csrmm_rowlane");
        const int64_t rblocks = (rows + BLK - 1) / BLK;
        // sparse-row matrices defer rows beyond 16 nonzeros; wide-panel use keeps ordinary rows inline
        const int32_t tm_want = nnz <= 2 * rows ? 16 : 512;
        const int32_t tm = defer ? (thr_long < tm_want ? thr_long : tm_want) : 0x7fffffff;
#define IG_ROWLANE_L(NC_, BM_, PK_, U_, BUF_)                                                      \
    hipLaunchKernelGGL((k_csrmm_rowlane<NC_, CONJ, BM_, PK_, U_, BUF_>), dim3((unsigned)rblocks),  \
                       dim3(BLK), 0, ctx->stream, rows, N, rowptr, colind, vals, X, sxc, sxr, Y, ldy, \
                       alpha, beta, xcd, wl, tm, thr_long, mask)
#define IG_ROWLANE(NC_)                                                                            \
    do {                                                                                           \
        if (packed && b0 && NC_ >= 2 && bufok) {                                                   \
            IG_ROWLANE_L(NC_, 0, true, 2, true);          /* two nonzeros per trip */             \
        } else if (packed && b0) IG_ROWLANE_L(NC_, 0, true, 1, false);                             \
        else if (packed) IG_ROWLANE_L(NC_, 1, true, 1, false);                                     \
        else if (b0) IG_ROWLANE_L(NC_, 0, false, 1, false);                                        \
        else IG_ROWLANE_L(NC_, 1, false, 1, false);                                                \
    } while (0)
        // buffer-descriptor loads need every array inside a 2 GB window
        const bool bufok = packed && nnz * 8 < 0x7fffffffLL && xrows * sxr * 8 < 0x7fffffffLL;
        // mostly-empty rows, packed panel of <= 8 columns, beta == 0: the dense-lane kernel
        constexpr int dense_thr = 32;
        // 64-column packed panel, short rows: lanes along the columns, 64 x 64 result tiles through LDS
        if (packed && sxc == 1 && N == 64 && sxr == 64 && !y_il && !mask.bits) {
            const int64_t tblocks = (rows + 63) / 64;
            IG_REQUIRE(ctx, tblocks <= 0x7fffffffLL, "csrmm: matrix too large for one launch");
            const int32_t tt = defer ? (thr_long < 512 ? thr_long : 512) : 0x7fffffff;
            if (b0) hipLaunchKernelGGL((k_csrmm_rowtile64<CONJ, 0>), dim3((unsigned)tblocks), dim3(BLK), 0, ctx->stream,
                                       rows, rowptr, colind, vals, X, Y, ldy, alpha, beta, wl, tt, thr_long);
            else    hipLaunchKernelGGL((k_csrmm_rowtile64<CONJ, 1>), dim3((unsigned)tblocks), dim3(BLK), 0, ctx->stream,
                                       rows, rowptr, colind, vals, X, Y, ldy, alpha, beta, wl, tt, thr_long);
        } else
        if (dense_thr > 0 && packed && b0 && bufok && nnz <= 2 * rows && N == sxr && (sxr == 8 || sxr == 4 || sxr == 2 || sxr == 1)) {
            const int32_t td = defer ? (thr_long < dense_thr ? thr_long : dense_thr) : 0x7fffffff;
            int64_t dblocks = ((rows + 63) / 64 + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
            // One task per wave by default: the hardware dispatcher balances the very uneven tasks better than a
            // static stride does (1.39 ms against 1.91 ms with 8 persistent workgroups per CU on the SENSE matrix).
            IG_REQUIRE(ctx, dblocks <= 0x7fffffffLL, "csrmm: matrix too large for one launch");
#define IG_DENSE(NC_) do { if (y_il) hipLaunchKernelGGL((k_csrmm_dense64<NC_, CONJ, true>), dim3((unsigned)dblocks), dim3(BLK), 0, ctx->stream, \
                                         rows, rowptr, colind, vals, X, Y, ldy, alpha, wl, td, thr_long, mask, dense_co);           \
                           else hipLaunchKernelGGL((k_csrmm_dense64<NC_, CONJ, false>), dim3((unsigned)dblocks), dim3(BLK), 0, ctx->stream, \
                                         rows, rowptr, colind, vals, X, Y, ldy, alpha, wl, td, thr_long, mask, dense_co); } while (0)
            constexpr int dense_co = 1;
            if (sxr == 8) IG_DENSE(8); else if (sxr == 4) IG_DENSE(4); else if (sxr == 2) IG_DENSE(2);
            else hipLaunchKernelGGL((k_csrmm_dense64<1, CONJ, false>), dim3((unsigned)dblocks), dim3(BLK), 0, ctx->stream,
                                    rows, rowptr, colind, vals, X, Y, ldy, alpha, wl, td, thr_long, mask, dense_co);
#undef IG_DENSE
        } else if (y_il) {
            return ig_fail(ctx, IG_ERR_UNSUPPORTED, "csrmm: an interleaved result panel needs the dense-lane kernel (2, 4 or 8 columns, beta = 0)");
        } else
        if (s.CL >= 8) IG_ROWLANE(8);
        else if (s.CL == 4) IG_ROWLANE(4);
        else if (s.CL == 2) IG_ROWLANE(2);
        else IG_ROWLANE(1);
#undef IG_ROWLANE
#undef IG_ROWLANE_L
        IG_LAUNCH_CHECK(ctx, "k_csrmm_rowlane");
    } else {
        IG_REQUIRE(ctx, !y_il, "csrmm: an interleaved result panel is only supported for matrices with mostly empty rows");
        ig_prof_scope prof(ctx, CONJ ? "csrmm_gather_conj" : "csrmm_gather");
        // row-major panel of 2, 4 or 8 columns (packed, or the coil-interleaved grid), rows of 8+ nonzeros on average:
        // several rows per wave with 16-byte panel loads (forward gridding, 8 coils: 0.50 ms against 0.97 ms)
        if (wide_v || (vw > 0 && packed && sxc == 1 && N == sxr && (N == 8 || N == 4 || N == 2) &&
                       nnz >= 8 * rows && (reinterpret_cast<uintptr_t>(X) & 15u) == 0)) {
#define IG_GV(VW_, CLV_, NL_) do {                                                                         \
            const int rpw_v = 64 / (CLV_ * NL_);                                                           \
            const int64_t vblocks = ((rows + rpw_v - 1) / rpw_v + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;  \
            if (b0) hipLaunchKernelGGL((k_csrmm_gather_v<VW_, CLV_, NL_, CONJ, 0>), dim3((unsigned)vblocks), dim3(BLK), 0, ctx->stream, \
                        rows, rowptr, colind, vals, X, sxr, Y, ldy, alpha, beta, xcd, wl, thr_mid, thr_long);   \
            else    hipLaunchKernelGGL((k_csrmm_gather_v<VW_, CLV_, NL_, CONJ, 1>), dim3((unsigned)vblocks), dim3(BLK), 0, ctx->stream, \
                        rows, rowptr, colind, vals, X, sxr, Y, ldy, alpha, beta, xcd, wl, thr_mid, thr_long); } while (0)
#define IG_GVR(VW_, CLV_, NL_) do {                                                                        \
            const int rpw_v = 64 / (CLV_ * NL_);                                                           \
            const int64_t vblocks = ((rows + rpw_v - 1) / rpw_v + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;  \
            if (b0) hipLaunchKernelGGL((k_csrmm_gather_v<VW_, CLV_, NL_, CONJ, 0, true>), dim3((unsigned)vblocks), dim3(BLK), 0, ctx->stream, \
                        rows, rowptr, colind, reinterpret_cast<const float2*>(rvals), X, sxr, Y, ldy, alpha, beta, xcd, wl, thr_mid, thr_long);   \
            else    hipLaunchKernelGGL((k_csrmm_gather_v<VW_, CLV_, NL_, CONJ, 1, true>), dim3((unsigned)vblocks), dim3(BLK), 0, ctx->stream, \
                        rows, rowptr, colind, reinterpret_cast<const float2*>(rvals), X, sxr, Y, ldy, alpha, beta, xcd, wl, thr_mid, thr_long); } while (0)
            if (rvals && N == 8 && vw >= 4 && vw < 8) IG_GVR(4, 2, 8);
            else if (rvals && N == 4) IG_GVR(2, 2, 8);
            else if (rvals && N == 2) IG_GVR(2, 1, 8);
            else
            if (N == 8) { if (vw >= 8) IG_GV(8, 1, 8); else if (vw >= 4) IG_GV(4, 2, 8); else IG_GV(2, 4, 8); }   // 8 / 4 / 2 rows per wave
            else if (N == 4) IG_GV(2, 2, 8);                                                               // 4 rows per wave
            else if (N == 2) IG_GV(2, 1, 8);                                                               // 8 rows per wave
            else if (nnz >= 8 * rows) {                       // long rows: spread a row's nonzeros over nonzero-lanes
                if (N == 16) IG_GV(4, 4, 8);                                                               // 2 rows per wave
                else if (N == 32) IG_GV(4, 8, 8);                                                          // a wave per row, 32 nonzeros per trip
                else if (!wl.yperm && nnz <= 0x7fffffffLL) {
                    // 64 columns: tiles of 64 rows per workgroup, software-pipelined (k_csrmm_gather_tile64)
                    const int64_t tblocks = (rows + 63) / 64;
                    IG_REQUIRE(ctx, tblocks <= 0x7fffffffLL, "csrmm: matrix too large for one launch");
                    // (waves per 64-row tile, measured: 4 -> 1.77-1.85 ms, 8 -> 1.82, 16 -> 1.97: one big workgroup per CU loses the
                    // overlap between workgroups and gains nothing from the smaller L2 footprint)
                    if (b0) hipLaunchKernelGGL((k_csrmm_gather_tile64<CONJ, 0, 4>), dim3((unsigned)tblocks), dim3(256), 0, ctx->stream,
                                rows, nnz, rowptr, colind, vals, X, Y, ldy, alpha, beta, xcd, wl, thr_long);
                    else    hipLaunchKernelGGL((k_csrmm_gather_tile64<CONJ, 1, 4>), dim3((unsigned)tblocks), dim3(256), 0, ctx->stream,
                                rows, nnz, rowptr, colind, vals, X, Y, ldy, alpha, beta, xcd, wl, thr_long);
                }
                else IG_GV(4, 16, 4);                                                                      // 64 columns: 16 nonzeros per trip
            } else {                                          // short rows (a transposed gridding matrix): one trip each
                if (N == 16) IG_GV(4, 4, 1);                                                               // 16 rows per wave
                else if (N == 32) IG_GV(4, 8, 1);                                                          // 8 rows per wave
                else IG_GV(4, 16, 1);                                                                      // 4 rows per wave
            }
#undef IG_GV
#undef IG_GVR
            IG_LAUNCH_CHECK(ctx, "k_csrmm_gather_v");
        } else {
#define IG_GATHER(CL_, NL_)                                                                        \
    do {                                                                                           \
        if (b0) hipLaunchKernelGGL((k_csrmm_gather<CL_, NL_, CONJ, 0>), dim3((unsigned)blocks),    \
                    dim3(BLK), 0, ctx->stream, rows, N, rowptr, colind, vals, X, sxc, sxr, Y, ldy, \
                    alpha, beta, xcd, wl, thr_mid, thr_long);                                      \
        else    hipLaunchKernelGGL((k_csrmm_gather<CL_, NL_, CONJ, 1>), dim3((unsigned)blocks),    \
                    dim3(BLK), 0, ctx->stream, rows, N, rowptr, colind, vals, X, sxc, sxr, Y, ldy, \
                    alpha, beta, xcd, wl, thr_mid, thr_long);                                      \
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
        }
    }
    if (defer) {
        const unsigned gw = ((unsigned)ctx->num_cu * 8 + WL_SUB - 1) / WL_SUB * WL_SUB, gb = ((unsigned)ctx->num_cu * 2 + WL_SUB - 1) / WL_SUB * WL_SUB;
#define IG_ROWS(CL_)                                                                               \
    do {                                                                                           \
        if (thr_mid < thr_long || rowlane) {                                                       \
            ig_prof_scope prof(ctx, "csrmm_rows_wave");                                            \
            if (b0) hipLaunchKernelGGL((k_csrmm_rows_wave<CL_, CONJ, 0>), dim3(gw), dim3(BLK), 0, ctx->stream, \
                        N, rowptr, colind, vals, X, sxc, sxr, Y, ldy, syr, alpha, beta, wl.rows[0], wl.count, wl.cap, wl.yperm);   \
            else    hipLaunchKernelGGL((k_csrmm_rows_wave<CL_, CONJ, 1>), dim3(gw), dim3(BLK), 0, ctx->stream, \
                        N, rowptr, colind, vals, X, sxc, sxr, Y, ldy, syr, alpha, beta, wl.rows[0], wl.count, wl.cap, wl.yperm);   \
        }                                                                                          \
        {                                                                                          \
            ig_prof_scope prof(ctx, "csrmm_rows_block");                                           \
            if (b0) hipLaunchKernelGGL((k_csrmm_rows_block<CL_, CONJ, 0>), dim3(gb), dim3(1024), 0, ctx->stream, \
                        N, rowptr, colind, vals, X, sxc, sxr, Y, ldy, syr, alpha, beta, wl.rows[1], wl.count + WL_SUB, wl.cap, wl.yperm); \
            else    hipLaunchKernelGGL((k_csrmm_rows_block<CL_, CONJ, 1>), dim3(gb), dim3(1024), 0, ctx->stream, \
                        N, rowptr, colind, vals, X, sxc, sxr, Y, ldy, syr, alpha, beta, wl.rows[1], wl.count + WL_SUB, wl.cap, wl.yperm); \
        }                                                                                          \
    } while (0)
        switch (s.CL) {
            case 1: IG_ROWS(1); break;   case 2: IG_ROWS(2); break;   case 4: IG_ROWS(4); break;
            case 8: IG_ROWS(8); break;   case 16: IG_ROWS(16); break; case 32: IG_ROWS(32); break;
            case 64: IG_ROWS(64); break;
        }
#undef IG_ROWS
        IG_LAUNCH_CHECK(ctx, "k_csrmm_rows");
    }
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
    const int xcd = 1;          // every XCD gets a contiguous range of row blocks (xcd_block)
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
        return launch_gather<false>(ctx, M, K, N, nnz, rowptr, colind, (const float2*)vals,
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
    return launch_gather<true>(ctx, K, M, N, nnz, rowptr_t, colind_t, (const float2*)vals_t,
                               (const float2*)X, ldx, (float2*)Y, ldy,
                               make_float2(ar, ai), make_float2(br, bi));
}

int ig_ccsrmm_rowperm(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                      float ar, float ai, const void* vals, const int32_t* colind, const int32_t* rowptr,
                      const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy,
                      const int32_t* yrow_perm) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_rowperm: ctx is NULL");
    if (int rc = check_panel_args(ctx, "ig_ccsrmm_rowperm", K, M, N, nnz, vals, colind, rowptr, X, ldx, Y, ldy)) return rc;
    if (N == 0 || M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    return launch_gather<false>(ctx, M, K, N, nnz, rowptr, colind, (const float2*)vals,
                                (const float2*)X, ldx, (float2*)Y, ldy, make_float2(ar, ai), make_float2(br, bi),
                                GridMask{nullptr, 0, 0}, yrow_perm, nullptr);
}

int ig_ccsrmm_xrows(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                    float ar, float ai, const void* vals, const int32_t* colind_c, const int32_t* rowptr,
                    const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy,
                    const int32_t* xrows, int64_t nxrows) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_xrows: ctx is NULL");
    if (int rc = check_panel_args(ctx, "ig_ccsrmm_xrows", K, M, N, nnz, vals, colind_c, rowptr, X, ldx, Y, ldy)) return rc;
    IG_REQUIRE(ctx, nxrows >= 0 && nxrows <= K && (nxrows == 0 || xrows), "ig_ccsrmm_xrows: bad row list");
    IG_REQUIRE(ctx, N >= 2 && N <= 64 || N == 0, "ig_ccsrmm_xrows: 2..64 panel columns (the subset is gathered while the panel is repacked)");
    if (N == 0 || M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    return launch_gather<false>(ctx, M, nxrows, N, nnz, rowptr, colind_c, (const float2*)vals,
                                (const float2*)X, ldx, (float2*)Y, ldy, make_float2(ar, ai), make_float2(br, bi),
                                GridMask{nullptr, 0, 0}, nullptr, xrows);
}

int ig_ccsrmm_t_grid(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                     float ar, float ai, const void* vals_t, const int32_t* colind_t, const int32_t* rowptr_t,
                     const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy,
                     const int16_t* support, int64_t n0, int64_t nm, const int32_t* xrow_perm) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_t_grid: ctx is NULL");
    if (int rc = check_panel_args(ctx, "ig_ccsrmm_t_grid", M, K, N, nnz, vals_t, colind_t, rowptr_t, X, ldx, Y, ldy)) return rc;
    IG_REQUIRE(ctx, !support || (n0 > 0 && nm > 0 && n0 % 16 == 0 && nm % 16 == 0 && nm <= 512 && K % (n0 * nm) == 0 && K < 0x7fffffffLL),
               "ig_ccsrmm_t_grid: rows (%lld) are not a grid of n0=%lld (multiple of 16) x nm=%lld (multiple of 16, <= 512) x ...",
               (long long)K, (long long)n0, (long long)nm);
    if (N == 0 || K == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    // third part of the support table: one bit per 16-row segment (after the z ranges and the y ranges)
    const int64_t ns = support ? K / (n0 * nm) : 0, nt = n0 / 16;
    GridMask mask{support ? reinterpret_cast<const uint32_t*>(support + 2 * (ns * nt + nt)) : nullptr, n0, nm};
    return launch_gather<true>(ctx, K, M, N, nnz, rowptr_t, colind_t, (const float2*)vals_t,
                               (const float2*)X, ldx, (float2*)Y, ldy,
                               make_float2(ar, ai), make_float2(br, bi), mask, nullptr, xrow_perm);
}

// ---- host-side structure analysis -------------------------------------------

int ig_ccsrmm_il(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                 float ar, float ai, const void* vals, const int32_t* colind, const int32_t* rowptr,
                 const void* X_il, float br, float bi, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_il: ctx is NULL");
    if (int rc = check_panel_args(ctx, "ig_ccsrmm_il", K, M, N, nnz, vals, colind, rowptr, X_il, K, Y, ldy)) return rc;
    IG_REQUIRE(ctx, N >= 1 && (N & (N - 1)) == 0 && N <= 64, "ig_ccsrmm_il: an interleaved panel needs a power-of-two column count <= 64 (got %lld)", (long long)N);
    if (M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    return launch_gather<false>(ctx, M, K, N, nnz, rowptr, colind, (const float2*)vals,
                                (const float2*)X_il, K, (float2*)Y, ldy, make_float2(ar, ai), make_float2(br, bi),
                                GridMask{nullptr, 0, 0}, nullptr, nullptr, true, false);
}

// The same product for a matrix whose weights are all real: `vals_re` = their real parts as floats (nnz x 4 bytes) beside the complex
// values (which the rows deferred to the long-row kernels, and every panel width other than 2, 4 or 8, still read).
int ig_ccsrmm_il_rw(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                    float ar, float ai, const void* vals, const float* vals_re, const int32_t* colind, const int32_t* rowptr,
                    const void* X_il, float br, float bi, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_il_rw: ctx is NULL");
    if (int rc = check_panel_args(ctx, "ig_ccsrmm_il_rw", K, M, N, nnz, vals, colind, rowptr, X_il, K, Y, ldy)) return rc;
    IG_REQUIRE(ctx, N >= 1 && (N & (N - 1)) == 0 && N <= 64, "ig_ccsrmm_il_rw: an interleaved panel needs a power-of-two column count <= 64 (got %lld)", (long long)N);
    IG_REQUIRE(ctx, nnz == 0 || vals_re, "ig_ccsrmm_il_rw: vals_re is NULL");
    if (M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    return launch_gather<false>(ctx, M, K, N, nnz, rowptr, colind, (const float2*)vals,
                                (const float2*)X_il, K, (float2*)Y, ldy, make_float2(ar, ai), make_float2(br, bi),
                                GridMask{nullptr, 0, 0}, nullptr, nullptr, true, false, vals_re);
}

int ig_ccsrmm_t_grid_il(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                        float ar, float ai, const void* vals_t, const int32_t* colind_t, const int32_t* rowptr_t,
                        const void* X, int64_t ldx, void* Y_il,
                        const int16_t* support, int64_t n0, int64_t nm) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_t_grid_il: ctx is NULL");
    if (int rc = check_panel_args(ctx, "ig_ccsrmm_t_grid_il", M, K, N, nnz, vals_t, colind_t, rowptr_t, X, ldx, Y_il, K)) return rc;
    IG_REQUIRE(ctx, N == 2 || N == 4 || N == 8, "ig_ccsrmm_t_grid_il: 2, 4 or 8 columns (got %lld)", (long long)N);
    IG_REQUIRE(ctx, !support || (n0 > 0 && nm > 0 && n0 % 16 == 0 && nm % 16 == 0 && nm <= 512 && K % (n0 * nm) == 0 && K < 0x7fffffffLL),
               "ig_ccsrmm_t_grid_il: rows (%lld) are not a grid of n0=%lld (multiple of 16) x nm=%lld (multiple of 16, <= 512) x ...",
               (long long)K, (long long)n0, (long long)nm);
    if (K == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const int64_t ns = support ? K / (n0 * nm) : 0, nt = n0 / 16;
    GridMask mask{support ? reinterpret_cast<const uint32_t*>(support + 2 * (ns * nt + nt)) : nullptr, n0, nm};
    return launch_gather<true>(ctx, K, M, N, nnz, rowptr_t, colind_t, (const float2*)vals_t,
                               (const float2*)X, ldx, (float2*)Y_il, K, make_float2(ar, ai), make_float2(0.f, 0.f),
                               mask, nullptr, nullptr, false, true);
}

namespace {
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline bool bricks_ok(int64_t n0, int64_t nm, int64_t ns, int bm, int bs, int unit) {
    return n0 > 0 && nm > 0 && ns > 0 && pow2(bm) && pow2(bs) && bm * bs <= 64 && n0 % 16 == 0 && nm % bm == 0 && ns % bs == 0 &&
           pow2(unit) && unit <= 64;
}
// the bricks one row touches and how many of its nonzeros fall into each (a row touches few bricks: linear search)
struct RowBricks {
    int64_t id[64]; int32_t cnt[64]; int n = 0;
    bool add(int64_t b) {
        for (int q = 0; q < n; ++q) if (id[q] == b) { ++cnt[q]; return true; }
        if (n == 64) return false;
        id[n] = b; cnt[n] = 1; ++n;
        return true;
    }
};
}  // namespace

// Both passes share the rows among a few host threads (contiguous row ranges, so that "in sample order inside a brick" is
// "thread 0's entries, then thread 1's, ..."): every thread counts its rows' padded entries per brick; the fill turns the
// per-thread counts into per-thread cursors.  (Single-threaded this was 4 s of the headline problem's 8 s of setup.)
namespace {
struct BrickGeom { int64_t n0, nm, ns, nbx, nbm, nbs, nb; int bm, bs, unit; };
inline int brick_threads(int64_t M) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? hw : 4);
    if (nt > 16) nt = 16;
    if (M < 16384) nt = 1;
    return nt;
}
// per-brick padded entry counts of rows [lo, hi); returns 0, or 1 (column outside the grid), 2 (a row touches > 64 bricks)
int count_rows(const BrickGeom& g, const int32_t* rowptr, const int32_t* colind, int64_t lo, int64_t hi, int32_t* cnt, int* overflow) {
    const int64_t P = g.n0 * g.nm * g.ns;
    for (int64_t t = lo; t < hi; ++t) {
        RowBricks rb;
        for (int32_t p = rowptr[t]; p < rowptr[t + 1]; ++p) {
            const int64_t col = colind[p];
            if (col < 0 || col >= P) return 1;
            const int64_t kx = col % g.n0, km = (col / g.n0) % g.nm, ks = col / (g.n0 * g.nm);
            if (!rb.add(kx / 16 + g.nbx * (km / g.bm + g.nbm * (ks / g.bs)))) return 2;
        }
        for (int q = 0; q < rb.n; ++q) {
            const int64_t padded = (rb.cnt[q] + g.unit - 1) / g.unit * g.unit;
            if ((int64_t)cnt[rb.id[q]] + padded > 0x7fffffffLL) { *overflow = 1; return 0; }
            cnt[rb.id[q]] += (int32_t)padded;
        }
    }
    return 0;
}
void run_threads(int nt, const std::function<void(int)>& body) {
    if (nt == 1) { body(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([t, &body]() { body(t); });
    for (auto& x : th) x.join();
}
}  // namespace

int ig_grid_bricks_count(int64_t M, const int32_t* rowptr, const int32_t* colind, int64_t n0, int64_t nm, int64_t ns,
                         int bm, int bs, int unit, int32_t* brick_entries /* (n0/16)*(nm/bm)*(ns/bs), zero-initialised by this call */) {
    if (M < 0 || !rowptr || !brick_entries || !bricks_ok(n0, nm, ns, bm, bs, unit))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_bricks_count: the grid must divide into 16 x bm x bs bricks (powers of two, bm*bs <= 64), unit a power of two <= 64");
    BrickGeom g{n0, nm, ns, n0 / 16, nm / bm, ns / bs, 0, bm, bs, unit};
    g.nb = g.nbx * g.nbm * g.nbs;
    const int nt = brick_threads(M);
    const int64_t per = (M + nt - 1) / nt;
    std::vector<std::vector<int32_t>> cnt((size_t)nt);
    std::vector<int> rc((size_t)nt, 0), ovf((size_t)nt, 0);
    run_threads(nt, [&](int t) {
        cnt[t].assign((size_t)g.nb, 0);
        const int64_t lo = std::min<int64_t>(M, t * per), hi = std::min<int64_t>(M, lo + per);
        rc[t] = count_rows(g, rowptr, colind, lo, hi, cnt[t].data(), &ovf[t]);
    });
    for (int t = 0; t < nt; ++t) {
        if (rc[t] == 1) return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_bricks_count: column index outside the grid");
        if (rc[t] == 2) return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_grid_bricks_count: a row touches more than 64 bricks");
    }
    std::atomic<int> bad{0};        // set from several host threads
    run_threads(nt, [&](int t) {                   // sum over the threads, brick ranges in parallel
        const int64_t pb = (g.nb + nt - 1) / nt, lo = std::min<int64_t>(g.nb, t * pb), hi = std::min<int64_t>(g.nb, lo + pb);
        for (int64_t b = lo; b < hi; ++b) {
            int64_t sum = 0;
            for (int u = 0; u < nt; ++u) sum += cnt[u][b];
            if (sum > 0x7fffffffLL) { bad = 1; sum = 0; }
            brick_entries[b] = (int32_t)sum;
        }
    });
    for (int t = 0; t < nt; ++t) if (ovf[t]) bad = 1;
    if (bad) return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_bricks_count: a brick exceeds 2^31 entries");
    return IG_OK;
}

int ig_grid_bricks_fill(int64_t M, const int32_t* rowptr, const int32_t* colind, const void* vals, int64_t n0, int64_t nm, int64_t ns,
                        int bm, int bs, int unit, const int64_t* brick_ptr /* exclusive prefix sums of the counts, nbricks + 1 */,
                        void* entries /* brick_ptr[nbricks] x 12 bytes: {uint32 cell in brick, float re, float im} */,
                        uint32_t* round_rows /* brick_ptr[nbricks] / unit: the row of each group of `unit` entries */) {
    if (M < 0 || !rowptr || !brick_ptr || !bricks_ok(n0, nm, ns, bm, bs, unit) || (rowptr[M] > rowptr[0] && (!colind || !vals || !entries || !round_rows)))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_bricks_fill: bad arguments");
    BrickGeom g{n0, nm, ns, n0 / 16, nm / bm, ns / bs, 0, bm, bs, unit};
    g.nb = g.nbx * g.nbm * g.nbs;
    const int64_t nbx = g.nbx, nbm = g.nbm;
    const int nt = brick_threads(M);
    const int64_t per = (M + nt - 1) / nt;
    // per-thread counts -> per-thread cursors (offsets from the brick's start)
    std::vector<std::vector<int32_t>> cur((size_t)nt);
    std::vector<int> rc((size_t)nt, 0), ovf((size_t)nt, 0);
    run_threads(nt, [&](int t) {
        cur[t].assign((size_t)g.nb, 0);
        const int64_t lo = std::min<int64_t>(M, t * per), hi = std::min<int64_t>(M, lo + per);
        rc[t] = count_rows(g, rowptr, colind, lo, hi, cur[t].data(), &ovf[t]);
    });
    for (int t = 0; t < nt; ++t)
        if (rc[t] || ovf[t]) return ig_fail(nullptr, rc[t] == 2 ? IG_ERR_UNSUPPORTED : IG_ERR_ARG, "ig_grid_bricks_fill: the matrix does not fit the brick format (see ig_grid_bricks_count)");
    std::atomic<int> mismatch{0};   // set from several host threads
    run_threads(nt, [&](int t) {                   // exclusive scan over the threads, brick ranges in parallel
        const int64_t pb = (g.nb + nt - 1) / nt, lo = std::min<int64_t>(g.nb, t * pb), hi = std::min<int64_t>(g.nb, lo + pb);
        for (int64_t b = lo; b < hi; ++b) {
            int64_t run = 0;
            for (int u = 0; u < nt; ++u) { const int32_t c = cur[u][b]; cur[u][b] = (int32_t)run; run += c; }
            if (run != brick_ptr[b + 1] - brick_ptr[b]) mismatch = 1;
        }
    });
    if (mismatch) return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_bricks_fill: brick_ptr does not come from ig_grid_bricks_count");
    const float2* v = (const float2*)vals;
    BrickEntry* out = (BrickEntry*)entries;
    run_threads(nt, [&](int th) {
        int32_t* cursor = cur[th].data();
        const int64_t lo = std::min<int64_t>(M, th * per), hi = std::min<int64_t>(M, lo + per);
        for (int64_t t = lo; t < hi; ++t) {
            RowBricks rb;
            int64_t start[64];
            for (int32_t p = rowptr[t]; p < rowptr[t + 1]; ++p) {
                const int64_t col = colind[p];
                const int64_t kx = col % n0, km = (col / n0) % nm, ks = col / (n0 * nm);
                const int64_t b = kx / 16 + nbx * (km / bm + nbm * (ks / bs));
                const int before = rb.n;
                rb.add(b);
                int q = 0;
                while (rb.id[q] != b) ++q;
                if (rb.n > before) start[q] = brick_ptr[b] + cursor[b];
                BrickEntry e;
                e.cell = (uint32_t)((kx % 16) + 16 * ((km % bm) + bm * (ks % bs)));
                e.re = v[p].x; e.im = v[p].y;
                out[start[q] + rb.cnt[q] - 1] = e;
            }
            for (int q = 0; q < rb.n; ++q) {            // pad this row's share of each brick to a multiple of `unit`
                const int64_t padded = (rb.cnt[q] + unit - 1) / unit * unit;
                for (int64_t i = rb.cnt[q]; i < padded; ++i) {
                    BrickEntry e; e.cell = 0xffffffffu; e.re = 0.f; e.im = 0.f;
                    out[start[q] + i] = e;
                }
                for (int64_t i = 0; i < padded; i += unit) round_rows[(start[q] + i) / unit] = (uint32_t)t;
                cursor[rb.id[q]] += (int32_t)padded;
            }
        }
    });
    return IG_OK;
}

int ig_ccsrmm_t_bricks(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, float ar, float ai,
                       const void* entries, const uint32_t* round_rows, const void* X, int64_t ldx, void* Y_il, const int16_t* support, int64_t n0, int64_t nm,
                       int bm, int bs, const int32_t* tasks, int64_t ntasks, const int32_t* brick_table,
                       const int32_t* shared_bricks, int64_t nshared, int support_tile, int support_zwords, int entry_words) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_t_bricks: ctx is NULL");
    IG_REQUIRE(ctx, entry_words == 3 || entry_words == 2, "ig_ccsrmm_t_bricks: entries of 3 words {cell, re, im} or 2 {cell, re} (got %d)", entry_words);
    const bool realw = entry_words == 2;
    const int zw = support_zwords > 0 ? support_zwords : 16;
    IG_REQUIRE(ctx, M >= 0 && K >= 0 && M <= 0x7fffffffLL, "ig_ccsrmm_t_bricks: bad dimensions");
    IG_REQUIRE(ctx, N == 4 || N == 8, "ig_ccsrmm_t_bricks: 4 or 8 columns (got %lld); entries must be padded to 64/N per row and brick", (long long)N);
    IG_REQUIRE(ctx, (ntasks == 0 || (entries && round_rows)) && (M == 0 || X) && (K == 0 || Y_il) && ldx >= M, "ig_ccsrmm_t_bricks: NULL array or short leading dimension");
    IG_REQUIRE(ctx, n0 > 0 && nm > 0 && K % (n0 * nm) == 0 && K < 0x7fffffffLL && bricks_ok(n0, nm, K / (n0 * nm), bm, bs, 8) && bm * bs <= 32,
               "ig_ccsrmm_t_bricks: rows (%lld) are not a grid of n0=%lld x nm=%lld x ... that divides into 16 x %d x %d bricks", (long long)K, (long long)n0, (long long)nm, bm, bs);
    const int64_t ns = K / (n0 * nm);
    IG_REQUIRE(ctx, !support || (zw <= 64 && nm <= 32 * (int64_t)zw), "ig_ccsrmm_t_bricks: the support bitmaps hold 32 bits in each of %d words: nm <= %d", zw, 32 * zw);
    IG_REQUIRE(ctx, support_tile == 16 || support_tile == 8 || support_tile == 4, "ig_ccsrmm_t_bricks: support_tile %d (kx points per entry of the support table: 16, 8 or 4)", support_tile);
    IG_REQUIRE(ctx, (16 / support_tile) * bm * bs <= 32, "ig_ccsrmm_t_bricks: at most 32 segments per brick (%d x %d x %d)", 16 / support_tile, bm, bs);
    const int st_log2 = support_tile == 16 ? 4 : support_tile == 8 ? 3 : 2;
    IG_REQUIRE(ctx, ntasks >= 0 && ntasks <= 0x7fffffffLL && (ntasks == 0 || (tasks && brick_table)) && nshared >= 0 && (nshared == 0 || shared_bricks), "ig_ccsrmm_t_bricks: bad task list");
    IG_REQUIRE(ctx, M * N * 8 < 0x7fffffffLL, "ig_ccsrmm_t_bricks: the panel (%lld x %lld) exceeds the 2 GB window of a buffer descriptor", (long long)M, (long long)N);
    if (K == 0 || ntasks == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const int64_t nt = n0 / 16;
    const int64_t snt = n0 / support_tile;                                // support entries per grid row
    const uint32_t* bits = support ? reinterpret_cast<const uint32_t*>(support + 2 * (ns * snt + snt)) : nullptr;
    const float2 alpha = make_float2(ar, ai);
    // the k-space panel as packed rows [t][N] (column-major in, as everywhere at the boundary)
    const size_t need = (size_t)M * N * 8;
    if (ctx->xpack_bytes < need) {
        if (ctx->d_xpack) { IG_HIP(ctx, hipStreamSynchronize(ctx->stream)); IG_HIP(ctx, hipFree(ctx->d_xpack)); ctx->d_xpack = nullptr; ctx->xpack_bytes = 0; }
        IG_HIP(ctx, hipMalloc((void**)&ctx->d_xpack, need));
        ctx->xpack_bytes = need;
    }
    float2* xp = (float2*)ctx->d_xpack;
    {
        ig_prof_scope prof(ctx, "pack_panel", 2.0 * (double)need);
        int64_t g = (M * N + BLK - 1) / BLK;
        const int64_t cap = (int64_t)ctx->num_cu * 16;
        if (g > cap) g = cap;
        if (N == 4) hipLaunchKernelGGL(k_pack_panel<4>, dim3((unsigned)g), dim3(BLK), 0, ctx->stream, M, N, (const float2*)X, ldx, xp, (const int32_t*)nullptr);
        else        hipLaunchKernelGGL(k_pack_panel<8>, dim3((unsigned)g), dim3(BLK), 0, ctx->stream, M, N, (const float2*)X, ldx, xp, (const int32_t*)nullptr);
        IG_LAUNCH_CHECK(ctx, "k_pack_panel");
    }
    int bm_log2 = 0, bs_log2 = 0;
    while ((1 << bm_log2) < bm) ++bm_log2;
    while ((1 << bs_log2) < bs) ++bs_log2;
    const int nbx = (int)nt, nbm = (int)(nm / bm);
    const size_t lds = (size_t)WAVES_PER_BLOCK * 16 * bm * bs * N * 8;          // one brick image per wave
    IG_REQUIRE(ctx, lds <= 64 * 1024, "ig_ccsrmm_t_bricks: bricks of 16 x %d x %d points x %lld columns need %zu bytes of LDS per workgroup (limit 64 KB)", bm, bs, (long long)N, lds);
    const unsigned blocks = (unsigned)((ntasks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
#define IG_BRICKS(NC_, NSEG_) IG_BRICKS_P(NC_, NSEG_, false)
#define IG_BRICKS_P(NC_, NSEG_, PAIR_) do {                                                                                            \
        if (nshared) {                                                                                                          \
            ig_prof_scope prof(ctx, "grid_bricks_zero");                                                                        \
            hipLaunchKernelGGL((k_grid_bricks_zero<NC_>), dim3((unsigned)nshared), dim3(BLK), 0, ctx->stream, shared_bricks, (float2*)Y_il, bits, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2, zw); } \
        ig_prof_scope prof(ctx, "csrmm_bricks_conj");                                                                           \
        if (realw) hipLaunchKernelGGL((k_grid_bricks<NC_, NSEG_, PAIR_, true>), dim3(blocks), dim3(BLK), lds, ctx->stream, (const BrickTask*)tasks, (int)ntasks, (const BrickRef*)brick_table, (const BrickEntry*)entries, round_rows, \
                           (const float2*)xp, (float2*)Y_il, alpha, bits, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2, zw); \
        else hipLaunchKernelGGL((k_grid_bricks<NC_, NSEG_, PAIR_, false>), dim3(blocks), dim3(BLK), lds, ctx->stream, (const BrickTask*)tasks, (int)ntasks, (const BrickRef*)brick_table, (const BrickEntry*)entries, round_rows, \
                           (const float2*)xp, (float2*)Y_il, alpha, bits, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2, zw); } while (0)
    const int nseg_total = (16 / support_tile) * bm * bs;
    if (N == 8 && nseg_total == 4) IG_BRICKS(8, 4);
    else if (N == 8 && nseg_total == 8) IG_BRICKS(8, 8);
    else if (N == 8 && nseg_total == 16 && support_tile == 4) IG_BRICKS_P(8, 8, true);      // the 4-point table: pairs of segments
    else if (N == 8) IG_BRICKS(8, 0);
    else IG_BRICKS(4, 0);
#undef IG_BRICKS
#undef IG_BRICKS_P
    IG_LAUNCH_CHECK(ctx, "k_grid_bricks");
    return IG_OK;
}

// Slot format of k_grid_slots from the UNPADDED brick format (ig_grid_bricks_count / _fill with unit = 1: entries12 in brick
// order, entry_rows = the sample of every entry, brick_ptr = prefix sums): inside every brick the entries are reordered by
// (occurrence of their cell, cell) and cut into slots of at most 64.  Outputs: entries16 (one per entry), brick_slots (slots
// per brick), slot_ptr (nslots + 1 offsets into entries16; sized nentries + 1 by the caller), *nslots.
int ig_grid_slots_build(int64_t nbricks, const int64_t* brick_ptr, const void* entries12, const uint32_t* entry_rows, int ncell,
                        void* entries16, int32_t* brick_slots, int32_t* slot_ptr, int64_t* nslots) {
    if (nbricks < 0 || !brick_ptr || ncell < 1 || ncell > 1024 || !brick_slots || !slot_ptr || !nslots ||
        (brick_ptr[nbricks] > 0 && (!entries12 || !entry_rows || !entries16)) || brick_ptr[nbricks] > 0x7fffffffLL)
        return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_slots_build: bad arguments");
    const BrickEntry* in = (const BrickEntry*)entries12;
    SlotEntry* out = (SlotEntry*)entries16;
    const int nt = brick_threads(nbricks * 8);
    const int64_t per = (nbricks + nt - 1) / nt;
    std::atomic<int> bad{0};        // set from several host threads
    // pass 1: slots per brick (the group sizes: how many cells are hit at least k + 1 times)
    run_threads(nt, [&](int th) {
        std::vector<int32_t> mult((size_t)ncell), gsize;
        const int64_t lo = std::min<int64_t>(nbricks, th * per), hi = std::min<int64_t>(nbricks, lo + per);
        for (int64_t b = lo; b < hi; ++b) {
            const int64_t p0 = brick_ptr[b], p1 = brick_ptr[b + 1];
            if (p1 == p0) { brick_slots[b] = 0; continue; }
            std::fill(mult.begin(), mult.end(), 0);
            gsize.clear();
            for (int64_t p = p0; p < p1; ++p) {
                const uint32_t c = in[p].cell;
                if (c >= (uint32_t)ncell) { bad = 1; continue; }
                const int k = mult[c]++;
                if ((size_t)k >= gsize.size()) gsize.push_back(0);
                ++gsize[k];
            }
            int64_t ns = 0;
            for (int32_t g : gsize) ns += (g + 63) / 64;
            brick_slots[b] = (int32_t)ns;
        }
    });
    if (bad) return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_slots_build: a cell index outside the brick");
    std::vector<int64_t> first((size_t)nbricks + 1, 0);
    for (int64_t b = 0; b < nbricks; ++b) first[b + 1] = first[b] + brick_slots[b];
    if (first[nbricks] > 0x7fffffffLL) return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_slots_build: more than 2^31 - 1 slots");
    *nslots = first[nbricks];
    // pass 2: reorder and emit the slot offsets
    run_threads(nt, [&](int th) {
        std::vector<int32_t> mult((size_t)ncell), gstart;
        std::vector<int32_t> gsize;
        const int64_t lo = std::min<int64_t>(nbricks, th * per), hi = std::min<int64_t>(nbricks, lo + per);
        for (int64_t b = lo; b < hi; ++b) {
            const int64_t p0 = brick_ptr[b], p1 = brick_ptr[b + 1];
            if (p1 == p0) continue;
            std::fill(mult.begin(), mult.end(), 0);
            gsize.clear();
            for (int64_t p = p0; p < p1; ++p) {
                const int k = mult[in[p].cell]++;
                if ((size_t)k >= gsize.size()) gsize.push_back(0);
                ++gsize[k];
            }
            gstart.assign(gsize.size() + 1, 0);
            for (size_t k = 0; k < gsize.size(); ++k) gstart[k + 1] = gstart[k] + gsize[k];
            int64_t s = first[b];
            for (size_t k = 0; k < gsize.size(); ++k)
                for (int32_t o = 0; o < gsize[k]; o += 64) slot_ptr[s++] = (int32_t)(p0 + gstart[k] + o);
            std::fill(mult.begin(), mult.end(), 0);
            std::vector<int32_t> cursor(gstart.begin(), gstart.end() - 1);
            for (int64_t p = p0; p < p1; ++p) {
                const int k = mult[in[p].cell]++;
                SlotEntry e; e.cell = in[p].cell; e.re = in[p].re; e.im = in[p].im; e.row = entry_rows[p];
                out[p0 + cursor[k]++] = e;
            }
        }
    });
    slot_ptr[first[nbricks]] = (int32_t)brick_ptr[nbricks];
    return IG_OK;
}

// Y_il(grid x N, rows interleaved) = alpha * G^H X for N = 1, 2 or 4 columns through the slot format (k_grid_slots); tasks and
// brick_table as for ig_ccsrmm_t_bricks with SLOTS in place of entries.  support / shared_bricks as there.
int ig_ccsrmm_t_slots(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, float ar, float ai,
                      const void* entries16, const int32_t* slot_ptr, const void* X, int64_t ldx, void* Y_il, const int16_t* support,
                      int64_t n0, int64_t nm, int bm, int bs, const int32_t* tasks, int64_t ntasks, const int32_t* brick_table,
                      const int32_t* shared_bricks, int64_t nshared, int support_tile, int support_zwords, int entry_words) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_t_slots: ctx is NULL");
    IG_REQUIRE(ctx, entry_words == 4 || entry_words == 3, "ig_ccsrmm_t_slots: entries of 4 words {cell, re, im, row} or 3 {cell, re, row} (got %d)", entry_words);
    const bool realw = entry_words == 3;
    const int zw = support_zwords > 0 ? support_zwords : 16;
    IG_REQUIRE(ctx, M >= 0 && K >= 0 && M <= 0x7fffffffLL, "ig_ccsrmm_t_slots: bad dimensions");
    IG_REQUIRE(ctx, N == 1 || N == 2 || N == 4, "ig_ccsrmm_t_slots: 1, 2 or 4 columns (got %lld; at 8 the round format of ig_ccsrmm_t_bricks is faster: 0.77 against 1.28 ms)", (long long)N);
    IG_REQUIRE(ctx, (ntasks == 0 || (entries16 && slot_ptr)) && (M == 0 || X) && (K == 0 || Y_il) && ldx >= M, "ig_ccsrmm_t_slots: NULL array or short leading dimension");
    IG_REQUIRE(ctx, n0 > 0 && nm > 0 && K % (n0 * nm) == 0 && K < 0x7fffffffLL && bricks_ok(n0, nm, K / (n0 * nm), bm, bs, 1) && bm * bs <= 32,
               "ig_ccsrmm_t_slots: rows (%lld) are not a grid of n0=%lld x nm=%lld x ... that divides into 16 x %d x %d bricks", (long long)K, (long long)n0, (long long)nm, bm, bs);
    const int64_t ns = K / (n0 * nm);
    IG_REQUIRE(ctx, !support || (zw <= 64 && nm <= 32 * (int64_t)zw), "ig_ccsrmm_t_slots: the support bitmaps hold 32 bits in each of %d words: nm <= %d", zw, 32 * zw);
    IG_REQUIRE(ctx, support_tile == 16 || support_tile == 8 || support_tile == 4, "ig_ccsrmm_t_slots: support_tile %d", support_tile);
    IG_REQUIRE(ctx, (16 / support_tile) * bm * bs <= 32, "ig_ccsrmm_t_slots: at most 32 segments per brick");
    const int st_log2 = support_tile == 16 ? 4 : support_tile == 8 ? 3 : 2;
    IG_REQUIRE(ctx, ntasks >= 0 && ntasks <= 0x7fffffffLL && (ntasks == 0 || (tasks && brick_table)) && nshared >= 0 && (nshared == 0 || shared_bricks), "ig_ccsrmm_t_slots: bad task list");
    IG_REQUIRE(ctx, M * N * 8 < 0x7fffffffLL, "ig_ccsrmm_t_slots: the panel exceeds the 2 GB window of a buffer descriptor");
    if (K == 0 || ntasks == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const int64_t snt = n0 / support_tile;
    const uint32_t* bits = support ? reinterpret_cast<const uint32_t*>(support + 2 * (ns * snt + snt)) : nullptr;
    const float2 alpha = make_float2(ar, ai);
    // the k-space panel as packed rows [t][N] (one column: it already is)
    const float2* xp = (const float2*)X;
    if (N > 1) {
        const size_t need = (size_t)M * N * 8;
        if (ctx->xpack_bytes < need) {
            if (ctx->d_xpack) { IG_HIP(ctx, hipStreamSynchronize(ctx->stream)); IG_HIP(ctx, hipFree(ctx->d_xpack)); ctx->d_xpack = nullptr; ctx->xpack_bytes = 0; }
            IG_HIP(ctx, hipMalloc((void**)&ctx->d_xpack, need));
            ctx->xpack_bytes = need;
        }
        ig_prof_scope prof(ctx, "pack_panel", 2.0 * (double)need);
        int64_t g = (M * N + BLK - 1) / BLK;
        const int64_t cap = (int64_t)ctx->num_cu * 16;
        if (g > cap) g = cap;
        if (N == 2) hipLaunchKernelGGL(k_pack_panel<2>, dim3((unsigned)g), dim3(BLK), 0, ctx->stream, M, N, (const float2*)X, ldx, (float2*)ctx->d_xpack, (const int32_t*)nullptr);
        else        hipLaunchKernelGGL(k_pack_panel<4>, dim3((unsigned)g), dim3(BLK), 0, ctx->stream, M, N, (const float2*)X, ldx, (float2*)ctx->d_xpack, (const int32_t*)nullptr);
        IG_LAUNCH_CHECK(ctx, "k_pack_panel");
        xp = (const float2*)ctx->d_xpack;
    }
    int bm_log2 = 0, bs_log2 = 0;
    while ((1 << bm_log2) < bm) ++bm_log2;
    while ((1 << bs_log2) < bs) ++bs_log2;
    const int nbx = (int)(n0 / 16), nbm = (int)(nm / bm);
    const size_t lds = (size_t)WAVES_PER_BLOCK * 16 * bm * bs * N * 8;
    IG_REQUIRE(ctx, lds <= 64 * 1024, "ig_ccsrmm_t_slots: brick images need %zu bytes of LDS per workgroup (limit 64 KB)", lds);
    const unsigned blocks = (unsigned)((ntasks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
#define IG_SLOTS(NC_) do {                                                                                                      \
        if (nshared) {                                                                                                          \
            ig_prof_scope prof(ctx, "grid_bricks_zero");                                                                        \
            hipLaunchKernelGGL((k_grid_bricks_zero<NC_>), dim3((unsigned)nshared), dim3(BLK), 0, ctx->stream, shared_bricks, (float2*)Y_il, bits, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2, zw); } \
        ig_prof_scope prof(ctx, "csrmm_slots_conj");                                                                            \
        if (realw) hipLaunchKernelGGL((k_grid_slots<NC_, true>), dim3(blocks), dim3(BLK), lds, ctx->stream, (const BrickTask*)tasks, (int)ntasks, (const BrickRef*)brick_table, slot_ptr, (const SlotEntry*)entries16, \
                           xp, (float2*)Y_il, alpha, bits, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2, zw); \
        else hipLaunchKernelGGL((k_grid_slots<NC_, false>), dim3(blocks), dim3(BLK), lds, ctx->stream, (const BrickTask*)tasks, (int)ntasks, (const BrickRef*)brick_table, slot_ptr, (const SlotEntry*)entries16, \
                           xp, (float2*)Y_il, alpha, bits, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2, zw); } while (0)
    if (N == 1) IG_SLOTS(1); else if (N == 2) IG_SLOTS(2); else IG_SLOTS(4);
#undef IG_SLOTS
    IG_LAUNCH_CHECK(ctx, "k_grid_slots");
    return IG_OK;
}

int ig_ccsrmm_t_bricks_wide(ig_ctx* ctx, int64_t M, int64_t K, float ar, float ai,
                            const void* entries, const uint32_t* entry_rows, const void* X, int64_t ldx, void* Y, int64_t ldy,
                            const int32_t* tasks, int64_t ntasks, const int32_t* brick_table, const uint32_t* owned_tiles) {
    return ig_ccsrmm_t_bricks_wide_grid(ctx, M, K, ar, ai, entries, entry_rows, X, ldx, Y, ldy, tasks, ntasks, brick_table, owned_tiles, 0, 0, 1, 1, 3);
}

int ig_ccsrmm_t_bricks_wide_grid(ig_ctx* ctx, int64_t M, int64_t K, float ar, float ai,
                                 const void* entries, const uint32_t* entry_rows, const void* X, int64_t ldx, void* Y, int64_t ldy,
                                 const int32_t* tasks, int64_t ntasks, const int32_t* brick_table, const uint32_t* owned_tiles,
                                 int64_t n0, int64_t nm, int bm, int bs, int entry_words) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_t_bricks_wide: ctx is NULL");
    const int64_t N = 64;
    const bool grid = bm * bs > 1;
    IG_REQUIRE(ctx, entry_words == 3 || (entry_words == 2 && grid), "ig_ccsrmm_t_bricks_wide_grid: entries of 3 words {cell, re, im}, or 2 {cell, re} (grid bricks only); got %d", entry_words);
    IG_REQUIRE(ctx, !grid || (n0 > 0 && nm > 0 && K % (n0 * nm) == 0 && bricks_ok(n0, nm, K / (n0 * nm), bm, bs, 1) && bm * bs <= 4),
               "ig_ccsrmm_t_bricks_wide_grid: rows (%lld) are not a grid of n0=%lld x nm=%lld x ... that divides into 16 x %d x %d bricks (bm * bs <= 4)",
               (long long)K, (long long)n0, (long long)nm, bm, bs);
    IG_REQUIRE(ctx, M >= 0 && K >= 0 && K % 16 == 0 && K <= 0x7fffffffLL, "ig_ccsrmm_t_bricks_wide: K (%lld) must be a multiple of 16", (long long)K);
    IG_REQUIRE(ctx, (ntasks == 0 || (entries && entry_rows && tasks && brick_table)) && (M == 0 || X) && (K == 0 || Y) && ldx >= M && ldy >= K,
               "ig_ccsrmm_t_bricks_wide: NULL array or short leading dimension");
    IG_REQUIRE(ctx, ntasks >= 0 && ntasks <= 0x7fffffffLL, "ig_ccsrmm_t_bricks_wide: bad task list");
    IG_REQUIRE(ctx, M * N * 8 < 0x7fffffffLL, "ig_ccsrmm_t_bricks_wide: the panel (%lld x 64) exceeds the 2 GB window of a buffer descriptor", (long long)M);
    if (K == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    if (owned_tiles && ntasks > 0 && M > 0 && (reinterpret_cast<uintptr_t>(Y) & 15u) == 0 && ldy % 2 == 0) {
        // every tile is written exactly once: here if no task owns it, by its owner's flush otherwise
        ig_prof_scope prof(ctx, "bricks_wide_zero");
        hipLaunchKernelGGL(k_wide_zero_unowned, dim3((unsigned)((K + 4095) / 4096)), dim3(BLK), 0, ctx->stream, owned_tiles, (float2*)Y, ldy, K);
        IG_LAUNCH_CHECK(ctx, "k_wide_zero_unowned");
    } else {
        ig_prof_scope prof(ctx, "bricks_wide_zero", (double)K * N * 8.0);
        if (ldy == K) IG_HIP(ctx, hipMemsetAsync(Y, 0, (size_t)K * N * 8, ctx->stream));
        else IG_HIP(ctx, hipMemset2DAsync(Y, (size_t)ldy * 8, 0, (size_t)K * 8, (size_t)N, ctx->stream));
    }
    if (ntasks == 0 || M == 0) return IG_OK;
    const size_t need = (size_t)M * N * 8;
    if (ctx->xpack_bytes < need) {
        if (ctx->d_xpack) { IG_HIP(ctx, hipStreamSynchronize(ctx->stream)); IG_HIP(ctx, hipFree(ctx->d_xpack)); ctx->d_xpack = nullptr; ctx->xpack_bytes = 0; }
        IG_HIP(ctx, hipMalloc((void**)&ctx->d_xpack, need));
        ctx->xpack_bytes = need;
    }
    float2* xp = (float2*)ctx->d_xpack;
    {
        ig_prof_scope prof(ctx, "pack_panel", 2.0 * (double)need);
        int64_t gt = (M + 63) / 64;
        const int64_t cap = (int64_t)ctx->num_cu * 16;
        if (gt > cap) gt = cap;
        hipLaunchKernelGGL(k_pack_panel_tiled<64>, dim3((unsigned)gt), dim3(BLK), 0, ctx->stream, M, N, (const float2*)X, ldx, xp, (const int32_t*)nullptr);
        IG_LAUNCH_CHECK(ctx, "k_pack_panel");
    }
    ig_prof_scope prof(ctx, "csrmm_bricks_wide_conj");
    const unsigned blocks = (unsigned)((ntasks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    if (grid) {
        int bm_log2 = 0, bs_log2 = 0;
        while ((1 << bm_log2) < bm) ++bm_log2;
        while ((1 << bs_log2) < bs) ++bs_log2;
        const int nbx = (int)(n0 / 16), nbm = (int)(nm / bm);
#define IG_WIDE_R(NT_) do { if (entry_words == 2) hipLaunchKernelGGL((k_bricks_wide64r<NT_, true>), dim3(blocks), dim3(BLK), 0, ctx->stream, (const BrickTask*)tasks, (int)ntasks, (const BrickRef*)brick_table, \
                       (const BrickEntry*)entries, entry_rows, (const float2*)xp, (float2*)Y, ldy, make_float2(ar, ai), (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm); \
                       else hipLaunchKernelGGL((k_bricks_wide64r<NT_, false>), dim3(blocks), dim3(BLK), 0, ctx->stream, (const BrickTask*)tasks, (int)ntasks, (const BrickRef*)brick_table, \
                       (const BrickEntry*)entries, entry_rows, (const float2*)xp, (float2*)Y, ldy, make_float2(ar, ai), (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm); } while (0)
        if (bm * bs == 2) IG_WIDE_R(2); else IG_WIDE_R(4);
#undef IG_WIDE_R
        IG_LAUNCH_CHECK(ctx, "k_bricks_wide64r");
        return IG_OK;
    }
    hipLaunchKernelGGL(k_bricks_wide64, dim3(blocks), dim3(BLK), 0, ctx->stream, (const BrickTask*)tasks, (int)ntasks, (const BrickRef*)brick_table,
                       (const BrickEntry*)entries, entry_rows, (const float2*)xp, (float2*)Y, ldy, make_float2(ar, ai));
    IG_LAUNCH_CHECK(ctx, "k_bricks_wide64");
    return IG_OK;
}

// The run format of k_csrmm_runs64r from a CSR with sorted rows: the nonzeros of every run of 16 consecutive rows grouped by
// column.  Two calls: dcols == NULL counts (run_dptr[nruns + 1] = prefix sums of the runs' distinct columns), the second fills
// dcols (run_dptr[nruns] words) and entries (nnz x 12 bytes).  *all_real = every value has a zero imaginary part.
// IG_ERR_UNSUPPORTED when a row holds a column twice or a column index needs more than 27 bits.
int ig_csr_runs_build(int64_t M, int64_t K, const int32_t* rowptr, const int32_t* colind, const void* vals,
                      int32_t* run_dptr, uint32_t* dcols, void* entries, int* all_real) {
    if (M < 0 || !rowptr || !run_dptr || (rowptr[M] > rowptr[0] && (!colind || !vals)) || (dcols && !entries))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_csr_runs_build: bad arguments");
    if (K > (1LL << 27)) return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_csr_runs_build: column indices need more than 27 bits");
    const int64_t nruns = (M + RUN_ROWS - 1) / RUN_ROWS;
    const float2* v = (const float2*)vals;
    RunEntry* out = (RunEntry*)entries;
    const int nt = brick_threads(M);
    const int64_t per = (nruns + nt - 1) / nt;
    std::atomic<int> bad{0}, cplx{0};
    if (!dcols) run_dptr[0] = 0;
    run_threads(nt, [&](int th) {
        struct Nz { int32_t col; uint32_t row; float2 val; };
        std::vector<Nz> buf;
        const int64_t lo = std::min<int64_t>(nruns, th * per), hi = std::min<int64_t>(nruns, lo + per);
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t rlo = r * RUN_ROWS, rhi = std::min<int64_t>(M, rlo + RUN_ROWS);
            buf.clear();
            for (int64_t t = rlo; t < rhi; ++t)
                for (int32_t p = rowptr[t]; p < rowptr[t + 1]; ++p) {
                    if (colind[p] < 0 || colind[p] >= K) { bad = 1; continue; }
                    buf.push_back(Nz{colind[p], (uint32_t)(t - rlo), v[p]});
                }
            std::stable_sort(buf.begin(), buf.end(), [](const Nz& a, const Nz& b) { return a.col < b.col; });
            int32_t nd = 0;
            for (size_t i = 0; i < buf.size(); ++i) {
                if (i == 0 || buf[i].col != buf[i - 1].col) ++nd;
                else if (buf[i].row == buf[i - 1].row) bad = 1;                       // a row holds a column twice
            }
            if (!dcols) { run_dptr[r + 1] = nd; continue; }                           // counting pass: sizes, prefix-summed below
            uint32_t* dc = dcols + run_dptr[r];
            RunEntry* e = out + (rowptr[rlo] - rowptr[0]);
            int32_t j = -1;
            for (size_t i = 0; i < buf.size(); ++i) {
                if (i == 0 || buf[i].col != buf[i - 1].col) { ++j; dc[j] = (uint32_t)buf[i].col; }
                else dc[j] += 1u << 27;
                e[i] = RunEntry{buf[i].row, buf[i].val.x, buf[i].val.y};
                if (buf[i].val.y != 0.f) cplx = 1;
            }
        }
    });
    if (bad) return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_csr_runs_build: a column index outside the matrix, or a row that holds a column twice");
    if (!dcols) {
        for (int64_t r = 0; r < nruns; ++r) {
            const int64_t sum = (int64_t)run_dptr[r] + run_dptr[r + 1];
            if (sum > 0x7fffffffLL) return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_csr_runs_build: more than 2^31 distinct columns");
            run_dptr[r + 1] = (int32_t)sum;
        }
    } else if (all_real) *all_real = cplx ? 0 : 1;
    return IG_OK;
}

// Y = beta*Y + alpha * A' * X[xrows, :] like ig_ccsrmm_xrows, for 64 columns, through the run format of A' (ig_csr_runs_build on the
// compact column indices): the panel's touched rows are repacked row-major, then k_csrmm_runs64r.
int ig_ccsrmm_xrows_runs(ig_ctx* ctx, int64_t M, int64_t K, int64_t nnz, float ar, float ai, const int32_t* rowptr,
                         const int32_t* run_dptr, const uint32_t* dcols, const void* entries, int all_real, const int32_t* run_order,
                         const void* X, int64_t ldx, float br, float bi, void* Y, int64_t ldy, const int32_t* xrows, int64_t nxrows) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_ccsrmm_xrows_runs: ctx is NULL");
    IG_REQUIRE(ctx, M >= 0 && K >= 0 && nnz >= 0 && nnz <= 0x7fffffffLL && M <= 0x7fffffffLL, "ig_ccsrmm_xrows_runs: bad dimensions");
    IG_REQUIRE(ctx, rowptr && run_dptr && (nnz == 0 || (dcols && entries)) && (M == 0 || Y) && ldy >= M && ldx >= K, "ig_ccsrmm_xrows_runs: NULL array or short leading dimension");
    IG_REQUIRE(ctx, nxrows >= 1 && nxrows <= K && xrows && X && nxrows * 512 < 0xffffffffLL, "ig_ccsrmm_xrows_runs: bad row list (1 .. 2^23 - 1 rows: the repacked panel is addressed with 32 bits)");
    IG_REQUIRE(ctx, nnz * 12 < 0x7fffffffLL, "ig_ccsrmm_xrows_runs: the entries exceed the 2 GB window of a buffer descriptor");
    if (M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const size_t need = (size_t)nxrows * 512;
    if (ctx->xpack_bytes < need) {
        if (ctx->d_xpack) { IG_HIP(ctx, hipStreamSynchronize(ctx->stream)); IG_HIP(ctx, hipFree(ctx->d_xpack)); ctx->d_xpack = nullptr; ctx->xpack_bytes = 0; }
        IG_HIP(ctx, hipMalloc((void**)&ctx->d_xpack, need));
        ctx->xpack_bytes = need;
    }
    float2* xp = (float2*)ctx->d_xpack;
    {
        ig_prof_scope prof(ctx, "pack_panel", (double)nxrows * 512.0 * 2.0);
        int64_t gt = (nxrows + 63) / 64;
        const int64_t cap = (int64_t)ctx->num_cu * 16;
        if (gt > cap) gt = cap;
        hipLaunchKernelGGL(k_pack_panel_tiled<64>, dim3((unsigned)gt), dim3(BLK), 0, ctx->stream, nxrows, (int64_t)64, (const float2*)X, ldx, xp, xrows);
        IG_LAUNCH_CHECK(ctx, "k_pack_panel_tiled");
    }
    const float2 alpha = make_float2(ar, ai), beta = make_float2(br, bi);
    const bool b0 = br == 0.f && bi == 0.f;
    const unsigned blocks = (unsigned)((M + 63) / 64);
    ig_prof_scope prof(ctx, "csrmm_runs");
#define IG_RUNS(BM_, RW_) hipLaunchKernelGGL((k_csrmm_runs64r<BM_, RW_>), dim3(blocks), dim3(256), 0, ctx->stream, M, rowptr, run_dptr, dcols, \
                                             (const RunEntry*)entries, run_order, (const float2*)xp, (float2*)Y, ldy, alpha, beta)
    if (b0) { if (all_real) IG_RUNS(0, true); else IG_RUNS(0, false); }
    else    { if (all_real) IG_RUNS(1, true); else IG_RUNS(1, false); }
#undef IG_RUNS
    IG_LAUNCH_CHECK(ctx, "k_csrmm_runs64r");
    return IG_OK;
}

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
