// Gridding straight from the SEPARABLE form of the gridding matrix (round 6): the taps of a sample are computed, not streamed.
//
// Reference: Backend.Interp / _interp3_mat (indigo/interp.py:18-60) builds w = wz * wy * wx per tap and stores every tap; the
// -O3 recipe folds the centred transform's modulation and scale into the stored values (examples/pics.py:104-177), and the products
// with that matrix run through Backend.ccsrmm (indigo/backends/backend.py:514-519, 569-585).  Here the matrix is one 64-byte record
// per sample (ig_interp3_sep, ig_interp.hip): first tap and tap count per axis, per-axis float32 weights with the modulation's
// sign folded in.  The two products of the fused SENSE leaf over a coil-interleaved grid panel (grid layout 2: element
// (coil, kx, km, ks) at coil + NC * (kx + n0 * (km + nm * ks))):
//
//   forward   Y[t, :]  = alpha * sum_taps w(t, tap) * X[cell(t, tap), :] + beta * Y[t, :]          k_grid_gather_sep
//   adjoint   Yg[cell, :] = alpha * sum_{t, tap -> cell} w(t, tap) * X[t, :]   (flagged segments)  k_grid_scatter_sep
//
// Both are HBM- and issue-bound gather / scatter work on 64-byte grid rows: no MFMA.
#include "ig_common.h"
#include <vector>
#include <thread>
#include <atomic>
#include <algorithm>
#include <cstring>

namespace {

constexpr int BLK = 256;
constexpr int WPB = BLK / 64;

__device__ __forceinline__ int64_t xcd_block(int64_t b, int64_t nb) {          // every XCD a contiguous range of blocks
    const int64_t q = nb >> 3, rem = nb & 7;
    const int64_t xcd = b & 7, idx = b >> 3;
    return (xcd < rem) ? xcd * (q + 1) + idx : rem * (q + 1) + (xcd - rem) * q + idx;
}

__host__ __device__ constexpr int sep_words(int tw) { return tw == 4 ? 16 : 32; }

// ---- forward -------------------------------------------------------------------------------------------------------------------
// A sample is worked on by LPS = XL x QL lanes: lane (i, q) owns x tap i (XL = 4 lanes for tw = 4, 8 for tw = 6 / 8) and the
// 16-byte piece q of a grid point's NC coils (QL = NC / 2), and walks the sample's (middle, slow) tap rows: per row ONE 16-byte
// load -- the XL x QL lanes of a sample read one contiguous run of XL grid points -- and four multiply-adds.  No index or value
// loads, no row pointers: a record's words are fetched by the lanes that need them (the 16 lanes of a sample hit one 64-byte
// line).  64 / LPS samples per wave; partial sums are folded across the x-tap lanes by shuffles.
// The loads of a group of CG slow-axis rows are all issued before the first multiply-add (no branch, no wait in between): a lane
// whose sample has fewer taps than the wave's widest sample re-reads its own last tap with weight zero -- a cell its sample
// touches anyway, so nothing foreign (an unflagged, never-written grid row) is ever read.
// BMODE: 0 => beta == 0 (Y not read), 1 => general beta.
template <int NC, int TW, int NBT, int CG>
__device__ __forceinline__ void gather_rows(float4& acc, const float4* __restrict__ X4, const uint32_t (&offm)[TW], const float (&w1)[TW],
                                            const float (&w2)[TW], float w0, int j2, int c2, int ns, uint32_t plane, int nc2u) {
#pragma unroll
    for (int c0 = 0; c0 < TW; c0 += CG) {
        if (c0 >= nc2u) break;                                            // (wave-uniform)
        float4 v[CG][NBT];
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            int cc = c0 + g;
            cc = cc < c2 ? cc : c2 - 1;                                   // (c2 >= 1 for every lane that reaches here)
            int js = j2 + cc;
            if (js >= ns) js -= ns;
            const uint32_t offs = (uint32_t)js * plane;
#pragma unroll
            for (int b = 0; b < NBT; ++b) v[g][b] = X4[(size_t)(offs + offm[b])];
        }
#pragma unroll
        for (int g = 0; g < CG; ++g) {
            const float ws = (c0 + g < c2) ? w0 * w2[c0 + g < TW ? c0 + g : 0] : 0.f;
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                const float w = ws * w1[b];
                acc.x = fmaf(w, v[g][b].x, acc.x); acc.y = fmaf(w, v[g][b].y, acc.y);
                acc.z = fmaf(w, v[g][b].z, acc.z); acc.w = fmaf(w, v[g][b].w, acc.w);
            }
        }
    }
}

template <int NC, int TW, int BMODE>
__global__ void __launch_bounds__(BLK)
k_grid_gather_sep(int64_t M, const uint32_t* __restrict__ rec, const float4* __restrict__ X4, int n0, int nm, int ns,
                  float2* __restrict__ Y, int64_t ldy, float2 alpha, float2 beta) {
    constexpr int QL = NC / 2, XL = TW <= 4 ? 4 : 8, LPS = QL * XL, SPW = 64 / LPS, RW = sep_words(TW);
    constexpr int CG = TW <= 4 ? 4 : 2;                                   // slow-axis rows whose loads are in flight together
    static_assert(LPS <= 64 && NC >= 2, "a sample's lanes fit a wave");
    const int lane = threadIdx.x & 63;
    const int q = lane % QL, i = (lane / QL) % XL, g = lane / LPS;
    const int64_t blk = xcd_block(blockIdx.x, gridDim.x);
    const int64_t wave = blk * WPB + (threadIdx.x >> 6);
    const int64_t t = wave * SPW + g;
    const bool ok = t < M;
    const uint32_t* __restrict__ r = rec + (size_t)(ok ? t : M - 1) * RW;
    const uint32_t h0 = r[3 * TW], h1 = r[3 * TW + 1];
    const int j0 = (int)(h0 & 0xffffu), j1 = (int)(h0 >> 16), j2 = (int)(h1 & 0xffffu);
    const int c0 = (int)((h1 >> 16) & 15u), c1 = (int)((h1 >> 20) & 15u), c2 = (int)((h1 >> 24) & 15u);
    // (a record's counts are >= 1: 2 width - 1 taps at least; lanes past the last sample work on the last record and store nothing)
    const int iv = i < c0 ? i : c0 - 1;                                   // x tap of this lane, or the sample's last one with weight 0
    const float w0 = (i < c0 && ok) ? __uint_as_float(r[iv]) : 0.f;
    float w1[TW], w2[TW];
#pragma unroll
    for (int b = 0; b < TW; ++b) { w1[b] = b < c1 ? __uint_as_float(r[TW + b]) : 0.f; w2[b] = __uint_as_float(r[2 * TW + b]); }
    int jx = j0 + iv;
    if (jx >= n0) jx -= n0;
    // float4 index of (tap b on the middle axis -- the sample's last one for b past its count --, this lane's x tap and coil piece)
    uint32_t offm[TW];
#pragma unroll
    for (int b = 0; b < TW; ++b) {
        int jm = j1 + (b < c1 ? b : c1 - 1);
        if (jm >= nm) jm -= nm;
        offm[b] = ((uint32_t)jm * (uint32_t)n0 + (uint32_t)jx) * QL + q;
    }
    const uint32_t plane = (uint32_t)nm * (uint32_t)n0 * QL;             // float4 per slow-axis step (the grid panel is < 2^32 float4)
    // the wave's widest sample: taps on the slow axis, and whether any sample has all TW taps on the middle axis
    int nc2u = 1;
#pragma unroll
    for (int v = TW; v > 1; --v)
        if (__builtin_amdgcn_ballot_w64(c2 >= v)) { nc2u = v; break; }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (__builtin_amdgcn_ballot_w64(c1 >= TW)) gather_rows<NC, TW, TW, CG>(acc, X4, offm, w1, w2, w0, j2, c2, ns, plane, nc2u);
    else                                       gather_rows<NC, TW, TW - 1, CG>(acc, X4, offm, w1, w2, w0, j2, c2, ns, plane, nc2u);
#pragma unroll
    for (int off = QL; off < LPS; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (i == 0 && ok) {
        float2* y0 = Y + (int64_t)(2 * q) * ldy + t;
        float2* y1 = y0 + ldy;
        float2 o0 = cmul(alpha, make_float2(acc.x, acc.y)), o1 = cmul(alpha, make_float2(acc.z, acc.w));
        if (BMODE == 1) { cfma(o0, beta, *y0); cfma(o1, beta, *y1); }
        *y0 = o0;
        *y1 = o1;
    }
}

// ---- adjoint -------------------------------------------------------------------------------------------------------------------
// The SCATTER, race-free by binning as in k_grid_bricks (ig_spmm.hip) -- a wave owns a run of grid bricks of 16 x BM x BS cells, keeps
// ONE brick image in LDS, accumulates with plain read-add-write and stores the image's flagged segments at each brick boundary --
// but what is binned are SHARES, not taps: a share = (sample, brick) for every brick the sample's footprint meets, 8 bytes:
//   word 0   sample
//   word 1   ox + 8 | (om + 8) << 5 | (os + 8) << 10 | blo << 15 | bhi << 18 | clo << 22 | chi << 25
//            tap (a, b, c) of the sample sits at brick cell (ox + a, om + b, os + c); the taps b in [blo, bhi), c in [clo, chi) and
//            those a with 0 <= ox + a < 16 are the ones inside this brick
// and the taps are computed from the sample's record.  A lane of the accumulation is (q, i, bl): 16-byte piece q of the NC coils,
// x tap i, middle-axis tap blo + bl (+ BL per sub-round); a round handles one slow-axis tap c: LDS read of 16 bytes, four
// multiply-adds, LDS write.  Record and k-space panel row of a share arrive as ONE load -- lane L < RW: record word L, lanes RW ..
// RW + 2 NC: the panel row's words -- issued a group of four shares ahead, and are handed to the lanes that need them by ds_bpermute
// (x tap weight, middle-axis weight, the four floats of the lane's coil pair) and v_readlane (slow-axis weight: wave-uniform).
// Against the stored-tap format: 8 bytes per share + 64 per record instead of 8 .. 12 bytes per tap padded to rounds (1.6 x);
// bricks of 256 cells instead of 64 (2.5 shares per sample instead of 4.5 panel-row fetches).
struct ShareTask { int32_t lo, hi, bt, nb_flags; };            // shares [lo, hi) = bricks table[bt .. bt + (nb_flags & 0xffff)); bit 16: shared
struct ShareBrick { int32_t brick, end; uint32_t mask_lo, mask_hi; };   // a non-empty brick, where its shares end, its flagged segments

typedef float v4f_t __attribute__((ext_vector_type(4)));

template <int NC, int TW>
__global__ void __launch_bounds__(BLK)
k_grid_scatter_sep(const ShareTask* __restrict__ tasks, int ntasks, const ShareBrick* __restrict__ btab, const uint2* __restrict__ shares,
                   const uint32_t* __restrict__ rec, const uint32_t* __restrict__ xp /* packed panel rows [t][NC] as words */,
                   float4* __restrict__ Y4, float2 alpha, int n0, int nm, int bm_log2, int bs_log2, int nbx, int nbm, int st_log2) {
    extern __shared__ float4 img_all[];                    // per wave: [cell][QL]
    constexpr int QL = NC / 2, XL = TW <= 4 ? 4 : 8, BL = 64 / (QL * XL), RW = sep_words(TW), XW = 2 * NC;
    static_assert(RW + XW <= 64, "record and panel row fit one lane-word load");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int task = blockIdx.x * WPB + wv;
    if (task >= ntasks) return;                            // (no workgroup barrier below: waves are independent)
    const ShareTask tk = tasks[task];
    const int nb = tk.nb_flags & 0xffff;
    const bool shared = (tk.nb_flags >> 16) & 1;
    const int nsh = tk.hi - tk.lo;
    const int BM = 1 << bm_log2, nrows = 1 << (bm_log2 + bs_log2), ncell = 16 * nrows;
    float4* __restrict__ img = img_all + (size_t)wv * ncell * QL;
    const int q = lane % QL, i = (lane / QL) % XL, bl = lane / (QL * XL);

    // the run's bricks, one per lane
    int my_end = 0x7fffffff;
    uint32_t my_mlo = 0xffffffffu, my_mhi = 0xffffffffu;
    int64_t my_pt = 0;
    if (lane < nb) {
        const ShareBrick br = btab[tk.bt + lane];
        if (!shared) my_end = br.end - tk.lo;
        my_mlo = br.mask_lo; my_mhi = br.mask_hi;
        const int bx = br.brick % nbx, bmi = (br.brick / nbx) % nbm, bsi = br.brick / (nbx * nbm);
        my_pt = (int64_t)bx * 16 + (int64_t)n0 * (((int64_t)bmi << bm_log2) + (int64_t)nm * ((int64_t)bsi << bs_log2));
    }
    for (int e = lane; e < ncell * QL; e += 64) img[e] = make_float4(0.f, 0.f, 0.f, 0.f);

    // lane-word source of a share's record + panel row: lane L < RW: record word L; RW <= L < RW + XW: panel word L - RW
    const bool is_rec = lane < RW;
    const uint32_t* lane_base = is_rec ? rec + lane : xp + (lane < RW + XW ? lane - RW : XW - 1);
    const uint32_t lane_stride = is_rec ? (uint32_t)RW : (uint32_t)XW;
    const int xs_log2 = 4 - st_log2;
    // flush roles: lane -> (row of the pass, x cell, coil piece)
    constexpr int LR = 16 * QL, RP = 64 / LR;              // lanes per brick row, brick rows per pass
    const int f_row = lane / LR, f_xq = lane % LR, f_x = f_xq / QL;

    int cur = 0;
    int cur_end = __builtin_amdgcn_readlane(my_end, 0);
    auto flush = [&]() __attribute__((always_inline)) {
        const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)my_mlo, cur), mhi = (uint32_t)__builtin_amdgcn_readlane((int)my_mhi, cur);
        const uint64_t mask = ((uint64_t)mhi << 32) | mlo;
        const int64_t pt = ((int64_t)__builtin_amdgcn_readlane((int)(my_pt >> 32), cur) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_pt, cur);
        for (int row0 = 0; row0 < nrows; row0 += RP) {
            const int row = row0 + f_row;
            const int seg = (f_x >> st_log2) + (row << xs_log2);
            const bool mine = (mask >> seg) & 1ull;
            float4* src = img + row * LR + f_xq;
            const float4 v = *src;
            *src = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mine) {
                const float2 o0 = cmul(alpha, make_float2(v.x, v.y)), o1 = cmul(alpha, make_float2(v.z, v.w));
                const int im = row & (BM - 1), is = row >> bm_log2;
                float4* dst = Y4 + (pt + (int64_t)n0 * (im + (int64_t)nm * is)) * QL + f_xq;
                // (stores and atomics as asm statements: the compiler's wait-count bookkeeping does not see them and so does not
                // drain the prefetched records in front of every share group; nothing here reads Y back)
                if (shared) {
                    asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %2, off offset:4\n\t"
                                 "global_atomic_add_f32 %0, %3, off offset:8\n\tglobal_atomic_add_f32 %0, %4, off offset:12"
                                 :: "v"(dst), "v"(o0.x), "v"(o0.y), "v"(o1.x), "v"(o1.y) : "memory");
                } else {
                    const v4f_t o = {o0.x, o0.y, o1.x, o1.y};
                    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(o) : "memory");
                }
            }
        }
        ++cur;
        cur_end = __builtin_amdgcn_readlane(my_end, cur & 63);
    };

    constexpr int G = 4;                                   // shares whose record + panel loads are in flight together
    for (int base = 0; base < nsh; base += 64) {
        const int nbatch = nsh - base < 64 ? nsh - base : 64;
        uint2 sh = make_uint2(0u, 0u);
        if (lane < nbatch) sh = shares[(size_t)tk.lo + base + lane];
        auto request = [&](uint32_t (&w)[G], int s0) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int s = s0 + k < nbatch ? s0 + k : nbatch - 1;
                const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)sh.x, s);
                w[k] = lane_base[(size_t)t * lane_stride];
            }
        };
        auto process = [&](const uint32_t (&w)[G], int s0) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int s = s0 + k;
                if (s >= nbatch) break;                    // (wave-uniform)
                if (base + s >= cur_end) flush();          // (every brick of the table holds at least one share)
                const uint32_t geo = (uint32_t)__builtin_amdgcn_readlane((int)sh.y, s);
                const int ox = (int)(geo & 31u) - 8, om = (int)((geo >> 5) & 31u) - 8, os = (int)((geo >> 10) & 31u) - 8;
                const int blo = (int)((geo >> 15) & 7u), bhi = (int)((geo >> 18) & 15u), clo = (int)((geo >> 22) & 7u), chi = (int)((geo >> 25) & 15u);
                const int wi = (int)w[k];
                const float w0 = __int_as_float(__builtin_amdgcn_ds_bpermute(i * 4, wi));
                float4 xv;
                xv.x = __int_as_float(__builtin_amdgcn_ds_bpermute((RW + 4 * q) * 4, wi));
                xv.y = __int_as_float(__builtin_amdgcn_ds_bpermute((RW + 4 * q + 1) * 4, wi));
                xv.z = __int_as_float(__builtin_amdgcn_ds_bpermute((RW + 4 * q + 2) * 4, wi));
                xv.w = __int_as_float(__builtin_amdgcn_ds_bpermute((RW + 4 * q + 3) * 4, wi));
                const int cx = ox + i;
                const bool vx = (unsigned)cx < 16u && i < TW;
                for (int b0 = blo; b0 < bhi; b0 += BL) {
                    const int b = b0 + bl;
                    const float w01 = w0 * __int_as_float(__builtin_amdgcn_ds_bpermute((TW + (b < TW ? b : 0)) * 4, wi));
                    if (vx && b < bhi) {
                        const int cell0 = cx + 16 * (om + b);
                        for (int c = clo; c < chi; ++c) {
                            const float w = w01 * __int_as_float(__builtin_amdgcn_readlane(wi, 2 * TW + c));
                            float4* a = img + (cell0 + (16 << bm_log2) * (os + c)) * QL + q;
                            float4 v = *a;                 // plain read-add-write: the lanes of a round hold distinct cells of ONE sample,
                            v.x = fmaf(w, xv.x, v.x); v.y = fmaf(w, xv.y, v.y);       // the image is this wave's, and a wave's LDS
                            v.z = fmaf(w, xv.z, v.z); v.w = fmaf(w, xv.w, v.w);       // operations execute in order
                            *a = v;
                        }
                    }
                }
            }
        };
        uint32_t wa[G], wb[G];
        request(wa, 0);
        for (int s0 = 0; s0 < nbatch; s0 += 2 * G) {
            request(wb, s0 + G);
            process(wa, s0);
            request(wa, s0 + 2 * G);
            process(wb, s0 + G);
        }
    }
    flush();
}

// zero the flagged segments of the bricks that several tasks add into
template <int NC>
__global__ void __launch_bounds__(BLK)
k_grid_sep_zero(const ShareBrick* __restrict__ bricks, float4* __restrict__ Y4, int n0, int nm, int bm_log2, int bs_log2, int nbx, int nbm, int st_log2) {
    constexpr int QL = NC / 2, LR = 16 * QL;
    const ShareBrick br = bricks[blockIdx.x];
    const uint64_t mask = ((uint64_t)br.mask_hi << 32) | br.mask_lo;
    const int bx = br.brick % nbx, bmi = (br.brick / nbx) % nbm, bsi = br.brick / (nbx * nbm);
    const int64_t pt = (int64_t)bx * 16 + (int64_t)n0 * (((int64_t)bmi << bm_log2) + (int64_t)nm * ((int64_t)bsi << bs_log2));
    const int nrows = 1 << (bm_log2 + bs_log2), xs_log2 = 4 - st_log2;
    for (int e = threadIdx.x; e < nrows * LR; e += BLK) {
        const int row = e / LR, xq = e % LR, x = xq / QL;
        if ((mask >> ((x >> st_log2) + (row << xs_log2))) & 1ull) {
            const int im = row & ((1 << bm_log2) - 1), is = row >> bm_log2;
            Y4[(pt + (int64_t)n0 * (im + (int64_t)nm * is)) * QL + xq] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// X (rows x NC, column-major, ld) -> packed rows [t][NC]
template <int NC>
__global__ void __launch_bounds__(BLK)
k_sep_pack_panel(int64_t rows, const float2* __restrict__ X, int64_t ld, float2* __restrict__ Xp) {
    for (int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x; e < rows * NC; e += (int64_t)gridDim.x * BLK)
        Xp[e] = X[(e % NC) * ld + e / NC];
}

// ---- host: shares of the samples by grid brick -----------------------------------------------------------------------------------
struct ShareGeom { int64_t n[3]; int bdim[3]; int64_t nbr[3]; int tw, rw; };

// the pieces of one axis: taps [lo, hi) of a sample whose first tap is j fall into brick `brick` at offset `o` + tap
struct AxisPiece { int brick, o, lo, hi; };
inline int axis_pieces(int j, int cnt, int64_t n, int bdim, AxisPiece* out) {
    int np = 0;
    int a = 0;
    while (a < cnt) {
        int64_t cell = j + a;
        if (cell >= n) cell -= n;
        const int brick = (int)(cell / bdim), off = (int)(cell % bdim);
        int run = bdim - off;                                              // taps until the brick (or the grid) ends
        if (run > cnt - a) run = cnt - a;
        if (cell + run > n) run = (int)(n - cell);
        out[np++] = AxisPiece{brick, off - a, a, a + run};
        a += run;
    }
    return np;
}

template <class F>
inline void for_shares(const ShareGeom& g, const uint32_t* r, F&& f) {
    const uint32_t h0 = r[3 * g.tw], h1 = r[3 * g.tw + 1];
    const int j[3] = {(int)(h0 & 0xffffu), (int)(h0 >> 16), (int)(h1 & 0xffffu)};
    const int cnt[3] = {(int)((h1 >> 16) & 15u), (int)((h1 >> 20) & 15u), (int)((h1 >> 24) & 15u)};
    AxisPiece px[9], pm[9], ps[9];
    const int nx = axis_pieces(j[0], cnt[0], g.n[0], g.bdim[0], px), nmm = axis_pieces(j[1], cnt[1], g.n[1], g.bdim[1], pm),
              nss = axis_pieces(j[2], cnt[2], g.n[2], g.bdim[2], ps);
    for (int c = 0; c < nss; ++c)
        for (int b = 0; b < nmm; ++b)
            for (int a = 0; a < nx; ++a) {
                const int64_t brick = px[a].brick + g.nbr[0] * (pm[b].brick + g.nbr[1] * (int64_t)ps[c].brick);
                const uint32_t geo = (uint32_t)(px[a].o + 8) | ((uint32_t)(pm[b].o + 8) << 5) | ((uint32_t)(ps[c].o + 8) << 10) |
                                     ((uint32_t)pm[b].lo << 15) | ((uint32_t)pm[b].hi << 18) | ((uint32_t)ps[c].lo << 22) | ((uint32_t)ps[c].hi << 25);
                f(brick, geo);
            }
}

inline int share_threads(int64_t M) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? hw : 4);
    if (nt > 16) nt = 16;
    if (M < 16384) nt = 1;
    return nt;
}

template <class F>
inline void run_threads(int nt, F&& body) {
    if (nt == 1) { body(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([t, &body]() { body(t); });
    for (auto& x : th) x.join();
}

inline bool share_geom(ShareGeom& g, int tw, int64_t n0, int64_t nm, int64_t ns, int bm, int bs) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (!(tw == 4 || tw == 6 || tw == 8) || n0 < 16 || n0 % 16 || nm < tw || ns < tw || !pow2(bm) || !pow2(bs) || bm > 16 || bs > 16 || nm % bm || ns % bs ||
        n0 > 65535 || nm > 65535 || ns > 65535)
        return false;
    g.n[0] = n0; g.n[1] = nm; g.n[2] = ns;
    g.bdim[0] = 16; g.bdim[1] = bm; g.bdim[2] = bs;
    g.nbr[0] = n0 / 16; g.nbr[1] = nm / bm; g.nbr[2] = ns / bs;
    g.tw = tw; g.rw = sep_words(tw);
    return g.nbr[0] * g.nbr[1] * g.nbr[2] < 0x7fffffffLL;
}

}  // namespace

extern "C" {

// Y (M x NC, column-major, ldy) = alpha * G X + beta * Y for the gridding matrix given by `records` (ig_interp3_sep with grid_order
// matching the panel: axes (n0, nm, ns) in memory order) and the coil-interleaved grid panel X (n0 * nm * ns rows of NC values).
int ig_grid_gather_sep(ig_ctx* ctx, int64_t M, int64_t NC, int tw, const void* records, const void* X_il, int64_t n0, int64_t nm, int64_t ns,
                       float ar, float ai, float br, float bi, void* Y, int64_t ldy) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_grid_gather_sep: ctx is NULL");
    IG_REQUIRE(ctx, M >= 0 && (NC == 2 || NC == 4 || NC == 8) && (tw == 4 || tw == 6 || tw == 8), "ig_grid_gather_sep: 2, 4 or 8 interleaved coils; tw 4, 6 or 8");
    IG_REQUIRE(ctx, n0 >= tw && nm >= tw && ns >= tw && n0 <= 65535 && nm <= 65535 && ns <= 65535 && n0 * nm * ns * (NC / 2) < (1LL << 32),
               "ig_grid_gather_sep: grid %lld x %lld x %lld x %lld coils out of range", (long long)n0, (long long)nm, (long long)ns, (long long)NC);
    IG_REQUIRE(ctx, M == 0 || (records && X_il && Y && ldy >= M), "ig_grid_gather_sep: NULL pointer or ldy < M");
    IG_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(X_il) & 15u) == 0 && (reinterpret_cast<uintptr_t>(records) & 15u) == 0, "ig_grid_gather_sep: 16-byte aligned panel and records");
    if (M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const float2 alpha = make_float2(ar, ai), beta = make_float2(br, bi);
    const bool b0 = br == 0.f && bi == 0.f;
    ig_prof_scope prof(ctx, "grid_gather_sep");
#define IG_GS(NC_, TW_) do {                                                                                                     \
        constexpr int spw = 64 / ((NC_ / 2) * (TW_ <= 4 ? 4 : 8));                                                               \
        const int64_t blocks = ((M + spw - 1) / spw + WPB - 1) / WPB;                                                            \
        IG_REQUIRE(ctx, blocks <= 0x7fffffffLL, "ig_grid_gather_sep: too many samples for one launch");                          \
        if (b0) hipLaunchKernelGGL((k_grid_gather_sep<NC_, TW_, 0>), dim3((unsigned)blocks), dim3(BLK), 0, ctx->stream, M,       \
                                   (const uint32_t*)records, (const float4*)X_il, (int)n0, (int)nm, (int)ns, (float2*)Y, ldy, alpha, beta); \
        else    hipLaunchKernelGGL((k_grid_gather_sep<NC_, TW_, 1>), dim3((unsigned)blocks), dim3(BLK), 0, ctx->stream, M,       \
                                   (const uint32_t*)records, (const float4*)X_il, (int)n0, (int)nm, (int)ns, (float2*)Y, ldy, alpha, beta); \
    } while (0)
#define IG_GS_TW(NC_) do { if (tw == 4) IG_GS(NC_, 4); else if (tw == 6) IG_GS(NC_, 6); else IG_GS(NC_, 8); } while (0)
    if (NC == 8) IG_GS_TW(8);
    else if (NC == 4) IG_GS_TW(4);
    else IG_GS_TW(2);
#undef IG_GS_TW
#undef IG_GS
    IG_LAUNCH_CHECK(ctx, "k_grid_gather_sep");
    return IG_OK;
}


// Host: the shares of every sample by grid brick (16 x bm x bs cells; records from ig_interp3_sep on the same grid axes).  Two passes as
// for the stored-tap bricks (ig_grid_bricks_count / _fill): counts per brick, then -- at the caller's exclusive prefix sums -- the
// 8-byte shares in brick order, sample order inside a brick (a few host threads own contiguous sample ranges).
int ig_grid_shares_count(int64_t M, const uint32_t* records, int tw, int64_t n0, int64_t nm, int64_t ns, int bm, int bs, int32_t* brick_shares) {
    ShareGeom g;
    if (M < 0 || (M > 0 && !records) || !brick_shares || !share_geom(g, tw, n0, nm, ns, bm, bs))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_shares_count: the grid must divide into 16 x bm x bs bricks (powers of two <= 16); tw 4, 6 or 8");
    const int64_t nb = g.nbr[0] * g.nbr[1] * g.nbr[2];
    const int nt = share_threads(M);
    const int64_t per = (M + nt - 1) / nt;
    std::vector<std::vector<int32_t>> cnt((size_t)nt);
    std::atomic<int> bad{0};
    run_threads(nt, [&](int t) {
        cnt[t].assign((size_t)nb, 0);
        const int64_t lo = std::min<int64_t>(M, t * per), hi = std::min<int64_t>(M, lo + per);
        for (int64_t s = lo; s < hi; ++s)
            for_shares(g, records + (size_t)s * g.rw, [&](int64_t brick, uint32_t) { if (++cnt[t][brick] < 0) bad = 1; });
    });
    run_threads(nt, [&](int t) {
        const int64_t pb = (nb + nt - 1) / nt, lo = std::min<int64_t>(nb, t * pb), hi = std::min<int64_t>(nb, lo + pb);
        for (int64_t b = lo; b < hi; ++b) {
            int64_t sum = 0;
            for (int u = 0; u < nt; ++u) sum += cnt[u][b];
            if (sum > 0x7fffffffLL) { bad = 1; sum = 0; }
            brick_shares[b] = (int32_t)sum;
        }
    });
    if (bad) return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_shares_count: a brick exceeds 2^31 shares");
    return IG_OK;
}

int ig_grid_shares_fill(int64_t M, const uint32_t* records, int tw, int64_t n0, int64_t nm, int64_t ns, int bm, int bs,
                        const int64_t* brick_ptr /* exclusive prefix sums of the counts, nbricks + 1 */, uint32_t* shares /* 2 words each */) {
    ShareGeom g;
    if (M < 0 || (M > 0 && (!records || !shares)) || !brick_ptr || !share_geom(g, tw, n0, nm, ns, bm, bs))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_shares_fill: bad arguments");
    const int64_t nb = g.nbr[0] * g.nbr[1] * g.nbr[2];
    const int nt = share_threads(M);
    const int64_t per = (M + nt - 1) / nt;
    std::vector<std::vector<int32_t>> cur((size_t)nt);
    run_threads(nt, [&](int t) {
        cur[t].assign((size_t)nb, 0);
        const int64_t lo = std::min<int64_t>(M, t * per), hi = std::min<int64_t>(M, lo + per);
        for (int64_t s = lo; s < hi; ++s)
            for_shares(g, records + (size_t)s * g.rw, [&](int64_t brick, uint32_t) { ++cur[t][brick]; });
    });
    std::atomic<int> mismatch{0};
    run_threads(nt, [&](int t) {                   // per-thread counts -> per-thread cursors
        const int64_t pb = (nb + nt - 1) / nt, lo = std::min<int64_t>(nb, t * pb), hi = std::min<int64_t>(nb, lo + pb);
        for (int64_t b = lo; b < hi; ++b) {
            int64_t run = 0;
            for (int u = 0; u < nt; ++u) { const int32_t c = cur[u][b]; cur[u][b] = (int32_t)run; run += c; }
            if (run != brick_ptr[b + 1] - brick_ptr[b]) mismatch = 1;
        }
    });
    if (mismatch) return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_shares_fill: brick_ptr does not come from ig_grid_shares_count");
    run_threads(nt, [&](int t) {
        int32_t* cursor = cur[t].data();
        const int64_t lo = std::min<int64_t>(M, t * per), hi = std::min<int64_t>(M, lo + per);
        for (int64_t s = lo; s < hi; ++s)
            for_shares(g, records + (size_t)s * g.rw, [&](int64_t brick, uint32_t geo) {
                uint32_t* o = shares + 2 * (size_t)(brick_ptr[brick] + cursor[brick]++);
                o[0] = (uint32_t)s; o[1] = geo;
            });
    });
    return IG_OK;
}

// Y_il (n0 * nm * ns grid points x NC interleaved coils) = alpha * G^H * X over the flagged segments of the bricks that hold a share;
// X (M x NC, column-major, ldx).  tasks / brick_table as for ig_ccsrmm_t_bricks with shares in place of entries and 16-byte table rows
// {brick, end of its shares, flagged segments: 64 bits, bit xs + (16 / support_tile) * (im + bm * is)}; shared_table: the rows of the
// bricks several tasks add into (zeroed first; those tasks add with float atomics).
int ig_grid_scatter_sep(ig_ctx* ctx, int64_t M, int64_t NC, int tw, const void* records, const void* shares, const void* X, int64_t ldx,
                        void* Y_il, int64_t n0, int64_t nm, int64_t ns, int bm, int bs, const int32_t* tasks, int64_t ntasks,
                        const int32_t* brick_table, const int32_t* shared_table, int64_t nshared, int support_tile, float ar, float ai) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_grid_scatter_sep: ctx is NULL");
    ShareGeom g;
    IG_REQUIRE(ctx, M >= 0 && M <= 0x7fffffffLL && (NC == 2 || NC == 4 || NC == 8) && share_geom(g, tw, n0, nm, ns, bm, bs),
               "ig_grid_scatter_sep: 2, 4 or 8 interleaved coils; a grid that divides into 16 x bm x bs bricks; tw 4, 6 or 8");
    IG_REQUIRE(ctx, support_tile == 16 || support_tile == 8 || support_tile == 4, "ig_grid_scatter_sep: support_tile 16, 8 or 4");
    IG_REQUIRE(ctx, (16 / support_tile) * bm * bs <= 64, "ig_grid_scatter_sep: at most 64 segments per brick");
    IG_REQUIRE(ctx, ntasks >= 0 && ntasks <= 0x7fffffffLL && (ntasks == 0 || (tasks && brick_table && records && shares && X && Y_il)) && ldx >= M &&
               nshared >= 0 && (nshared == 0 || shared_table), "ig_grid_scatter_sep: bad task list or NULL array");
    IG_REQUIRE(ctx, n0 * nm * ns * (NC / 2) < (1LL << 32), "ig_grid_scatter_sep: grid panel too large");
    if (ntasks == 0 || M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const float2 alpha = make_float2(ar, ai);
    const size_t need = (size_t)M * NC * 8;
    if (ctx->xpack_bytes < need) {
        if (ctx->d_xpack) { IG_HIP(ctx, hipStreamSynchronize(ctx->stream)); IG_HIP(ctx, hipFree(ctx->d_xpack)); ctx->d_xpack = nullptr; ctx->xpack_bytes = 0; }
        IG_HIP(ctx, hipMalloc((void**)&ctx->d_xpack, need));
        ctx->xpack_bytes = need;
    }
    float2* xpk = (float2*)ctx->d_xpack;
    int bm_log2 = 0, bs_log2 = 0;
    while ((1 << bm_log2) < bm) ++bm_log2;
    while ((1 << bs_log2) < bs) ++bs_log2;
    const int st_log2 = support_tile == 16 ? 4 : support_tile == 8 ? 3 : 2;
    const int nbx = (int)(n0 / 16), nbm = (int)(nm / bm);
    const size_t lds = (size_t)WPB * 16 * bm * bs * NC * 8;                // one brick image per wave
    IG_REQUIRE(ctx, lds <= 160 * 1024, "ig_grid_scatter_sep: bricks of 16 x %d x %d cells x %lld coils need %zu bytes of LDS per workgroup", bm, bs, (long long)NC, lds);
    const unsigned blocks = (unsigned)((ntasks + WPB - 1) / WPB);
#define IG_SS(NC_, TW_) do {                                                                                                             \
        {   ig_prof_scope prof(ctx, "pack_panel", 2.0 * (double)need);                                                                   \
            int64_t gp = (M * NC_ + BLK - 1) / BLK;                                                                                      \
            const int64_t cap = (int64_t)ctx->num_cu * 16;                                                                               \
            if (gp > cap) gp = cap;                                                                                                      \
            hipLaunchKernelGGL((k_sep_pack_panel<NC_>), dim3((unsigned)gp), dim3(BLK), 0, ctx->stream, M, (const float2*)X, ldx, xpk); } \
        if (nshared) {                                                                                                                   \
            ig_prof_scope prof(ctx, "grid_sep_zero");                                                                                    \
            hipLaunchKernelGGL((k_grid_sep_zero<NC_>), dim3((unsigned)nshared), dim3(BLK), 0, ctx->stream, (const ShareBrick*)shared_table, \
                               (float4*)Y_il, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2); }                                  \
        if (lds > 64 * 1024)                                                                                                             \
            IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_scatter_sep<NC_, TW_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        ig_prof_scope prof(ctx, "grid_scatter_sep");                                                                                     \
        hipLaunchKernelGGL((k_grid_scatter_sep<NC_, TW_>), dim3(blocks), dim3(BLK), lds, ctx->stream, (const ShareTask*)tasks, (int)ntasks, \
                           (const ShareBrick*)brick_table, (const uint2*)shares, (const uint32_t*)records, (const uint32_t*)xpk, (float4*)Y_il, \
                           alpha, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2);                                                \
    } while (0)
#define IG_SS_TW(NC_) do { if (tw == 4) IG_SS(NC_, 4); else if (tw == 6) IG_SS(NC_, 6); else IG_SS(NC_, 8); } while (0)
    if (NC == 8) IG_SS_TW(8);
    else if (NC == 4) IG_SS_TW(4);
    else IG_SS_TW(2);
#undef IG_SS_TW
#undef IG_SS
    IG_LAUNCH_CHECK(ctx, "k_grid_scatter_sep");
    return IG_OK;
}

}  // extern "C"
