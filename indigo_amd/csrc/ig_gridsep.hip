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
//   adjoint   Yg[cell, :] = alpha * sum_{t, tap -> cell} w(t, tap) * X[t, :]   (flagged segments)  k_grid_scatter_mfma
//
// The forward is bound by the vector-memory return path and HBM; the adjoint by instruction issue -- which is why its accumulation runs
// as outer products on the matrix cores (fp32 in, fp32 accumulate).
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
k_grid_gather_sep(int64_t M, const uint32_t* __restrict__ rec, int rs /* words per record (>= sep_words) */, const float4* __restrict__ X4, int n0, int nm, int ns,
                  float2* __restrict__ Y, int64_t ldy, float2 alpha, float2 beta, const uint32_t* __restrict__ order) {
    constexpr int QL = NC / 2, XL = TW <= 4 ? 4 : 8, LPS = QL * XL, SPW = 64 / LPS;
    constexpr int CG = TW <= 4 ? 4 : 2;                                   // slow-axis rows whose loads are in flight together
    static_assert(LPS <= 64 && NC >= 2, "a sample's lanes fit a wave");
    const int lane = threadIdx.x & 63;
    const int q = lane % QL, i = (lane / QL) % XL, g = lane / LPS;
    // Which group of WPB * SPW consecutive samples this workgroup takes: by default in trajectory order; with `order` (a permutation of the
    // groups, sorted by where their samples lie on the grid) neighbouring workgroups -- every XCD a contiguous range of them -- read
    // neighbouring grid rows: a densely sampled trajectory, whose spokes cross the same cells far apart in trajectory order, re-fetches its
    // grid rows 3.3 x from HBM otherwise (profiles/r06_gather_order.txt).  Same samples, same arithmetic, same bits.
    int64_t blk = xcd_block(blockIdx.x, gridDim.x);
    if (order) { const uint32_t o = order[blk]; blk = o < gridDim.x ? (int64_t)o : blk; }
    const int64_t wave = blk * WPB + (threadIdx.x >> 6);
    const int64_t t = wave * SPW + g;
    const bool ok = t < M;
    const uint32_t* __restrict__ r = rec + (size_t)(ok ? t : M - 1) * (uint32_t)rs;
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
// The SCATTER, race-free by binning as in k_grid_bricks (ig_spmm.hip) -- a wave owns a run of grid bricks of 16 x 4 x 4 cells, keeps
// ONE brick image, accumulates into it and stores the image's flagged segments at each brick boundary; bricks of the k-space centre
// are cut into pieces whose waves add with float atomics -- but what is binned are SHARES, not taps: a share = (sample, brick) for
// every brick the sample's footprint meets, 8 bytes:
//   word 0   sample | the brick's slow-axis cells that hold a tap of the share << 28 (bricks of <= 4 slow cells; else 15)
//   word 1   ox + 8 | (om + 8) << 5 | (os + 8) << 10 | blo << 15 | bhi << 18 | clo << 22 | chi << 25
//            tap (a, b, c) of the sample sits at brick cell (ox + a, om + b, os + c); the taps b in [blo, bhi), c in [clo, chi) and
//            those a with 0 <= ox + a < 16 are the ones inside this brick
// and the taps are computed from the sample's record: 8 bytes per share + 64 per record instead of 8 .. 12 bytes per tap padded to
// rounds (1.6 x); bricks of 256 cells instead of 64 (2.5 shares per sample instead of 4.5 panel-row fetches at kernel half-width 2).
//
// The brick image lives in REGISTERS and the accumulation runs as outer products on the MFMA pipe.  The stored-tap scatter spends
// ~130 vector instructions per sample to issue the 27 x 16 multiply-adds a sample is (7 wave instructions' worth): it is bound by
// instruction issue, not by HBM (1.8 TB/s on a densely sampled trajectory, profiles/r06_scatter_forms.txt; a first form of this
// kernel with the image in LDS and vector multiply-adds was slower still and is gone).  A share's contribution to a brick IS a sum of
// outer products: for slow-axis cell g and middle-axis cell beta
//       image[x, beta, g][f] += wx[x - ox] * (wm[beta - om] * ws[g - os] * X[t, f]),        f = 0 .. 2 NC - 1 floats of the NC coils
// which v_mfma_f32_16x16x1_4b_f32 computes for all 16 x, all 4 beta (its four blocks) and all 16 f in ONE instruction of 32 clocks:
// A = wx[x - ox] (lane = x + 16 beta), B = wm[beta - om] ws[g - os] X[t, f] (lane = f + 16 beta), D = the 16 registers of group g.
// A share costs one MFMA per slow-axis cell it touches whatever the number of taps inside it -- 27 (kernel half-width 2) or 125 (the
// reference's default 3) -- plus a handful of LDS reads (x weight, middle weight, panel value, slow weights).  fp32 in, fp32
// accumulate: the arithmetic of the vector kernels.  Record and panel row of a sample are ONE 128-byte line (`recx`: the pack kernel
// writes the panel row behind the record), fetched by one lane-word load per share, two groups of four shares ahead, through a ring
// in LDS; share headers come 64 at a time, two batches deep in LDS.
struct ShareTask { int32_t lo, hi, bt, nb_flags; };            // shares [lo, hi) = bricks table[bt .. bt + (nb_flags & 0xffff)); bit 16: shared
struct ShareBrick { int32_t brick, end; uint32_t mask_lo, mask_hi; };   // a non-empty brick, where its shares end, its flagged segments

typedef float v16f_t __attribute__((ext_vector_type(16)));

template <bool SHARED, int NG, int XW>
__device__ __forceinline__ void mfma_flush(v16f_t (&acc)[NG], float* __restrict__ Yf, int64_t pt, uint64_t mask, float2 alpha, float sgn_ai, bool f_ok,
                                           int beta, int mn, int n0, int nm, int bm, int bs, int st_log2) {
    const int xs_log2 = 4 - st_log2;
    const int xseg = (4 * beta) >> st_log2;                // (a lane's four cells 4 beta .. 4 beta + 3 lie in one segment: segments are >= 4 cells)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g < bs) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < bm) {
                    // D of block b: register 4 b + r of lane (f, xq) holds cell x = 4 xq + r, float f
                    float* dst = Yf + (pt + 4 * beta + (int64_t)n0 * (b + (int64_t)nm * g)) * XW + mn;
                    const bool mine = f_ok && ((mask >> (xseg + ((b + bm * g) << xs_log2))) & 1ull);
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[g][4 * b + r];
                        const float pv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
                        o[r] = fmaf(alpha.x, v, sgn_ai * pv);
                    }
                    if (mine) {
                        // (asm: the compiler's wait-count bookkeeping does not see these and so leaves the prefetched records alone; the
                        // cell's offset rides as the instruction's immediate: one address per brick row)
                        if (SHARED)
                            asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %2, off offset:%5\n\t"
                                         "global_atomic_add_f32 %0, %3, off offset:%6\n\tglobal_atomic_add_f32 %0, %4, off offset:%7"
                                         :: "v"(dst), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "i"(XW * 4), "i"(2 * XW * 4), "i"(3 * XW * 4) : "memory");
                        else
                            asm volatile("global_store_dword %0, %1, off\n\tglobal_store_dword %0, %2, off offset:%5\n\t"
                                         "global_store_dword %0, %3, off offset:%6\n\tglobal_store_dword %0, %4, off offset:%7"
                                         :: "v"(dst), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "i"(XW * 4), "i"(2 * XW * 4), "i"(3 * XW * 4) : "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);     // (keep the 16 rows' addresses from all being formed up front)
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;
    }
}

template <int NC, int TW, int NG>
__global__ void __launch_bounds__(BLK, 3)                  // (three waves per SIMD: 168 registers, 64 of them the brick image)
k_grid_scatter_mfma(const ShareTask* __restrict__ tasks, int ntasks, const ShareBrick* __restrict__ btab, const uint2* __restrict__ shares,
                    const uint32_t* __restrict__ recx, int rs /* words per sample: record, then the packed panel row */,
                    float* __restrict__ Yf, float2 alpha, int n0, int nm, int bm_log2, int bs_log2, int nbx, int nbm, int st_log2) {
    constexpr int RW = sep_words(TW), XW = 2 * NC, ZL = 3 * TW + 2;          // ZL: a record word that is always zero
    constexpr int PW = RW + 16;                            // ring slot words [PW, PW + TW + 6): 0 0 0, the slow-axis weights, 0 0 0
    constexpr int G = 4, R = 16;                           // shares per group (their loads are in flight together); ring slots
    static_assert(PW + TW + 6 <= 64 && XW <= 16 && NG == 4, "a ring slot holds record, panel row and the padded slow-axis weights");
    __shared__ uint32_t ring_all[WPB][R][64];              // lane-word images of the next shares' record + panel row
    __shared__ uint2 hdr_all[WPB][128];                    // two batches of share headers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int task = blockIdx.x * WPB + wv;
    if (task >= ntasks) return;                            // (no workgroup barrier below: waves are independent)
    const ShareTask tk = tasks[task];
    const int nb = tk.nb_flags & 0xffff;
    const bool shared = (tk.nb_flags >> 16) & 1;
    const int nsh = tk.hi - tk.lo;
    const int bm = 1 << bm_log2, bs = 1 << bs_log2;
    const int beta = lane >> 4, mn = lane & 15;            // block (middle-axis cell of the brick); x cell (A, D rows) / float (B, D columns)
    uint32_t* __restrict__ ring = &ring_all[wv][0][0];

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
    v16f_t acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;

    // what lane L fetches of a sample: L < RW the record, RW <= L < RW + XW the panel row, PW + 3 <= L < PW + 3 + TW the slow-axis
    // weights once more (between zeros: a share reads four of these words starting at PW + 3 - os, whatever os is), else a zero word
    const int lw = lane < RW + XW ? lane : (lane >= PW + 3 && lane < PW + 3 + TW) ? 2 * TW + (lane - PW - 3) : ZL;
    const uint32_t* lane_base = recx + lw;
    const float sgn_ai = (mn & 1) ? alpha.y : -alpha.y;    // out = ar * v + sgn_ai * (the other part of the complex value: the neighbouring lane's)
    const bool f_ok = mn < XW;
    const int xv_off = (f_ok ? RW + mn : ZL) * 4, mn4 = mn * 4, beta4 = (TW + beta) * 4;

    int cur = 0;
    int cur_end = __builtin_amdgcn_readlane(my_end, 0);
    auto brick_of = [&](int& pt_hi, int& pt_lo, uint32_t& mlo, uint32_t& mhi) __attribute__((always_inline)) {
        mlo = (uint32_t)__builtin_amdgcn_readlane((int)my_mlo, cur); mhi = (uint32_t)__builtin_amdgcn_readlane((int)my_mhi, cur);
        pt_hi = __builtin_amdgcn_readlane((int)(my_pt >> 32), cur); pt_lo = __builtin_amdgcn_readlane((int)(uint32_t)my_pt, cur);
    };

    // Share headers: 64 per batch, lane-parallel (one coalesced 512-byte load), kept in LDS two batches deep with a third in
    // flight -- every later access is an LDS read, so nothing in the loop waits for a scalar or vector load it has just issued.
    const uint2* __restrict__ shp = shares + tk.lo;
    uint2* __restrict__ hdrs = &hdr_all[wv][0];
    auto load_batch = [&](int b) __attribute__((always_inline)) {
        const int idx = b * 64 + lane;
        return idx < nsh ? shp[idx] : make_uint2(0u, 0u);
    };
    {
        const uint2 b0 = load_batch(0), b1 = load_batch(1);
        hdrs[lane] = b0;
        hdrs[64 + lane] = b1;
    }
    uint2 sh_next = load_batch(2);
    int cb = 0;                                            // LDS holds batches cb and cb + 1; sh_next holds batch cb + 2

    // The pipeline of a group of G shares:  Q  the lane-word loads of record + panel row  ->  S  the loaded words into the ring  ->
    // P  the outer products.  Step i runs Q(i + 2), P(i), S(i + 1): a load has two groups' worth of work to arrive in.
    auto request = [&](uint32_t (&w)[G], int gi) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int sraw = G * gi + k;
            const int sc = sraw < nsh ? sraw : nsh - 1;
            const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdrs[sc & 127].x) & 0x0fffffffu;          // (wave-uniform address)
            w[k] = lane_base[(size_t)t * (uint32_t)rs];
        }
    };
    auto stage = [&](const uint32_t (&w)[G], int gi) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < G; ++k) ring[((G * gi + k) & (R - 1)) * 64 + lane] = w[k];
    };
    auto process = [&](int gi) __attribute__((always_inline)) {
        for (int k = 0; k < G; ++k) {                      // (a run-time loop: ONE share site, one flush site)
            const int s = G * gi + k;
            if (s >= nsh) break;                           // (wave-uniform)
            if (s >= cur_end) {                            // (every brick of the table holds at least one share; shared tasks never get here)
                int ph, pl; uint32_t mlo, mhi;
                brick_of(ph, pl, mlo, mhi);
                mfma_flush<false, NG, XW>(acc, Yf, ((int64_t)ph << 32) | (uint32_t)pl, ((uint64_t)mhi << 32) | mlo, alpha, sgn_ai, f_ok, beta, mn, n0, nm, bm, bs, st_log2);
                ++cur;
                cur_end = __builtin_amdgcn_readlane(my_end, cur & 63);
            }
            const uint2 hd = hdrs[s & 127];
            const uint32_t geo = (uint32_t)__builtin_amdgcn_readfirstlane((int)hd.y), gmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)hd.x) >> 28;
            const char* slot = reinterpret_cast<const char*>(ring) + (s & (R - 1)) * 256;
            // byte offsets into the slot: x weight of cell mn, middle weight of block beta (a zero word outside the taps), panel value
            const int ox4 = (int)(geo & 31u) * 4 - 32, om4 = (int)((geo >> 5) & 31u) * 4 - 32, os4 = (int)((geo >> 10) & 31u) * 4 - 32;
            const unsigned dx = (unsigned)(mn4 - ox4), db = (unsigned)(beta4 - om4);
            const float ax = __uint_as_float(*reinterpret_cast<const uint32_t*>(slot + (dx < (unsigned)(4 * TW) ? dx : (unsigned)(4 * ZL))));
            const float wm = __uint_as_float(*reinterpret_cast<const uint32_t*>(slot + (db - 4u * TW < (unsigned)(4 * TW) ? db : (unsigned)(4 * ZL))));
            const float xv = __uint_as_float(*reinterpret_cast<const uint32_t*>(slot + xv_off));
            const uint32_t* wsp = reinterpret_cast<const uint32_t*>(slot + (PW + 3) * 4 - os4);             // (wave-uniform: broadcast reads)
            const float bmx = wm * xv;
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if ((gmask >> g) & 1u)                     // (wave-uniform: slow-axis cells without a tap of this share cost nothing)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x1f32(ax, bmx * __uint_as_float(wsp[g]), acc[g], 0, 0, 0);
        }
    };

    const int ngroups = (nsh + G - 1) / G;
    uint32_t wa[G], wb[G];
    request(wa, 0);
    request(wb, 1);
    stage(wa, 0);
    for (int gi = 0; gi < ngroups; gi += 2) {
        if (((G * gi) >> 6) != cb) {                       // processing enters batch cb + 1: batch cb + 2 takes the place of batch cb
            ++cb;
            hdrs[((cb + 1) & 1) * 64 + lane] = sh_next;
            sh_next = load_batch(cb + 2);
        }
        request(wa, gi + 2);
        process(gi);
        stage(wb, gi + 1);
        request(wb, gi + 3);
        process(gi + 1);
        stage(wa, gi + 2);
    }
    {
        int ph, pl; uint32_t mlo, mhi;
        brick_of(ph, pl, mlo, mhi);
        const int64_t pt = ((int64_t)ph << 32) | (uint32_t)pl;
        const uint64_t mask = ((uint64_t)mhi << 32) | mlo;
        if (shared) mfma_flush<true, NG, XW>(acc, Yf, pt, mask, alpha, sgn_ai, f_ok, beta, mn, n0, nm, bm, bs, st_log2);
        else        mfma_flush<false, NG, XW>(acc, Yf, pt, mask, alpha, sgn_ai, f_ok, beta, mn, n0, nm, bm, bs, st_log2);
    }
}

// the record of every sample with room for its panel row behind it: recx[t * rs + 0 .. RW) = record, [RW, RW + 2 NC) = X[t, :]
template <int NC>
__global__ void __launch_bounds__(BLK)
k_sep_pack_recx(int64_t rows, const float2* __restrict__ X, int64_t ld, float2* __restrict__ recx2 /* recx as float2 */, int rs2 /* rs / 2 */, int rw2 /* RW / 2 */) {
    for (int64_t e = (int64_t)blockIdx.x * BLK + threadIdx.x; e < rows * NC; e += (int64_t)gridDim.x * BLK)
        recx2[(e / NC) * rs2 + rw2 + e % NC] = X[(e % NC) * ld + e / NC];
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
                // the slow-axis cells of the brick that hold a tap of this share (bricks of at most 4 slow cells: the MFMA form)
                const uint32_t gmask = g.bdim[2] <= 4 ? ((((1u << (ps[c].hi - ps[c].lo)) - 1u) << (ps[c].o + ps[c].lo)) & 15u) : 15u;
                f(brick, geo, gmask);
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
    // (bricks need not divide the middle and slow axes: the last brick of an axis is then partly outside the grid -- no share ever holds a
    // tap there, and the caller's segment masks never flag it.  277 = 69 * 4 + 1: the reference driver's default grid)
    if (!(tw == 4 || tw == 6 || tw == 8) || n0 < 16 || n0 % 16 || nm < tw || ns < tw || !pow2(bm) || !pow2(bs) || bm > 16 || bs > 16 ||
        n0 > 65535 || nm > 65535 || ns > 65535)
        return false;
    g.n[0] = n0; g.n[1] = nm; g.n[2] = ns;
    g.bdim[0] = 16; g.bdim[1] = bm; g.bdim[2] = bs;
    g.nbr[0] = n0 / 16; g.nbr[1] = (nm + bm - 1) / bm; g.nbr[2] = (ns + bs - 1) / bs;
    g.tw = tw; g.rw = sep_words(tw);
    return g.nbr[0] * g.nbr[1] * g.nbr[2] < 0x7fffffffLL;
}

}  // namespace

extern "C" {

// Y (M x NC, column-major, ldy) = alpha * G X + beta * Y for the gridding matrix given by `records` (ig_interp3_sep with grid_order
// matching the panel: axes (n0, nm, ns) in memory order) and the coil-interleaved grid panel X (n0 * nm * ns rows of NC values).
int ig_grid_gather_sep_group(int64_t NC, int tw) {
    if (!(NC == 2 || NC == 4 || NC == 8) || !(tw == 4 || tw == 6 || tw == 8)) return 0;
    return WPB * (64 / ((int)(NC / 2) * (tw <= 4 ? 4 : 8)));
}

int ig_grid_gather_sep(ig_ctx* ctx, int64_t M, int64_t NC, int tw, const void* records, int64_t rec_stride, const void* X_il, int64_t n0, int64_t nm, int64_t ns,
                       float ar, float ai, float br, float bi, void* Y, int64_t ldy, const uint32_t* group_order) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_grid_gather_sep: ctx is NULL");
    IG_REQUIRE(ctx, M >= 0 && (NC == 2 || NC == 4 || NC == 8) && (tw == 4 || tw == 6 || tw == 8), "ig_grid_gather_sep: 2, 4 or 8 interleaved coils; tw 4, 6 or 8");
    IG_REQUIRE(ctx, n0 >= tw && nm >= tw && ns >= tw && n0 <= 65535 && nm <= 65535 && ns <= 65535 && n0 * nm * ns * (NC / 2) < (1LL << 32),
               "ig_grid_gather_sep: grid %lld x %lld x %lld x %lld coils out of range", (long long)n0, (long long)nm, (long long)ns, (long long)NC);
    IG_REQUIRE(ctx, M == 0 || (records && X_il && Y && ldy >= M), "ig_grid_gather_sep: NULL pointer or ldy < M");
    IG_REQUIRE(ctx, rec_stride >= sep_words(tw) && rec_stride % 4 == 0 && rec_stride <= 1024, "ig_grid_gather_sep: record stride %lld words", (long long)rec_stride);
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
                                   (const uint32_t*)records, (int)rec_stride, (const float4*)X_il, (int)n0, (int)nm, (int)ns, (float2*)Y, ldy, alpha, beta, group_order); \
        else    hipLaunchKernelGGL((k_grid_gather_sep<NC_, TW_, 1>), dim3((unsigned)blocks), dim3(BLK), 0, ctx->stream, M,       \
                                   (const uint32_t*)records, (int)rec_stride, (const float4*)X_il, (int)n0, (int)nm, (int)ns, (float2*)Y, ldy, alpha, beta, group_order); \
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
        return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_shares_count: a grid of a multiple of 16 points along x, bricks of 16 x bm x bs cells (powers of two <= 16); tw 4, 6 or 8");
    const int64_t nb = g.nbr[0] * g.nbr[1] * g.nbr[2];
    const int nt = share_threads(M);
    const int64_t per = (M + nt - 1) / nt;
    std::vector<std::vector<int32_t>> cnt((size_t)nt);
    std::atomic<int> bad{0};
    run_threads(nt, [&](int t) {
        cnt[t].assign((size_t)nb, 0);
        const int64_t lo = std::min<int64_t>(M, t * per), hi = std::min<int64_t>(M, lo + per);
        for (int64_t s = lo; s < hi; ++s)
            for_shares(g, records + (size_t)s * g.rw, [&](int64_t brick, uint32_t, uint32_t) { if (++cnt[t][brick] < 0) bad = 1; });
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
    if (M < 0 || M >= (1LL << 28) || (M > 0 && (!records || !shares)) || !brick_ptr || !share_geom(g, tw, n0, nm, ns, bm, bs))
        return ig_fail(nullptr, IG_ERR_ARG, "ig_grid_shares_fill: bad arguments (fewer than 2^28 samples)");
    const int64_t nb = g.nbr[0] * g.nbr[1] * g.nbr[2];
    const int nt = share_threads(M);
    const int64_t per = (M + nt - 1) / nt;
    std::vector<std::vector<int32_t>> cur((size_t)nt);
    run_threads(nt, [&](int t) {
        cur[t].assign((size_t)nb, 0);
        const int64_t lo = std::min<int64_t>(M, t * per), hi = std::min<int64_t>(M, lo + per);
        for (int64_t s = lo; s < hi; ++s)
            for_shares(g, records + (size_t)s * g.rw, [&](int64_t brick, uint32_t, uint32_t) { ++cur[t][brick]; });
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
            for_shares(g, records + (size_t)s * g.rw, [&](int64_t brick, uint32_t geo, uint32_t gmask) {
                uint32_t* o = shares + 2 * (size_t)(brick_ptr[brick] + cursor[brick]++);
                o[0] = (uint32_t)s | (gmask << 28); o[1] = geo;
            });
    });
    return IG_OK;
}

// Y_il (n0 * nm * ns grid points x NC interleaved coils) = alpha * G^H * X over the flagged segments of the bricks that hold a share;
// X (M x NC, column-major, ldx).  tasks / brick_table as for ig_ccsrmm_t_bricks with shares in place of entries and 16-byte table rows
// {brick, end of its shares, flagged segments: 64 bits, bit xs + (16 / support_tile) * (im + bm * is)}; shared_table: the rows of the
// bricks several tasks add into (zeroed first; those tasks add with float atomics).
int ig_grid_scatter_sep(ig_ctx* ctx, int64_t M, int64_t NC, int tw, void* records, int64_t rec_stride, const void* shares, const void* X, int64_t ldx,
                        void* Y_il, int64_t n0, int64_t nm, int64_t ns, int bm, int bs, const int32_t* tasks, int64_t ntasks,
                        const int32_t* brick_table, const int32_t* shared_table, int64_t nshared, int support_tile, float ar, float ai) {
    IG_REQUIRE(ctx, ctx != nullptr, "ig_grid_scatter_sep: ctx is NULL");
    ShareGeom g;
    IG_REQUIRE(ctx, M >= 0 && M < (1LL << 28) && (NC == 4 || NC == 8) && share_geom(g, tw, n0, nm, ns, bm, bs) && bm <= 4 && bs <= 4,
               "ig_grid_scatter_sep: 4 or 8 interleaved coils; a multiple of 16 points along x; bricks of 16 x (bm <= 4) x (bs <= 4) cells; tw 4, 6 or 8");
    IG_REQUIRE(ctx, support_tile == 16 || support_tile == 8 || support_tile == 4, "ig_grid_scatter_sep: support_tile 16, 8 or 4");
    IG_REQUIRE(ctx, ntasks >= 0 && ntasks <= 0x7fffffffLL && (ntasks == 0 || (tasks && brick_table && records && shares && X && Y_il)) && ldx >= M &&
               nshared >= 0 && (nshared == 0 || shared_table), "ig_grid_scatter_sep: bad task list or NULL array");
    IG_REQUIRE(ctx, n0 * nm * ns * (NC / 2) < (1LL << 32), "ig_grid_scatter_sep: grid panel too large");
    const int rw = sep_words(tw);
    IG_REQUIRE(ctx, rec_stride >= rw + 2 * NC && rec_stride % 4 == 0, "ig_grid_scatter_sep: record stride %lld words: the record (%d) and the panel row behind it", (long long)rec_stride, rw);
    if (ntasks == 0 || M == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    const float2 alpha = make_float2(ar, ai);
    int bm_log2 = 0, bs_log2 = 0;
    while ((1 << bm_log2) < bm) ++bm_log2;
    while ((1 << bs_log2) < bs) ++bs_log2;
    const int st_log2 = support_tile == 16 ? 4 : support_tile == 8 ? 3 : 2;
    const int nbx = (int)(n0 / 16), nbm = (int)((nm + bm - 1) / bm);
    const unsigned blocks = (unsigned)((ntasks + WPB - 1) / WPB);
    const size_t need = (size_t)M * NC * 8;
    int64_t gp = (M * NC + BLK - 1) / BLK;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (gp > cap) gp = cap;
#define IG_SM(NC_, TW_) do {                                                                                                             \
        {   ig_prof_scope prof(ctx, "pack_panel", 2.0 * (double)need);                                                                   \
            hipLaunchKernelGGL((k_sep_pack_recx<NC_>), dim3((unsigned)gp), dim3(BLK), 0, ctx->stream, M, (const float2*)X, ldx, (float2*)records, (int)(rec_stride / 2), rw / 2); } \
        if (nshared) {                                                                                                                   \
            ig_prof_scope prof(ctx, "grid_sep_zero");                                                                                    \
            hipLaunchKernelGGL((k_grid_sep_zero<NC_>), dim3((unsigned)nshared), dim3(BLK), 0, ctx->stream, (const ShareBrick*)shared_table, \
                               (float4*)Y_il, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2); }                                  \
        ig_prof_scope prof(ctx, "grid_scatter_sep");                                                                                     \
        hipLaunchKernelGGL((k_grid_scatter_mfma<NC_, TW_, 4>), dim3(blocks), dim3(BLK), 0, ctx->stream, (const ShareTask*)tasks, (int)ntasks, \
                           (const ShareBrick*)brick_table, (const uint2*)shares, (const uint32_t*)records, (int)rec_stride, (float*)Y_il, \
                           alpha, (int)n0, (int)nm, bm_log2, bs_log2, nbx, nbm, st_log2);                                                \
    } while (0)
#define IG_SM_TW(NC_) do { if (tw == 4) IG_SM(NC_, 4); else if (tw == 6) IG_SM(NC_, 6); else IG_SM(NC_, 8); } while (0)
    if (NC == 8) IG_SM_TW(8); else IG_SM_TW(4);
#undef IG_SM_TW
#undef IG_SM
    IG_LAUNCH_CHECK(ctx, "k_grid_scatter_mfma");
    return IG_OK;
}

}  // extern "C"
