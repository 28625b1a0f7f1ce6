// Batched complex-to-complex FFT for complex64 on gfx950 -- hand-written
// Stockham autosort kernels, no vendor FFT library.
//
// Contract (Backend.fftn/ifftn, indigo/backends/backend.py:497-509; oracle
// np.py:102-115; cuFFT usage cuda.py:470-498): Fortran-ordered volume
// dims[0..rank-1] (dims[0] contiguous) times `batch` contiguous volumes,
// unnormalised in both directions.
//
// One pass per axis.  For an axis of length n the array is viewed as
// [outer][n][inner] (inner contiguous), i.e. inner*outer independent
// "columns" of n elements with stride `inner`.
//
//  * two-stage register kernel k_fft_2stage (n = 512 = 32x16, n = 256 = 16x16; every pass of the SENSE path):
//    a thread loads its 32 (16) inputs straight from HBM, runs a register DFT, hands the results over through LDS
//    in two 32 KB rounds, runs the second register DFT and stores straight back.  The same kernel serves the
//    zero-pad-aware forward / cropped inverse transforms (boxes, diagonal weights, k-space support bitmap,
//    coil-interleaved layouts, coil combination) through one pass descriptor; see PassDesc and the kernel below.
//
//  * LDS kernel (n <= 4096 with prime factors in {2,3,5,7}):
//    a workgroup owns a tile of W neighbouring columns.  The tile is loaded
//    with coalesced global reads (for axis 0, where a column is a contiguous
//    line, the tile is one contiguous block and is transposed on the way in),
//    kept in LDS as [n][W+1] (one pad element per row against bank conflicts)
//    next to an LDS copy of the n twiddles, transformed by radix-8/4/2/3/5/7
//    Stockham stages (each thread holds up to 8 elements in registers between
//    the read barrier and the write barrier, so one LDS buffer suffices), and
//    written back the way it came.  Tiles are disjoint, so an axis can run in
//    place.  HBM traffic: exactly one read and one write of the volume per axis.
//
//  * generic kernel (anything else: large prime factors, n > 4096):
//    one Stockham stage per launch through global memory, one thread per
//    output element, any radix (a prime radix p is a direct p-point DFT).
//    Slow, but exact in structure for every size; needs a ping-pong workspace.
//
// The inverse transform is computed as conj(FFT(conj(x))).
#include "ig_common.h"
#include "ig_packed.h"
#include "ig_fft_ab.h"
#include "ig_fft_ab_list.h"
#include <vector>
#include <functional>
#include <algorithm>
#include <cmath>
#include <cmath>
#include <cstring>
#include <string>
#include <memory>
#include <complex>

namespace {

// Choices measured in rounds 1-3 and now fixed in the code (the rejected alternatives are recorded in DESIGN.md section 3.1; what
// is left to switch lives in tools/kernel_lab.py, not here):
//  * non-temporal (streaming) loads and stores in the 2-stage axis passes -- every byte of a pass is touched exactly once
//    (256^3 x 8 SENSE evaluation: 8.95 ms with either one off, 8.58 ms with both on);
//  * wave-uniform skipping of the load instructions no lane wants, and scalar-gated stores (one opaque asm block each) on
//    unweighted passes with a run-time output box;
//  * the strided half-OUTPUT variants are not capped at 128 VGPRs (the cap spilled and was slower).
constexpr int MAX_STAGES = 16;
constexpr int E = 8;                 // complex elements a thread holds per LDS stage
constexpr int LDS_NMAX = 4096;
constexpr size_t LDS_BUDGET = 150 * 1024;

struct Radices { int r[MAX_STAGES]; };

// lane exchange inside a row of 16 lanes by a DPP control word (out-of-row sources read as zero)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// ---- butterflies --------------------------------------------------------------
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

// streaming (non-temporal) 8-byte accesses: an axis pass touches every byte exactly once
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ float2 ld_stream(const float2* p) {
    if (NT) { const v2f_t v = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(p)); return make_float2(v.x, v.y); }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st_stream(float2* p, float2 a) {
    if (NT) { v2f_t v; v.x = a.x; v.y = a.y; __builtin_nontemporal_store(v, reinterpret_cast<v2f_t*>(p)); }
    else *p = a;
}

__device__ __forceinline__ void bfly2(float2& a, float2& b) {
    const float2 t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// forward 4-point DFT in place, natural output order
__device__ __forceinline__ void bfly4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2);
    const float2 t2 = cadd(a1, a3), t3 = mul_mi(csub(a1, a3));
    a0 = cadd(t0, t2); a2 = csub(t0, t2);
    a1 = cadd(t1, t3); a3 = csub(t1, t3);
}

template <int R>
struct Bfly {
    // generic small prime: direct R-point DFT, roots read from the LDS twiddle
    // table of the enclosing length-n transform (R divides n)
    __device__ static __forceinline__ void run(float2 (&v)[R], const float2* __restrict__ tws, int n) {
        float2 o[R];
        const int step = n / R;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            float2 acc = v[0];
#pragma unroll
            for (int q = 1; q < R; ++q) cfma(acc, v[q], tws[((q * k) % R) * step]);
            o[k] = acc;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = o[k];
    }
};
template <> struct Bfly<2> {
    __device__ static __forceinline__ void run(float2 (&v)[2], const float2*, int) { bfly2(v[0], v[1]); }
};
template <> struct Bfly<4> {
    __device__ static __forceinline__ void run(float2 (&v)[4], const float2*, int) { bfly4(v[0], v[1], v[2], v[3]); }
};
template <> struct Bfly<8> {
    __device__ static __forceinline__ void run(float2 (&v)[8], const float2*, int) {
        // two 4-point DFTs on even / odd inputs, then the 8th-root twiddles
        bfly4(v[0], v[2], v[4], v[6]);
        bfly4(v[1], v[3], v[5], v[7]);
        const float h = 0.70710678118654752440f;
        const float2 w1 = make_float2(h, -h), w3 = make_float2(-h, -h);
        v[3] = cmul(v[3], w1);
        v[5] = mul_mi(v[5]);
        v[7] = cmul(v[7], w3);
        // after bfly4 the even half holds E[0..3] in v[0],v[2],v[4],v[6] and the odd half O[0..3] in v[1],v[3],v[5],v[7]
        bfly2(v[0], v[1]);   // X0, X4
        bfly2(v[2], v[3]);   // X1, X5
        bfly2(v[4], v[5]);   // X2, X6
        bfly2(v[6], v[7]);   // X3, X7
        // reorder to natural output order X0..X7
        const float2 x4 = v[1], x1 = v[2], x5 = v[3], x2 = v[4], x6 = v[5], x3 = v[6];
        v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
    }
};

// One Stockham stage over the LDS tile.  lds[j*WP + w] is element j of column w.
// Thread (w, t) handles butterflies b = t, t+T, ... (< n/R) of column w.
template <int R>
__device__ __forceinline__ void lds_stage(float2* __restrict__ lds, const float2* __restrict__ tws,
                                          int n, int Ns, int T, int t, int w, int WP) {
    constexpr int BPT = E / R;
    const int nb = n / R;
    float2 v[BPT][R];
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int b = t + q * T;
        if (b < nb) {
#pragma unroll
            for (int k = 0; k < R; ++k) v[q][k] = lds[(b + k * nb) * WP + w];
        }
    }
    __syncthreads();
    const int tstride = nb / Ns;     // n / (Ns * R)
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        const int b = t + q * T;
        if (b < nb) {
            const int m = b % Ns;
            if (Ns > 1) {
#pragma unroll
                for (int k = 1; k < R; ++k) v[q][k] = cmul(v[q][k], tws[k * m * tstride]);
            }
            Bfly<R>::run(v[q], tws, n);
            const int base = (b - m) * R + m;
#pragma unroll
            for (int k = 0; k < R; ++k) lds[(base + k * Ns) * WP + w] = v[q][k];
        }
    }
    __syncthreads();
}

// ---- register-resident DFTs of 16 and 32 points --------------------------------
// cos(2 pi j / 32); compile-time foldable so that unrolled code carries literals
__host__ __device__ constexpr float cos32(int j) {
    j &= 31;
    if (j > 16) j = 32 - j;
    const float T[9] = {1.0f, 0.980785251f, 0.923879504f, 0.831469595f, 0.707106769f,
                        0.555570245f, 0.382683426f, 0.195090324f, 0.0f};
    return j <= 8 ? T[j] : -T[16 - j];
}
// a * exp(-2 pi i j / 32), trivial rotations without multiplies (j is a constant after unrolling)
__device__ __forceinline__ float2 mul_w32(float2 a, int j) {
    j &= 31;
    if (j == 0) return a;
    if (j == 8) return make_float2(a.y, -a.x);
    if (j == 16) return make_float2(-a.x, -a.y);
    if (j == 24) return make_float2(-a.y, a.x);
    const float c = cos32(j), s = -cos32(j + 24);      // w = c + i s, s = -sin(2 pi j/32)
    return make_float2(fmaf(a.x, c, -a.y * s), fmaf(a.x, s, a.y * c));
}

template <int N> struct RegFFT;       // in-place forward DFT of N register values, natural order out
template <> struct RegFFT<4> {
    __device__ static __forceinline__ void run(float2 (&x)[4]) { bfly4(x[0], x[1], x[2], x[3]); }
};
template <> struct RegFFT<8> {
    __device__ static __forceinline__ void run(float2 (&x)[8]) { Bfly<8>::run(x, nullptr, 0); }
};
template <int N> struct RegFFT {      // N = 16, 32: radix-4 decimation in frequency over N/4-point DFTs
    __device__ static __forceinline__ void run(float2 (&x)[N]) {
        constexpr int M = N / 4;
        float2 z[4][M];
#pragma unroll
        for (int r = 0; r < M; ++r) {
            bfly4(x[r], x[r + M], x[r + 2 * M], x[r + 3 * M]);
            z[0][r] = x[r];
#pragma unroll
            for (int q = 1; q < 4; ++q) z[q][r] = mul_w32(x[r + q * M], r * q * (32 / N));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) RegFFT<M>::run(z[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < M; ++k) x[4 * k + q] = z[q][k];
    }
};

// The same DFT when only the middle half of the inputs, x[N/4 .. 3N/4), is non-zero (the image box of a
// 2x-oversampled grid): the first radix-4 layer collapses to one add/sub pair per butterfly and the outer
// quarters of x are never read, so they cost neither loads nor registers.
template <int N> struct RegFFTHalfIn {
    __device__ static __forceinline__ void run(float2 (&x)[N]) {
        constexpr int M = N / 4;
        float2 z[4][M];
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const float2 a1 = x[r + M], a2 = x[r + 2 * M];
            const float2 t3 = mul_mi(a1);
            z[0][r] = cadd(a2, a1);
            z[1][r] = mul_w32(csub(t3, a2), r * (32 / N));
            z[2][r] = mul_w32(csub(a2, a1), r * 2 * (32 / N));
            const float2 s = cadd(a2, t3);
            z[3][r] = mul_w32(make_float2(-s.x, -s.y), r * 3 * (32 / N));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) RegFFT<M>::run(z[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < M; ++k) x[4 * k + q] = z[q][k];
    }
};

// (packed complex arithmetic -- one 64-bit register pair per complex number, v_pk_* instructions: ig_packed.h)
__device__ __forceinline__ void pbfly2(cx& a, cx& b) { const cx t = a - b; a = a + b; b = t; }
// Direction as arithmetic: INV = true is the same butterfly network with every constant conjugated (the unnormalised inverse
// DFT), so an inverse pass needs no conjugation of its inputs and outputs -- that cost the cropped passes two VALU
// instructions per element on each side (a packed negate and a move), a tenth of their vector instructions.
// (ASM = false keeps the compiler's own lowering of the -+i products: the second launch of the two-launch 256^3 transform is
// scheduled around it -- with the assembly forms it went from 94 to 128 registers and spilled.)
template <bool INV = false, bool ASM = true>
__device__ __forceinline__ void pbfly4(cx& a0, cx& a1, cx& a2, cx& a3) {
    const cx t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, dd = a1 - a3;
    a0 = t0 + t2; a2 = t0 - t2;
    if (ASM) {
        a1 = INV ? padd_pi(t1, dd) : padd_mi(t1, dd);          // t1 -+ i dd
        a3 = INV ? padd_mi(t1, dd) : padd_pi(t1, dd);
    } else {
        const cx t3 = INV ? cmul_pi_c(dd) : cmul_mi_c(dd);
        a1 = t1 + t3; a3 = t1 - t3;
    }
}
// a * exp(-+2 pi i j / 32) with a literal root (j constant after unrolling), trivial rotations without multiplies
template <bool INV = false, bool ASM = true>
__device__ __forceinline__ cx pmul_w32(cx a, int j) {
    j &= 31;
    if (INV) j = (32 - j) & 31;
    if (j == 0) return a;
    if (j == 8) return ASM ? cmul_mi(a) : cmul_mi_c(a);
    if (j == 16) return cneg(a);
    if (j == 24) return ASM ? cmul_pi(a) : cmul_pi_c(a);
    const float c = cos32(j), s = -cos32(j + 24);      // w = c + i s
    return cxmul(a, mk(c, s));
}
template <int N, bool INV = false, bool ASM = true> struct PFFT;          // in-place DFT of N packed register values, natural order out
template <bool INV, bool ASM> struct PFFT<4, INV, ASM> {
    __device__ static __forceinline__ void run(cx (&x)[4]) { pbfly4<INV, ASM>(x[0], x[1], x[2], x[3]); }
};
template <bool INV, bool ASM> struct PFFT<8, INV, ASM> {
    __device__ static __forceinline__ void run(cx (&v)[8]) {
        pbfly4<INV, ASM>(v[0], v[2], v[4], v[6]);
        pbfly4<INV, ASM>(v[1], v[3], v[5], v[7]);
        const float h = 0.70710678118654752440f;
        v[3] = cxmul(v[3], mk(h, INV ? h : -h));
        v[7] = cxmul(v[7], mk(-h, INV ? h : -h));
        pbfly2(v[0], v[1]); pbfly2(v[2], v[3]); pbfly2(v[6], v[7]);
        if (ASM) {   // v[4] +- (-+i) v[5]
            const cx e = v[4], o = v[5];
            v[4] = INV ? padd_pi(e, o) : padd_mi(e, o);
            v[5] = INV ? padd_mi(e, o) : padd_pi(e, o);
        } else {
            v[5] = INV ? cmul_pi_c(v[5]) : cmul_mi_c(v[5]);
            pbfly2(v[4], v[5]);
        }
        const cx x4 = v[1], x1 = v[2], x5 = v[3], x2 = v[4], x6 = v[5], x3 = v[6];
        v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
    }
};
template <int N, bool INV, bool ASM> struct PFFT {         // N = 16, 32: radix-4 decimation in frequency over N/4-point DFTs
    __device__ static __forceinline__ void run(cx (&x)[N]) {
        constexpr int M = N / 4;
        cx z[4][M];
#pragma unroll
        for (int r = 0; r < M; ++r) {
            pbfly4<INV, ASM>(x[r], x[r + M], x[r + 2 * M], x[r + 3 * M]);
            z[0][r] = x[r];
#pragma unroll
            for (int q = 1; q < 4; ++q) z[q][r] = pmul_w32<INV, ASM>(x[r + q * M], r * q * (32 / N));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) PFFT<M, INV, ASM>::run(z[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < M; ++k) x[4 * k + q] = z[q][k];
    }
};
// only the middle half of the inputs, x[N/4 .. 3N/4), is non-zero: the first radix-4 layer collapses
template <int N> struct PFFTHalfIn {
    __device__ static __forceinline__ void run(cx (&x)[N]) {
        constexpr int M = N / 4;
        cx z[4][M];
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const cx a1 = x[r + M], a2 = x[r + 2 * M];
            z[0][r] = a2 + a1;
            z[1][r] = pmul_w32(cneg(padd_pi(a2, a1)), r * (32 / N));          // (-i) a1 - a2
            z[2][r] = pmul_w32(a2 - a1, r * 2 * (32 / N));
            z[3][r] = pmul_w32(cneg(padd_mi(a2, a1)), r * 3 * (32 / N));      // -(a2 + (-i) a1)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) PFFT<M>::run(z[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < M; ++k) x[4 * k + q] = z[q][k];
    }
};

// Two-stage kernel for n = R1*R2 (256 = 16x16, 512 = 32x16): the whole column lives in registers
// twice -- stage 1 reads its R1 inputs straight from global memory, stage 2 writes its R2 outputs
// straight back -- with ONE LDS exchange in between (Stockham index map, inter-stage twiddles applied
// on the way into LDS).  A workgroup = W columns x T threads; lanes run along the direction that is
// contiguous in memory (columns for strided axes, elements for axis 0), so every wave access is a
// set of 128-byte segments and every LDS access is conflict-free at the minimum cycle count
// (axis 0 uses a 16-element XOR swizzle of the line).  Inputs outside [in_lo, in_hi) are zeros that
// are never loaded and outputs outside [out_lo, out_hi) are never stored: that is what makes the
// zero-padded forward / cropped inverse transforms cheap (SENSE: 1/8 of the grid is non-zero).
// HALF selects boxes known at compile time (the image box of a 2x-oversampled grid is [n/4, 3n/4)):
//   1: input half, output box at run time      3: input half, output full
//   2: output half, input box at run time      4: output half, input full
// Half inputs prune the first butterfly layer; compile-time boxes need no predicates or bounds registers.
// Loads and stores carry the non-temporal (streaming) hint: every byte of a pass is touched exactly once.
template <int R1, int R2, int T, int W, bool AXIS0, int WMODE, bool BOXED, int HALF>
__global__ void __launch_bounds__(W * T, (!AXIS0 && R1 == 32 && (W == 32 || HALF == 1 || HALF == 3)) ? 4 : 1)
k_fft_2stage(PassDesc d, const float2* __restrict__ tw) {
    constexpr int n = R1 * R2, B1 = R2 / T, B2 = R1 / T, NT = W * T;
    constexpr bool NT_LD = true, NT_ST = true;      // streaming hints on both sides
    constexpr bool HALF_IN = HALF == 1 || HALF == 3, HALF_OUT = HALF == 2 || HALF == 4;
    constexpr int SUMW = WMODE >= 3 ? (1 << (WMODE - 3)) : 0;       // WMODE 3 + log2(coils): 3 -> 1 (no sum), 4 -> 2, 5 -> 4, 6 -> 8, 7 -> 16
    // direction: the half-input variants only serve forward (zero-padded) passes and the half-output variants only
    // inverse (cropped) ones, so their conjugations are sign modifiers, not a select per element
    // (SINV: the inverse transform by conjugated constants, PFFT<N, true>; the run-time direction of the general variants by
    // conjugating inputs and outputs around the forward network)
    constexpr bool SINV = HALF_OUT;
    const bool inv = (HALF_OUT || HALF_IN) ? false : (d.inverse != 0);
    static_assert(R2 % T == 0 && R1 % T == 0 && T == 16 && B1 == 1, "lane groups of 16, one stage-1 butterfly per thread");
    extern __shared__ float2 lds[];
    float2* __restrict__ tws = lds + (AXIS0 ? 16 * 17 * W : 16 * T * W);
    const int tid = threadIdx.x;
    // inter-stage twiddles in the order the threads read them: entry t * R1 + k = w_n^(t k) (the second half of the plan's table),
    // so a thread's R1 - 1 reads are one base address plus compile-time offsets, two values per LDS instruction
    // (NT == n: one value per thread, requested here and written to LDS only after the stage-1 loads have been issued -- written
    // straight away, the copy is a round trip of its own at the head of every workgroup: the compiler waits for it before anything else)
    constexpr int TWN = n / NT;                      // 1 (32-column tiles) or 2
    constexpr bool TW_LATE = (n % NT == 0) && TWN <= 2;
    float2 tw_mine[TW_LATE ? TWN : 1];
    if (TW_LATE) {
#pragma unroll
        for (int i = 0; i < TWN; ++i) tw_mine[i] = tw[n + tid + i * NT];
    } else for (int k = tid; k < n; k += NT) tws[k] = tw[n + k];

    const int t = AXIS0 ? (tid % T) : (tid / W);
    const int w = AXIS0 ? (tid / T) : (tid % W);
    // ---- the workgroup's tile: W consecutive k0 of one (k1, k2) row; everything below is wave-uniform
    // (Workgroups are dealt round-robin to the 8 XCDs.  Giving each XCD a contiguous range of tiles instead was measured
    // 10 % SLOWER on the z passes: with the default dealing all XCDs stream through the same DRAM pages together, and no
    // tile shares a cache line with another anyway.)
    // (tile of the row, k1, k2): pass_tile, ig_fft_ab.h
    unsigned tr, k1, k2;
    pass_tile(d, tr, k1, k2);
    const int64_t k0u = (int64_t)tr * ((!AXIS0 && d.cw) ? (d.cw_log2 >= 0 ? W >> d.cw_log2 : W / d.cw) : W);
    int in_lo = d.in_lo, in_hi = d.in_hi, out_lo = d.out_lo, out_hi = d.out_hi;
    // the tile's support records, requested here and tested behind the descriptor set-up (pass_records, ig_fft_ab.h)
    short2 k1r = make_short2(0, 0x7fff), trg = make_short2(0, 0x7fff);
    PassRecords rcd{0u, 0u, 0x7fff0000u, 0xffffffffu, false, false, false, true, false};
    if (BOXED && !AXIS0) rcd = pass_records(d, tw, tr, k1, 16, t, true);
    const bool has_k1r = rcd.has_k1r, has_trg = rcd.has_trg, has_zb = rcd.has_zb;
    // Buffer descriptors based at the tile's first column: every access is descriptor + a wave-uniform byte
    // offset (SGPR, or an immediate on axis 0) + ONE per-lane 32-bit offset.  A lane offset of IG_OOB fails the
    // hardware range check -- the load returns zero, the store is dropped -- which is how boxes and the ragged
    // last tile are predicated without a branch, so all loads of a stage issue back to back.
    // Strided passes re-base the descriptor for every group of 16 elements along the axis (a pure SGPR add), so a
    // column may span far more than the 2 GB window (y pass of the interleaved layout: 16 MB per element step).
    const float2* const b_in = d.in + (k0u * d.in_s[0] + (int64_t)k1 * d.in_s[1] + (int64_t)k2 * d.in_s[2]);
    float2* const b_out = d.out + (k0u * d.out_s[0] + (int64_t)k1 * d.out_s[1] + (int64_t)k2 * d.out_s[2]);
    const float2* const b_w = WMODE ? d.w + (k0u * d.w_s[0] + (int64_t)k1 * d.w_s[1] + (int64_t)k2 * d.w_s[2]) : nullptr;
    const rsrc_t r_in = make_rsrc(b_in), r_out = make_rsrc(b_out), r_w = make_rsrc(b_w);
    // x-axis passes run along contiguous memory by construction: unit strides known at compile time
    const unsigned isj = AXIS0 ? 1u : (unsigned)d.in_sj, osj = AXIS0 ? 1u : (unsigned)d.out_sj, wsj = AXIS0 ? 1u : (unsigned)d.w_sj;
    bool valid;
    unsigned l_in, l_out, l_w;
    if (!AXIS0 && d.cw) {
        const unsigned yl = d.cw_log2 >= 0 ? (unsigned)w >> d.cw_log2 : (unsigned)w / (unsigned)d.cw, a = (unsigned)w - yl * (unsigned)d.cw;
        valid = k0u + yl < d.ext0;
        l_in = (a * (unsigned)d.in_sa + yl * (unsigned)d.in_s[0] + (unsigned)t * isj) * 8u;
        l_out = (a * (unsigned)d.out_sa + yl * (unsigned)d.out_s[0] + (unsigned)t * osj) * 8u;
        l_w = (a * (unsigned)d.w_sa + yl * (unsigned)d.w_s[0] + (unsigned)t * wsj) * 8u;
    } else {
        valid = k0u + w < d.ext0;
        l_in = ((unsigned)w * (unsigned)d.in_s[0] + (unsigned)t * isj) * 8u;
        l_out = ((unsigned)w * (unsigned)d.out_s[0] + (unsigned)t * osj) * 8u;
        l_w = ((unsigned)w * (unsigned)d.w_s[0] + (unsigned)t * wsj) * 8u;
    }
    if (!valid) l_in = l_out = l_w = IG_OOB;
    if (!WMODE) l_w = IG_OOB;
    if (SUMW && (w % SUMW) != 0) l_out = IG_OOB;            // only a column's first sub-column stores the coil sum
    // ---- the records are needed from here on (everything above overlapped their round trip)
    if (has_k1r) k1r = rcd.k1_hull();
    if (has_k1r) {
        if ((int)k1 < k1r.x || (int)k1 >= k1r.y) return;
    }
    // (A half-input pass does not wait for its range here where it need not: its loads need nothing of it -- the output side is
    // narrowed behind the loads.  The y pass, whose hulls do not depend on k1 and are never empty inside a support; and a z pass
    // that has the ky hulls for its early exit above: an empty range inside the hull only leaves no store flagged.)
    const bool defer_out = HALF_IN && has_trg && d.tile_range_mode == 1 && (rcd.scalar || has_k1r);
    if (has_trg && !defer_out) {
        const short2 r = trg = rcd.range();
        if (d.tile_range_mode == 1) {
            out_lo = out_lo > r.x ? out_lo : r.x;
            out_hi = out_hi < r.y ? out_hi : r.y;
            if (out_hi <= out_lo) return;                             // nothing of this tile is ever read
        } else {
            in_lo = in_lo > r.x ? in_lo : r.x;
            in_hi = in_hi < r.y ? in_hi : r.y;
        }
    }

    // Element j = t + 16*m of this thread's column <-> bit m of a 32-bit word: ibits flags the inputs to read
    // (stage 1 loads m = k), obits the outputs to keep (stage 2 stores m = q + r*R1/16).  Boxes [lo, hi) become
    // bit ranges; a per-tile bitmap (k-space support at 16-row granularity) is and-ed in.
    auto ceil16 = [](int a) -> int { a += 15; return a <= 0 ? 0 : (a >= 512 ? 32 : a >> 4); };
    auto below = [](int h) -> uint32_t { return h >= 32 ? 0xffffffffu : ((1u << h) - 1u); };
    uint32_t ibits = 0xffffffffu, obits = 0xffffffffu;
    if (BOXED) {
        ibits = below(ceil16(in_hi - t)) & ~below(ceil16(in_lo - t));
        if (has_zb && d.tile_range_mode != 1) ibits &= rcd.bits();
    }
    // Wave-uniform version of the input mask (the OR over the wave's 64 / W values of t): where a bit is clear NO lane of the
    // wave wants that element, and the load instruction itself is skipped by a scalar branch -- a k-space column is
    // mostly unflagged rows (72 % on the headline problem), and an instruction whose lanes are all out of range still
    // costs its issue slot and its pass through the address unit.
    // (Unweighted passes only: the weighted run-time-box x pass went from 155 to 198 registers with them -- three waves per SIMD
    // to two, 1.08 -> 1.86 ms on config 5.  C++ branches around the STORES of the padded z pass cost it its register
    // allocation -- 172 bytes of spills at the 128-register cap -- and predicating them through the execution mask, one opaque
    // v_and / v_cmp / s_and_saveexec / store / s_mov exec block per element, was slower than the out-of-range offsets
    // (padded y pass 1.08 -> 1.14 ms, config 5 41.9 -> 44.3 ms).  What works for the stores is the scalar branch INSIDE one
    // opaque block per store, buf_st_gated below: no control flow for the compiler, padded z pass 1.21 -> 1.14 ms.)
    uint32_t gin = 0xffffffffu;
    if (BOXED && !AXIS0 && WMODE == 0 && !HALF_IN && HALF != 4) {
        gin = 0;
#pragma unroll
        for (int l = 0; l < 64; l += W) gin |= (uint32_t)__builtin_amdgcn_readlane((int)ibits, l);
    }
    constexpr bool GATE_ST = BOXED && !AXIS0 && WMODE == 0 && !HALF_OUT && HALF != 3;
    uint32_t gout = 0xffffffffu;
    // ---- stage 1: radix R1 on inputs j = t + k*R2, results (times w_n^{t k}) to the exchange
    // Strided passes re-base their descriptors once per GRP elements -- one 64-bit scalar add, hidden from the compiler, which
    // otherwise recomputes base + k * step with two 32-bit multiplies, a high multiply and their adds for every element -- and
    // reach the elements in between through the instruction's scalar offset (1, 2, 3 steps: three registers per stream, computed
    // once).  Per-element re-basing was a third of the scalar instructions of these kernels, and a wave issues one instruction
    // of any kind per turn.  (GRP * 16 - 1 element steps plus the tile's lanes must fit the 2 GB window: launch_2stage checks.)
    constexpr int GRP = 4;
    auto opaque = [](const float2*& q) { asm("" : "+s"(q)); };
    cx v[R1];
    {
        cx wv[R1];
        constexpr int K0 = HALF_IN ? R1 / 4 : 0, K1 = HALF_IN ? 3 * R1 / 4 : R1;
        const int64_t st_in = (int64_t)R2 * d.in_sj, st_w = WMODE == 1 ? (int64_t)R2 * d.w_sj : 0;
        const unsigned so_in = (unsigned)st_in * 8u, so_w = (unsigned)st_w * 8u;
        const float2* p_in = AXIS0 ? b_in : b_in + K0 * st_in;
        const float2* p_w = (AXIS0 || WMODE != 1) ? b_w : b_w + K0 * st_w;
        rsrc_t g_in = make_rsrc(p_in), g_w = make_rsrc(p_w);
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            const int g = (k - K0) % GRP;
            if (!AXIS0 && g == 0 && k > K0) {
                p_in += GRP * st_in; opaque(p_in); g_in = make_rsrc(p_in);
                if (WMODE == 1) { p_w += GRP * st_w; opaque(p_w); g_w = make_rsrc(p_w); }
            }
            const bool stat = !BOXED || HALF_IN || HALF == 4;                 // box known at compile time
            // off = all ones (out of range) where the element is not wanted: one bit-field extract + one or
            const unsigned off = stat ? 0u : (unsigned)__builtin_amdgcn_sbfe((int)~ibits, k, 1);
            if (!stat && !AXIS0 && WMODE == 0 && !((gin >> k) & 1u)) v[k] = mk(0.f, 0.f);
            else if (AXIS0) {
                v[k] = from2(buf_ld<NT_LD>(r_in, l_in | off, (unsigned)(k * R2) * 8u));
                if (WMODE == 1) wv[k] = from2(buf_ld<false>(r_w, l_w | off, (unsigned)(k * R2) * 8u));
            } else {
                v[k] = from2(buf_ld<NT_LD>(g_in, l_in | off, (unsigned)g * so_in));
                if (WMODE == 1) wv[k] = from2(buf_ld<false>(g_w, l_w | off, (unsigned)g * so_w));
            }
        }
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            if (WMODE == 1) v[k] = cxmul_r(v[k], wv[k]);
            if (inv) v[k] = cconj(v[k]);
        }
    }
    // ---- the output side of the boxes, behind the loads (the half-input y pass has not waited for its record until here)
    if (defer_out) {
        trg = rcd.range();
        out_lo = out_lo > trg.x ? out_lo : trg.x;
        out_hi = out_hi < trg.y ? out_hi : trg.y;
    }
    if (BOXED) {
        obits = below(ceil16(out_hi - t)) & ~below(ceil16(out_lo - t));
        if (has_zb && d.tile_range_mode == 1) obits &= rcd.bits();
    }
    // the wave-uniform mask for the stores of unweighted passes with a run-time output box -- as ONE opaque block per store (buf_st_gated)
    if (GATE_ST) {
        gout = 0;
#pragma unroll
        for (int l = 0; l < 64; l += W) gout |= (uint32_t)__builtin_amdgcn_readlane((int)obits, l);
    }
    if (TW_LATE) {
#pragma unroll
        for (int i = 0; i < TWN; ++i) tws[tid + i * NT] = tw_mine[i];
    }
    __syncthreads();            // twiddle table visible (the global loads above are already in flight)
    if (HALF_IN) PFFTHalfIn<R1>::run(v);
    else PFFT<R1, SINV>::run(v);
    {
        const float2* __restrict__ twt = tws + t * R1;
#pragma unroll
        for (int k = 1; k < R1; ++k) v[k] = SINV ? cxmulc(from2(twt[k]), v[k]) : cxmul_r(v[k], from2(twt[k]));
    }

    // ---- exchange + stage 2 in B2 = R1/16 rounds.  Stage-2 butterfly b2 = t + 16*q needs, from every
    // stage-1 thread b, exactly its output k = b2: round q therefore moves only the outputs
    // k in [16q, 16q+16) through LDS, i.e. 16 x 16 x W elements = 32 KB per round whatever R1 is.
    // Exchange element (b, kk) lives at row b, slot kk.  Strided axes: [b][kk][w] (lanes along w).  Axis 0: column
    // group w owns 16 rows of 17 slots -- the odd row stride makes both the row-wise writes and the column-wise
    // reads conflict-free, and every address is one per-thread base plus a compile-time offset.
    // (Round 5, not adopted: handing a round over in two halves of 8 slots makes the 32-column tiles' image 32 KB -- three workgroups
    // per CU by LDS -- but a thread then holds its 24 outputs still to hand over beside the 16 inputs of its second transform:
    // at the 80 registers six waves per SIMD allow the kernels spill 18 ... 24 registers.  Not measured.)
    float2* __restrict__ lw = AXIS0 ? lds + w * (16 * 17) + t * 17 : lds + (t * 16) * W + w;     // + kk * (AXIS0 ? 1 : W)
    float2* __restrict__ lr = AXIS0 ? lds + w * (16 * 17) + t      : lds + t * W + w;            // + k2 * (AXIS0 ? 17 : 16 * W)
#pragma unroll
    for (int q = 0; q < B2; ++q) {
        if (q > 0) __syncthreads();             // the previous round's reads are done
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) lw[kk * (AXIS0 ? 1 : W)] = to2(v[16 * q + kk]);
        __syncthreads();
        cx u[R2];
#pragma unroll
        for (int r = 0; r < R2; ++r) u[r] = from2(lr[r * (AXIS0 ? 17 : 16 * W)]);
        PFFT<R2, SINV>::run(u);
        constexpr int Q0 = HALF_OUT ? R2 / 4 : 0, Q1 = HALF_OUT ? 3 * R2 / 4 : R2;          // outputs outside are never stored
        const int64_t st_out = (int64_t)R1 * d.out_sj, st_w = WMODE >= 2 ? (int64_t)R1 * d.w_sj : 0;
        const unsigned so_out = (unsigned)st_out * 8u, so_w = (unsigned)st_w * 8u;
        const float2* p_out = AXIS0 ? b_out : b_out + (q * T) * d.out_sj + Q0 * st_out;
        const float2* p_w = (AXIS0 || WMODE < 2) ? b_w : b_w + (q * T) * d.w_sj + Q0 * st_w;
        cx wv[R2];          // kept outputs j = t + 16*(q + r*B2): bit q + r*B2 of obits
        if (WMODE >= 2) {
            rsrc_t g_w = make_rsrc(p_w);
#pragma unroll
            for (int r = Q0; r < Q1; ++r) {
                const int g = (r - Q0) % GRP;
                if (!AXIS0 && g == 0 && r > Q0) { p_w += GRP * st_w; opaque(p_w); g_w = make_rsrc(p_w); }
                const bool stat = !BOXED || HALF_OUT || HALF == 3;
                const unsigned off = stat ? 0u : (unsigned)__builtin_amdgcn_sbfe((int)~obits, q + r * B2, 1);
                if (AXIS0) wv[r] = from2(buf_ld<false>(r_w, l_w | off, (unsigned)(q * T + r * R1) * 8u));
                else wv[r] = from2(buf_ld<false>(g_w, l_w | off, (unsigned)g * so_w));
            }
        }
        rsrc_t g_out = make_rsrc(p_out);
        v4i_t gw_out = make_rsrc_words(p_out);
#pragma unroll
        for (int r = Q0; r < Q1; ++r) {
            const int g = (r - Q0) % GRP;
            if (!AXIS0 && g == 0 && r > Q0) {
                p_out += GRP * st_out; opaque(p_out);
                if (GATE_ST) gw_out = make_rsrc_words(p_out); else g_out = make_rsrc(p_out);
            }
            const bool stat = !BOXED || HALF_OUT || HALF == 3;
            cx ac = u[r];
            if (inv) ac = cconj(ac);
            if (WMODE >= 2) ac = cxmulc(wv[r], ac);
            float2 a = to2(ac);
            const unsigned off = stat ? 0u : (unsigned)__builtin_amdgcn_sbfe((int)~obits, q + r * B2, 1);
            if (SUMW) {
                // coil combination: the SUMW sub-columns (coils) of a column sit in SUMW consecutive lanes; data-parallel
                // primitives move the partial sums (no LDS traffic, no extra registers): afterwards the group's
                // first lane holds the total (the other lanes' l_out is out of range)
                if (SUMW >= 2)  { a.x += dpp_f<0xB1>(a.x);  a.y += dpp_f<0xB1>(a.y); }      // quad_perm [1,0,3,2]
                if (SUMW >= 4)  { a.x += dpp_f<0x4E>(a.x);  a.y += dpp_f<0x4E>(a.y); }      // quad_perm [2,3,0,1]
                if (SUMW >= 8)  { a.x += dpp_f<0x104>(a.x); a.y += dpp_f<0x104>(a.y); }     // row_shl:4
                if (SUMW >= 16) { a.x += dpp_f<0x108>(a.x); a.y += dpp_f<0x108>(a.y); }     // row_shl:8
            }
            if (AXIS0) buf_st<NT_ST>(r_out, l_out | off, (unsigned)(q * T + r * R1) * 8u, a);
            else if (GATE_ST) buf_st_gated<NT_ST>(gw_out, l_out | off, (unsigned)g * so_out, a, (gout >> (q + r * B2)) & 1u);
            else buf_st<NT_ST>(g_out, l_out | off, (unsigned)g * so_out, a);
        }
    }
}

// (Round 5, measured and removed: two neighbouring tiles per workgroup, software-pipelined -- the second tile's loads issued as soon
// as the first tile's stage-1 outputs sat in the exchange image, 110 registers, no spills, bit-identical results.  Same box,
// alternating runs: 6.652 / 6.657 ms per evaluation against 6.664 / 6.678 ms; two of the four passes 2-5 % faster, two 1-3 % slower.
// profiles/r05_fft_pass_experiments.txt.)

// ---- 256^3 in TWO launches ------------------------------------------------------------------------------------
// The plain 3-D transform above is three passes = 6 x the volume in HBM traffic; the reference's own accounting
// (benchmark.py:55) prices a 3-D transform at 4 x.  For 256^3 volumes (BASELINE config 2) two launches suffice if a
// workgroup owns 16384 elements (2 x 64 registers per thread pair ... 32 complex values per thread, 512 threads) and the
// y axis is split 64 x 4 between the launches:
//
//   launch A, workgroup (z, n2):  the 64 lines y = 4*n1 + n2 of plane z.  y stage 1 (64-point DFT over n1 = 8 x 8), the
//             inter-stage twiddle w256^(n2 k1), then the full 256-point x transform (4 x 8 x 8).  Lines go out to
//             y' = k1 + 64*n2: one contiguous 128 KB block per workgroup.  Out of place.
//   launch B, workgroup (16 x, k1): the 4 rows y' = k1 + 64*n2 for all 256 z.  y stage 2 (4-point DFT over n2, giving
//             ky = k1 + 64*k2 in place), then the full 256-point z transform (8 x 32).  In place.
//
// Every stage is a register DFT on values the thread already holds; between stages the workgroup redistributes through
// LDS in four 32 KB rounds (launch A: twice across waves and once inside each wave; launch B: once).  All LDS address
// maps below are conflict-free (one lane per 8-byte slot modulo 64); tools/fft2pass_model.py is a thread-level numpy
// model of exactly these maps, checked against numpy.fft.fftn.
// Streaming hints of the two launches, as measured: launch A loads with the hint, stores WITHOUT it -- it stores 16-byte pieces,
// two instructions per 128-byte line, and a streaming hint on partial-line stores is ruinous (1.48 against 0.78 ms) --, launch B
// uses none (1.746 against 1.788 ms per 256^3 x 16 transform with them).  The blocks of both launches are renumbered so that every
// XCD walks a contiguous range.
constexpr bool F3A_NT_LD = true, F3A_NT_ST = false, F3B_NT_LD = false, F3B_NT_ST = false;
constexpr int F3_LDS_ELEMS = 64 * 72;            // largest exchange image (launch A, second exchange)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(512, 4)
k_fft3d_a(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw, int inverse) {
    extern __shared__ float2 lds[];
    float2* __restrict__ tws = lds + F3_LDS_ELEMS;
    const int tid = threadIdx.x;
    for (int k = tid; k < 256; k += 512) tws[k] = tw[k];       // (round 5: written behind the loads instead, as in k_fft_2stage, the transform took 1.73 against 1.69 ms)
    // each XCD walks a contiguous range of (n2, z, volume) triples: the four n2 workgroups of a plane -- interleaved 2 KB lines
    // of the same 512 KB -- run behind one L2 at about the same time
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), tot = gridDim.x * gridDim.y * gridDim.z;
    const unsigned logical = (lin & 7u) * (tot >> 3) + (lin >> 3);            // (tot is a multiple of 8: 256 * 4 * batch)
    const int n2 = (int)(logical & 3u), z = (int)((logical >> 2) & 255u);
    const int64_t base = ((int64_t)(logical >> 10) << 24) + ((int64_t)z << 16);
    const int xl = tid & 63, a = tid >> 6;
    const bool inv = inverse != 0;

    // role 0: lane = x mod 64, wave = a.  v[j][b]: line n1 = a + 8b, x = xl + 64j (a wave load = 512 contiguous bytes)
    // (buffer loads: descriptor at the wave's first line, one 32-bit lane offset, the (b, j) displacement as a scalar / immediate
    // offset -- 32 loads in flight without 32 address register pairs)
    cx v[4][8];
    {
        const int a_u = __builtin_amdgcn_readfirstlane(a);              // the wave index is wave-uniform
        const rsrc_t r_in = make_rsrc(in + base + 256 * n2 + 1024 * a_u);
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j][b] = from2(buf_ld<F3A_NT_LD>(r_in, (unsigned)xl * 8u, (unsigned)(64 * j + 8192 * b) * 8u));
                if (inv) v[j][b] = cconj(v[j][b]);
            }
    }
    __syncthreads();                                   // twiddle table visible
    // y stage 1a: 8-point DFT over b, twiddle w64^(a kb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        PFFT<8, false, false>::run(v[j]);
#pragma unroll
        for (int kb = 1; kb < 8; ++kb) v[j][kb] = cxmul(v[j][kb], from2(tws[(4 * a * kb) & 255]));
    }
    // exchange 1 (a <-> kb between waves, lane kept): image [a][kb][xl], two images (two j) per round
    cx r[4][8];
    float2* __restrict__ img1 = lds + F3_LDS_ELEMS + 256;         // second image (behind the twiddle table)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        if (jp) __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            lds[(a * 8 + kb) * 64 + xl] = to2(v[2 * jp][kb]);
            img1[(a * 8 + kb) * 64 + xl] = to2(v[2 * jp + 1][kb]);
        }
        __syncthreads();
#pragma unroll
        for (int ap = 0; ap < 8; ++ap) {
            r[2 * jp][ap] = from2(lds[(ap * 8 + a) * 64 + xl]);
            r[2 * jp + 1][ap] = from2(img1[(ap * 8 + a) * 64 + xl]);
        }
    }
    const int kb = a;                                  // role 1: the wave index now names kb
    // y stage 1b: 8-point DFT over a -> line k1 = kb + 8 ka; inter-launch twiddle w256^(n2 k1); x stage 1: 4-point DFT over j
    cx p[8][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) PFFT<8, false, false>::run(r[j]);
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) {
        const cx wy = from2(tws[(n2 * (kb + 8 * ka)) & 255]);
        cx t4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t4[j] = cxmul(r[j][ka], wy);
        PFFT<4, false, false>::run(t4);
        p[ka][0] = t4[0];
#pragma unroll
        for (int kj = 1; kj < 4; ++kj) p[ka][kj] = cxmul(t4[kj], from2(tws[(xl * kj) & 255]));
    }
    // exchange 2: image [line][72] (x within a line); role 2 = (c = x mod 8, line): lane = c + 8*(line mod 8), wave = line / 8
    const int c = tid & 7, l = (tid >> 3) & 7, line = (tid >> 3);
    cx q[4][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        __syncthreads();
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
            lds[(kb + 8 * ka) * 72 + xl] = to2(p[ka][2 * jp]);
            img1[(kb + 8 * ka) * 72 + xl] = to2(p[ka][2 * jp + 1]);
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            q[2 * jp][d] = from2(lds[line * 72 + c + 8 * d]);
            q[2 * jp + 1][d] = from2(img1[line * 72 + c + 8 * d]);
        }
    }
    // x stage 2: 8-point DFT over d, twiddle w64^(c kd)
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
        PFFT<8, false, false>::run(q[kj]);
#pragma unroll
        for (int kd = 1; kd < 8; ++kd) q[kj][kd] = cxmul(q[kj][kd], from2(tws[(4 * c * kd) & 255]));
    }
    // exchange 3 (c <-> kd among the 8 lanes of a line: inside the wave): image [line][64], swizzled both ways
    __syncthreads();                                   // the last round of exchange 2 has been read everywhere
    const int kd3 = c;                                 // role 3: the low lane bits now name kd
    cx t3[4][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        if (jp) wave_sync();
#pragma unroll
        for (int kd = 0; kd < 8; ++kd) {
            lds[line * 64 + 8 * ((c + l) & 7) + ((kd + l) & 7)] = to2(q[2 * jp][kd]);
            img1[line * 64 + 8 * ((c + l) & 7) + ((kd + l) & 7)] = to2(q[2 * jp + 1][kd]);
        }
        wave_sync();
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            t3[2 * jp][cc] = from2(lds[line * 64 + 8 * ((cc + l) & 7) + ((kd3 + l) & 7)]);
            t3[2 * jp + 1][cc] = from2(img1[line * 64 + 8 * ((cc + l) & 7) + ((kd3 + l) & 7)]);
        }
    }
    // x stage 3: 8-point DFT over c -> kx = kj + 4 kd + 32 kc; four adjacent kx per thread and kc: two 16-byte stores
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) PFFT<8, false, false>::run(t3[kj]);
    const rsrc_t r_out = make_rsrc(out + base + 256 * 64 * n2);
    const unsigned l_out = (unsigned)(256 * line + 4 * kd3) * 8u;
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            cx e0 = t3[2 * h][kc], e1 = t3[2 * h + 1][kc];
            if (inv) { e0 = cconj(e0); e1 = cconj(e1); }
            // the displacement goes into the instruction's immediate offset (lane offset + constant), NOT into the scalar
            // offset operand: a 16-byte buffer store with a scalar-register offset whose data registers are overwritten by the
            // next VALU instruction stored stale upper lanes on gfx950 (the compiler only pads that hazard for immediate offsets)
            buf_st_f4<F3A_NT_ST>(r_out, l_out + (unsigned)(32 * kc + 2 * h) * 8u, make_float4(e0.v.x, e0.v.y, e1.v.x, e1.v.y));
        }
    }
}

__global__ void __launch_bounds__(512, 4)
k_fft3d_b(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw, int inverse) {
    extern __shared__ float2 lds[];
    float2* __restrict__ tws = lds + 2 * F3_LDS_ELEMS;
    const int tid = threadIdx.x;
    for (int k = tid; k < 256; k += 512) tws[k] = tw[k];       // (round 5: written behind the loads instead, as in k_fft_2stage, the transform took 1.73 against 1.69 ms)
    // blocks are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of (x tile, k1, volume) triples, so that the
    // sixteen x tiles of a row group -- adjacent 128-byte pieces of the same 2 KB rows -- run behind one L2 at about the same time
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), tot = gridDim.x * gridDim.y * gridDim.z;
    const unsigned logical = (lin & 7u) * (tot >> 3) + (lin >> 3);            // (tot is a multiple of 8: 16 * 64 * batch)
    const int xs = (int)(logical & 15u) * 16, k1 = (int)((logical >> 4) & 63u);
    const int64_t base = ((int64_t)(logical >> 10) << 24) + xs;
    // role 0: h = lane bit 0 picks the rows n2 = h, h + 2 (the other two live in the neighbouring lane); w = x; t = z mod 16
    const int h = tid & 1, w = (tid >> 1) & 15, t = tid >> 5;
    const bool inv = inverse != 0;
    cx v[2][16];                                       // [m][k]: row n2 = h + 2m, z = t + 16k
    const unsigned l_io = ((unsigned)w + ((unsigned)t << 16) + (unsigned)h * (256u * 64u)) * 8u;   // x + 65536*(z mod 16) + the row of h
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            // descriptor re-based per (k, m): a scalar add; the lane offset stays below 8.6 MB
            v[m][k] = from2(buf_ld<F3B_NT_LD>(make_rsrc(in + base + 256 * (k1 + 128 * m) + ((int64_t)(16 * k) << 16)), l_io, 0));
            if (inv) v[m][k] = cconj(v[m][k]);
        }
    __syncthreads();
    // y stage 2: 4-point DFT over n2 = h + 2m -> ky = k1 + 64 k2, k2 = q + 2h.  In the thread: the sum / difference over m
    // (q = 0, 1) and the twiddle w4^(h q) = -i for h = q = 1; then one exchange with the neighbouring lane (h ^ 1):
    // h = 0 keeps own + partner (k2 = q), h = 1 keeps partner - own (k2 = q + 2).
    cx y[2][16];                                       // [q][k]
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        cx a0 = v[0][k] + v[1][k], a1 = v[0][k] - v[1][k];
        if (h) a1 = cmul_mi_c(a1);
        cx p0, p1;
        p0.v = v2f{dpp_f<0xB1>(a0.v.x), dpp_f<0xB1>(a0.v.y)};      // quad_perm [1,0,3,2]: the lane with the other h
        p1.v = v2f{dpp_f<0xB1>(a1.v.x), dpp_f<0xB1>(a1.v.y)};
        y[0][k] = h ? p0 - a0 : a0 + p0;
        y[1][k] = h ? p1 - a1 : a1 + p1;
    }
    __builtin_amdgcn_sched_barrier(0);
    // z stage 1: 16-point DFT over k, twiddle w256^(t kk); exchange (t <-> kk among the threads of one (x, h)); z stage 2:
    // 16-point DFT over t -> kz = kk + 16 rr.  One round per q: every thread writes 16 and reads 16 values.
    const int kk1 = t;                                 // role 1: the same thread index now names kk
    const int k2_base = 2 * h;
    const unsigned l_st = ((unsigned)w + ((unsigned)kk1 << 16) + (unsigned)k2_base * (256u * 64u)) * 8u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        PFFT<16, false, false>::run(y[q]);
        if (q) __syncthreads();
        // twiddle and hand over four values at a time; the compiler-level fences keep the scheduler from hoisting all
        // fifteen twiddle loads (30 registers) above the transform, which spilled y[1] at the 128-register cap
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int kk = 4 * g; kk < 4 * g + 4; ++kk) {
                if (kk) y[q][kk] = cxmul(y[q][kk], from2(tws[(t * kk) & 255]));
                lds[t * 544 + kk * 32 + h * 16 + w] = to2(y[q][kk]);
            }
            asm volatile("" ::: "memory");
        }
        __syncthreads();
        cx r[16];
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) r[tt] = from2(lds[tt * 544 + kk1 * 32 + h * 16 + w]);
        PFFT<16, false, false>::run(r);
        const float2* const ob = out + base + 256 * (k1 + 64 * q);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            cx e = r[rr];
            if (inv) e = cconj(e);
            buf_st<F3B_NT_ST>(make_rsrc(ob + ((int64_t)(16 * rr) << 16)), l_st, 0, to2(e));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// AXIS0: inner == 1 (columns are contiguous lines of n elements).
template <bool AXIS0>
__global__ void __launch_bounds__(1024)
k_fft_lds(const float2* __restrict__ x, float2* __restrict__ y, const float2* __restrict__ tw,
          int n, int64_t inner, int64_t ncols, int W, int T, int nstages, Radices rad, int inverse) {
    extern __shared__ float2 lds[];
    const int WP = W + 1;
    float2* __restrict__ tws = lds + (size_t)n * WP;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int64_t c0 = (int64_t)blockIdx.x * W;
    const int wcount = (int)((ncols - c0 < W) ? (ncols - c0) : W);

    for (int k = tid; k < n; k += nth) tws[k] = tw[k];

    const int w = tid % W, t = tid / W;
    int64_t colbase = 0;
    if (AXIS0) {
        const float2* __restrict__ src = x + c0 * n;
        const int cnt = wcount * n;
        for (int e = tid; e < cnt; e += nth) {
            const int ww = e / n, j = e - ww * n;
            float2 v = src[e];
            if (inverse) v.y = -v.y;
            lds[j * WP + ww] = v;
        }
    } else {
        if (w < wcount) {
            const int64_t col = c0 + w;
            const int64_t o = col / inner, i = col - o * inner;
            colbase = i + inner * n * o;
            const float2* __restrict__ src = x + colbase;
            for (int j = t; j < n; j += T) {
                float2 v = src[(int64_t)j * inner];
                if (inverse) v.y = -v.y;
                lds[j * WP + w] = v;
            }
        }
    }
    __syncthreads();

    int Ns = 1;
    for (int s = 0; s < nstages; ++s) {
        const int R = rad.r[s];
        switch (R) {
            case 8: lds_stage<8>(lds, tws, n, Ns, T, t, w, WP); break;
            case 4: lds_stage<4>(lds, tws, n, Ns, T, t, w, WP); break;
            case 2: lds_stage<2>(lds, tws, n, Ns, T, t, w, WP); break;
            case 3: lds_stage<3>(lds, tws, n, Ns, T, t, w, WP); break;
            case 5: lds_stage<5>(lds, tws, n, Ns, T, t, w, WP); break;
            case 7: lds_stage<7>(lds, tws, n, Ns, T, t, w, WP); break;
            default: break;   // the planner never emits other radices for this kernel
        }
        Ns *= R;
    }

    if (AXIS0) {
        float2* __restrict__ dst = y + c0 * n;
        const int cnt = wcount * n;
        for (int e = tid; e < cnt; e += nth) {
            const int ww = e / n, j = e - ww * n;
            float2 v = lds[j * WP + ww];
            if (inverse) v.y = -v.y;
            dst[e] = v;
        }
    } else {
        if (w < wcount) {
            float2* __restrict__ dst = y + colbase;
            for (int j = t; j < n; j += T) {
                float2 v = lds[j * WP + w];
                if (inverse) v.y = -v.y;
                dst[(int64_t)j * inner] = v;
            }
        }
    }
}

// One Stockham stage through global memory, one thread per output element.
__global__ void __launch_bounds__(256)
k_fft_generic_stage(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw,
                    int64_t n, int64_t inner, int64_t total, int R, int64_t Ns, int inverse) {
    const int64_t nb = n / R;
    const int64_t tstride = nb / Ns;
    const int64_t rstride = n / R;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t i = idx % inner;
        const int64_t rest = idx / inner;
        const int64_t jo = rest % n;
        const int64_t o = rest / n;
        const int64_t m = jo % Ns;
        const int64_t k = (jo / Ns) % R;
        const int64_t b = (jo / (Ns * R)) * Ns + m;
        const float2* __restrict__ src = in + i + inner * n * o;
        float2 acc = make_float2(0.f, 0.f);
        for (int q = 0; q < R; ++q) {
            const float2 v = src[(b + q * nb) * inner];
            const int64_t ti = (q * m * tstride + (int64_t)((q * k) % R) * rstride) % n;
            float2 wv = tw[ti];
            if (inverse) wv.y = -wv.y;
            cfma(acc, v, wv);
        }
        out[idx] = acc;
    }
}

// ---- planning -----------------------------------------------------------------

struct Chirp;
struct AxisPlan {
    int64_t n = 1, inner = 1, outer = 1;
    int kind = 2;                      // 0 LDS kernel, 1 generic stages, 2 nothing to do (n == 1), 3 two-stage (256, 512), 4 two-stage A x B,
                                       // 5 chirp-z (Bluestein) over a smooth length m >= 2 n - 1: lengths with a prime factor > 7
    std::shared_ptr<Chirp> chirp;      // kind 5
    int nstages = 0;
    int ab_A = 0, ab_B = 0;            // kind 4: n = A * B (ig_fft_ab.h)
    Radices rad{};
    std::vector<int64_t> gen_radices;  // generic path may carry large prime radices
    int W = 16, T = 1;
    size_t lds_bytes = 0;
    float2* d_tw = nullptr;
};

// Bluestein: X_k = b_k sum_j (x_j b_j) conj(b)_{k-j}, b_j = exp(-i pi j^2 / n) -- a cyclic convolution of length m >= 2 n - 1, done
// with two transforms of a length the fast kernels have.  Tables for both directions (the inverse's are the conjugates):
//   b[dir]    m entries: the chirp (input weights; only j < n is ever read)
//   bhat[dir] m entries: F_m of the wrapped conjugate chirp (the convolution kernel in the frequency domain)
//   bout[dir] m entries: b_k / m (output weights of the inverse sub-transform, which is unnormalised)
struct Chirp {
    int64_t m = 0;
    float2* d_b[2] = {nullptr, nullptr};
    float2* d_bhat[2] = {nullptr, nullptr};
    float2* d_bout[2] = {nullptr, nullptr};
    float2* d_hat_out[2] = {nullptr, nullptr};     // bhat then bout in one array (the one-launch kernel's second table)
    // (round 6) the tables of the one-launch kernel for a transform that carries a circular shift by `shift` points on its image side
    // (ig_fft_set_axis_shift): input weights, then [transformed kernel, output weights / m]
    int64_t shift = 0;
    float2* d_in_s[2] = {nullptr, nullptr};
    float2* d_hat_out_s[2] = {nullptr, nullptr};
    AxisPlan sub;                      // the length-m axis with this axis's inner / outer extents
    bool fused = false;                // ONE launch per pass (k_fft_chirp: both length-m transforms in registers / LDS; strided axes, m = A x B)
    ~Chirp() {
        for (int d = 0; d < 2; ++d) { if (d_b[d]) (void)hipFree(d_b[d]); if (d_bhat[d]) (void)hipFree(d_bhat[d]); if (d_bout[d]) (void)hipFree(d_bout[d]);
                                      if (d_hat_out[d]) (void)hipFree(d_hat_out[d]);
                                      if (d_in_s[d]) (void)hipFree(d_in_s[d]); if (d_hat_out_s[d]) (void)hipFree(d_hat_out_s[d]); }
        if (sub.d_tw) (void)hipFree(sub.d_tw);
    }
};

bool factor_lds(int64_t n, Radices& rad, int& nstages) {
    nstages = 0;
    int64_t m = n;
    auto push = [&](int r) { rad.r[nstages++] = r; };
    while (m % 8 == 0 && nstages < MAX_STAGES) { push(8); m /= 8; }
    while (m % 4 == 0 && nstages < MAX_STAGES) { push(4); m /= 4; }
    while (m % 2 == 0 && nstages < MAX_STAGES) { push(2); m /= 2; }
    for (int p : {3, 5, 7})
        while (m % p == 0 && nstages < MAX_STAGES) { push(p); m /= p; }
    return m == 1;
}

void factor_generic(int64_t n, std::vector<int64_t>& out) {
    out.clear();
    int64_t m = n;
    while (m % 8 == 0) { out.push_back(8); m /= 8; }
    while (m % 4 == 0) { out.push_back(4); m /= 4; }
    while (m % 2 == 0) { out.push_back(2); m /= 2; }
    for (int64_t p = 3; p * p <= m; p += 2)
        while (m % p == 0) { out.push_back(p); m /= p; }
    if (m > 1) out.push_back(m);
}

}  // namespace

struct ig_fft {
    ig_ctx* ctx = nullptr;
    bool is_chirp_sub = false;       // a throw-away plan made to plan the length-m axis of a chirp-z axis (no nesting)
    int rank = 0;
    int64_t dims[3] = {1, 1, 1};
    int64_t batch = 1;
    int64_t total = 0;               // elements in all volumes
    AxisPlan axis[3];
    size_t workspace_bytes = 0;
    std::string desc;
    // zero-padded / cropped plans (ig_fft_plan_padded): the image occupies box_lo .. box_lo+box_dims of the grid
    bool two_launch = false;         // 256^3 volumes: k_fft3d_a + k_fft3d_b instead of three axis passes
    size_t inplace_workspace_bytes = 0;   // two-launch transform called in place: staging volumes in the CALLER's workspace (ig_fft_inplace_workspace)
    bool padded = false;
    bool has_ab_axis = false;        // a zero-padded plan with an A x B axis (160 ... 640)
    bool has_chirp_axis = false;     // ... with a chirp-z axis (y or z): no k-space support table
    int zw_in = 16, zw_out = 16;     // words per entry of the k-space support table's bitmaps on the z axis (ig_fft_support_words)
    int layout = 0;                  // memory order of the grid: 0 = (x, y, z), 1 = (x, z, y)
    int support_tile = 16;           // kx points per entry of the k-space support table (layout 2: ig_fft_set_support_tile)
    int64_t box_lo[3] = {0, 0, 0}, box_dims[3] = {1, 1, 1};
};

namespace {

// ---- two-stage A x B passes (ig_fft_ab.h): the instantiated splits (ig_fft_ab_list.h, generated by tools/gen_ab_list.py; the
// kernels themselves are compiled in four side translation units, ig_fft_abd0..3.hip) -------------------------------------------
}  // namespace
#define IG_ABD_DECL(K_)                                                                                                              \
    int ig_ab_launch_part##K_(hipStream_t, int64_t, bool, dim3, dim3, size_t, const float2*, float2*, const float2*, int64_t, int64_t, int); \
    int ig_abd_launch_part##K_(hipStream_t, int64_t, int, dim3, dim3, const PassDesc&, const float2*);                                     \
    int ig_abz_launch_part##K_(hipStream_t, int64_t, dim3, dim3, const PassDesc&, const float2*);
IG_ABD_DECL(0) IG_ABD_DECL(1) IG_ABD_DECL(2) IG_ABD_DECL(3)
#undef IG_ABD_DECL
namespace {
bool ab_split(int64_t n, int& A, int& B) {
#define IG_AB_CASE(A_, B_, R_) if (n == (A_) * (B_)) { A = A_; B = B_; return true; }
    IG_AB_LIST(IG_AB_CASE)
#undef IG_AB_CASE
    return false;
}
size_t ab_lds(int64_t n, bool axis0) {
#define IG_AB_CASE(A_, B_, R_) if (n == (A_) * (B_)) return axis0 ? anyfft::ab_lds_bytes<A_, B_, R_, true>() : anyfft::ab_lds_bytes<A_, B_, R_, false>();
    IG_AB_LIST(IG_AB_CASE)
#undef IG_AB_CASE
    return 0;
}
int launch_ab(ig_ctx* ctx, const AxisPlan& ax, const float2* in, float2* out, int inverse) {
    const int64_t ncols = ax.inner * ax.outer;
    const int64_t blocks = (ncols + anyfft::AB_W - 1) / anyfft::AB_W;
    IG_REQUIRE(ctx, blocks <= 0x7fffffffLL, "ig_fft_exec: too many tiles");
    const dim3 grid((unsigned)blocks), block((unsigned)(anyfft::AB_W * ax.ab_B));
    const bool ax0 = ax.inner == 1;
    int r = ig_ab_launch_part0(ctx->stream, ax.n, ax0, grid, block, ax.lds_bytes, in, out, ax.d_tw, ax.inner, ncols, inverse);
    if (!r) r = ig_ab_launch_part1(ctx->stream, ax.n, ax0, grid, block, ax.lds_bytes, in, out, ax.d_tw, ax.inner, ncols, inverse);
    if (!r) r = ig_ab_launch_part2(ctx->stream, ax.n, ax0, grid, block, ax.lds_bytes, in, out, ax.d_tw, ax.inner, ncols, inverse);
    if (!r) r = ig_ab_launch_part3(ctx->stream, ax.n, ax0, grid, block, ax.lds_bytes, in, out, ax.d_tw, ax.inner, ncols, inverse);
    if (r == 2) return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_fft_exec: device %d refused %zu bytes of LDS for the %lld-point kernel", ctx->device, ax.lds_bytes, (long long)ax.n);
    if (!r) return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_fft_exec: no A x B kernel for n = %lld", (long long)ax.n);
    IG_LAUNCH_CHECK(ctx, "k_fft_ab");
    return IG_OK;
}

int plan_chirp(ig_ctx* ctx, AxisPlan& ax, int64_t m);

// The A x B length m >= 2 n - 1 a chirp-z axis of n points runs over (0: none up to 1024), and its split.
// (measured on 640 x 277 x 410 x 8: what a pass costs follows m AND the split -- 410 over m = 840 = 28 x 30 9.6 ms, over
// 864 = 27 x 32 8.6 ms, over 896 = 28 x 32 8.5 ms; 277 over 560 = 20 x 28 5.7 ms, over 576 = 24 x 24 7.5 ms: whole waves
// (B a multiple of 4) beat the shortest m -- so: the first length within 8 % of 2 n - 1 whose B is a multiple of 4, else the shortest)
int64_t chirp_ab_length(int64_t n, int& A, int& B) {
    int64_t m = 0;
    for (int64_t c = 2 * n - 1; c <= 1024 && c * 100 <= (2 * n - 1) * 108; ++c)
        if (ab_split(c, A, B)) { if (!m) m = c; if (B % 4 == 0) { m = c; break; } }
    for (int64_t c = 2 * n - 1; c <= 1024 && !m; ++c) if (ab_split(c, A, B)) m = c;
    if (m) ab_split(m, A, B);
    return m;
}

int plan_axis(ig_ctx* ctx, ig_fft* p, int a) {
    AxisPlan& ax = p->axis[a];
    ax.n = p->dims[a];
    ax.inner = 1;
    for (int d = 0; d < a; ++d) ax.inner *= p->dims[d];
    ax.outer = p->total / (ax.n * ax.inner);
    if (ax.n == 1) { ax.kind = 2; return IG_OK; }

    const bool force_generic = ctx->opt_fft_kernels == 2;
    Radices rad{};
    int ns = 0;
    const bool two_stage = !force_generic && (ax.n == 512 || ax.n == 256) &&
                           32 * ax.inner * 8 < 0x7fffffffLL;              // 2 GB descriptor window per 16 elements of a column
    if (two_stage) {
        ax.kind = 3; ax.W = 16; ax.T = 16; ax.nstages = 2;
        ax.rad.r[0] = ax.n == 512 ? 32 : 16; ax.rad.r[1] = 16;
        ax.lds_bytes = ((size_t)16 * 17 * ax.W + ax.n) * 8;     // one exchange round (padded rows, x passes) + the twiddle table
        // 36 KB of dynamic LDS: below the 64 KB every kernel may use without an attribute
    }
    // lengths with a register-resident A x B split (the oversampled grids of the reference's example and their like)
    const bool use_ab = ctx->opt_fft_kernels == 0;
    bool ab = false;
    if (!two_stage && !force_generic && use_ab) {
        int A = 0, B = 0;
        if (ab_split(ax.n, A, B)) {
            ab = true;
            ax.kind = 4; ax.ab_A = A; ax.ab_B = B; ax.W = anyfft::AB_W; ax.T = B; ax.nstages = 2;
            ax.rad.r[0] = A; ax.rad.r[1] = B;
            ax.lds_bytes = ab_lds(ax.n, ax.inner == 1);
        }
    }
    bool lds_ok = !ab && !two_stage && !force_generic && ax.n <= LDS_NMAX && factor_lds(ax.n, rad, ns);
    if (lds_ok) {
        int T = 1;
        for (int s = 0; s < ns; ++s) {
            const int R = rad.r[s];
            const int bpt = E / R;
            const int nb = (int)(ax.n / R);
            const int need = (nb + bpt - 1) / bpt;
            if (need > T) T = need;
        }
        // widest tile (multiple of 16 columns = 128-byte segments for strided axes) within the
        // thread and LDS limits; narrow it for long transforms
        int W = 16;
        while (W > 1 && ((int64_t)W * T > 1024 || ((size_t)ax.n * (W + 1) + ax.n) * 8 > 72 * 1024)) W >>= 1;
        while (W < 64 && (int64_t)(W * 2) * T <= 256 && ((size_t)ax.n * (2 * W + 1) + ax.n) * 8 <= 72 * 1024) W <<= 1;
        const size_t lds = ((size_t)ax.n * (W + 1) + ax.n) * 8;
        if ((int64_t)W * T > 1024 || lds > LDS_BUDGET) lds_ok = false;
        else {
            ax.kind = 0; ax.rad = rad; ax.nstages = ns; ax.W = W; ax.T = T; ax.lds_bytes = lds;
            if (lds > 48 * 1024) {
                // more than the default dynamic-LDS limit: opt in (gfx950 has 160 KiB per CU)
                IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_lds<true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
                IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_lds<false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET));
            }
        }
    }
    if (!lds_ok && !two_stage && !ab) {
        ax.kind = 1;
        factor_generic(ax.n, ax.gen_radices);
        ax.nstages = (int)ax.gen_radices.size();
        // a prime factor above 7 (277, 410 = 2 * 5 * 41: what int(N * osf) of the reference's driver produces, backend.py:427-430):
        // chirp-z over the smallest smooth length m >= 2 n - 1 that a fast kernel transforms -- an A x B length up to 1024 where
        // one exists (two fused launches on strided axes), else a length of the LDS kernel up to 4096
        int64_t big = 1;
        for (int64_t r : ax.gen_radices) big = std::max(big, r);
        if (big > 7 && ax.n >= 32 && ctx->opt_fft_kernels == 0 && !p->is_chirp_sub) {        // (shorter ones: the direct stages are fine)
            int A = 0, B = 0;
            int64_t m = chirp_ab_length(ax.n, A, B);
            Radices r2{};
            int ns2 = 0;
            for (int64_t c = 2 * ax.n - 1; c <= LDS_NMAX && !m; ++c) if (factor_lds(c, r2, ns2)) m = c;
            if (m) {
                if (int rc = plan_chirp(ctx, ax, m)) return rc;
                ax.kind = 5;
            }
        }
    }
    // twiddles exp(-2 pi i k / n), rounded once from double
    // (two-stage axes: a second table behind the first, the inter-stage twiddles in the order k_fft_2stage's threads read them --
    // entry n + t * R1 + k = w_n^(t k) for thread t < 16 and stage-1 output k < R1 = n / 16)
    const size_t ntw = (size_t)ax.n * (ax.kind == 3 ? 2 : 1);
    std::vector<float2> tw(ntw);
    for (int64_t k = 0; k < ax.n; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)ax.n;
        tw[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    if (ax.kind == 3) {
        const int64_t R1 = ax.n / 16;
        for (int64_t t = 0; t < 16; ++t)
            for (int64_t k = 0; k < R1; ++k) tw[ax.n + t * R1 + k] = tw[(t * k) % ax.n];
    }
    IG_HIP(ctx, hipMalloc((void**)&ax.d_tw, sizeof(float2) * ntw));
    IG_HIP(ctx, hipMemcpy(ax.d_tw, tw.data(), sizeof(float2) * ntw, hipMemcpyHostToDevice));
    return IG_OK;
}

// The launch grid of a pass kernel: (tiles of a row, k1, k2) in three dimensions where the extents allow (the dispatch order of the
// linear numbering, without the divisions at the head of every workgroup -- pass_tile, ig_fft_ab.h), else linear.  Also notes log2 of
// a power-of-two lane split.
static dim3 pass_grid(PassDesc& d, int64_t tpr) {
    d.tpr = (unsigned)tpr;
    d.cw_log2 = -1;
    if (d.cw > 0 && (d.cw & (d.cw - 1)) == 0) { d.cw_log2 = 0; while ((1 << d.cw_log2) < d.cw) ++d.cw_log2; }
    const int64_t rows = d.ncols / d.ext0, ext2 = d.ext1 > 0 ? rows / d.ext1 : 0;
    d.grid3 = (d.ext1 >= 1 && d.ext1 <= 65535 && ext2 >= 1 && ext2 <= 65535 && d.ext1 * ext2 == rows) ? 1 : 0;
    return d.grid3 ? dim3((unsigned)tpr, (unsigned)d.ext1, (unsigned)ext2) : dim3((unsigned)(tpr * rows));
}

// launch one 2-stage axis pass (n in {256, 512}); axis0 selects the lane mapping for contiguous columns
int launch_2stage(ig_ctx* ctx, const AxisPlan& ax, const PassDesc& d_in, bool axis0, int wmode) {
    PassDesc d = d_in;
    if (d.ncols == 0) return IG_OK;
    IG_REQUIRE(ctx, !d.tile_bits || d.tile_words == 16 || d.tile_words == 0, "ig_fft: the two-stage kernel reads support bitmaps of 16 words per entry (got %d)", d.tile_words);
    const int64_t cpt = (!axis0 && d.cw) ? ax.W / d.cw : ax.W;           // columns (k0 values) per tile
    IG_REQUIRE(ctx, !d.cw || (!axis0 && d.cw <= ax.W && ax.W % d.cw == 0), "ig_fft: bad lane split");
    const int64_t tpr = (d.ext0 + cpt - 1) / cpt;
    const int64_t blocks = tpr * (d.ncols / d.ext0);           // ncols = ext0 * ext1 * ext2
    IG_REQUIRE(ctx, blocks <= 0x7fffffffLL && d.ext1 <= 0x7fffffffLL, "ig_fft: too many tiles");
    const dim3 grid = pass_grid(d, tpr);
    {   // every in-tile byte offset must stay inside the 2 GB descriptor window
        const int64_t lim = 0x7fffffffLL / 8;
        // (strided passes re-base once per 4 groups of 16 elements and reach the three groups in between through the scalar
        // offset -- at 32 elements per step on the output side of a 512-point axis: 3 * 32 + 15 element steps plus the tile's lanes must fit)
        const int64_t reach = axis0 ? ax.n + 15 : 3 * (ax.n / 16) + 15;
        const int64_t span_in = reach * d.in_sj + 15 * d.in_s[0] + 15 * d.in_sa, span_out = reach * d.out_sj + 15 * d.out_s[0] + 15 * d.out_sa;
        const int64_t span_w = wmode ? reach * d.w_sj + 15 * d.w_s[0] + 15 * d.w_sa : 0;
        IG_REQUIRE(ctx, d.in_sj >= 0 && d.out_sj >= 0 && d.in_s[0] >= 0 && d.out_s[0] >= 0 && span_in < lim && span_out < lim && span_w < lim,
                   "ig_fft: axis stride too large for the two-stage kernel");
    }
    const dim3 block((unsigned)(ax.W * ax.T));
#define IG_2S(R1_, AX0_, WM_, BX_, HF_)                                                             \
    hipLaunchKernelGGL((k_fft_2stage<R1_, 16, 16, 16, AX0_, WM_, BX_, HF_>), grid, block, ax.lds_bytes, ctx->stream, d, ax.d_tw)
#define IG_2S_W(R1_, AX0_)                                                                           \
    do {                                                                                             \
        if (!boxed && wmode == 0) IG_2S(R1_, AX0_, 0, false, 0);                                     \
        else if (wmode == 0) {                                                                       \
            if (half == 1) IG_2S(R1_, AX0_, 0, true, 1); else if (half == 2) IG_2S(R1_, AX0_, 0, true, 2);   \
            else if (half == 3) IG_2S(R1_, AX0_, 0, true, 3); else if (half == 4) IG_2S(R1_, AX0_, 0, true, 4); \
            else IG_2S(R1_, AX0_, 0, true, 0); }                                                     \
        else if (wmode == 1) { if (half == 3) IG_2S(R1_, AX0_, 1, true, 3); else IG_2S(R1_, AX0_, 1, true, 0); } \
        else { if (half == 4) IG_2S(R1_, AX0_, 2, true, 4); else IG_2S(R1_, AX0_, 2, true, 0); }      \
    } while (0)
    const bool boxed = d.tile_range || !(d.in_lo <= 0 && d.in_hi >= (int)ax.n && d.out_lo <= 0 && d.out_hi >= (int)ax.n);
    // the image box of a 2x-oversampled grid sits at [n/4, 3n/4): compile-time-pruned variants
    const int qn = (int)ax.n / 4;
    int half = 0;
    if (boxed) {
        const bool in_full = d.in_lo <= 0 && d.in_hi >= (int)ax.n, out_full = d.out_lo <= 0 && d.out_hi >= (int)ax.n;
        if (d.in_lo == qn && d.in_hi == 3 * qn && !(d.tile_range && d.tile_range_mode == 2))
            half = (out_full && !d.tile_range && !d.tile_bits) ? 3 : 1;
        else if (d.out_lo == qn && d.out_hi == 3 * qn && !(d.tile_range && d.tile_range_mode == 1))
            half = (in_full && !d.tile_range && !d.tile_bits) ? 4 : 2;
        if (wmode == 1 && half != 3) half = 0;      // weighted variants exist for the fully static boxes only
        if (wmode >= 2 && half != 4) half = 0;
        if (((half == 1 || half == 3) && d.inverse) || ((half == 2 || half == 4) && !d.inverse)) half = 0;   // direction is baked in
    }
    // 32-column tiles (256-byte segments) pay where a side of the pass runs at a huge stride (y passes of the
    // interleaved layout, 16 MB per element: -11...-15 %), for the half-input variants, which fit 128 VGPRs, and -- since
    // round 3 skips the load instructions no lane wants -- for the half-output variants too (cropped z pass 1.28 -> 1.20 ms;
    // before that skip the wider tile lost there: 1.63 against 1.50 ms).
    const bool big_stride = (d.in_sj > d.out_sj ? d.in_sj : d.out_sj) * 8 >= (1 << 20);
    // boxed passes WITHOUT a compile-time half box (e.g. the 320-point box of a 512-point axis, oversampling 1.6) also take
    // 32-column tiles when one side runs at a huge stride: cropped y pass of config 5 0.85 -> 0.74 ms, padded y pass unchanged
    // ... and so do plain (unboxed) passes at a huge stride, through the same run-time-box variant: the z pass of a plain 512^3
    // transform steps 2 MB per element (3.82 -> 3.08 ms for 512^3 x 8)
    // ... and the z passes of such a box once the support bitmap gates their loads / stores (config 5: see DESIGN 3.1)
    const bool w32_generic = half == 0 && (big_stride || d.tile_bits != nullptr);
    if (ax.n == 512 && !axis0 && wmode == 0 && !d.cw && d.ext0 % 32 == 0 &&
        (half == 1 || half == 3 || half == 2 || half == 4 || w32_generic) &&
        (!d.tile_range || d.tile_shift >= 1)) {
        // 32-column tiles: 256-byte segments per row, 512 threads, 69.6 KB of LDS (2 workgroups per CU)
        PassDesc d2 = d;
        if (d2.tile_range) d2.tile_shift = d.tile_shift - 1;
        const dim3 g2 = pass_grid(d2, d.ext0 / 32), b2(512);
        const size_t lds2 = ((size_t)16 * 16 * 32 + 512) * 8;
        if (!ctx->fft_w32_attr) {          // per context (= per device): the attribute is a per-device property
            IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_2stage<32, 16, 16, 32, false, 0, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_2stage<32, 16, 16, 32, false, 0, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_2stage<32, 16, 16, 32, false, 0, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_2stage<32, 16, 16, 32, false, 0, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft_2stage<32, 16, 16, 32, false, 0, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            ctx->fft_w32_attr = true;
        }
        if (half == 0) hipLaunchKernelGGL((k_fft_2stage<32, 16, 16, 32, false, 0, true, 0>), g2, b2, lds2, ctx->stream, d2, ax.d_tw);
        else if (half == 1) hipLaunchKernelGGL((k_fft_2stage<32, 16, 16, 32, false, 0, true, 1>), g2, b2, lds2, ctx->stream, d2, ax.d_tw);
        else if (half == 3) hipLaunchKernelGGL((k_fft_2stage<32, 16, 16, 32, false, 0, true, 3>), g2, b2, lds2, ctx->stream, d2, ax.d_tw);
        else if (half == 2) hipLaunchKernelGGL((k_fft_2stage<32, 16, 16, 32, false, 0, true, 2>), g2, b2, lds2, ctx->stream, d2, ax.d_tw);
        else hipLaunchKernelGGL((k_fft_2stage<32, 16, 16, 32, false, 0, true, 4>), g2, b2, lds2, ctx->stream, d2, ax.d_tw);
    } else
    if (wmode == 3) {
        IG_REQUIRE(ctx, !axis0 && d.cw >= 1, "ig_fft: the coil-summing pass is a strided pass with a lane split");
#define IG_2S_SUM(R1_, HF_) do { switch (d.cw) { case 1: IG_2S(R1_, false, 3, true, HF_); break; case 2: IG_2S(R1_, false, 4, true, HF_); break; \
                                 case 4: IG_2S(R1_, false, 5, true, HF_); break; case 8: IG_2S(R1_, false, 6, true, HF_); break;              \
                                 default: IG_2S(R1_, false, 7, true, HF_); break; } } while (0)
        if (ax.n == 512) { if (half == 4) IG_2S_SUM(32, 4); else IG_2S_SUM(32, 0); }
        else             { if (half == 4) IG_2S_SUM(16, 4); else IG_2S_SUM(16, 0); }
#undef IG_2S_SUM
    } else
    if (ax.n == 512) { if (axis0) IG_2S_W(32, true); else IG_2S_W(32, false); }
    else             { if (axis0) IG_2S_W(16, true); else IG_2S_W(16, false); }
#undef IG_2S_W
#undef IG_2S
    IG_LAUNCH_CHECK(ctx, "k_fft_2stage");
    return IG_OK;
}

// ---- zero-pad-aware passes on A x B axes (k_fft_ab_desc): the lengths with a descriptor-driven instantiation -------------
bool abd_supported(int64_t n) {
#define IG_ABD_CASE(A_, B_, R_) if (n == (A_) * (B_)) return true;
    IG_ABD_LIST(IG_ABD_CASE)
#undef IG_ABD_CASE
    return false;
}
int launch_ab_desc(ig_ctx* ctx, const AxisPlan& ax, const PassDesc& d_in, int wmode) {
    PassDesc d = d_in;
    if (d.ncols == 0) return IG_OK;
    IG_REQUIRE(ctx, !d.cw || (d.cw <= anyfft::AB_W && anyfft::AB_W % d.cw == 0), "ig_fft: bad lane split");
    IG_REQUIRE(ctx, !d.tile_bits || d.tile_words == (d.tile_range_mode == 1 ? ax.ab_A : ax.ab_B), "ig_fft: the support bitmap has %d words per entry, this pass needs %d",
               d.tile_words, d.tile_range_mode == 1 ? ax.ab_A : ax.ab_B);
    const int64_t cpt = d.cw ? anyfft::AB_W / d.cw : anyfft::AB_W;
    const int64_t tpr = (d.ext0 + cpt - 1) / cpt;
    const int64_t blocks = tpr * (d.ncols / d.ext0);
    IG_REQUIRE(ctx, blocks <= 0x7fffffffLL && d.ext1 <= 0x7fffffffLL, "ig_fft: too many tiles");
    const dim3 grid = pass_grid(d, tpr);
    {   // a thread's lane offset reaches (B - 1) element steps + the tile's lanes: inside the 2 GB descriptor window
        const int64_t lim = 0x7fffffffLL / 8, reach = ax.ab_B - 1;
        const int64_t span_in = reach * d.in_sj + 15 * d.in_s[0] + 15 * d.in_sa, span_out = reach * d.out_sj + 15 * d.out_s[0] + 15 * d.out_sa;
        const int64_t span_w = wmode ? reach * d.w_sj + 15 * d.w_s[0] + 15 * d.w_sa : 0;
        IG_REQUIRE(ctx, d.in_sj >= 0 && d.out_sj >= 0 && d.in_s[0] >= 0 && d.out_s[0] >= 0 && span_in < lim && span_out < lim && span_w < lim,
                   "ig_fft: axis stride too large for the two-stage kernel");
    }
    int wm = wmode;
    if (wmode == 3) {
        IG_REQUIRE(ctx, d.cw == 2 || d.cw == 4 || d.cw == 8 || d.cw == 16, "ig_fft: the coil-summing A x B pass takes 2, 4, 8 or 16 interleaved coils");
        wm = d.cw == 2 ? 4 : d.cw == 4 ? 5 : d.cw == 8 ? 6 : 7;
    }
    // (Round 5, measured and removed: 32-column tiles on these passes, as on the power-of-two kernel.  Image 256^3 x 8 coils on the
    // 320^3 grid, same box, alternating runs: 4.03 / 4.08 ms per evaluation against 3.84 / 3.86 ms on 16-column tiles -- padded z
    // 0.56 against 0.47 ms, padded y 0.61-0.63 against 0.57: 20 threads per column make a 32-column workgroup 10 waves with an 80 KB
    // exchange image.  profiles/r05_lab_osf125_ab_w32_{1,0}.json)
    const dim3 block((unsigned)(anyfft::AB_W * ax.ab_B));
    int r = ig_abd_launch_part0(ctx->stream, ax.n, wm, grid, block, d, ax.d_tw);
    if (!r) r = ig_abd_launch_part1(ctx->stream, ax.n, wm, grid, block, d, ax.d_tw);
    if (!r) r = ig_abd_launch_part2(ctx->stream, ax.n, wm, grid, block, d, ax.d_tw);
    if (!r) r = ig_abd_launch_part3(ctx->stream, ax.n, wm, grid, block, d, ax.d_tw);
    if (r == 2) return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_fft: device %d refused the LDS the %lld-point zero-pad-aware kernel needs", ctx->device, (long long)ax.n);
    if (!r)
        return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_fft: no zero-pad-aware kernel for n = %lld", (long long)ax.n);
    IG_LAUNCH_CHECK(ctx, "k_fft_ab_desc");
    return IG_OK;
}

int plan_chirp(ig_ctx* ctx, AxisPlan& ax, int64_t m) {
    auto ch = std::make_shared<Chirp>();
    ch->m = m;
    const int64_t n = ax.n;
    {   // the length-m axis: planned like any other axis of an (inner, m, outer) array
        ig_fft tmp;
        tmp.ctx = ctx; tmp.is_chirp_sub = true; tmp.rank = 3; tmp.batch = 1;
        tmp.dims[0] = ax.inner; tmp.dims[1] = m; tmp.dims[2] = ax.outer;
        tmp.total = ax.inner * m * ax.outer;
        if (int rc = plan_axis(ctx, &tmp, 1)) return rc;
        ch->sub = tmp.axis[1];
        tmp.axis[1].d_tw = nullptr;
    }
    IG_REQUIRE(ctx, ch->sub.kind == 0 || ch->sub.kind == 3 || ch->sub.kind == 4, "ig_fft_plan: no fast kernel for the chirp-z length %lld", (long long)m);
    ch->fused = ch->sub.kind == 4 && ax.inner >= 16;
    // tables in double, rounded once
    typedef std::complex<double> cd;
    std::vector<cd> b((size_t)n), h((size_t)m, cd(0, 0)), H((size_t)m);
    for (int64_t j = 0; j < n; ++j) {
        const double ang = -M_PI * (double)((j * j) % (2 * n)) / (double)n;       // j^2 mod 2n keeps the argument small
        b[j] = cd(cos(ang), sin(ang));
    }
    h[0] = cd(1, 0);
    for (int64_t j = 1; j < n; ++j) h[j] = h[m - j] = std::conj(b[j]);
    {   // H = F_m(h): h is even, so H_k = h_0 + 2 sum_{j=1}^{n-1} h_j cos(2 pi j k / m)
        std::vector<double> cs((size_t)m);
        for (int64_t k = 0; k < m; ++k) cs[k] = cos(2.0 * M_PI * (double)k / (double)m);
        for (int64_t k = 0; k < m; ++k) {
            cd acc = h[0];
            for (int64_t j = 1; j < n; ++j) acc += 2.0 * h[j] * cs[(j * k) % m];
            H[k] = acc;
        }
    }
    std::vector<float2> t((size_t)m);
    auto upload = [&](float2*& dst, const std::function<cd(int64_t)>& f) -> int {
        for (int64_t k = 0; k < m; ++k) { const cd v = f(k); t[k] = make_float2((float)v.real(), (float)v.imag()); }
        IG_HIP(ctx, hipMalloc((void**)&dst, sizeof(float2) * (size_t)m));
        IG_HIP(ctx, hipMemcpy(dst, t.data(), sizeof(float2) * (size_t)m, hipMemcpyHostToDevice));
        return IG_OK;
    };
    for (int dir = 0; dir < 2; ++dir) {
        auto cj = [dir](cd v) { return dir ? std::conj(v) : v; };
        if (int rc = upload(ch->d_b[dir], [&](int64_t k) { return k < n ? cj(b[k]) : cd(0, 0); })) return rc;
        if (int rc = upload(ch->d_bhat[dir], [&](int64_t k) { return cj(H[k]); })) return rc;
        if (int rc = upload(ch->d_bout[dir], [&](int64_t k) { return k < n ? cj(b[k]) / (double)m : cd(0, 0); })) return rc;
        IG_HIP(ctx, hipMalloc((void**)&ch->d_hat_out[dir], sizeof(float2) * 2 * (size_t)m));
        IG_HIP(ctx, hipMemcpy(ch->d_hat_out[dir], ch->d_bhat[dir], sizeof(float2) * (size_t)m, hipMemcpyDeviceToDevice));
        IG_HIP(ctx, hipMemcpy(ch->d_hat_out[dir] + m, ch->d_bout[dir], sizeof(float2) * (size_t)m, hipMemcpyDeviceToDevice));
    }
    ax.chirp = ch;
    return IG_OK;
}

// The tables of a chirp-z axis whose zero-padded / cropped passes carry a circular shift by c points on the image side -- equivalently
// the modulation exp(2 pi i k c / n) on the k-space side: what a centred transform puts on an ODD axis (c = n / 2 rounded down; on an even
// axis it is a sign, which the gridding matrix absorbs).  With j' = j - c:
//    forward   Y_k = sum_j x_j exp(-2 pi i j' k / n) = b_k sum_j (x_j b_j') conj(b)_(k - j'),      b_t = exp(-i pi t^2 / n), any integer t
//    inverse   x_j = sum_k Y_k exp(+2 pi i j' k / n) = conj(b)_j' sum_k (Y_k conj(b)_k) b_(j' - k)     (its adjoint)
// i.e. the same convolution with the input (forward) or output (inverse) weights and the kernel's origin moved by c.
int plan_chirp_shift(ig_ctx* ctx, Chirp& ch, int64_t n, int64_t c) {
    typedef std::complex<double> cd;
    const int64_t m = ch.m;
    for (int d = 0; d < 2; ++d) {
        if (ch.d_in_s[d]) { (void)hipFree(ch.d_in_s[d]); ch.d_in_s[d] = nullptr; }
        if (ch.d_hat_out_s[d]) { (void)hipFree(ch.d_hat_out_s[d]); ch.d_hat_out_s[d] = nullptr; }
    }
    ch.shift = c;
    if (c == 0) return IG_OK;
    auto bt = [n](int64_t t) { const double ang = -M_PI * (double)((t * t) % (2 * n)) / (double)n; return cd(cos(ang), sin(ang)); };
    std::vector<double> cs((size_t)m), sn((size_t)m);
    for (int64_t k = 0; k < m; ++k) { cs[k] = cos(2.0 * M_PI * (double)k / (double)m); sn[k] = sin(2.0 * M_PI * (double)k / (double)m); }
    std::vector<float2> t((size_t)(2 * m));
    for (int dir = 0; dir < 2; ++dir) {
        // the convolution kernel h[u mod m], u = (output index) - (input index) in (-n, n)
        std::vector<cd> h((size_t)m, cd(0, 0)), H((size_t)m);
        for (int64_t u = -(n - 1); u <= n - 1; ++u)
            h[(size_t)((u % m + m) % m)] = dir == 0 ? std::conj(bt(u + c)) : bt(u - c);
        for (int64_t k = 0; k < m; ++k) {
            cd acc(0, 0);
            for (int64_t u = 0; u < m; ++u) {
                if (h[u] == cd(0, 0)) continue;
                const int64_t q = (u * k) % m;
                acc += h[u] * cd(cs[q], -sn[q]);
            }
            H[k] = acc;
        }
        // input weights (m entries; those at and beyond n are never used: inputs there are zeros that are never loaded)
        for (int64_t j = 0; j < m; ++j) {
            const cd v = j < n ? (dir == 0 ? bt(j - c) : std::conj(bt(j))) : cd(0, 0);
            t[j] = make_float2((float)v.real(), (float)v.imag());
        }
        IG_HIP(ctx, hipMalloc((void**)&ch.d_in_s[dir], sizeof(float2) * (size_t)m));
        IG_HIP(ctx, hipMemcpy(ch.d_in_s[dir], t.data(), sizeof(float2) * (size_t)m, hipMemcpyHostToDevice));
        // [transformed kernel, output weights / m]
        for (int64_t k = 0; k < m; ++k) {
            t[k] = make_float2((float)H[k].real(), (float)H[k].imag());
            const cd v = k < n ? (dir == 0 ? bt(k) : std::conj(bt(k - c))) / (double)m : cd(0, 0);
            t[m + k] = make_float2((float)v.real(), (float)v.imag());
        }
        IG_HIP(ctx, hipMalloc((void**)&ch.d_hat_out_s[dir], sizeof(float2) * 2 * (size_t)m));
        IG_HIP(ctx, hipMemcpy(ch.d_hat_out_s[dir], t.data(), sizeof(float2) * 2 * (size_t)m, hipMemcpyHostToDevice));
    }
    return IG_OK;
}

// elementwise steps of the unfused chirp-z route (contiguous axes, lengths beyond the A x B kernel)
__global__ void __launch_bounds__(256)
k_chirp_pre(const float2* __restrict__ x, float2* __restrict__ W, const float2* __restrict__ b, int64_t n, int64_t m, int64_t inner, int64_t total_m) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total_m; idx += (int64_t)gridDim.x * 256) {
        const int64_t i = idx % inner, rest = idx / inner, j = rest % m, o = rest / m;
        W[idx] = j < n ? cmul(x[i + inner * (j + n * o)], b[j]) : make_float2(0.f, 0.f);
    }
}
__global__ void __launch_bounds__(256)
k_chirp_mul(float2* __restrict__ W, const float2* __restrict__ bhat, int64_t m, int64_t inner, int64_t total_m) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total_m; idx += (int64_t)gridDim.x * 256)
        W[idx] = cmul(W[idx], bhat[(idx / inner) % m]);
}
__global__ void __launch_bounds__(256)
k_chirp_post(const float2* __restrict__ W, float2* __restrict__ y, const float2* __restrict__ bout, int64_t n, int64_t m, int64_t inner, int64_t total_n) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total_n; idx += (int64_t)gridDim.x * 256) {
        const int64_t i = idx % inner, rest = idx / inner, k = rest % n, o = rest / n;
        y[idx] = cmul(W[i + inner * (k + m * o)], bout[k]);
    }
}

// A chirp-z pass described by a PassDesc (boxes on both sides, strided columns, no weights of its own): ONE launch of k_fft_chirp
// (ig_fft_ab.h) -- both length-m transforms of the convolution run in the registers and the LDS of the workgroup that holds the
// column; nothing of length m ever reaches memory.
int launch_chirp_desc(ig_ctx* ctx, const AxisPlan& ax, const PassDesc& d_in) {
    const Chirp& ch = *ax.chirp;
    IG_REQUIRE(ctx, ch.fused, "ig_fft: this chirp-z axis has no one-launch route");
    IG_REQUIRE(ctx, !d_in.cw && (!d_in.tile_bits || d_in.tile_words == ch.sub.ab_B), "ig_fft: a chirp-z pass takes no lane split, and support bitmaps of B = %d words per entry", ch.sub.ab_B);
    IG_REQUIRE(ctx, d_in.in_lo >= 0 && d_in.in_hi <= ax.n && d_in.out_lo >= 0 && d_in.out_hi <= ax.n, "ig_fft: chirp-z boxes must lie inside the axis");
    PassDesc d = d_in;
    if (d.ncols == 0) return IG_OK;
    const int dir = d.inverse ? 1 : 0;
    d.w = ch.d_b[dir]; d.w2 = ch.d_hat_out[dir]; d.chirp_out = 0;
    if (ch.shift) { d.w = ch.d_in_s[dir]; d.w2 = ch.d_hat_out_s[dir]; d.chirp_out = 1; }
    const int64_t tpr = (d.ext0 + anyfft::AB_W - 1) / anyfft::AB_W;
    const int64_t blocks = tpr * (d.ncols / d.ext0);
    IG_REQUIRE(ctx, blocks <= 0x7fffffffLL && d.ext1 <= 0x7fffffffLL, "ig_fft: too many tiles");
    const dim3 grid = pass_grid(d, tpr);
    {
        const int64_t lim = 0x7fffffffLL / 8, reach = ch.sub.ab_B - 1;
        IG_REQUIRE(ctx, d.in_sj >= 0 && d.out_sj >= 0 && d.in_s[0] >= 0 && d.out_s[0] >= 0 && reach * d.in_sj + 15 * d.in_s[0] < lim && reach * d.out_sj + 15 * d.out_s[0] < lim,
                   "ig_fft: axis stride too large for the chirp-z kernel");
    }
    const dim3 block((unsigned)(anyfft::AB_W * ch.sub.ab_B));
    int r = ig_abz_launch_part0(ctx->stream, ch.m, grid, block, d, ch.sub.d_tw);
    if (!r) r = ig_abz_launch_part1(ctx->stream, ch.m, grid, block, d, ch.sub.d_tw);
    if (!r) r = ig_abz_launch_part2(ctx->stream, ch.m, grid, block, d, ch.sub.d_tw);
    if (!r) r = ig_abz_launch_part3(ctx->stream, ch.m, grid, block, d, ch.sub.d_tw);
    if (r == 2) return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_fft: device %d refused the LDS the chirp-z kernel over %lld points needs", ctx->device, (long long)ch.m);
    if (!r)
        return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_fft: no chirp-z kernel for m = %lld", (long long)ch.m);
    IG_LAUNCH_CHECK(ctx, "k_fft_chirp");
    return IG_OK;
}

// one pass of a zero-padded / cropped transform: the power-of-two kernel, the A x B one or a chirp-z pair, by the axis plan
int launch_pass(ig_ctx* ctx, const AxisPlan& ax, const PassDesc& d, bool axis0, int wmode) {
    if (ax.kind == 3) return launch_2stage(ctx, ax, d, axis0, wmode);
    if (ax.kind == 5) {
        IG_REQUIRE(ctx, !axis0 && wmode == 0, "ig_fft: a chirp-z axis takes unweighted strided passes only");
        return launch_chirp_desc(ctx, ax, d);
    }
    IG_REQUIRE(ctx, ax.kind == 4 && !axis0, "ig_fft: a zero-padded pass on an axis without a two-stage kernel");
    return launch_ab_desc(ctx, ax, d, wmode);
}

}  // namespace

extern "C" {

int ig_fft_plan(ig_ctx* ctx, int rank, const int64_t* dims, int64_t batch, ig_fft** plan, size_t* workspace_bytes) {
    IG_REQUIRE(ctx, ctx && dims && plan, "ig_fft_plan: bad arguments");
    IG_REQUIRE(ctx, rank >= 1 && rank <= 3, "ig_fft_plan: rank %d not in 1..3", rank);
    IG_REQUIRE(ctx, batch >= 1, "ig_fft_plan: batch must be >= 1");
    *plan = nullptr;
    if (int rc = ig_set_device(ctx)) return rc;
    ig_fft* p = new ig_fft();
    p->ctx = ctx; p->rank = rank; p->batch = batch;
    p->total = batch;
    for (int a = 0; a < rank; ++a) {
        if (dims[a] < 1) { delete p; return ig_fail(ctx, IG_ERR_ARG, "ig_fft_plan: dims[%d] = %lld", a, (long long)dims[a]); }
        p->dims[a] = dims[a];
        p->total *= dims[a];
    }
    bool need_ws = false;
    for (int a = 0; a < rank; ++a) {
        int rc = plan_axis(ctx, p, a);
        if (rc != IG_OK) { ig_fft_destroy(p); return rc; }
        if (p->axis[a].kind == 1) need_ws = true;
    }
    p->workspace_bytes = need_ws ? (size_t)p->total * 8 : 0;
    for (int a = 0; a < rank; ++a)           // a chirp-z axis stages its length-m columns in the workspace
        if (p->axis[a].kind == 5 && !p->axis[a].chirp->fused)
            p->workspace_bytes = std::max(p->workspace_bytes, (size_t)(p->total / p->axis[a].n * p->axis[a].chirp->m) * 8);
    if (workspace_bytes) *workspace_bytes = p->workspace_bytes;
    p->two_launch = rank == 3 && dims[0] == 256 && dims[1] == 256 && dims[2] == 256 && p->axis[0].kind == 3;
    if (p->two_launch) p->inplace_workspace_bytes = (size_t)p->total * 8;

    char buf[256];
    p->desc.clear();
    for (int a = 0; a < rank; ++a) {
        const AxisPlan& ax = p->axis[a];
        if (ax.kind == 2) { snprintf(buf, sizeof(buf), "axis%d n=1 skip; ", a); p->desc += buf; continue; }
        snprintf(buf, sizeof(buf), "axis%d n=%lld %s", a, (long long)ax.n, ax.kind == 0 ? "lds" : ax.kind == 3 ? "2stage" : ax.kind == 4 ? "AxB" : ax.kind == 5 ? "chirp-z" : "generic");
        p->desc += buf;
        if (ax.kind == 5) {
            snprintf(buf, sizeof(buf), " m=%lld (%s, %s); ", (long long)ax.chirp->m, ax.chirp->sub.kind == 4 ? "AxB" : ax.chirp->sub.kind == 3 ? "2stage" : "lds",
                     ax.chirp->fused ? "one fused launch" : "five steps");
            p->desc += buf;
            continue;
        }
        if (ax.kind == 0 || ax.kind == 3 || ax.kind == 4) {
            snprintf(buf, sizeof(buf), " W=%d T=%d lds=%zuB radices=", ax.W, ax.T, ax.lds_bytes);
            p->desc += buf;
            for (int s = 0; s < ax.nstages; ++s) { snprintf(buf, sizeof(buf), s ? "x%d" : "%d", ax.rad.r[s]); p->desc += buf; }
        } else {
            p->desc += " radices=";
            for (size_t s = 0; s < ax.gen_radices.size(); ++s) { snprintf(buf, sizeof(buf), s ? "x%lld" : "%lld", (long long)ax.gen_radices[s]); p->desc += buf; }
        }
        p->desc += "; ";
    }
    if (p->two_launch) p->desc = "two launches (x + y/64 | y/4 + z), 512 threads x 32 values, LDS exchanges; fallback: " + p->desc;
    *plan = p;
    return IG_OK;
}

int ig_fft_exec(ig_fft* p, const void* xv, void* yv, int direction, void* workspace) {
    if (!p) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_exec: plan is NULL");
    ig_ctx* ctx = p->ctx;
    IG_REQUIRE(ctx, xv && yv, "ig_fft_exec: NULL array");
    IG_REQUIRE(ctx, direction == -1 || direction == 1, "ig_fft_exec: direction must be -1 or +1");
    IG_REQUIRE(ctx, p->workspace_bytes == 0 || workspace != nullptr, "ig_fft_exec: plan needs %zu bytes of workspace", p->workspace_bytes);
    IG_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(xv) & 7u) == 0 && (reinterpret_cast<uintptr_t>(yv) & 7u) == 0,
               "ig_fft_exec: arrays must be 8-byte aligned");
    if (int rc = ig_set_device(ctx)) return rc;
    const int inverse = direction > 0 ? 1 : 0;
    const float2* cur = (const float2*)xv;
    float2* y = (float2*)yv;
    float2* work = (float2*)workspace;

    // launch A cannot run in place: an in-place call stages through the caller's workspace (ig_fft_inplace_workspace bytes);
    // without one it takes the three axis passes below, which do run in place -- the library allocates nothing behind the
    // caller's back
    if (p->two_launch && !(xv == yv && !workspace)) {
        // two exchange images + the twiddle table: 75.7 KB per workgroup (two workgroups per CU), above the 64 KB a kernel may
        // use without opting in
        const size_t lds_b = (size_t)(2 * F3_LDS_ELEMS + 256) * sizeof(float2);
        if (!ctx->fft3d_attr) {
            IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft3d_a), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
            IG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fft3d_b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
            ctx->fft3d_attr = true;
        }
        const float2* src = (const float2*)xv;
        float2* mid = y;
        if (xv == yv) mid = (float2*)workspace;
        const double half_bytes = 2.0 * (double)p->total * 8.0;        // benchmark.py:55: 4 * nbytes per transform
        {
            ig_prof_scope prof(ctx, "fft3d_xy", half_bytes);
            hipLaunchKernelGGL(k_fft3d_a, dim3(256, 4, (unsigned)p->batch), dim3(512), lds_b, ctx->stream, src, mid, p->axis[0].d_tw, inverse);
            IG_LAUNCH_CHECK(ctx, "k_fft3d_a");
        }
        {
            ig_prof_scope prof(ctx, "fft3d_yz", half_bytes);
            hipLaunchKernelGGL(k_fft3d_b, dim3(16, 64, (unsigned)p->batch), dim3(512), lds_b, ctx->stream, (const float2*)mid, y, p->axis[0].d_tw, inverse);
            IG_LAUNCH_CHECK(ctx, "k_fft3d_b");
        }
        return IG_OK;
    }

    int live_axes = 0;
    for (int a = 0; a < p->rank; ++a) if (p->axis[a].kind != 2) ++live_axes;
    // the reference's convention books 4 * nbytes per multi-dimensional FFT (benchmark.py:55); each axis
    // pass gets an equal share of it
    const double pass_bytes = live_axes ? 4.0 * (double)p->total * 8.0 / live_axes : 0.0;

    // (Measured and rejected: a volume-at-a-time schedule whose passes hand a 134 MB volume to each other through the 256 MB
    // Infinity Cache -- the kernels themselves ran 7 % faster on 256^3 x 16, but 48 small dependent launches lost more than
    // that in the gaps between them: 2.45 against 2.37 ms.)
    // one axis pass with a fast kernel (kinds 0, 3, 4), in -> out (in place allowed: a workgroup holds its columns before it stores)
    auto fast_axis = [&](const AxisPlan& ax, const float2* in, float2* out, int inv, int a, double bytes) -> int {
        if (ax.kind == 3) {
            ig_prof_scope prof(ctx, a == 0 ? "fft_2stage_axis0" : a == 1 ? "fft_2stage_axis1" : "fft_2stage_axis2", bytes);
            PassDesc d{};
            d.in = in; d.out = out; d.w = nullptr;
            d.in_sj = d.out_sj = ax.inner; d.w_sj = 0;
            d.ncols = ax.inner * ax.outer;
            if (ax.inner == 1) {            // contiguous lines: the lines themselves are the tile's W columns
                d.ext0 = ax.outer; d.ext1 = 1; d.in_s[0] = d.out_s[0] = ax.n; d.in_s[1] = d.out_s[1] = 0;
            } else {
                d.ext0 = ax.inner; d.ext1 = ax.outer; d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = ax.inner * ax.n;
            }
            d.in_s[2] = d.out_s[2] = 0;
            d.in_lo = d.out_lo = 0; d.in_hi = d.out_hi = (int)ax.n; d.inverse = inv;
            return launch_2stage(ctx, ax, d, ax.inner == 1, 0);
        }
        if (ax.kind == 4) {
            ig_prof_scope prof(ctx, a == 0 ? "fft_ab_axis0" : a == 1 ? "fft_ab_axis1" : "fft_ab_axis2", bytes);
            return launch_ab(ctx, ax, in, out, inv);
        }
        IG_REQUIRE(ctx, ax.kind == 0, "ig_fft_exec: not a fast axis");
        ig_prof_scope prof(ctx, a == 0 ? "fft_lds_axis0" : a == 1 ? "fft_lds_axis1" : "fft_lds_axis2", bytes);
        const int64_t ncols = ax.inner * ax.outer;
        const int64_t blocks = (ncols + ax.W - 1) / ax.W;
        IG_REQUIRE(ctx, blocks <= 0x7fffffffLL, "ig_fft_exec: too many tiles");
        const dim3 grid((unsigned)blocks), block((unsigned)(ax.W * ax.T));
        if (ax.inner == 1)
            hipLaunchKernelGGL(k_fft_lds<true>, grid, block, ax.lds_bytes, ctx->stream, in, out, ax.d_tw,
                               (int)ax.n, ax.inner, ncols, ax.W, ax.T, ax.nstages, ax.rad, inv);
        else
            hipLaunchKernelGGL(k_fft_lds<false>, grid, block, ax.lds_bytes, ctx->stream, in, out, ax.d_tw,
                               (int)ax.n, ax.inner, ncols, ax.W, ax.T, ax.nstages, ax.rad, inv);
        IG_LAUNCH_CHECK(ctx, "k_fft_lds");
        return IG_OK;
    };
    for (int a = 0; a < p->rank; ++a) {
        const AxisPlan& ax = p->axis[a];
        if (ax.kind == 2) continue;
        if (ax.kind == 0 || ax.kind == 3 || ax.kind == 4) {
            if (int rc = fast_axis(ax, cur, y, inverse, a, pass_bytes)) return rc;
            cur = y;
        } else if (ax.kind == 5) {
            // chirp-z through the workspace: W holds the columns at length m
            const Chirp& ch = *ax.chirp;
            const int dir = inverse ? 1 : 0;
            if (cur == work) {              // (an earlier generic axis left its result there)
                IG_HIP(ctx, hipMemcpyAsync(y, cur, (size_t)p->total * 8, hipMemcpyDeviceToDevice, ctx->stream));
                cur = y;
            }
            if (ch.fused) {
                ig_prof_scope prof(ctx, a == 1 ? "fft_chirp_axis1" : "fft_chirp_axis2", pass_bytes);
                PassDesc d{};
                d.in = cur; d.in_sj = ax.inner; d.in_s[0] = 1; d.in_s[1] = ax.inner * ax.n;
                d.out = y; d.out_sj = ax.inner; d.out_s[0] = 1; d.out_s[1] = ax.inner * ax.n;
                d.ext0 = ax.inner; d.ext1 = ax.outer; d.ncols = ax.inner * ax.outer;
                d.in_lo = d.out_lo = 0; d.in_hi = d.out_hi = (int)ax.n; d.inverse = inverse;
                if (int rc = launch_chirp_desc(ctx, ax, d)) return rc;
            } else {
                const int64_t total_m = p->total / ax.n * ch.m;
                int64_t g = (total_m + 255) / 256;
                const int64_t cap = (int64_t)ctx->num_cu * 16;
                if (g > cap) g = cap;
                {
                    ig_prof_scope prof(ctx, "fft_chirp_pre");
                    hipLaunchKernelGGL(k_chirp_pre, dim3((unsigned)g), dim3(256), 0, ctx->stream, cur, work, ch.d_b[dir], ax.n, ch.m, ax.inner, total_m);
                    IG_LAUNCH_CHECK(ctx, "k_chirp_pre");
                }
                if (int rc = fast_axis(ch.sub, work, work, 0, a, 0.0)) return rc;
                {
                    ig_prof_scope prof(ctx, "fft_chirp_mul");
                    hipLaunchKernelGGL(k_chirp_mul, dim3((unsigned)g), dim3(256), 0, ctx->stream, work, ch.d_bhat[dir], ch.m, ax.inner, total_m);
                    IG_LAUNCH_CHECK(ctx, "k_chirp_mul");
                }
                if (int rc = fast_axis(ch.sub, work, work, 1, a, 0.0)) return rc;
                {
                    ig_prof_scope prof(ctx, "fft_chirp_post");
                    hipLaunchKernelGGL(k_chirp_post, dim3((unsigned)g), dim3(256), 0, ctx->stream, (const float2*)work, y, ch.d_bout[dir], ax.n, ch.m, ax.inner, p->total);
                    IG_LAUNCH_CHECK(ctx, "k_chirp_post");
                }
            }
            cur = y;
        } else {
            int64_t Ns = 1;
            int64_t g = (p->total + 255) / 256;
            const int64_t cap = (int64_t)ctx->num_cu * 16;
            if (g > cap) g = cap;
            for (size_t s = 0; s < ax.gen_radices.size(); ++s) {
                const int64_t R = ax.gen_radices[s];
                IG_REQUIRE(ctx, R <= 0x7fffffffLL, "ig_fft_exec: radix too large");
                float2* dst = (cur == y) ? work : y;
                ig_prof_scope prof(ctx, "fft_generic_stage", pass_bytes / (double)ax.gen_radices.size());
                hipLaunchKernelGGL(k_fft_generic_stage, dim3((unsigned)g), dim3(256), 0, ctx->stream,
                                   cur, dst, ax.d_tw, ax.n, ax.inner, p->total, (int)R, Ns, inverse);
                IG_LAUNCH_CHECK(ctx, "k_fft_generic_stage");
                cur = dst;
                Ns *= R;
            }
        }
    }
    if (cur != y) {
        // all axes were length 1 (cur == x) or the last generic stage landed in the workspace
        IG_HIP(ctx, hipMemcpyAsync(y, cur, (size_t)p->total * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return IG_OK;
}

// ---- zero-padded forward / cropped inverse 3-D transforms ------------------------------
// The gridding FFT of a non-Cartesian SENSE operator never sees a dense grid: its input is the image
// zero-padded into an oversampled grid (backend.py:371-387 Zpad, :432-442 NUFFT) and, on the adjoint,
// only the image box of its output is kept.  An axis pass then only has to touch the columns that
// are non-zero (forward) or kept (inverse), and only the in-box part of each column:
//     forward : pass x reads b0*b1*b2      writes n0*b1*b2      (per batch member)
//               pass y reads n0*b1*b2      writes n0*n1*b2
//               pass z reads n0*n1*b2      writes n0*n1*n2
// i.e. 2.6x the grid volume instead of 6x at 2x oversampling, and the diagonal factors of the tree
// (maps * roll-off * modulation) ride along as load/store weights instead of a (C*P)-row CSR matrix.

int ig_fft_plan_padded(ig_ctx* ctx, const int64_t* dims, const int64_t* box_lo, const int64_t* box_dims,
                       int64_t batch, int grid_layout, ig_fft** plan, size_t* workspace_bytes) {
    IG_REQUIRE(ctx, ctx && dims && box_lo && box_dims && plan, "ig_fft_plan_padded: bad arguments");
    IG_REQUIRE(ctx, grid_layout >= 0 && grid_layout <= 2, "ig_fft_plan_padded: grid_layout must be 0 (x,y,z), 1 (x,z,y) or 2 (c,x,z,y)");
    IG_REQUIRE(ctx, grid_layout != 2 || batch == 1 || batch == 2 || batch == 4 || batch == 8 || batch == 16,
               "ig_fft_plan_padded: the coil-interleaved layout needs a batch of 1, 2, 4, 8 or 16 (got %lld)", (long long)batch);
    for (int a = 0; a < 3; ++a) {
        IG_REQUIRE(ctx, box_lo[a] >= 0 && box_dims[a] >= 1 && box_lo[a] + box_dims[a] <= dims[a],
                   "ig_fft_plan_padded: box [%lld, %lld) outside grid axis %d of length %lld", (long long)box_lo[a],
                   (long long)(box_lo[a] + box_dims[a]), a, (long long)dims[a]);
    }
    size_t ws = 0;
    int rc = ig_fft_plan(ctx, 3, dims, batch, plan, &ws);
    if (rc != IG_OK) return rc;
    ig_fft* p = *plan;
    for (int a = 0; a < 3; ++a) {
        // chirp-z axes (a prime factor above 7: 277, 410 ...) take the unweighted strided passes, i.e. the y and z axes of the
        // coil-interleaved layout (round 5: with the k-space support table -- hulls, and bitmaps of B words per entry on a chirp-z z axis)
        const bool cz_ok = p->axis[a].kind == 5 && p->axis[a].chirp->sub.kind == 4 && a >= 1 && grid_layout == 2 && batch >= 2;
        if (cz_ok) p->axis[a].chirp->fused = true;
        const bool ab_ok = (p->axis[a].kind == 4 && abd_supported(dims[a]) && grid_layout == 2 && batch >= 2) || cz_ok;
        if (p->axis[a].kind != 3 && !ab_ok) {
            ig_fft_destroy(p);
            *plan = nullptr;
            return ig_fail(ctx, IG_ERR_UNSUPPORTED,
                           "ig_fft_plan_padded: grid axis %d has length %lld; the padded path needs 256 or 512 -- or, with 2, 4, 8 or 16 "
                           "coil-interleaved batch members (grid_layout 2: a chunk of any coil count padded to such a width, indigo_amd.fused.plan_chunks), "
                           "any length from 128 to 640 with factors 2, 3, 5, 7 only that splits as A x B with A, B <= 32, or on the y / z axes a length "
                           "with a larger prime factor (chirp-z)", a, (long long)dims[a]);
        }
        if (ab_ok) p->has_ab_axis = true;
        // (a chirp-z z axis: its threads hold rows b + B a of the length-m transform on both sides -- B words per entry either way)
        if (a == 2) { p->zw_in = cz_ok ? p->axis[a].chirp->sub.ab_B : ab_ok ? p->axis[a].ab_B : 16; p->zw_out = cz_ok ? p->axis[a].chirp->sub.ab_B : ab_ok ? p->axis[a].ab_A : 16; }
        if (cz_ok) p->has_chirp_axis = true;
        p->box_lo[a] = box_lo[a];
        p->box_dims[a] = box_dims[a];
    }
    p->padded = true;
    p->layout = grid_layout;
    // one full grid x batch array (the cropped inverse keeps its input intact) + one partially
    // transformed compact array n0 x b1 x b2 x batch (layout 1 routes the y pass through it)
    p->workspace_bytes = ((size_t)p->total + (size_t)(dims[0] * box_dims[1] * box_dims[2] * batch)) * 8;
    if (workspace_bytes) *workspace_bytes = p->workspace_bytes;
    p->desc = std::string("padded layout=") + (grid_layout == 2 ? "cxzy " : grid_layout ? "xzy " : "xyz ") + p->desc;
    return IG_OK;
}

// Grid layout 1 stores the grid as (x, z, y): the element (kx, ky, kz) of batch member c lives at
// kx + n0*kz + n0*n2*ky + vol*c.  With it the z pass -- the largest one, the whole grid -- runs at a
// 4 KB stride on both sides, and the only 2 MB-stride traffic left is the y pass's output (forward) or
// input (inverse), half a grid.  The y pass reads (writes) its other side from a compact
// n0 x b1 x b2 array in the workspace.  Whoever consumes the grid must index it the same way
// (SenseProblem.fused_interp(layout=1) permutes the gridding matrix's columns).
static int exec_padded_layout1(ig_fft* p, const float2* x, int64_t x_bstride, const float2* w, float2* y, float2* work,
                               const short2* support) {
    ig_ctx* ctx = p->ctx;
    const int64_t n0 = p->dims[0], n1 = p->dims[1], n2 = p->dims[2];
    const int64_t b0 = p->box_dims[0], b1 = p->box_dims[1], b2 = p->box_dims[2];
    const int64_t l0 = p->box_lo[0], l1 = p->box_lo[1], l2 = p->box_lo[2];
    const int64_t vol = n0 * n1 * n2, bvol = b0 * b1 * b2, C = p->batch;
    const int64_t cvol = n0 * b1 * b2;                 // compact intermediate per batch member
    float2* L1 = work + (size_t)p->total;              // behind the full-size part of the workspace
    {   // pass x: compact weighted image rows -> compact [kx][y'][z']
        ig_prof_scope prof(ctx, "fft_pad_x", (double)(bvol + (w ? bvol : 0) + cvol) * C * 8.0);
        PassDesc d{};
        d.in = x - l0; d.in_sj = 1; d.in_s[0] = b0; d.in_s[1] = b0 * b1; d.in_s[2] = x_bstride;
        d.w = w ? w - l0 : nullptr; d.w_sj = 1; d.w_s[0] = b0; d.w_s[1] = b0 * b1; d.w_s[2] = bvol;
        d.out = L1; d.out_sj = 1; d.out_s[0] = n0; d.out_s[1] = n0 * b1; d.out_s[2] = cvol;
        d.ext0 = b1; d.ext1 = b2; d.ncols = b1 * b2 * C;
        d.in_lo = (int)l0; d.in_hi = (int)(l0 + b0); d.out_lo = 0; d.out_hi = (int)n0; d.inverse = 0;
        if (int rc = launch_2stage(ctx, p->axis[0], d, true, w ? 1 : 0)) return rc;
    }
    {   // pass y: columns (kx, z'), compact in (4 KB stride), grid out at z = l2 + z' (stride n0*n2)
        ig_prof_scope prof(ctx, "fft_pad_y", (double)(cvol + n0 * n1 * b2) * C * 8.0);
        PassDesc d{};
        d.in = L1 - l1 * n0; d.in_sj = n0; d.in_s[0] = 1; d.in_s[1] = n0 * b1; d.in_s[2] = cvol;
        d.out = y + l2 * n0; d.out_sj = n0 * n2; d.out_s[0] = 1; d.out_s[1] = n0; d.out_s[2] = vol;
        d.ext0 = n0; d.ext1 = b2; d.ncols = n0 * b2 * C;
        d.in_lo = (int)l1; d.in_hi = (int)(l1 + b1); d.out_lo = 0; d.out_hi = (int)n1; d.inverse = 0;
        // ky outside the support of this kx tile is never transformed along z: do not produce it
        if (support) { d.tile_range = support + n1 * (n0 / 16); d.tile_range_mode = 1; d.tile_range_k1 = 0; }
        if (int rc = launch_2stage(ctx, p->axis[1], d, false, 0)) return rc;
    }
    {   // pass z: all columns (kx, ky), in place, stride n0 on both sides
        ig_prof_scope prof(ctx, "fft_pad_z", (double)(n0 * n1 * b2 + vol) * C * 8.0);
        PassDesc d{};
        d.in = d.out = y; d.in_sj = d.out_sj = n0;
        d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = n0 * n2; d.in_s[2] = d.out_s[2] = vol;
        d.ext0 = n0; d.ext1 = n1; d.ncols = n0 * n1 * C;
        d.in_lo = (int)l2; d.in_hi = (int)(l2 + b2); d.out_lo = 0; d.out_hi = (int)n2; d.inverse = 0;
        d.tile_range = support; d.tile_range_mode = 1; d.tile_range_k1 = n0 / 16;   // only the support is ever gridded from
        if (support) { d.tile_bits = reinterpret_cast<const uint32_t*>(support + n1 * (n0 / 16) + n0 / 16); d.tile_words = 16; }
        if (int rc = launch_2stage(ctx, p->axis[2], d, false, 0)) return rc;
    }
    return IG_OK;
}

// phases: bit 0 = the z pass (whole grid), bit 1 = the y and x passes, restricted to the image planes z0 <= z' < z1
static int exec_cropped_layout1(ig_fft* p, const float2* y, const float2* w, float2* x, int64_t x_bstride, float2* work,
                                const short2* support, int phases = 3, int64_t z0 = 0, int64_t z1 = -1) {
    ig_ctx* ctx = p->ctx;
    const int64_t n0 = p->dims[0], n1 = p->dims[1], n2 = p->dims[2];
    const int64_t b0 = p->box_dims[0], b1 = p->box_dims[1], b2 = p->box_dims[2];
    const int64_t l0 = p->box_lo[0], l1 = p->box_lo[1], l2 = p->box_lo[2];
    const int64_t vol = n0 * n1 * n2, bvol = b0 * b1 * b2, C = p->batch;
    const int64_t cvol = n0 * b1 * b2;
    float2* L1 = work + (size_t)p->total;
    if (z1 < 0) z1 = b2;
    const int64_t nz = z1 - z0;                      // image planes the y and x passes cover
    if (phases & 1) {   // pass z: all columns (kx, ky), keep z in box, stride n0; input intact, result into the workspace
        ig_prof_scope prof(ctx, "fft_crop_z", (double)(vol + n0 * n1 * b2) * C * 8.0);
        PassDesc d{};
        d.in = y; d.out = work; d.in_sj = d.out_sj = n0;
        d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = n0 * n2; d.in_s[2] = d.out_s[2] = vol;
        d.ext0 = n0; d.ext1 = n1; d.ncols = n0 * n1 * C;
        d.in_lo = 0; d.in_hi = (int)n2; d.out_lo = (int)l2; d.out_hi = (int)(l2 + b2); d.inverse = 1;
        d.tile_range = support; d.tile_range_mode = 2; d.tile_range_k1 = n0 / 16;   // the adjoint gridding only wrote the support
        if (support) { d.tile_bits = reinterpret_cast<const uint32_t*>(support + n1 * (n0 / 16) + n0 / 16); d.tile_words = 16; }
        if (support) d.k1_range = support + n1 * (n0 / 16);                         // ky the y pass will never read
        if (int rc = launch_2stage(ctx, p->axis[2], d, false, 0)) return rc;
    }
    if (!(phases & 2) || nz <= 0) return IG_OK;
    {   // pass y: columns (kx, z'), grid in (stride n0*n2), compact out, keep y in box
        ig_prof_scope prof(ctx, "fft_crop_y", (double)(n0 * n1 * nz + n0 * b1 * nz) * C * 8.0);
        PassDesc d{};
        d.in = work + (l2 + z0) * n0; d.in_sj = n0 * n2; d.in_s[0] = 1; d.in_s[1] = n0; d.in_s[2] = vol;
        d.out = L1 - l1 * n0 + z0 * n0 * b1; d.out_sj = n0; d.out_s[0] = 1; d.out_s[1] = n0 * b1; d.out_s[2] = cvol;
        d.ext0 = n0; d.ext1 = nz; d.ncols = n0 * nz * C;
        d.in_lo = 0; d.in_hi = (int)n1; d.out_lo = (int)l1; d.out_hi = (int)(l1 + b1); d.inverse = 1;
        if (support) { d.tile_range = support + n1 * (n0 / 16); d.tile_range_mode = 2; d.tile_range_k1 = 0; }
        if (int rc = launch_2stage(ctx, p->axis[1], d, false, 0)) return rc;
    }
    {   // pass x: compact rows, keep x in box, times conj(w), into the compact image array
        ig_prof_scope prof(ctx, "fft_crop_x", (double)(cvol + bvol + (w ? bvol : 0)) * C * 8.0 * (double)nz / (double)b2);
        PassDesc d{};
        d.in = L1 + z0 * n0 * b1; d.in_sj = 1; d.in_s[0] = n0; d.in_s[1] = n0 * b1; d.in_s[2] = cvol;
        d.out = x - l0 + z0 * b0 * b1; d.out_sj = 1; d.out_s[0] = b0; d.out_s[1] = b0 * b1; d.out_s[2] = x_bstride;
        d.w = w ? w - l0 + z0 * b0 * b1 : nullptr; d.w_sj = 1; d.w_s[0] = b0; d.w_s[1] = b0 * b1; d.w_s[2] = bvol;
        d.ext0 = b1; d.ext1 = nz; d.ncols = b1 * nz * C;
        d.in_lo = 0; d.in_hi = (int)n0; d.out_lo = (int)l0; d.out_hi = (int)(l0 + b0); d.inverse = 1;
        if (int rc = launch_2stage(ctx, p->axis[0], d, true, w ? 2 : 0)) return rc;
    }
    return IG_OK;
}

// Grid layout 2 interleaves the batch (coils) below layout 1: element (c, kx, ky, kz) lives at
// c + C*(kx + n0*kz + n0*n2*ky).  The gridding matrix then reads / writes all C coils of a grid point as ONE
// contiguous C*8-byte row instead of C separate 8-byte gathers a gigabyte apart (forward gridding of the 256^3 x 8
// SENSE problem: 0.98 ms instead of 2.05 ms).  Every pass is a strided pass whose 16 lanes run over the combined
// (c, kx) index -- for the x passes over (c, line) pairs, with the transform axis at stride C -- so all accesses stay
// 128-byte (y, z) or 64*C/8-byte (x) contiguous.  The compact intermediate and, for the cropped transform, the
// compact result are interleaved the same way; the weights must be too (w[(i)*C + c]).
static int lg2(int64_t v) { int s = 0; while ((1LL << s) < v) ++s; return s; }

static int exec_padded_layout2(ig_fft* p, const float2* x, int64_t x_bstride, const float2* w, float2* y, float2* work,
                               const short2* support) {
    ig_ctx* ctx = p->ctx;
    const int64_t n0 = p->dims[0], n1 = p->dims[1], n2 = p->dims[2];
    const int64_t snt = n0 / p->support_tile;                        // support entries per grid row
    const int sshift = lg2((int)(p->batch * p->support_tile / 16));    // 16-column tiles per support entry
    // (a one-row pitch on the z axis, to break the 16 MB power-of-two stride of the y pass, was measured: no gain once
    // the y pass runs on 32-column tiles)
    const int64_t b0 = p->box_dims[0], b1 = p->box_dims[1], b2 = p->box_dims[2];
    const int64_t l0 = p->box_lo[0], l1 = p->box_lo[1], l2 = p->box_lo[2];
    const int64_t vol = n0 * n1 * n2, bvol = b0 * b1 * b2, C = p->batch;
    const int64_t cvol = n0 * b1 * b2;
    float2* L1 = work + (size_t)p->total;
    {   // pass x: lines (c, y', z') of the weighted image -> interleaved compact [c][kx][y'][z']
        ig_prof_scope prof(ctx, "fft_pad_x", (double)(bvol + (w ? bvol : 0) + cvol) * C * 8.0);
        PassDesc d{};
        d.cw = (int)C;
        d.in = x - l0; d.in_sj = 1; d.in_sa = x_bstride; d.in_s[0] = b0; d.in_s[1] = b0 * b1;
        d.w = w ? w - l0 * C : nullptr; d.w_sj = C; d.w_sa = 1; d.w_s[0] = b0 * C; d.w_s[1] = b0 * b1 * C;
        d.out = L1; d.out_sj = C; d.out_sa = 1; d.out_s[0] = C * n0; d.out_s[1] = C * n0 * b1;
        d.ext0 = b1; d.ext1 = b2; d.ncols = b1 * b2;
        d.in_lo = (int)l0; d.in_hi = (int)(l0 + b0); d.out_lo = 0; d.out_hi = (int)n0; d.inverse = 0;
        if (int rc = launch_pass(ctx, p->axis[0], d, false, w ? 1 : 0)) return rc;
    }
    {   // pass y: columns (c + C*kx, z')
        ig_prof_scope prof(ctx, "fft_pad_y", (double)(cvol + n0 * n1 * b2) * C * 8.0);
        PassDesc d{};
        d.in = L1 - l1 * C * n0; d.in_sj = C * n0; d.in_s[0] = 1; d.in_s[1] = C * n0 * b1;
        d.out = y + l2 * C * n0; d.out_sj = C * n0 * n2; d.out_s[0] = 1; d.out_s[1] = C * n0;
        d.ext0 = C * n0; d.ext1 = b2; d.ncols = C * n0 * b2;
        d.in_lo = (int)l1; d.in_hi = (int)(l1 + b1); d.out_lo = 0; d.out_hi = (int)n1; d.inverse = 0;
        if (support) { d.tile_range = support + n1 * snt; d.tile_range_mode = 1; d.tile_range_k1 = 0; d.tile_shift = sshift; }
        if (int rc = launch_pass(ctx, p->axis[1], d, false, 0)) return rc;
    }
    {   // pass z: columns (c + C*kx, ky), in place
        ig_prof_scope prof(ctx, "fft_pad_z", (double)(n0 * n1 * b2 + vol) * C * 8.0);
        PassDesc d{};
        d.in = d.out = y; d.in_sj = d.out_sj = C * n0;
        d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = C * n0 * n2;
        d.ext0 = C * n0; d.ext1 = n1; d.ncols = C * n0 * n1;
        d.in_lo = (int)l2; d.in_hi = (int)(l2 + b2); d.out_lo = 0; d.out_hi = (int)n2; d.inverse = 0;
        d.tile_range = support; d.tile_range_mode = 1; d.tile_range_k1 = snt; d.tile_shift = sshift;
        // (round 5, measured: with the ky hulls for the early exit and range and bitmap read behind the loads, this pass took 1.029 ms against 1.010)
        if (support) {      // the output-side form of the bitmaps (it follows the input-side form where the two differ)
            d.tile_bits = reinterpret_cast<const uint32_t*>(support + n1 * snt + snt) + (p->zw_out != p->zw_in ? n1 * snt * p->zw_in : 0);
            d.tile_words = p->zw_out;
        }
        if (int rc = launch_pass(ctx, p->axis[2], d, false, 0)) return rc;
    }
    return IG_OK;
}

// phases: bit 0 = the z pass (whole grid), bit 1 = the y and x passes, restricted to the image planes z0 <= z' < z1
static int exec_cropped_layout2(ig_fft* p, const float2* y, const float2* w, float2* x, float2* work, const short2* support,
                                bool sum_coils = false, int phases = 3, int64_t z0 = 0, int64_t z1 = -1) {
    ig_ctx* ctx = p->ctx;
    const int64_t n0 = p->dims[0], n1 = p->dims[1], n2 = p->dims[2];
    const int64_t snt = n0 / p->support_tile;                        // support entries per grid row
    const int sshift = lg2((int)(p->batch * p->support_tile / 16));    // 16-column tiles per support entry
    const int64_t b0 = p->box_dims[0], b1 = p->box_dims[1], b2 = p->box_dims[2];
    const int64_t l0 = p->box_lo[0], l1 = p->box_lo[1], l2 = p->box_lo[2];
    const int64_t vol = n0 * n1 * n2, bvol = b0 * b1 * b2, C = p->batch;
    const int64_t cvol = n0 * b1 * b2;
    float2* L1 = work + (size_t)p->total;
    if (z1 < 0) z1 = b2;
    const int64_t nz = z1 - z0;                      // image planes the y and x passes cover
    if (phases & 1) {   // pass z: input intact, result into the workspace
        ig_prof_scope prof(ctx, "fft_crop_z", (double)(vol + n0 * n1 * b2) * C * 8.0);
        PassDesc d{};
        d.in = y; d.out = work; d.in_sj = d.out_sj = C * n0;
        d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = C * n0 * n2;
        d.ext0 = C * n0; d.ext1 = n1; d.ncols = C * n0 * n1;
        d.in_lo = 0; d.in_hi = (int)n2; d.out_lo = (int)l2; d.out_hi = (int)(l2 + b2); d.inverse = 1;
        d.tile_range = support; d.tile_range_mode = 2; d.tile_range_k1 = snt; d.tile_shift = sshift;
        if (support) { d.tile_bits = reinterpret_cast<const uint32_t*>(support + n1 * snt + snt); d.tile_words = p->zw_in; }
        if (support) d.k1_range = support + n1 * snt;                         // ky the y pass will never read
        if (int rc = launch_pass(ctx, p->axis[2], d, false, 0)) return rc;
    }
    if (!(phases & 2) || nz <= 0) return IG_OK;
    {   // pass y
        ig_prof_scope prof(ctx, "fft_crop_y", (double)(n0 * n1 * nz + n0 * b1 * nz) * C * 8.0);
        PassDesc d{};
        d.in = work + (l2 + z0) * C * n0; d.in_sj = C * n0 * n2; d.in_s[0] = 1; d.in_s[1] = C * n0;
        d.out = L1 - l1 * C * n0 + z0 * C * n0 * b1; d.out_sj = C * n0; d.out_s[0] = 1; d.out_s[1] = C * n0 * b1;
        d.ext0 = C * n0; d.ext1 = nz; d.ncols = C * n0 * nz;
        d.in_lo = 0; d.in_hi = (int)n1; d.out_lo = (int)l1; d.out_hi = (int)(l1 + b1); d.inverse = 1;
        if (support) { d.tile_range = support + n1 * snt; d.tile_range_mode = 2; d.tile_range_k1 = 0; d.tile_shift = sshift; }
        if (int rc = launch_pass(ctx, p->axis[1], d, false, 0)) return rc;
    }
    if (sum_coils) {   // pass x with the coil combination: x = sum_c conj(w_c) .* crop(...), one image box
        ig_prof_scope prof(ctx, "fft_crop_x", ((double)(cvol + bvol) * C * 8.0 + (double)bvol * 8.0) * (double)nz / (double)b2);
        PassDesc d{};
        d.cw = (int)C;
        d.in = L1 + z0 * C * n0 * b1; d.in_sj = C; d.in_sa = 1; d.in_s[0] = C * n0; d.in_s[1] = C * n0 * b1;
        d.out = x - l0 + z0 * b0 * b1; d.out_sj = 1; d.out_sa = 0; d.out_s[0] = b0; d.out_s[1] = b0 * b1;
        d.w = w - l0 * C + z0 * b0 * b1 * C; d.w_sj = C; d.w_sa = 1; d.w_s[0] = b0 * C; d.w_s[1] = b0 * b1 * C;
        d.ext0 = b1; d.ext1 = nz; d.ncols = b1 * nz;
        d.in_lo = 0; d.in_hi = (int)n0; d.out_lo = (int)l0; d.out_hi = (int)(l0 + b0); d.inverse = 1;
        if (int rc = launch_pass(ctx, p->axis[0], d, false, 3)) return rc;
    } else
    {   // pass x: interleaved compact rows -> interleaved compact image box, times conj(w)
        ig_prof_scope prof(ctx, "fft_crop_x", (double)(cvol + bvol + (w ? bvol : 0)) * C * 8.0);
        PassDesc d{};
        d.cw = (int)C;
        d.in = L1; d.in_sj = C; d.in_sa = 1; d.in_s[0] = C * n0; d.in_s[1] = C * n0 * b1;
        d.out = x - l0 * C; d.out_sj = C; d.out_sa = 1; d.out_s[0] = b0 * C; d.out_s[1] = b0 * b1 * C;
        d.w = w ? w - l0 * C : nullptr; d.w_sj = C; d.w_sa = 1; d.w_s[0] = b0 * C; d.w_s[1] = b0 * b1 * C;
        d.ext0 = b1; d.ext1 = b2; d.ncols = b1 * b2;
        d.in_lo = 0; d.in_hi = (int)n0; d.out_lo = (int)l0; d.out_hi = (int)(l0 + b0); d.inverse = 1;
        if (int rc = launch_pass(ctx, p->axis[0], d, false, w ? 2 : 0)) return rc;
    }
    return IG_OK;
}

int ig_fft_support_words(int64_t n, int* zw_in, int* zw_out) {
    if (!zw_in || !zw_out) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_support_words: bad arguments");
    int A = 0, B = 0;
    if (n == 256 || n == 512) { *zw_in = *zw_out = 16; return IG_OK; }
    if (abd_supported(n) && ab_split(n, A, B)) { *zw_in = B; *zw_out = A; return IG_OK; }
    {   // a chirp-z axis (a prime factor above 7) over an A x B length: B words on both sides
        std::vector<int64_t> f;
        factor_generic(n, f);
        int64_t big = 1;
        for (int64_t r : f) big = std::max(big, r);
        if (big > 7 && n >= 32 && chirp_ab_length(n, A, B) && n <= 32 * (int64_t)B) { *zw_in = *zw_out = B; return IG_OK; }
    }
    return ig_fail(nullptr, IG_ERR_UNSUPPORTED, "ig_fft_support_words: no zero-pad-aware z pass for an axis of %lld points", (long long)n);
}

int ig_fft_padded_axis_kind(int64_t n, int* kind) {
    if (!kind || n < 1) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_padded_axis_kind: bad arguments");
    int A = 0, B = 0;
    *kind = 0;
    if (n == 256 || n == 512) { *kind = 3; return IG_OK; }
    if (abd_supported(n)) { *kind = 4; return IG_OK; }
    // chirp-z: a length with a prime factor above 7 whose 2 n - 1 is covered by an A x B length
    std::vector<int64_t> f;
    factor_generic(n, f);
    int64_t big = 1;
    for (int64_t r : f) big = std::max(big, r);
    if (big > 7 && n >= 32)
        for (int64_t c = 2 * n - 1; c <= 1024; ++c) if (ab_split(c, A, B)) { *kind = 5; break; }
    return IG_OK;
}

int ig_fft_set_axis_shift(ig_fft* p, int axis, int64_t shift) {
    if (!p) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_set_axis_shift: plan is NULL");
    ig_ctx* ctx = p->ctx;
    IG_REQUIRE(ctx, p->padded && p->layout == 2 && axis >= 1 && axis < p->rank, "ig_fft_set_axis_shift: the y or z axis of a zero-padded plan of the coil-interleaved layout");
    AxisPlan& ax = p->axis[axis];
    IG_REQUIRE(ctx, shift >= 0 && shift < ax.n, "ig_fft_set_axis_shift: shift %lld outside [0, %lld)", (long long)shift, (long long)ax.n);
    if (shift == 0 && !(ax.kind == 5 && ax.chirp && ax.chirp->shift)) return IG_OK;
    if (!(ax.kind == 5 && ax.chirp && ax.chirp->fused))
        return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_fft_set_axis_shift: axis %d (%lld points) is no one-launch chirp-z axis", axis, (long long)ax.n);
    if (int rc = ig_set_device(ctx)) return rc;
    IG_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (tables of a pass still in flight are about to be freed)
    return plan_chirp_shift(ctx, *ax.chirp, ax.n, shift);
}

int ig_fft_set_support_tile(ig_fft* p, int tile) {
    if (!p) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_set_support_tile: plan is NULL");
    ig_ctx* ctx = p->ctx;
    IG_REQUIRE(ctx, p->padded && p->layout == 2, "ig_fft_set_support_tile: a zero-padded plan of the coil-interleaved layout");
    IG_REQUIRE(ctx, (tile == 2 || tile == 4 || tile == 8 || tile == 16) && p->batch * tile >= 16 && p->dims[0] % tile == 0,
               "ig_fft_set_support_tile: tile %d (2, 4, 8 or 16 kx points; coils * tile >= 16)", tile);
    p->support_tile = tile;
    return IG_OK;
}

int ig_fft_exec_padded(ig_fft* p, const void* xv, int64_t x_bstride, const void* wv, void* yv, void* workspace,
                       const int16_t* support) {
    if (!p) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_exec_padded: plan is NULL");
    ig_ctx* ctx = p->ctx;
    IG_REQUIRE(ctx, p->padded, "ig_fft_exec_padded: plan was not made by ig_fft_plan_padded");
    IG_REQUIRE(ctx, xv && yv, "ig_fft_exec_padded: NULL array");
    IG_REQUIRE(ctx, p->layout == 0 || workspace, "ig_fft_exec_padded: grid layouts 1 and 2 need the workspace");
    if (int rc = ig_set_device(ctx)) return rc;
    IG_REQUIRE(ctx, !support || (p->layout >= 1 && (!p->has_chirp_axis || p->layout == 2)), "ig_fft_exec_padded: a support table needs grid layout 1 or 2 (chirp-z axes: layout 2, hulls only)");
    if (p->layout == 2)
        return exec_padded_layout2(p, (const float2*)xv, x_bstride, (const float2*)wv, (float2*)yv, (float2*)workspace,
                                   (const short2*)support);
    if (p->layout == 1)
        return exec_padded_layout1(p, (const float2*)xv, x_bstride, (const float2*)wv, (float2*)yv, (float2*)workspace,
                                   (const short2*)support);
    const int64_t n0 = p->dims[0], n1 = p->dims[1], n2 = p->dims[2];
    const int64_t b0 = p->box_dims[0], b1 = p->box_dims[1], b2 = p->box_dims[2];
    const int64_t l0 = p->box_lo[0], l1 = p->box_lo[1], l2 = p->box_lo[2];
    const int64_t vol = n0 * n1 * n2, bvol = b0 * b1 * b2, C = p->batch;
    const float2* x = (const float2*)xv;
    const float2* w = (const float2*)wv;
    float2* y = (float2*)yv;
    {   // pass x: compact (weighted) image rows -> full-length rows at the box's (y, z) positions
        ig_prof_scope prof(ctx, "fft_pad_x", (double)(bvol + (w ? bvol : 0) + n0 * b1 * b2) * C * 8.0);
        PassDesc d{};
        d.in = x - l0; d.in_sj = 1; d.in_s[0] = b0; d.in_s[1] = b0 * b1; d.in_s[2] = x_bstride;
        d.w = w ? w - l0 : nullptr; d.w_sj = 1; d.w_s[0] = b0; d.w_s[1] = b0 * b1; d.w_s[2] = bvol;
        d.out = y + l1 * n0 + l2 * n0 * n1; d.out_sj = 1; d.out_s[0] = n0; d.out_s[1] = n0 * n1; d.out_s[2] = vol;
        d.ext0 = b1; d.ext1 = b2; d.ncols = b1 * b2 * C;
        d.in_lo = (int)l0; d.in_hi = (int)(l0 + b0); d.out_lo = 0; d.out_hi = (int)n0; d.inverse = 0;
        if (int rc = launch_2stage(ctx, p->axis[0], d, true, w ? 1 : 0)) return rc;
    }
    {   // pass y: columns (kx, z in box), inputs y in box
        ig_prof_scope prof(ctx, "fft_pad_y", (double)(n0 * b1 * b2 + n0 * n1 * b2) * C * 8.0);
        PassDesc d{};
        d.in = d.out = y + l2 * n0 * n1; d.in_sj = d.out_sj = n0;
        d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = n0 * n1; d.in_s[2] = d.out_s[2] = vol;
        d.ext0 = n0; d.ext1 = b2; d.ncols = n0 * b2 * C;
        d.in_lo = (int)l1; d.in_hi = (int)(l1 + b1); d.out_lo = 0; d.out_hi = (int)n1; d.inverse = 0;
        if (int rc = launch_2stage(ctx, p->axis[1], d, false, 0)) return rc;
    }
    {   // pass z: all columns (kx, ky), inputs z in box
        ig_prof_scope prof(ctx, "fft_pad_z", (double)(n0 * n1 * b2 + vol) * C * 8.0);
        PassDesc d{};
        d.in = d.out = y; d.in_sj = d.out_sj = n0 * n1;
        d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = n0; d.in_s[2] = d.out_s[2] = vol;
        d.ext0 = n0; d.ext1 = n1; d.ncols = n0 * n1 * C;
        d.in_lo = (int)l2; d.in_hi = (int)(l2 + b2); d.out_lo = 0; d.out_hi = (int)n2; d.inverse = 0;
        if (int rc = launch_2stage(ctx, p->axis[2], d, false, 0)) return rc;
    }
    return IG_OK;
}

int ig_fft_exec_cropped(ig_fft* p, const void* yv, const void* wv, void* xv, int64_t x_bstride, void* workspace,
                        const int16_t* support) {
    if (!p) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_exec_cropped: plan is NULL");
    ig_ctx* ctx = p->ctx;
    IG_REQUIRE(ctx, p->padded, "ig_fft_exec_cropped: plan was not made by ig_fft_plan_padded");
    IG_REQUIRE(ctx, xv && yv && workspace, "ig_fft_exec_cropped: NULL array");
    if (int rc = ig_set_device(ctx)) return rc;
    IG_REQUIRE(ctx, !support || (p->layout >= 1 && (!p->has_chirp_axis || p->layout == 2)), "ig_fft_exec_cropped: a support table needs grid layout 1 or 2 (chirp-z axes: layout 2, hulls only)");
    if (p->layout == 2)
        return exec_cropped_layout2(p, (const float2*)yv, (const float2*)wv, (float2*)xv, (float2*)workspace,
                                    (const short2*)support);
    if (p->layout == 1)
        return exec_cropped_layout1(p, (const float2*)yv, (const float2*)wv, (float2*)xv, x_bstride, (float2*)workspace,
                                    (const short2*)support);
    const int64_t n0 = p->dims[0], n1 = p->dims[1], n2 = p->dims[2];
    const int64_t b0 = p->box_dims[0], b1 = p->box_dims[1], b2 = p->box_dims[2];
    const int64_t l0 = p->box_lo[0], l1 = p->box_lo[1], l2 = p->box_lo[2];
    const int64_t vol = n0 * n1 * n2, bvol = b0 * b1 * b2, C = p->batch;
    const float2* y = (const float2*)yv;
    const float2* w = (const float2*)wv;
    float2* x = (float2*)xv;
    float2* work = (float2*)workspace;
    {   // pass z: all columns, keep z in box (input stays intact: written to the workspace)
        ig_prof_scope prof(ctx, "fft_crop_z", (double)(vol + n0 * n1 * b2) * C * 8.0);
        PassDesc d{};
        d.in = y; d.out = work; d.in_sj = d.out_sj = n0 * n1;
        d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = n0; d.in_s[2] = d.out_s[2] = vol;
        d.ext0 = n0; d.ext1 = n1; d.ncols = n0 * n1 * C;
        d.in_lo = 0; d.in_hi = (int)n2; d.out_lo = (int)l2; d.out_hi = (int)(l2 + b2); d.inverse = 1;
        if (int rc = launch_2stage(ctx, p->axis[2], d, false, 0)) return rc;
    }
    {   // pass y: columns (kx, z in box), keep y in box
        ig_prof_scope prof(ctx, "fft_crop_y", (double)(n0 * n1 * b2 + n0 * b1 * b2) * C * 8.0);
        PassDesc d{};
        d.in = d.out = work + l2 * n0 * n1; d.in_sj = d.out_sj = n0;
        d.in_s[0] = d.out_s[0] = 1; d.in_s[1] = d.out_s[1] = n0 * n1; d.in_s[2] = d.out_s[2] = vol;
        d.ext0 = n0; d.ext1 = b2; d.ncols = n0 * b2 * C;
        d.in_lo = 0; d.in_hi = (int)n1; d.out_lo = (int)l1; d.out_hi = (int)(l1 + b1); d.inverse = 1;
        if (int rc = launch_2stage(ctx, p->axis[1], d, false, 0)) return rc;
    }
    {   // pass x: rows (y, z in box), keep x in box, times conj(w), into the compact array
        ig_prof_scope prof(ctx, "fft_crop_x", (double)(n0 * b1 * b2 + bvol + (w ? bvol : 0)) * C * 8.0);
        PassDesc d{};
        d.in = work + l1 * n0 + l2 * n0 * n1; d.in_sj = 1; d.in_s[0] = n0; d.in_s[1] = n0 * n1; d.in_s[2] = vol;
        d.out = x - l0; d.out_sj = 1; d.out_s[0] = b0; d.out_s[1] = b0 * b1; d.out_s[2] = x_bstride;
        d.w = w ? w - l0 : nullptr; d.w_sj = 1; d.w_s[0] = b0; d.w_s[1] = b0 * b1; d.w_s[2] = bvol;
        d.ext0 = b1; d.ext1 = b2; d.ncols = b1 * b2 * C;
        d.in_lo = 0; d.in_hi = (int)n0; d.out_lo = (int)l0; d.out_hi = (int)(l0 + b0); d.inverse = 1;
        if (int rc = launch_2stage(ctx, p->axis[0], d, true, w ? 2 : 0)) return rc;
    }
    return IG_OK;
}

int ig_fft_exec_cropped_slab(ig_fft* p, const void* yv, const void* wv, void* xv, int64_t x_bstride, void* workspace,
                             const int16_t* support, int phase, int64_t z0, int64_t z1) {
    if (!p) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_exec_cropped_slab: plan is NULL");
    ig_ctx* ctx = p->ctx;
    IG_REQUIRE(ctx, p->padded && p->layout == 1, "ig_fft_exec_cropped_slab: needs a plan of ig_fft_plan_padded with grid_layout 1 (layout 2: ig_fft_exec_cropped_sum_slab)");
    IG_REQUIRE(ctx, xv && yv && workspace, "ig_fft_exec_cropped_slab: NULL array");
    IG_REQUIRE(ctx, phase == 0 || phase == 1, "ig_fft_exec_cropped_slab: phase must be 0 (z pass) or 1 (y and x passes of a slab)");
    IG_REQUIRE(ctx, phase == 0 || (0 <= z0 && z0 <= z1 && z1 <= p->box_dims[2]),
               "ig_fft_exec_cropped_slab: slab [%lld, %lld) outside the image's %lld planes", (long long)z0, (long long)z1, (long long)p->box_dims[2]);
    if (int rc = ig_set_device(ctx)) return rc;
    return exec_cropped_layout1(p, (const float2*)yv, (const float2*)wv, (float2*)xv, x_bstride, (float2*)workspace,
                                (const short2*)support, phase == 0 ? 1 : 2, z0, z1);
}

int ig_fft_exec_cropped_sum(ig_fft* p, const void* yv, const void* wv, void* xv, void* workspace, const int16_t* support) {
    if (!p) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_exec_cropped_sum: plan is NULL");
    ig_ctx* ctx = p->ctx;
    IG_REQUIRE(ctx, p->padded && p->layout == 2, "ig_fft_exec_cropped_sum: needs a plan of ig_fft_plan_padded with grid_layout 2");
    IG_REQUIRE(ctx, xv && yv && wv && workspace, "ig_fft_exec_cropped_sum: NULL array");
    // (a grid with a chirp-z axis takes the table's hulls; its bitmaps are not read)
    if (int rc = ig_set_device(ctx)) return rc;
    return exec_cropped_layout2(p, (const float2*)yv, (const float2*)wv, (float2*)xv, (float2*)workspace,
                                (const short2*)support, true);
}

int ig_fft_exec_cropped_sum_slab(ig_fft* p, const void* yv, const void* wv, void* xv, void* workspace, const int16_t* support,
                                 int phase, int64_t z0, int64_t z1) {
    if (!p) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_exec_cropped_sum_slab: plan is NULL");
    ig_ctx* ctx = p->ctx;
    IG_REQUIRE(ctx, p->padded && p->layout == 2, "ig_fft_exec_cropped_sum_slab: needs a plan of ig_fft_plan_padded with grid_layout 2");
    IG_REQUIRE(ctx, xv && yv && wv && workspace, "ig_fft_exec_cropped_sum_slab: NULL array");
    IG_REQUIRE(ctx, phase == 0 || phase == 1, "ig_fft_exec_cropped_sum_slab: phase must be 0 (z pass) or 1 (y and x passes of a slab)");
    IG_REQUIRE(ctx, phase == 0 || (0 <= z0 && z0 <= z1 && z1 <= p->box_dims[2]),
               "ig_fft_exec_cropped_sum_slab: slab [%lld, %lld) outside the image's %lld planes", (long long)z0, (long long)z1, (long long)p->box_dims[2]);

    if (int rc = ig_set_device(ctx)) return rc;
    return exec_cropped_layout2(p, (const float2*)yv, (const float2*)wv, (float2*)xv, (float2*)workspace,
                                (const short2*)support, true, phase == 0 ? 1 : 2, z0, z1);
}

int ig_fft_inplace_workspace(ig_fft* p, size_t* bytes) {
    if (!p || !bytes) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_inplace_workspace: bad arguments");
    *bytes = p->inplace_workspace_bytes > p->workspace_bytes ? p->inplace_workspace_bytes : p->workspace_bytes;
    return IG_OK;
}

int ig_fft_describe(ig_fft* p, char* buf, size_t len) {
    if (!p || !buf || len == 0) return ig_fail(nullptr, IG_ERR_ARG, "ig_fft_describe: bad arguments");
    snprintf(buf, len, "%s", p->desc.c_str());
    return IG_OK;
}

int ig_fft_destroy(ig_fft* p) {
    if (!p) return IG_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    for (int a = 0; a < 3; ++a)
        if (p->axis[a].d_tw) (void)hipFree(p->axis[a].d_tw);
    delete p;
    return IG_OK;
}

}  // extern "C"
