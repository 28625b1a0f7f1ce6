// Two-stage axis passes for transform lengths n = A * B that are not powers of two (160 ... 640: 320, 432, 480, 640 are the
// oversampled grids of the reference's own example, examples/pics.py:87-90 -- 640/480, 270/208, 432/308, 288/208, 400/308,
// 600/480, 392/308).  Same shape as the power-of-two kernel k_fft_2stage (ig_fft.hip) -- a column lives in the registers
// of B threads, one LDS exchange between two register-resident DFTs, one read and one write of every element per pass --
// but the small DFTs are generated at compile time for any length with factors 2, 3, 5, 7 (mixed-radix Cooley-Tukey,
// fully unrolled, twiddles as literals), and the two stages need not use the same number of threads:
//
//   j = b + B a  (a < A, b < B),   k = k1 + A k2  (k1 < A, k2 < B)
//   stage 1, thread b      : Y[b][k1] = sum_a x[b + B a] wA^(a k1)      (A-point DFT)   then  Y[b][k1] *= wn^(b k1)
//   exchange through LDS   : thread k1 collects Y[.][k1]
//   stage 2, thread k1 < A : X[k1 + A k2] = sum_b Y[b][k1] wB^(b k2)    (B-point DFT; the B - A other threads idle)
//
// A <= B and as close as the factors allow (16 x 20, 18 x 24, 20 x 24, 20 x 32, ...), so at most a third of the
// threads idle in the second DFT; the memory side stays balanced (B threads load A values, A threads store B).
// k_fft_ab: plain passes (the fftn / ifftn contract, backend.py:497-509); k_fft_ab_desc: the zero-pad-aware SENSE passes.
#pragma once
#include <type_traits>

// One axis pass, described generally enough for plain, zero-padded and cropped transforms.
// Columns are enumerated by three indices (k0 fastest, then k1, k2); element j of column k lives at
//   in  + k0*in_s[0]  + k1*in_s[1]  + k2*in_s[2]  + j*in_sj      (read only for in_lo  <= j < in_hi, else 0)
//   out + k0*out_s[0] + k1*out_s[1] + k2*out_s[2] + j*out_sj     (stored only for out_lo <= j < out_hi)
// The base pointers are pre-offset on the host (index origins of boxes / compact arrays), so they may
// point outside the buffers; they are only dereferenced inside the boxes.  Optional diagonal weights
// `w` (same indexing, own strides): WMODE 1 multiplies the inputs by w, WMODE 2 the outputs by conj(w); WMODE
// 3 + log2(cw) additionally sums the weighted outputs over the cw sub-columns of a column (SENSE coil combination)
// and stores the sum once, through the addressing of sub-column 0.
struct PassDesc {
    const float2* in; float2* out; const float2* w;
    // chirp-z passes (k_fft_chirp): w and w2 are ONE-dimensional tables along the transform axis (no column dependence)
    const float2* w2;
    int chirp_out;              // k_fft_chirp: 1 = the output weights are a table of their own, w2[m ..] (an axis whose transform carries a
                                // circular shift, ig_fft_set_axis_shift); 0 = they are the input chirp w times 1 / m
    int64_t in_sj, out_sj, w_sj;
    int64_t in_s[3], out_s[3], w_s[3];
    int64_t ext0, ext1, ncols;
    unsigned tpr;               // tiles per (k1, k2) row = ceil(ext0 / W); filled in by the launcher
    int grid3;                  // k_fft_2stage: the launch grid is (tpr, ext1, ext2) -- no divisions at the head of a workgroup; filled in by the launcher
    int cw_log2;                // log2(cw) where cw is a power of two, else -1; filled in by the launcher
    // Optional split of a tile's W lanes (strided passes): cw > 0 makes lane w address sub-column a = w % cw (element
    // strides in_sa / out_sa / w_sa) of column k0 = tile*(W/cw) + w/cw, i.e. a tile is W/cw columns of cw contiguous
    // sub-columns.  The coil-interleaved grid layout uses it for its x passes (cw = coils, columns = lines).
    int cw;
    int64_t in_sa, out_sa, w_sa;
    int tile_shift;             // tile_range / tile_bits entries are shared by 2^tile_shift consecutive tiles
    int in_lo, in_hi, out_lo, out_hi;
    int inverse;
    // optional per-tile override of the box along the transform axis (strided passes, W | ext0):
    // tile_range[k1 * (ext0 / W) + k0 / W] = (lo, hi); mode 1 narrows the OUTPUT box (tiles with an empty
    // range are skipped altogether), mode 2 narrows the INPUT box (everything outside reads as zero)
    const short2* tile_range;
    int tile_range_mode;
    int64_t tile_range_k1;      // table row stride per k1 (tiles per row), or 0 if the ranges do not depend on k1
    // optional refinement of tile_range (same mode, same indexing, tile_words words per tile; 0 means 16): bit m of word t is
    // set iff element j = t + tile_words*m of the tile's columns is needed (mode 1) / was ever written (mode 2).  The
    // power-of-two kernel holds the elements t + 16 m on both sides; the A x B kernel b + B a on its input side and
    // k1 + A k2 on its output side (tile_words = B in mode 2, A in mode 1: ig_grid_support writes both forms)
    const uint32_t* tile_bits;
    int tile_words;
    // optional: tiles whose k1 lies outside k1_range[tile >> tile_shift] = [lo, hi) are skipped altogether (the
    // cropped z pass: the y pass that follows never reads ky outside the kx tile's ky hull)
    const short2* k1_range;
};


#include "ig_packed.h"

// ---- the head of a pass workgroup (k_fft_2stage, k_fft_ab_desc, k_fft_chirp) -----------------------------------------------
// Round 5: everything between the start of a workgroup and its first grid row in flight is time in which the workgroup keeps
// nothing of HBM busy, and only two of these workgroups fit a CU.  Measured on the headline (same box, alternating runs): the
// twiddle copy written to LDS on the spot (a round trip the compiler waits for before anything else), the two integer divisions
// of the linear tile number and the support records fetched one after the other cost 2 ... 6 % of every y / z pass.
// (1) The launcher passes (tile of the row, k1, k2) as a three-dimensional grid where the extents allow: the same dispatch order
//     as the linear numbering, no divisions.
__device__ __forceinline__ void pass_tile(const PassDesc& d, unsigned& tr, unsigned& k1, unsigned& k2) {
    if (d.grid3) { tr = blockIdx.x; k1 = blockIdx.y; k2 = blockIdx.z; }
    else { const unsigned tile = blockIdx.x, rest = tile / d.tpr; tr = tile % d.tpr; k1 = rest % (unsigned)d.ext1; k2 = rest / (unsigned)d.ext1; }
}
// (2) The tile's support records -- ky hull of its kx tile, z range, bitmap word of this thread -- are INDEPENDENT loads, all
//     addressed by the tile index alone: requested together at the top of the kernel, used as late as possible.  The ky hull does
//     not depend on k1 (one word per kx tile: the same for a whole column of workgroups): a scalar load, served by the scalar
//     cache; so is a range that does not depend on k1 (the y passes).  A range per (k1, tile) and the bitmap word are vector
//     loads, the range read back with v_readlane where it is needed -- as a scalar load the compiler sank it into the block of
//     its first use, a dependent trip of its own behind the hull's.
struct PassRecords {
    uint32_t rec_v, rec_s, hull_s, zb_raw;
    bool has_k1r, has_trg, has_zb, word_ok, scalar;
    __device__ __forceinline__ short2 k1_hull() const { return make_short2((short)(hull_s & 0xffffu), (short)(hull_s >> 16)); }
    __device__ __forceinline__ short2 range() const {
        const uint32_t a = scalar ? rec_s : (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_v);
        return make_short2((short)(a & 0xffffu), (short)(a >> 16));
    }
    // this thread's bitmap word (all ones without a bitmap); called where the word is USED: a select on the loaded value right
    // behind the load makes the compiler wait for it there
    __device__ __forceinline__ uint32_t bits() const { return has_zb ? (word_ok ? zb_raw : 0u) : 0xffffffffu; }
};
__device__ __forceinline__ PassRecords pass_records(const PassDesc& d, const void* safe, unsigned tr, unsigned k1, int words, int word, bool word_ok) {
    PassRecords r;
    r.rec_v = 0; r.rec_s = 0; r.hull_s = 0x7fff0000u; r.zb_raw = 0xffffffffu; r.word_ok = word_ok;
    r.has_k1r = d.k1_range != nullptr; r.has_trg = d.tile_range != nullptr; r.has_zb = d.tile_bits != nullptr;
    r.scalar = r.has_trg && d.tile_range_k1 == 0;
    if (r.has_k1r) r.hull_s = *reinterpret_cast<const uint32_t*>(d.k1_range + (tr >> d.tile_shift));
    if (r.scalar) r.rec_s = *reinterpret_cast<const uint32_t*>(d.tile_range + (tr >> d.tile_shift));
    if ((r.has_trg && !r.scalar) || r.has_zb) {
        const int64_t tidx = (int64_t)k1 * d.tile_range_k1 + (tr >> d.tile_shift);
        const uint32_t* pt = (r.has_trg && !r.scalar) ? reinterpret_cast<const uint32_t*>(d.tile_range + tidx) : reinterpret_cast<const uint32_t*>(safe);
        const uint32_t* pb = (r.has_zb && word_ok) ? d.tile_bits + (tidx * words + word) : reinterpret_cast<const uint32_t*>(safe);
        r.zb_raw = *pb;
        r.rec_v = *pt;
    }
    return r;
}

namespace anyfft {

constexpr double kPi = 3.14159265358979323846264338327950288;

// cos / sin of 2 pi m / n for integers, evaluated at compile time (Taylor series on (-pi, pi], double precision)
constexpr double c_cos_x(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int k = 1; k <= 16; ++k) { term *= -x2 / ((2.0 * k - 1.0) * (2.0 * k)); sum += term; }
    return sum;
}
constexpr double c_sin_x(double x) {
    double x2 = x * x, term = x, sum = x;
    for (int k = 1; k <= 16; ++k) { term *= -x2 / ((2.0 * k) * (2.0 * k + 1.0)); sum += term; }
    return sum;
}
constexpr int c_mod(int m, int n) { return ((m % n) + n) % n; }
constexpr double c_angle(int m, int n) {            // 2 pi m / n reduced to (-pi, pi]
    int r = c_mod(m, n);
    if (2 * r > n) r -= n;
    return 2.0 * kPi * (double)r / (double)n;
}
constexpr float cos2pi(int m, int n) {
    const int r = c_mod(m, n);
    if (r == 0) return 1.0f;
    if (2 * r == n) return -1.0f;
    if (4 * r == n || 4 * r == 3 * n) return 0.0f;
    return (float)c_cos_x(c_angle(m, n));
}
constexpr float sin2pi(int m, int n) {
    const int r = c_mod(m, n);
    if (r == 0 || 2 * r == n) return 0.0f;
    if (4 * r == n) return 1.0f;
    if (4 * r == 3 * n) return -1.0f;
    return (float)c_sin_x(c_angle(m, n));
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- the element type of the generated DFTs: float2 (scalar arithmetic: the bandwidth-bound A x B passes) or cx (ig_packed.h:
// one v_pk_* instruction per complex add, two per complex product: the compute-bound chirp-z passes) ----------------------------
__device__ __forceinline__ float2 d_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 d_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 d_addmi(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }      // a + (-i) b
__device__ __forceinline__ float2 d_addpi(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }      // a + (+i) b
__device__ __forceinline__ float2 d_fma_r(float c, float2 t, float2 acc) { return make_float2(fmaf(c, t.x, acc.x), fmaf(c, t.y, acc.y)); }
__device__ __forceinline__ float2 d_mul_r(float c, float2 t) { return make_float2(c * t.x, c * t.y); }
__device__ __forceinline__ cx d_add(cx a, cx b) { return a + b; }
__device__ __forceinline__ cx d_sub(cx a, cx b) { return a - b; }
__device__ __forceinline__ cx d_addmi(cx a, cx b) { return padd_mi(a, b); }
__device__ __forceinline__ cx d_addpi(cx a, cx b) { return padd_pi(a, b); }
__device__ __forceinline__ cx d_fma_r(float c, cx t, cx acc) { return cfma_r(c, t, acc); }
__device__ __forceinline__ cx d_mul_r(float c, cx t) { cx r; r.v = t.v * v2f{c, c}; return r; }

// a * exp(-2 pi i M / N) with M, N compile-time: rotations by multiples of a quarter turn cost no multiply
template <int M, int N>
__device__ __forceinline__ float2 rot(float2 a) {
    constexpr int r = c_mod(M, N);
    if constexpr (r == 0) return a;
    else if constexpr (4 * r == N) return make_float2(a.y, -a.x);
    else if constexpr (2 * r == N) return make_float2(-a.x, -a.y);
    else if constexpr (4 * r == 3 * N) return make_float2(-a.y, a.x);
    else {
        constexpr float c = cos2pi(r, N), s = -sin2pi(r, N);          // w = c + i s
        return make_float2(fmaf(a.x, c, -a.y * s), fmaf(a.x, s, a.y * c));
    }
}
template <int M, int N>
__device__ __forceinline__ cx rot(cx a) {
    constexpr int r = c_mod(M, N);
    if constexpr (r == 0) return a;
    else if constexpr (4 * r == N) return cmul_mi(a);
    else if constexpr (2 * r == N) return cneg(a);
    else if constexpr (4 * r == 3 * N) return cmul_pi(a);
    else {
        constexpr float c = cos2pi(r, N), s = -sin2pi(r, N);          // w = c + i s
        return cxmul(a, mk(c, s));
    }
}

constexpr int pick_radix(int n) { return n % 4 == 0 ? 4 : n % 2 == 0 ? 2 : n % 3 == 0 ? 3 : n % 5 == 0 ? 5 : n % 7 == 0 ? 7 : n; }

// forward DFT of N register values of type T (float2 or cx), in place, natural order
template <int N, typename T = float2, bool BASE = (pick_radix(N) == N)>
struct RegDFT;

template <typename T> struct RegDFT<1, T, true> { __device__ static __forceinline__ void run(T (&)[1]) {} };
template <typename T> struct RegDFT<2, T, true> {
    __device__ static __forceinline__ void run(T (&x)[2]) {
        const T a = x[0], b = x[1];
        x[0] = d_add(a, b); x[1] = d_sub(a, b);
    }
};
template <typename T> struct RegDFT<4, T, true> {
    __device__ static __forceinline__ void run(T (&x)[4]) {
        const T t0 = d_add(x[0], x[2]), t1 = d_sub(x[0], x[2]), t2 = d_add(x[1], x[3]), d = d_sub(x[1], x[3]);
        x[0] = d_add(t0, t2); x[2] = d_sub(t0, t2);
        x[1] = d_addmi(t1, d); x[3] = d_addpi(t1, d);              // t1 -+ i (x1 - x3)
    }
};
// odd primes (3, 5, 7): X_k, X_{P-k} = a_k -/+ i b_k with a_k = x0 + sum_q cos(2 pi q k / P) (x_q + x_{P-q}),
// b_k = sum_q sin(2 pi q k / P) (x_q - x_{P-q})
template <int P, typename T>
struct RegDFT<P, T, true> {
    static_assert(P == 3 || P == 5 || P == 7, "RegDFT: lengths with factors 2, 3, 5, 7 only");
    __device__ static __forceinline__ void run(T (&x)[P]) {
        constexpr int H = (P - 1) / 2;
        T t[H], d[H];
        static_for<0, H>([&](auto q_) {
            constexpr int q = decltype(q_)::value + 1;
            t[q - 1] = d_add(x[q], x[P - q]);
            d[q - 1] = d_sub(x[q], x[P - q]);
        });
        const T x0 = x[0];
        T s0 = x0;
        static_for<0, H>([&](auto q_) { s0 = d_add(s0, t[decltype(q_)::value]); });
        x[0] = s0;
        static_for<0, H>([&](auto k_) {
            constexpr int k = decltype(k_)::value + 1;
            T a = x0, b = d_mul_r(sin2pi(k, P), d[0]);
            static_for<0, H>([&](auto q_) {
                constexpr int q = decltype(q_)::value + 1;
                constexpr float c = cos2pi(q * k, P), s = sin2pi(q * k, P);
                a = d_fma_r(c, t[q - 1], a);
                if constexpr (q > 1) b = d_fma_r(s, d[q - 1], b);
            });
            x[k] = d_addmi(a, b);               // a - i b
            x[P - k] = d_addpi(a, b);
        });
    }
};
// composite N = P Q:  n = Q n1 + n2,  k = k1 + P k2
template <int N, typename T>
struct RegDFT<N, T, false> {
    __device__ static __forceinline__ void run(T (&x)[N]) {
        constexpr int P = pick_radix(N), Q = N / P;
        T y[Q][P];
        static_for<0, Q>([&](auto n2_) {
            constexpr int n2 = decltype(n2_)::value;
            T a[P];
            static_for<0, P>([&](auto n1_) { constexpr int n1 = decltype(n1_)::value; a[n1] = x[Q * n1 + n2]; });
            RegDFT<P, T>::run(a);
            static_for<0, P>([&](auto k1_) { constexpr int k1 = decltype(k1_)::value; y[n2][k1] = rot<n2 * k1, N>(a[k1]); });
        });
        static_for<0, P>([&](auto k1_) {
            constexpr int k1 = decltype(k1_)::value;
            T b[Q];
            static_for<0, Q>([&](auto n2_) { constexpr int n2 = decltype(n2_)::value; b[n2] = y[n2][k1]; });
            RegDFT<Q, T>::run(b);
            static_for<0, Q>([&](auto k2_) { constexpr int k2 = decltype(k2_)::value; x[k1 + P * k2] = b[k2]; });
        });
    }
};

constexpr int AB_W = 16;             // columns per workgroup: 128-byte segments on strided axes

// LDS exchange slot of Y[b][k1] of column w (k1 relative to the round's first k1; AR = k1 values per round)
template <int AR, int B, bool AXIS0>
__device__ __forceinline__ int ab_slot(int k1, int b, int w) {
    // strided axes: lanes run over w (then b): a wave writes 64 consecutive slots, reads 16-slot runs
    // contiguous lines: lanes run over b (writing) / k1 (reading): b fastest, rows padded to an odd length
    if (AXIS0) return (w * AR + k1) * (B | 1) + b;
    return (k1 * B + b) * AB_W + w;
}
template <int A, int B, int ROUNDS, bool AXIS0>
constexpr size_t ab_lds_bytes() {
    constexpr int AR = (A + ROUNDS - 1) / ROUNDS;
    return ((size_t)(AXIS0 ? AB_W * AR * (B | 1) : AR * B * AB_W) + (size_t)A * B) * 8;
}

// One axis pass.  AXIS0: the columns are contiguous lines of n elements (inner == 1).  ROUNDS = 2 halves the LDS
// footprint (the exchange runs once per half of the k1 range) for the long lengths.
template <int A, int B, int ROUNDS, bool AXIS0>
__global__ void __launch_bounds__(AB_W * B)
k_fft_ab(const float2* __restrict__ x, float2* __restrict__ y, const float2* __restrict__ tw,
         int64_t inner, int64_t ncols, int inverse) {
    constexpr int N = A * B, AR = (A + ROUNDS - 1) / ROUNDS;
    extern __shared__ float2 lds[];
    float2* __restrict__ tws = lds + (AXIS0 ? AB_W * AR * (B | 1) : AR * B * AB_W);
    const int tid = threadIdx.x;
    // the twiddles: requested here, written to LDS behind the stage-1 loads (see pass_tile above: written on the spot, the copy is a
    // round trip of its own at the head of every workgroup)
    constexpr int TWN = (N + AB_W * B - 1) / (AB_W * B);
    float2 tw_mine[TWN];
#pragma unroll
    for (int i = 0; i < TWN; ++i) { const int k = tid + i * AB_W * B; tw_mine[i] = tw[k < N ? k : N - 1]; }
    const int b = AXIS0 ? tid % B : tid / AB_W, w = AXIS0 ? tid / B : tid % AB_W;
    const int64_t col = (int64_t)blockIdx.x * AB_W + w;
    const bool valid = col < ncols;
    int64_t base, sj;
    if (AXIS0) { base = col * N; sj = 1; }
    else { const int64_t o = col / inner, i = col - o * inner; base = i + inner * N * o; sj = inner; }
    const bool inv = inverse != 0;

    float2 v[A];
    if (valid) {
        const float2* __restrict__ src = x + base + (int64_t)b * sj;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float2 e = src[(int64_t)(B * a) * sj];
            if (inv) e.y = -e.y;
            v[a] = e;
        }
    } else {
#pragma unroll
        for (int a = 0; a < A; ++a) v[a] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TWN; ++i) if (tid + i * AB_W * B < N) tws[tid + i * AB_W * B] = tw_mine[i];
    RegDFT<A>::run(v);
    __syncthreads();                                   // the twiddle table is in place
#pragma unroll
    for (int k1 = 1; k1 < A; ++k1) {                   // b k1 < n: no reduction needed
        const float2 t = tws[b * k1];
        v[k1] = make_float2(fmaf(v[k1].x, t.x, -v[k1].y * t.y), fmaf(v[k1].x, t.y, v[k1].y * t.x));
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r) __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1)
            if (k1 >= r * AR && k1 < (r + 1) * AR) lds[ab_slot<AR, B, AXIS0>(k1 - r * AR, b, w)] = v[k1];
        __syncthreads();
        const int k1 = b;                              // stage 2: this thread's output residue
        if (k1 >= r * AR && k1 < (r + 1) * AR && k1 < A) {
            float2 u[B];
#pragma unroll
            for (int bb = 0; bb < B; ++bb) u[bb] = lds[ab_slot<AR, B, AXIS0>(k1 - r * AR, bb, w)];
            RegDFT<B>::run(u);
            if (valid) {
                float2* __restrict__ dst = y + base + (int64_t)k1 * sj;
#pragma unroll
                for (int k2 = 0; k2 < B; ++k2) {
                    float2 e = u[k2];
                    if (inv) e.y = -e.y;
                    dst[(int64_t)(A * k2) * sj] = e;
                }
            }
        }
    }
}

// ---- the same pass through a PassDesc: zero-padded / cropped / weighted / coil-summing passes for n = A * B -----------------
// What k_fft_2stage does for the 256 / 512-point axes of the SENSE transform, for the oversampled grids of the reference's own
// driver (320, 384, 400, 432, 480, 640: examples/pics.py:87-90): inputs outside [in_lo, in_hi) are zeros that are never
// loaded, outputs outside [out_lo, out_hi) are never stored (buffer-descriptor range checks, no branches per element; whole
// load instructions no lane of the wave wants are skipped by a scalar branch), WMODE 1 multiplies the inputs by the weights,
// 2 the outputs by their conjugates, 3 + log2(cw) also sums the cw sub-columns (coils) of a column with DPP adds; `cw` splits
// a tile's 16 lanes into 16 / cw columns x cw sub-columns (the x passes of the coil-interleaved layout).  Strided passes only
// (every pass of that layout is one).  The k-space support table narrows the boxes per tile exactly as in k_fft_2stage
// (tile_range / k1_range: tiles outside the hulls leave at once; tile_bits: one word per thread and-ed into its element mask),
// and unweighted passes skip the store instructions no lane of a wave wants by a scalar branch (buf_st_gated).
template <int CTRL>
__device__ __forceinline__ float ab_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ int ab_ceil_div_clamp(int num, int den, int hi) {       // ceil(num / den) clamped to [0, hi]
    if (num <= 0) return 0;
    const int q = (num + den - 1) / den;
    return q > hi ? hi : q;
}

template <int AR, int B, int W>
__device__ __forceinline__ int abd_slot(int k1, int b, int w) { return (k1 * B + b) * W + w; }

template <int A, int B, int ROUNDS, int WMODE, int W = AB_W>          // W columns per workgroup (16: 32 were measured in round 5 and lost)
__global__ void __launch_bounds__(W * B)
k_fft_ab_desc(PassDesc d, const float2* __restrict__ tw) {
    constexpr int N = A * B, AR = (A + ROUNDS - 1) / ROUNDS;
    constexpr int SUMW = WMODE >= 3 ? (1 << (WMODE - 3)) : 0;
    static_assert(A <= 32 && B <= 32, "element masks are 32-bit words");
    extern __shared__ float2 lds[];
    float2* __restrict__ tws = lds + AR * B * W;
    const int tid = threadIdx.x;
    // the twiddles: requested here, written to LDS behind the stage-1 loads (written on the spot they are a round trip of their own)
    constexpr int TWN = (N + W * B - 1) / (W * B);
    float2 tw_mine[TWN];
#pragma unroll
    for (int i = 0; i < TWN; ++i) { const int k = tid + i * W * B; tw_mine[i] = tw[k < N ? k : N - 1]; }       // (no branch: the load must not be waited for here)
    const int b = tid / W, w = tid % W;
    const bool inv = d.inverse != 0;
    // the workgroup's tile: 16 consecutive k0 (or 16 / cw columns x cw sub-columns) of one (k1, k2) row
    unsigned tr, k1i, k2i;
    pass_tile(d, tr, k1i, k2i);
    const int tw_ = d.tile_words ? d.tile_words : 16;
    const PassRecords rcd = pass_records(d, tw, tr, k1i, tw_, b, b < tw_);
    const int64_t k0u = (int64_t)tr * (d.cw ? (d.cw_log2 >= 0 ? W >> d.cw_log2 : W / d.cw) : W);
    const float2* const b_in = d.in + (k0u * d.in_s[0] + (int64_t)k1i * d.in_s[1] + (int64_t)k2i * d.in_s[2]);
    float2* const b_out = d.out + (k0u * d.out_s[0] + (int64_t)k1i * d.out_s[1] + (int64_t)k2i * d.out_s[2]);
    const float2* const b_w = WMODE ? d.w + (k0u * d.w_s[0] + (int64_t)k1i * d.w_s[1] + (int64_t)k2i * d.w_s[2]) : nullptr;
    const unsigned isj = (unsigned)d.in_sj, osj = (unsigned)d.out_sj, wsj = (unsigned)d.w_sj;
    bool valid;
    unsigned l_in, l_out, l_w;
    if (d.cw) {
        const unsigned yl = d.cw_log2 >= 0 ? (unsigned)w >> d.cw_log2 : (unsigned)w / (unsigned)d.cw, a = (unsigned)w - yl * (unsigned)d.cw;
        valid = k0u + yl < d.ext0;
        l_in = (a * (unsigned)d.in_sa + yl * (unsigned)d.in_s[0] + (unsigned)b * isj) * 8u;
        l_out = (a * (unsigned)d.out_sa + yl * (unsigned)d.out_s[0] + (unsigned)b * osj) * 8u;
        l_w = (a * (unsigned)d.w_sa + yl * (unsigned)d.w_s[0] + (unsigned)b * wsj) * 8u;
    } else {
        valid = k0u + w < d.ext0;
        l_in = ((unsigned)w * (unsigned)d.in_s[0] + (unsigned)b * isj) * 8u;
        l_out = ((unsigned)w * (unsigned)d.out_s[0] + (unsigned)b * osj) * 8u;
        l_w = ((unsigned)w * (unsigned)d.w_s[0] + (unsigned)b * wsj) * 8u;
    }
    if (!valid) l_in = l_out = l_w = IG_OOB;
    if (!WMODE) l_w = IG_OOB;
    if (SUMW && (w % SUMW) != 0) l_out = IG_OOB;            // only a column's first sub-column stores the coil sum
    // the tile's support records (requested at the top: pass_records), tested here
    int in_lo = d.in_lo, in_hi = d.in_hi, out_lo = d.out_lo, out_hi = d.out_hi;
    const bool defer_out = rcd.has_trg && d.tile_range_mode == 1 && (rcd.scalar || rcd.has_k1r);
    {
        if (rcd.has_k1r) { const short2 k1r = rcd.k1_hull(); if ((int)k1i < k1r.x || (int)k1i >= k1r.y) return; }          // (wave- and workgroup-uniform: before any barrier)
        // (a zero-padded pass does not wait for its range here where it need not: its loads need nothing of it, the output side is
        // narrowed behind them.  The y pass -- output hull of the kx tile, never empty inside a support -- and a z pass whose tiles
        // outside the ky hull of their kx tile have left above: an empty range inside the hull only leaves no store flagged.)
        if (rcd.has_trg && !defer_out) {
            const short2 trg = rcd.range();
            if (d.tile_range_mode == 1) {
                out_lo = out_lo > trg.x ? out_lo : trg.x;
                out_hi = out_hi < trg.y ? out_hi : trg.y;
                if (out_hi <= out_lo) return;                                           // nothing of this tile is ever read
            } else {
                in_lo = in_lo > trg.x ? in_lo : trg.x;
                in_hi = in_hi < trg.y ? in_hi : trg.y;
            }
        }
    }
    // element j = b + B a <-> bit a of ibits (stage 1 loads); output k = b + A k2 <-> bit k2 of obits (stage 2 stores, b < A)
    auto below = [](int h) -> uint32_t { return h >= 32 ? 0xffffffffu : ((1u << h) - 1u); };
    uint32_t ibits = below(ab_ceil_div_clamp(in_hi - b, B, A)) & ~below(ab_ceil_div_clamp(in_lo - b, B, A));
    if (d.tile_range_mode != 1) ibits &= rcd.bits();
    uint32_t gin = 0;                                          // wave-uniform: elements SOME lane of this wave wants
#pragma unroll
    for (int l = 0; l < 64; l += W) gin |= (uint32_t)__builtin_amdgcn_readlane((int)ibits, l);

    float2 v[A];
    {
        float2 wv[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            if (!((gin >> a) & 1u)) { v[a] = make_float2(0.f, 0.f); if (WMODE == 1) wv[a] = make_float2(0.f, 0.f); continue; }
            const unsigned off = (unsigned)__builtin_amdgcn_sbfe((int)~ibits, a, 1);
            v[a] = buf_ld<true>(make_rsrc(b_in + (int64_t)(B * a) * d.in_sj), l_in | off, 0);
            if (WMODE == 1) wv[a] = buf_ld<false>(make_rsrc(b_w + (int64_t)(B * a) * d.w_sj), l_w | off, 0);
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            if (WMODE == 1) v[a] = make_float2(fmaf(v[a].x, wv[a].x, -v[a].y * wv[a].y), fmaf(v[a].x, wv[a].y, v[a].y * wv[a].x));
            if (inv) v[a].y = -v[a].y;
        }
    }
    // ---- the output side of the boxes, behind the loads
    if (defer_out) {
        const short2 trg = rcd.range();
        out_lo = out_lo > trg.x ? out_lo : trg.x;
        out_hi = out_hi < trg.y ? out_hi : trg.y;
    }
    uint32_t obits = below(ab_ceil_div_clamp(out_hi - b, A, B)) & ~below(ab_ceil_div_clamp(out_lo - b, A, B));
    if (d.tile_range_mode == 1) obits &= rcd.bits();
    uint32_t gout = 0xffffffffu;                               // wave-uniform: outputs SOME lane of this wave keeps (unweighted passes gate their stores with it)
    if (WMODE == 0) {
        gout = 0;
#pragma unroll
        for (int l = 0; l < 64; l += W) gout |= (uint32_t)__builtin_amdgcn_readlane((int)obits, l);
    }
#pragma unroll
    for (int i = 0; i < TWN; ++i) if (tid + i * W * B < N) tws[tid + i * W * B] = tw_mine[i];
    RegDFT<A>::run(v);
    __syncthreads();                                   // the twiddle table is in place
#pragma unroll
    for (int k1 = 1; k1 < A; ++k1) {
        const float2 t = tws[b * k1];
        v[k1] = make_float2(fmaf(v[k1].x, t.x, -v[k1].y * t.y), fmaf(v[k1].x, t.y, v[k1].y * t.x));
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r) __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1)
            if (k1 >= r * AR && k1 < (r + 1) * AR) lds[abd_slot<AR, B, W>(k1 - r * AR, b, w)] = v[k1];
        __syncthreads();
        const int k1 = b;                              // stage 2: this thread's output residue
        if (k1 >= r * AR && k1 < (r + 1) * AR && k1 < A) {
            float2 u[B];
#pragma unroll
            for (int bb = 0; bb < B; ++bb) u[bb] = lds[abd_slot<AR, B, W>(k1 - r * AR, bb, w)];
            RegDFT<B>::run(u);
            float2 wv[B];
            if (WMODE >= 2) {
#pragma unroll
                for (int k2 = 0; k2 < B; ++k2) {
                    const unsigned off = (unsigned)__builtin_amdgcn_sbfe((int)~obits, k2, 1);
                    wv[k2] = buf_ld<false>(make_rsrc(b_w + (int64_t)(A * k2) * d.w_sj), l_w | off, 0);
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < B; ++k2) {
                float2 e = u[k2];
                if (inv) e.y = -e.y;
                if (WMODE >= 2) e = make_float2(fmaf(wv[k2].x, e.x, wv[k2].y * e.y), fmaf(wv[k2].x, e.y, -wv[k2].y * e.x));      // conj(w) * e
                if (SUMW >= 2)  { e.x += ab_dpp<0xB1>(e.x);  e.y += ab_dpp<0xB1>(e.y); }      // quad_perm [1,0,3,2]
                if (SUMW >= 4)  { e.x += ab_dpp<0x4E>(e.x);  e.y += ab_dpp<0x4E>(e.y); }      // quad_perm [2,3,0,1]
                if (SUMW >= 8)  { e.x += ab_dpp<0x104>(e.x); e.y += ab_dpp<0x104>(e.y); }     // row_shl:4
                if (SUMW >= 16) { e.x += ab_dpp<0x108>(e.x); e.y += ab_dpp<0x108>(e.y); }     // row_shl:8
                const unsigned off = (unsigned)__builtin_amdgcn_sbfe((int)~obits, k2, 1);
                if (WMODE == 0) buf_st_gated<true>(make_rsrc_words(b_out + (int64_t)(A * k2) * d.out_sj), l_out | off, 0u, e, (gout >> k2) & 1u);
                else buf_st<true>(make_rsrc(b_out + (int64_t)(A * k2) * d.out_sj), l_out | off, 0, e);
            }
        }
    }
}

// ---- chirp-z (Bluestein) pass in ONE launch: an axis of n points with a prime factor above 7 (277, 410 = 2 * 5 * 41: what
// int(N * osf) of the reference's driver produces, indigo/backends/backend.py:427-430) as a cyclic convolution of length
// m = A * B >= 2 n - 1,
//      X_k = b_k sum_j (x_j b_j) conj(b)_{k - j},      b_j = exp(-i pi j^2 / n),
// i.e. F_m^-1( F_m(x b) . F_m(conj b) ) times b.  Both length-m transforms run back to back in the registers and the LDS of the
// workgroup that holds the column -- a column is read once (n elements) and written once (n elements), like any other axis pass;
// the length-m intermediates never reach memory.  The OUTPUT layout of the (A, B) split -- thread k1 < A holds X[k1 + A k2] --
// is the INPUT layout of the (B, A) split -- thread b' < A holds x[b' + A a'] --, so the second transform is the same two-stage
// scheme with the roles of A and B exchanged: B-point DFTs on the A threads that hold the first result, one exchange, A-point
// DFTs on all B threads, whose outputs k1' + B k2' land where the first transform's inputs came from.
//   d.w   b_j        (m entries; inputs j >= n are zeros that are never loaded: the input box must lie inside [0, n))
//   d.w2  F_m of the wrapped conjugate chirp (m entries), then b_k / m (m entries): tables of the direction d.inverse asks for
// Boxes as everywhere: inputs outside [in_lo, in_hi) are not read, outputs outside [out_lo, out_hi) (inside [0, n)) not stored.
template <int A, int B, int ROUNDS>
__global__ void __launch_bounds__(AB_W * B)
k_fft_chirp(PassDesc d, const float2* __restrict__ tw) {
    constexpr int N = A * B, AR = (A + ROUNDS - 1) / ROUNDS, BR = (B + ROUNDS - 1) / ROUNDS;
    static_assert(A <= 32 && B <= 32 && A <= B, "element masks are 32-bit words");
    constexpr int EX1 = AR * B * AB_W, EX2 = BR * A * AB_W;
    extern __shared__ float2 lds[];
    float2* __restrict__ tws = lds + (EX1 > EX2 ? EX1 : EX2);
    // The chirp, its transform and the output weights in LDS beside the twiddles (round 5): a thread reads 2 A + B of them, the same
    // for all 16 columns of its row -- as global loads they were three times the tile's own loads and stores in vector-memory
    // instructions, every one a 16-lane broadcast through the address unit.
    float2* __restrict__ t_b = tws + N;
    float2* __restrict__ t_hat = t_b + N;
    // (round 6: the output weights b_k / m are the chirp itself times 1 / m -- no table of their own: the 6.9 KB it took at m = 864 were
    // what kept a second workgroup off the CU there, and a third at m = 560)
    // (an axis with a circular shift folded in, ig_fft_set_axis_shift, has input weights b_(j - c) and output weights b_k: a fourth table
    // behind the others, which the launcher adds to the dynamic LDS only then)
    constexpr float INV_M = 1.0f / (float)N;
    const bool outtab = d.chirp_out != 0;
    float2* __restrict__ t_o = t_hat + N;
    const int tid = threadIdx.x;
    // (requested here, written to LDS behind the first transform's loads: written on the spot the copies are a round trip of their own
    // at the head of a workgroup of which one or two fit a CU)
    constexpr int TWN = (N + AB_W * B - 1) / (AB_W * B);
    float2 tab_mine[TWN][4];
#pragma unroll
    for (int i = 0; i < TWN; ++i) {
        const int k = tid + i * AB_W * B, kk = k < N ? k : N - 1;
        tab_mine[i][0] = tw[kk]; tab_mine[i][1] = d.w[kk]; tab_mine[i][2] = d.w2[kk];
        tab_mine[i][3] = d.w2[outtab ? N + kk : kk];          // (no branch: the load must not be waited for here)
    }
    const int b = tid / AB_W, w = tid % AB_W;
    unsigned tr, k1i, k2i;
    pass_tile(d, tr, k1i, k2i);
    const PassRecords rcd = pass_records(d, tw, tr, k1i, B, b, true);
    const int64_t k0u = (int64_t)tr * AB_W;
    const float2* const b_in = d.in + (k0u * d.in_s[0] + (int64_t)k1i * d.in_s[1] + (int64_t)k2i * d.in_s[2]);
    float2* const b_out = d.out + (k0u * d.out_s[0] + (int64_t)k1i * d.out_s[1] + (int64_t)k2i * d.out_s[2]);
    const bool valid = k0u + w < d.ext0;
    unsigned l_in = ((unsigned)w * (unsigned)d.in_s[0] + (unsigned)b * (unsigned)d.in_sj) * 8u;
    unsigned l_out = ((unsigned)w * (unsigned)d.out_s[0] + (unsigned)b * (unsigned)d.out_sj) * 8u;
    if (!valid) l_in = l_out = IG_OOB;
    // the k-space support table (round 5): tiles outside the ky hull of their kx tile, or with an empty z range, leave at once;
    // inside, the range narrows the output box (zero-padded forward pass) or the input box (cropped inverse pass), and the segment
    // bitmap masks single rows.  A thread holds the rows b + B a on its input side and b + B k2 on its output side -- of a
    // transform of m = A B >= 2 n - 1 points, of which only rows below n exist --, so the bitmaps of a chirp-z axis are B words per
    // entry on both sides: bit kz / B of word kz % B (ig_fft_support_words).  All of it wave- and workgroup-uniform, before any barrier.
    int in_lo = d.in_lo, in_hi = d.in_hi, out_lo = d.out_lo, out_hi = d.out_hi;
    {
        if (rcd.has_k1r) { const short2 k1r = rcd.k1_hull(); if ((int)k1i < k1r.x || (int)k1i >= k1r.y) return; }
        if (rcd.has_trg) {
            const short2 trg = rcd.range();
            if (d.tile_range_mode == 1) {
                out_lo = out_lo > trg.x ? out_lo : trg.x;
                out_hi = out_hi < trg.y ? out_hi : trg.y;
                if (out_hi <= out_lo) return;
            } else {
                in_lo = in_lo > trg.x ? in_lo : trg.x;
                in_hi = in_hi < trg.y ? in_hi : trg.y;
            }
        }
    }
    auto below = [](int h) -> uint32_t { return h >= 32 ? 0xffffffffu : ((1u << h) - 1u); };
    // input j = b + B a <-> bit a; output k = b + B k2 (k2 < A) <-> bit k2
    uint32_t ibits = below(ab_ceil_div_clamp(in_hi - b, B, A)) & ~below(ab_ceil_div_clamp(in_lo - b, B, A));
    uint32_t obits = below(ab_ceil_div_clamp(out_hi - b, B, A)) & ~below(ab_ceil_div_clamp(out_lo - b, B, A));
    if (d.tile_range_mode == 1) obits &= rcd.bits(); else ibits &= rcd.bits();
    uint32_t gin = 0, gout = 0;
#pragma unroll
    for (int l = 0; l < 64; l += AB_W) { gin |= (uint32_t)__builtin_amdgcn_readlane((int)ibits, l); gout |= (uint32_t)__builtin_amdgcn_readlane((int)obits, l); }
    // Packed arithmetic throughout (cx, ig_packed.h): these passes are bound by their two length-m transforms per column, not by
    // memory -- one v_pk_* instruction per complex add, two per product, the conjugations folded into the products.

    // ---- first transform, (A, B): v_j = x_j b_j
    cx v[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        if (!((gin >> a) & 1u)) { v[a] = mk(0.f, 0.f); continue; }
        const unsigned off = (unsigned)__builtin_amdgcn_sbfe((int)~ibits, a, 1);
        v[a] = from2(buf_ld<true>(make_rsrc(b_in + (int64_t)(B * a) * d.in_sj), l_in | off, 0));
    }
#pragma unroll
    for (int i = 0; i < TWN; ++i) {
        const int k = tid + i * AB_W * B;
        if (k < N) { tws[k] = tab_mine[i][0]; t_b[k] = tab_mine[i][1]; t_hat[k] = tab_mine[i][2]; if (outtab) t_o[k] = tab_mine[i][3]; }
    }
    __syncthreads();                                   // the tables are in place (the loads above are in flight)
#pragma unroll
    for (int a = 0; a < A; ++a)
        if ((gin >> a) & 1u) v[a] = cxmul_r(v[a], from2(t_b[b + B * a]));
    RegDFT<A, cx>::run(v);
#pragma unroll
    for (int k1 = 1; k1 < A; ++k1) v[k1] = cxmul_r(v[k1], from2(tws[b * k1]));
    cx u[B];                                           // threads b < A: U[b + A k2] after the first transform
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r) __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1)
            if (k1 >= r * AR && k1 < (r + 1) * AR) lds[ab_slot<AR, B, false>(k1 - r * AR, b, w)] = to2(v[k1]);
        __syncthreads();
        if (b >= r * AR && b < (r + 1) * AR && b < A) {
#pragma unroll
            for (int bb = 0; bb < B; ++bb) u[bb] = from2(lds[ab_slot<AR, B, false>(b - r * AR, bb, w)]);
        }
    }
    const float2* __restrict__ t_ow = outtab ? t_o : t_b;          // output weights: their own table (already / m) or the chirp times 1 / m
    const float om = outtab ? 1.0f : INV_M;
    // ---- the convolution in the frequency domain, and the second (inverse) transform, (B, A): conj, forward, conj
    if (b < A) {
        RegDFT<B, cx>::run(u);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) u[k2] = cxmul_cc(u[k2], from2(t_hat[b + A * k2]));      // conj(U . hat)
        RegDFT<B, cx>::run(u);                         // thread b' = b < A holds the B inputs b' + A a'
#pragma unroll
        for (int k1 = 1; k1 < B; ++k1) u[k1] = cxmul_r(u[k1], from2(tws[b * k1]));       // w_m^(b' k1')
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        __syncthreads();                               // (the previous exchange has been read)
        if (b < A) {
#pragma unroll
            for (int k1 = 0; k1 < B; ++k1)
                if (k1 >= r * BR && k1 < (r + 1) * BR) lds[ab_slot<BR, A, false>(k1 - r * BR, b, w)] = to2(u[k1]);
        }
        __syncthreads();
        if (b >= r * BR && b < (r + 1) * BR) {         // stage 2 on all B threads: output residue k1' = b
            cx y[A];
#pragma unroll
            for (int bb = 0; bb < A; ++bb) y[bb] = from2(lds[ab_slot<BR, A, false>(b - r * BR, bb, w)]);
            RegDFT<A, cx>::run(y);
#pragma unroll
            for (int k2 = 0; k2 < A; ++k2) {           // X[b + B k2], k2 < A: only k < n is ever kept
                if (!((gout >> k2) & 1u)) continue;
                const float2 ob = t_ow[b + B * k2];
                const cx e = cxmulc(y[k2], from2(make_float2(ob.x * om, ob.y * om)));            // conj(y) . b_k / m
                const unsigned off = (unsigned)__builtin_amdgcn_sbfe((int)~obits, k2, 1);
                buf_st<true>(make_rsrc(b_out + (int64_t)(B * k2) * d.out_sj), l_out | off, 0, to2(e));
            }
        }
    }
}
template <int A, int B, int ROUNDS>
constexpr size_t chirp_lds_bytes() {
    constexpr int AR = (A + ROUNDS - 1) / ROUNDS, BR = (B + ROUNDS - 1) / ROUNDS;
    constexpr int EX1 = AR * B * AB_W, EX2 = BR * A * AB_W;
    return ((size_t)(EX1 > EX2 ? EX1 : EX2) + (size_t)3 * A * B) * 8;          // exchange image + twiddles + the chirp and its transform
}
// exchange rounds of the chirp-z kernel: as many (up to three) as let one more workgroup onto the CU's 160 KB, else one
template <int A, int B>
constexpr int chirp_rounds() {
    constexpr size_t L1 = chirp_lds_bytes<A, B, 1>(), L2 = chirp_lds_bytes<A, B, 2>(), L3 = chirp_lds_bytes<A, B, 3>(), CU = 160 * 1024;
    constexpr int w1 = (int)(CU / L1), w2 = (int)(CU / L2), w3 = (int)(CU / L3);
    return (w2 > w1 && w2 <= 3) ? ((w3 > w2 && w3 <= 3) ? 3 : 2) : 1;
}

}  // namespace anyfft
