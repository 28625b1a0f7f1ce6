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
// Plain passes only (the fftn / ifftn contract, backend.py:497-509); the zero-pad-aware SENSE passes need n in {256, 512}.
#pragma once
#include <type_traits>

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

constexpr int pick_radix(int n) { return n % 4 == 0 ? 4 : n % 2 == 0 ? 2 : n % 3 == 0 ? 3 : n % 5 == 0 ? 5 : n % 7 == 0 ? 7 : n; }

// forward DFT of N register values, in place, natural order
template <int N, bool BASE = (pick_radix(N) == N)>
struct RegDFT;

template <> struct RegDFT<1, true> { __device__ static __forceinline__ void run(float2 (&)[1]) {} };
template <> struct RegDFT<2, true> {
    __device__ static __forceinline__ void run(float2 (&x)[2]) {
        const float2 a = x[0], b = x[1];
        x[0] = make_float2(a.x + b.x, a.y + b.y); x[1] = make_float2(a.x - b.x, a.y - b.y);
    }
};
template <> struct RegDFT<4, true> {
    __device__ static __forceinline__ void run(float2 (&x)[4]) {
        const float2 t0 = make_float2(x[0].x + x[2].x, x[0].y + x[2].y), t1 = make_float2(x[0].x - x[2].x, x[0].y - x[2].y);
        const float2 t2 = make_float2(x[1].x + x[3].x, x[1].y + x[3].y);
        const float2 d = make_float2(x[1].x - x[3].x, x[1].y - x[3].y), t3 = make_float2(d.y, -d.x);      // -i (x1 - x3)
        x[0] = make_float2(t0.x + t2.x, t0.y + t2.y); x[2] = make_float2(t0.x - t2.x, t0.y - t2.y);
        x[1] = make_float2(t1.x + t3.x, t1.y + t3.y); x[3] = make_float2(t1.x - t3.x, t1.y - t3.y);
    }
};
// odd primes (3, 5, 7): X_k, X_{P-k} = a_k -/+ i b_k with a_k = x0 + sum_q cos(2 pi q k / P) (x_q + x_{P-q}),
// b_k = sum_q sin(2 pi q k / P) (x_q - x_{P-q})
template <int P>
struct RegDFT<P, true> {
    static_assert(P == 3 || P == 5 || P == 7, "RegDFT: lengths with factors 2, 3, 5, 7 only");
    __device__ static __forceinline__ void run(float2 (&x)[P]) {
        constexpr int H = (P - 1) / 2;
        float2 t[H], d[H];
        static_for<0, H>([&](auto q_) {
            constexpr int q = decltype(q_)::value + 1;
            t[q - 1] = make_float2(x[q].x + x[P - q].x, x[q].y + x[P - q].y);
            d[q - 1] = make_float2(x[q].x - x[P - q].x, x[q].y - x[P - q].y);
        });
        const float2 x0 = x[0];
        float2 s0 = x0;
        static_for<0, H>([&](auto q_) { s0.x += t[decltype(q_)::value].x; s0.y += t[decltype(q_)::value].y; });
        x[0] = s0;
        static_for<0, H>([&](auto k_) {
            constexpr int k = decltype(k_)::value + 1;
            float2 a = x0, b = make_float2(0.f, 0.f);
            static_for<0, H>([&](auto q_) {
                constexpr int q = decltype(q_)::value + 1;
                constexpr float c = cos2pi(q * k, P), s = sin2pi(q * k, P);
                a.x = fmaf(c, t[q - 1].x, a.x); a.y = fmaf(c, t[q - 1].y, a.y);
                b.x = fmaf(s, d[q - 1].x, b.x); b.y = fmaf(s, d[q - 1].y, b.y);
            });
            // -i b = (b.y, -b.x)
            x[k] = make_float2(a.x + b.y, a.y - b.x);
            x[P - k] = make_float2(a.x - b.y, a.y + b.x);
        });
    }
};
// composite N = P Q:  n = Q n1 + n2,  k = k1 + P k2
template <int N>
struct RegDFT<N, false> {
    __device__ static __forceinline__ void run(float2 (&x)[N]) {
        constexpr int P = pick_radix(N), Q = N / P;
        float2 y[Q][P];
        static_for<0, Q>([&](auto n2_) {
            constexpr int n2 = decltype(n2_)::value;
            float2 a[P];
            static_for<0, P>([&](auto n1_) { constexpr int n1 = decltype(n1_)::value; a[n1] = x[Q * n1 + n2]; });
            RegDFT<P>::run(a);
            static_for<0, P>([&](auto k1_) { constexpr int k1 = decltype(k1_)::value; y[n2][k1] = rot<n2 * k1, N>(a[k1]); });
        });
        static_for<0, P>([&](auto k1_) {
            constexpr int k1 = decltype(k1_)::value;
            float2 b[Q];
            static_for<0, Q>([&](auto n2_) { constexpr int n2 = decltype(n2_)::value; b[n2] = y[n2][k1]; });
            RegDFT<Q>::run(b);
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
    for (int k = tid; k < N; k += AB_W * B) tws[k] = tw[k];
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

}  // namespace anyfft
