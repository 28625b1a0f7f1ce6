// Packed complex arithmetic for the register-resident transforms (shared by ig_fft.hip and ig_fft_ab.h).
#pragma once
#include <hip/hip_runtime.h>

// A complex number is ONE 64-bit register pair (native 2-vector): complex add/sub is a single v_pk_add_f32, a complex
// multiply is v_pk_mul_f32 + v_pk_fma_f32 with operand swizzles, multiplication by -i is a swizzle with a sign
// modifier.  Written on float2 structs the same butterflies compile to scalar v_add/v_mul/v_fmac, twice the VALU
// instructions -- and these kernels are bound by instruction issue as much as by HBM (SIMDs ~85 % busy).
typedef float v2f __attribute__((ext_vector_type(2)));
struct cx { v2f v; };
__device__ __forceinline__ cx mk(float re, float im) { cx r; r.v = v2f{re, im}; return r; }
__device__ __forceinline__ cx from2(float2 a) { return mk(a.x, a.y); }
__device__ __forceinline__ float2 to2(cx a) { return make_float2(a.v.x, a.v.y); }
__device__ __forceinline__ cx operator+(cx a, cx b) { cx r; r.v = a.v + b.v; return r; }
__device__ __forceinline__ cx operator-(cx a, cx b) { cx r; r.v = a.v - b.v; return r; }
__device__ __forceinline__ cx cneg(cx a) { cx r; r.v = -a.v; return r; }
__device__ __forceinline__ cx cconj(cx a) { cx r; r.v = v2f{a.v.x, -a.v.y}; return r; }
// Multiplications by -+i and the twiddle products with a RUN-TIME root are single packed instructions with operand swizzles
// (op_sel) and sign modifiers -- written as assembly, because from vector shuffles the compiler builds the rotated operand
// (-w.y, w.x) with a v_xor and a v_mov first: two extra vector instructions per twiddle and per -i, 130 of the 800 of a 512-point
// pass.  (Pure register operations: the compiler schedules them like any other instruction.)
__device__ __forceinline__ cx cmul_mi(cx a) {          // a * (-i) = (a.y, -a.x)
    cx r; asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r.v) : "v"(a.v)); return r;
}
__device__ __forceinline__ cx cmul_pi(cx a) {          // a * (+i) = (-a.y, a.x)
    cx r; asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r.v) : "v"(a.v)); return r;
}
__device__ __forceinline__ cx cmul_mi_c(cx a) { cx r; r.v = v2f{a.v.y, -a.v.x}; return r; }          // (compiler-lowered forms: k_fft3d_b)
__device__ __forceinline__ cx cmul_pi_c(cx a) { cx r; r.v = v2f{-a.v.y, a.v.x}; return r; }
__device__ __forceinline__ cx padd_mi(cx a, cx b) {    // a + (-i) b = (a.x + b.y, a.y - b.x)
    cx r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r.v) : "v"(a.v), "v"(b.v)); return r;
}
__device__ __forceinline__ cx padd_pi(cx a, cx b) {    // a + (+i) b = (a.x - b.y, a.y + b.x)
    cx r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r.v) : "v"(a.v), "v"(b.v)); return r;
}
// a * w = a.xx * (w.x, w.y) + a.yy * (-w.y, w.x)   (w a literal: the compiler folds the rotation into the constants)
__device__ __forceinline__ cx cxmul(cx a, cx w) {
    cx r;
    r.v = __builtin_shufflevector(a.v, a.v, 0, 0) * w.v + __builtin_shufflevector(a.v, a.v, 1, 1) * v2f{-w.v.y, w.v.x};
    return r;
}
// the same with w in registers (twiddles out of LDS, weights out of memory): two instructions, no rotated copy of w
__device__ __forceinline__ cx cxmul_r(cx a, cx w) {
    cx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t.v) : "v"(a.v), "v"(w.v));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r.v) : "v"(a.v), "v"(w.v), "v"(t.v));
    return r;
}
// conj(w) * a = a.xx * (w.x, -w.y) + a.yy * (w.y, w.x), w in registers
__device__ __forceinline__ cx cxmulc(cx w, cx a) {
    cx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t.v) : "v"(a.v), "v"(w.v));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r.v) : "v"(a.v), "v"(w.v), "v"(t.v));
    return r;
}
// conj(a * w) = (a.x w.x - a.y w.y, -(a.x w.y + a.y w.x)), w in registers (the chirp-z convolution's product followed by the
// conjugation that turns the next forward transform into an inverse one: two instructions, like any product)
__device__ __forceinline__ cx cxmul_cc(cx a, cx w) {
    cx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t.v) : "v"(a.v), "v"(w.v));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r.v) : "v"(a.v), "v"(w.v), "v"(t.v));
    return r;
}
// acc + c * t for a REAL compile-time constant c (the odd-prime butterflies): one packed multiply-add
__device__ __forceinline__ cx cfma_r(float c, cx t, cx acc) { cx r; r.v = t.v * v2f{c, c} + acc.v; return r; }
