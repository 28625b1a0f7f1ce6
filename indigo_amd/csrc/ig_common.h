// Internal definitions shared by the HIP translation units of libindigo_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>
#include <vector>

#include "indigo_hip.h"

// one timed kernel launch (profile mode): events recorded on the context's stream
struct ig_prof_rec {
    const char* name;
    double      bytes;     // caller-supplied algorithmic bytes of this launch (0 if not given)
    hipEvent_t  e0, e1;
};

struct ig_ctx {
    int          device      = 0;
    hipStream_t  stream      = nullptr;
    bool         own_stream  = false;
    int          num_cu      = 256;
    std::string  err;
    // small device/pinned scratch used by the reductions (dot, nrm2)
    double*      d_partials  = nullptr;   // (IG_MAX_RED_BLOCKS + 1) * 2 doubles
    double*      h_result    = nullptr;   // pinned, 2 doubles
    double*      d_scalars   = nullptr;   // IG_NUM_SCALARS device-resident doubles for solver scalars (ig_scalars)
    void*        d_xpack     = nullptr;   // SpMM repacked-panel scratch (grown on demand)
    size_t       xpack_bytes = 0;
    int32_t*     d_worklist  = nullptr;   // SpMM deferred-row lists + counters (allocated on first use)
    size_t       worklist_bytes = 0, partials_bytes = 0, scalars_bytes = 0;   // what was really allocated (ig_library_bytes)
    // plan options (ig_set_option): which transform kernels a NEW plan may use.  0 = all; 1 = no register-resident A x B passes
    // (their lengths fall back to the multi-stage LDS kernel); 2 = only the one-stage-per-launch generic kernel
    int          opt_fft_kernels = 0;
    bool         bricks_attr = false;     // the brick-binned gridding kernel's dynamic-LDS opt-in was applied on this device
    bool         fft3d_attr = false;      // the two-launch 256^3 transform's dynamic-LDS opt-in was applied on this device
    bool         fft_w32_attr = false;    // the 32-column FFT kernels' dynamic-LDS opt-in was applied on this device
    // profile mode (ig_prof_enable): every kernel launch is bracketed by two events
    bool                     prof_on = false;
    bool                     capturing = false;    // ig_graph_begin ... ig_graph_end / _abort: launches are recorded, not executed
    std::vector<ig_prof_rec> prof;
    std::vector<hipEvent_t>  prof_pool;    // recycled events
};

// RAII bracket around a kernel launch; a no-op unless profile mode is on.
struct ig_prof_scope {
    ig_ctx* ctx; size_t idx; bool on;
    ig_prof_scope(ig_ctx* c, const char* name, double bytes = 0.0) : ctx(c), idx(0), on(c && c->prof_on) {
        if (!on) return;
        ig_prof_rec r; r.name = name; r.bytes = bytes;
        auto get = [&]() { hipEvent_t e = nullptr;
            if (!ctx->prof_pool.empty()) { e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e; };
        r.e0 = get(); r.e1 = get();
        (void)hipEventRecord(r.e0, ctx->stream);
        idx = ctx->prof.size();
        ctx->prof.push_back(r);
    }
    ~ig_prof_scope() { if (on) (void)hipEventRecord(ctx->prof[idx].e1, ctx->stream); }
};

struct ig_graph {
    ig_ctx*        ctx  = nullptr;
    hipGraph_t     graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct ig_event {
    ig_ctx*    ctx = nullptr;
    hipEvent_t ev  = nullptr;
};

constexpr int IG_MAX_RED_BLOCKS = 2048;
constexpr int IG_NUM_SCALARS = 1024;

// thread-local error for calls without a context
std::string& ig_tls_error();

int ig_fail(ig_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

#define IG_HIP(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return ig_fail((ctx), IG_ERR_HIP, "%s failed: %s (%s:%d)", #call,          \
                           hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)

#define IG_REQUIRE(ctx, cond, ...)                                                     \
    do {                                                                               \
        if (!(cond)) return ig_fail((ctx), IG_ERR_ARG, __VA_ARGS__);                   \
    } while (0)

// checks the launch itself (configuration errors); execution errors surface at the next sync
#define IG_LAUNCH_CHECK(ctx, what)                                                     \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess)                                                          \
            return ig_fail((ctx), IG_ERR_HIP, "launch of %s failed: %s", (what),       \
                           hipGetErrorString(e_));                                     \
    } while (0)

static inline int ig_set_device(ig_ctx* ctx) {
    IG_HIP(ctx, hipSetDevice(ctx->device));
    return IG_OK;
}

// ---- tiny complex helpers (device) -----------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// conj(a) * b
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.x, b.y, -a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// acc += a*b
__device__ __forceinline__ void cfma(float2& acc, float2 a, float2 b) {
    acc.x = fmaf(a.x, b.x, acc.x); acc.x = fmaf(-a.y, b.y, acc.x);
    acc.y = fmaf(a.x, b.y, acc.y); acc.y = fmaf(a.y, b.x, acc.y);
}

// Raw buffer accesses: 128-bit descriptor (wave-uniform base, 2 GB window) + per-lane 32-bit byte offset + uniform
// byte offset.  A lane offset >= the window (IG_OOB) is out of range: loads return 0, stores are dropped.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
#define IG_OOB 0x80000000u
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)IG_OOB, 0x00020000);
}
template <bool NT>
__device__ __forceinline__ float2 buf_ld(rsrc_t r, unsigned voff, unsigned soff) {
    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, NT ? 2 : 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
template <bool NT>
__device__ __forceinline__ void buf_st(rsrc_t r, unsigned voff, unsigned soff, float2 a) {
    v2u_t v; v.x = __float_as_uint(a.x); v.y = __float_as_uint(a.y);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, NT ? 2 : 0);
}

// An 8-byte buffer store that is SKIPPED -- by a scalar branch inside one opaque block, so the compiler sees no control flow and
// allocates registers as before -- when the wave-uniform flag `wanted` is 0: no lane of the wave keeps the element
// (the lanes that do not are out of range anyway), and a store whose lanes are all out of range would still take its issue slot
// and its pass through the address unit.  A store reads its data registers at issue; nothing here reads the target back.
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i_t make_rsrc_words(const void* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    return v4i_t{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)IG_OOB, 0x00020000};
}
template <bool NT>
__device__ __forceinline__ void buf_st_gated(v4i_t r, unsigned voff, unsigned soff /* wave-uniform byte offset */, float2 a,
                                             unsigned wanted /* wave-uniform: 0 = skip */) {
    v2u_t v; v.x = __float_as_uint(a.x); v.y = __float_as_uint(a.y);
    if (NT)
        asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lig_st_skip%=\n\tbuffer_store_dwordx2 %0, %1, %2, %4 offen nt\n.Lig_st_skip%=:"
                     :: "v"(v), "v"(voff), "s"(r), "s"(wanted), "s"(soff) : "scc", "memory");
    else
        asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lig_st_skip%=\n\tbuffer_store_dwordx2 %0, %1, %2, %4 offen\n.Lig_st_skip%=:"
                     :: "v"(v), "v"(voff), "s"(r), "s"(wanted), "s"(soff) : "scc", "memory");
}

typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int buf_ld_i32(rsrc_t r, unsigned voff) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
}
// 16-byte store.  No scalar-offset operand on purpose: on gfx950 a 16-byte buffer store with a scalar-REGISTER offset,
// followed directly by a VALU write of its data registers, stored stale data in the upper lanes of every 16-lane row
// (measured: the two-launch FFT's imaginary parts); the compiler pads that write-after-read hazard only for
// immediate offsets.  Fold constant displacements into `voff` (they become the instruction's immediate offset).
template <bool NT>
__device__ __forceinline__ void buf_st_f4(rsrc_t r, unsigned voff, float4 a) {
    v4u_t v; v.x = __float_as_uint(a.x); v.y = __float_as_uint(a.y); v.z = __float_as_uint(a.z); v.w = __float_as_uint(a.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, NT ? 2 : 0);
}
typedef unsigned int v3u_t __attribute__((ext_vector_type(3)));
struct u3 { unsigned x, y, z; };
__device__ __forceinline__ u3 buf_ld_u3(rsrc_t r, unsigned voff) {          // 12 bytes (4-byte aligned)
    const v3u_t v = __builtin_amdgcn_raw_buffer_load_b96(r, voff, 0, 0);
    return u3{v.x, v.y, v.z};
}
__device__ __forceinline__ float4 buf_ld_f4(rsrc_t r, unsigned voff) {
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

