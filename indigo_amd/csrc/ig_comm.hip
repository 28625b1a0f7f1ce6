// Collective of the coil-sharded SENSE path: the ONE all-reduce (sum) of the image per A^H A evaluation
// (reference hook: indigo/backends/backend.py:469-479 pdot/pnorm2 -> comm.allreduce, and the coil sum
// of VStack._eval_adjoint, indigo/operators.py:440-447, which a coil-sharded run finishes across GPUs).
//
// One process per GPU, one communicator per context, RCCL over xGMI.  RCCL is bound at run time (dlopen) so that
// libindigo_hip.so has no link-time dependency on it: single-GPU users never load it, and a process that already
// holds a copy (e.g. the one PyTorch ships) shares that copy instead of loading a second one.
//
// Two ways to issue the all-reduce:
//   ig_allreduce_sum_f32        in order with the context's stream (what a caller without further knowledge wants)
//   ig_allreduce_sum_f32_side   ordered after the work enqueued so far on the context's stream, but NOT waited for by it;
//                               ig_comm_join makes the context's stream wait.  The cropped transform produces the image
//                               slab by slab (ig_fft_exec_cropped_sum_slab), so the all-reduce of slab s runs over xGMI
//                               while slab s+1 is still being transformed.
// EVERY collective of a communicator is enqueued on the communicator's own stream (`side`), in call order; the two forms
// differ only in the event dependencies around it.  One communicator therefore only ever sees ONE stream: nothing here
// relies on how RCCL orders launches that reach one communicator from two streams.
#include "ig_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <mutex>

namespace {

// the handful of RCCL entry points used, resolved with dlsym; types and enums come from ROCm's rccl.h
typedef decltype(&ncclGetUniqueId) fn_get_unique_id;
typedef decltype(&ncclCommInitRank) fn_comm_init_rank;
typedef decltype(&ncclCommDestroy) fn_comm_destroy;
typedef decltype(&ncclAllReduce) fn_all_reduce;
typedef decltype(&ncclGetErrorString) fn_error_string;
typedef decltype(&ncclGetVersion) fn_get_version;
static_assert(sizeof(ncclUniqueId) == IG_COMM_ID_BYTES, "ig_comm_unique_id hands out an ncclUniqueId");
typedef ncclUniqueId rccl_unique_id;
typedef ncclComm_t rccl_comm_t;
constexpr ncclDataType_t RCCL_FLOAT32 = ncclFloat32, RCCL_FLOAT64 = ncclFloat64;
constexpr ncclRedOp_t RCCL_SUM = ncclSum, RCCL_MAX = ncclMax;

struct Rccl {
    void* handle = nullptr;
    std::string path;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    fn_get_version get_version = nullptr;
    int version = 0;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl(ig_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return IG_OK;
    const char* env = getenv("INDIGO_HIP_RCCL_LIB");
    void* h = nullptr;
    std::string tried;
    if (env && *env) {
        h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        tried = env;
        if (h) g_rccl.path = env;
    } else {
        // a copy this process already holds wins (RTLD_NOLOAD), then the loader's search path, then ROCm's own
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (int pass = 0; pass < 2 && !h; ++pass)
            for (const char* n : names) {
                if (pass == 0 && n[0] == '/') continue;
                h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (h) { g_rccl.path = std::string(n) + (pass == 0 ? " (already loaded)" : ""); break; }
                if (pass == 1) { tried += n; tried += ' '; }
            }
    }
    if (!h) return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_comm: cannot load RCCL (tried %s): %s", tried.c_str(), dlerror());
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    g_rccl.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    g_rccl.get_version = (fn_get_version)dlsym(h, "ncclGetVersion");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce || !g_rccl.error_string) {
        dlclose(h);
        return ig_fail(ctx, IG_ERR_UNSUPPORTED, "ig_comm: %s lacks the ncclCommInitRank / ncclAllReduce entry points", g_rccl.path.c_str());
    }
    if (g_rccl.get_version) (void)g_rccl.get_version(&g_rccl.version);
    g_rccl.handle = h;
    return IG_OK;
}

}  // namespace

struct ig_comm {
    ig_ctx*     ctx = nullptr;
    rccl_comm_t comm = nullptr;
    int         rank = 0, nranks = 1;
    hipStream_t side = nullptr;          // the communicator's own stream (overlapped all-reduces)
    hipEvent_t  ev_work = nullptr;       // "the context's stream got this far"
    hipEvent_t  ev_side = nullptr;       // "the side stream got this far"
    bool        side_busy = false;
    double*     d_scalar = nullptr;      // one double for the host-scalar reductions
};

#define IG_RCCL(ctx, call)                                                             \
    do {                                                                               \
        ncclResult_t r_ = (call);                                                      \
        if (r_ != ncclSuccess)                                                                 \
            return ig_fail((ctx), IG_ERR_HIP, "%s failed: %s (%s:%d)", #call,          \
                           g_rccl.error_string(r_), __FILE__, __LINE__);               \
    } while (0)

extern "C" {

// Everything of the bring-up that can fail on ONE rank alone (no RCCL to load, entry points missing): to be called -- and
// its result agreed on by all ranks (indigo_amd/dist.py: exchange_id) -- BEFORE anyone enters ncclCommInitRank, which
// only returns when every rank has entered it.
int ig_comm_preflight(void) { return load_rccl(nullptr); }

int ig_comm_unique_id(void* id_out) {
    if (!id_out) return ig_fail(nullptr, IG_ERR_ARG, "ig_comm_unique_id: id_out is NULL");
    if (int rc = load_rccl(nullptr)) return rc;
    rccl_unique_id id;
    std::memset(&id, 0, sizeof(id));
    IG_RCCL(nullptr, g_rccl.get_unique_id(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return IG_OK;
}

int ig_comm_init_rank(ig_ctx* ctx, int nranks, int rank, const void* id, ig_comm** out) {
    IG_REQUIRE(ctx, ctx && id && out, "ig_comm_init_rank: bad arguments");
    IG_REQUIRE(ctx, nranks >= 1 && rank >= 0 && rank < nranks, "ig_comm_init_rank: rank %d of %d", rank, nranks);
    *out = nullptr;
    if (int rc = load_rccl(ctx)) return rc;
    if (int rc = ig_set_device(ctx)) return rc;
    ig_comm* c = new ig_comm();
    c->ctx = ctx; c->rank = rank; c->nranks = nranks;
    rccl_unique_id uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = g_rccl.comm_init_rank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return ig_fail(ctx, IG_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.error_string(r));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_work, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_scalar, sizeof(double));
    if (e != hipSuccess) {
        int rc = ig_fail(ctx, IG_ERR_HIP, "ig_comm_init_rank: stream/event/scratch creation failed: %s", hipGetErrorString(e));
        ig_comm_destroy(c);
        return rc;
    }
    *out = c;
    return IG_OK;
}

int ig_comm_info(ig_comm* c, int* rank, int* nranks, char* lib, size_t len) {
    if (!c) return ig_fail(nullptr, IG_ERR_ARG, "ig_comm_info: comm is NULL");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (lib && len) snprintf(lib, len, "%s version %d", g_rccl.path.c_str(), g_rccl.version);
    return IG_OK;
}

int ig_allreduce_sum_f32(ig_comm* c, void* buf, int64_t nfloats) {
    if (!c) return ig_fail(nullptr, IG_ERR_ARG, "ig_allreduce_sum_f32: comm is NULL");
    ig_ctx* ctx = c->ctx;
    IG_REQUIRE(ctx, nfloats >= 0 && (nfloats == 0 || buf), "ig_allreduce_sum_f32: bad buffer");
    if (nfloats == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    ig_prof_scope prof(ctx, "allreduce", (double)nfloats * 4.0);
    // context's stream -> side stream -> collective -> context's stream: in order with both
    IG_HIP(ctx, hipEventRecord(c->ev_work, ctx->stream));
    IG_HIP(ctx, hipStreamWaitEvent(c->side, c->ev_work, 0));
    IG_RCCL(ctx, g_rccl.all_reduce(buf, buf, (size_t)nfloats, RCCL_FLOAT32, RCCL_SUM, c->comm, c->side));
    IG_HIP(ctx, hipEventRecord(c->ev_side, c->side));
    IG_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_side, 0));
    c->side_busy = false;                  // (the context's stream now waits for everything the side stream holds)
    return IG_OK;
}

int ig_allreduce_sum_f32_side(ig_comm* c, void* buf, int64_t nfloats) {
    if (!c) return ig_fail(nullptr, IG_ERR_ARG, "ig_allreduce_sum_f32_side: comm is NULL");
    ig_ctx* ctx = c->ctx;
    IG_REQUIRE(ctx, nfloats >= 0 && (nfloats == 0 || buf), "ig_allreduce_sum_f32_side: bad buffer");
    if (nfloats == 0) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    // everything enqueued so far on the context's stream (the kernels that produced buf) comes first
    IG_HIP(ctx, hipEventRecord(c->ev_work, ctx->stream));
    IG_HIP(ctx, hipStreamWaitEvent(c->side, c->ev_work, 0));
    IG_RCCL(ctx, g_rccl.all_reduce(buf, buf, (size_t)nfloats, RCCL_FLOAT32, RCCL_SUM, c->comm, c->side));
    c->side_busy = true;
    return IG_OK;
}

int ig_comm_join(ig_comm* c) {
    if (!c) return ig_fail(nullptr, IG_ERR_ARG, "ig_comm_join: comm is NULL");
    ig_ctx* ctx = c->ctx;
    if (!c->side_busy) return IG_OK;
    if (int rc = ig_set_device(ctx)) return rc;
    IG_HIP(ctx, hipEventRecord(c->ev_side, c->side));
    IG_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_side, 0));
    c->side_busy = false;
    return IG_OK;
}

// max / sum of ONE host double over the ranks (timing and convergence scalars): synchronous
static int host_scalar(ig_comm* c, double* v, ncclRedOp_t op, const char* who) {
    if (!c || !v) return ig_fail(nullptr, IG_ERR_ARG, "%s: bad arguments", who);
    ig_ctx* ctx = c->ctx;
    if (int rc = ig_set_device(ctx)) return rc;
    // on the communicator's stream like every collective, after whatever the context's stream holds; synchronous
    IG_HIP(ctx, hipEventRecord(c->ev_work, ctx->stream));
    IG_HIP(ctx, hipStreamWaitEvent(c->side, c->ev_work, 0));
    IG_HIP(ctx, hipMemcpyAsync(c->d_scalar, v, sizeof(double), hipMemcpyHostToDevice, c->side));
    IG_RCCL(ctx, g_rccl.all_reduce(c->d_scalar, c->d_scalar, 1, RCCL_FLOAT64, op, c->comm, c->side));
    IG_HIP(ctx, hipMemcpyAsync(v, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, c->side));
    IG_HIP(ctx, hipStreamSynchronize(c->side));
    IG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    c->side_busy = false;
    return IG_OK;
}

int ig_allreduce_max_f64_host(ig_comm* c, double* inout) { return host_scalar(c, inout, RCCL_MAX, "ig_allreduce_max_f64_host"); }
int ig_allreduce_sum_f64_host(ig_comm* c, double* inout) { return host_scalar(c, inout, RCCL_SUM, "ig_allreduce_sum_f64_host"); }

int ig_comm_barrier(ig_comm* c) {
    double one = 1.0;
    if (int rc = ig_comm_join(c)) return rc;
    return host_scalar(c, &one, RCCL_SUM, "ig_comm_barrier");
}

int ig_comm_destroy(ig_comm* c) {
    if (!c) return IG_OK;
    (void)hipSetDevice(c->ctx->device);
    if (c->side) (void)hipStreamSynchronize(c->side);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->comm);
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    if (c->ev_work) (void)hipEventDestroy(c->ev_work);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return IG_OK;
}

}  // extern "C"
