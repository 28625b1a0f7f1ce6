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
#include <atomic>
#include <chrono>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

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

// ---- the DIRECT communicator (round 6): no RCCL -----------------------------------------------------------------------------------
// One node, one process per GPU.  Every rank owns a WINDOW of device memory whose IPC handle (hipIpcGetMemHandle) the others map
// (hipIpcOpenMemHandle): the all-reduce is then a reduce-scatter + all-gather over peer-mapped memory in which every rank moves its
// 1 / N slab to and from ALL N - 1 peers at once -- over all of a GPU's xGMI links, where a ring is bound by one (0.43 ms
// against 3.0 ms for the 262 MB image of config 5 on 8 GPUs).  A small POSIX shared-memory segment carries the handles, a barrier
// with a timeout and the host scalars.  Synchronisation is on the HOST (stream sync + shared-memory barrier): no kernel ever spins
// on a flag another process has to set, so a rank that dies costs its peers a timeout, not a hung GPU.
constexpr int IG_DIRECT_MAX_RANKS = 16;
struct DirectShm {
    std::atomic<uint32_t> magic;              // set last by rank 0
    uint32_t nranks;
    std::atomic<uint32_t> arrived, generation, error;
    double slots[IG_DIRECT_MAX_RANKS];
    hipIpcMemHandle_t win[IG_DIRECT_MAX_RANKS];
};
constexpr uint32_t IG_DIRECT_MAGIC = 0x1d160c06u;

struct ig_comm {
    ig_ctx*     ctx = nullptr;
    rccl_comm_t comm = nullptr;
    // direct form
    bool        direct = false;
    DirectShm*  shm = nullptr;
    std::string shm_name;
    void*       win = nullptr;               // this rank's window
    void*       peer[IG_DIRECT_MAX_RANKS] = {nullptr};       // every rank's window as mapped here (peer[rank] == win)
    float**     d_peers = nullptr;           // the same table on the device
    size_t      win_bytes = 0;
    double      timeout_s = 60.0;
    int         rank = 0, nranks = 1;
    hipStream_t side = nullptr;          // the communicator's own stream (overlapped all-reduces)
    hipEvent_t  ev_work = nullptr;       // "the context's stream got this far"
    hipEvent_t  ev_side = nullptr;       // "the side stream got this far"
    bool        side_busy = false;
    double*     d_scalar = nullptr;      // one double for the host-scalar reductions
};

namespace {

// peer-visible accesses are system-scope (sc0 sc1 on gfx950): the windows are ordinary device memory, and what a peer wrote must
// not be served from -- nor what a peer will read be left in -- an XCD's L2
__device__ __forceinline__ float ld_sys(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ void __launch_bounds__(256) k_direct_put(float* __restrict__ win, const float* __restrict__ buf, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) st_sys(win + i, buf[i]);
}
__global__ void __launch_bounds__(256) k_direct_get(float* __restrict__ buf, const float* __restrict__ win, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) buf[i] = ld_sys(win + i);
}
// this rank's slab [lo, hi) of every window: the sum in rank order (the same bits whatever rank computes it), written back to all
__global__ void __launch_bounds__(256) k_direct_reduce(float* const* __restrict__ peers, int nranks, int64_t lo, int64_t hi) {
    for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int p = 0; p < nranks; ++p) s += ld_sys(peers[p] + i);
        for (int p = 0; p < nranks; ++p) st_sys(peers[p] + i, s);
    }
}

int direct_barrier(ig_comm* c, const char* who) {
    DirectShm* sh = c->shm;
    if (sh->error.load()) return ig_fail(c->ctx, IG_ERR_HIP, "%s: another rank of the direct communicator failed", who);
    const uint32_t gen = sh->generation.load();
    if (sh->arrived.fetch_add(1) + 1 == (uint32_t)c->nranks) {
        sh->arrived.store(0);
        sh->generation.fetch_add(1);
        return IG_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (sh->generation.load() == gen) {
        if (sh->error.load()) return ig_fail(c->ctx, IG_ERR_HIP, "%s: another rank of the direct communicator failed", who);
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
            sh->error.store(1);
            return ig_fail(c->ctx, IG_ERR_HIP, "%s: rank %d waited %.0f s for the other ranks (one of them died or never arrived)", who, c->rank, c->timeout_s);
        }
    }
    return IG_OK;
}

int direct_host_scalar(ig_comm* c, double* v, bool is_max, const char* who) {
    DirectShm* sh = c->shm;
    sh->slots[c->rank] = *v;
    if (int rc = direct_barrier(c, who)) return rc;
    double r = sh->slots[0];
    for (int p = 1; p < c->nranks; ++p) r = is_max ? (sh->slots[p] > r ? sh->slots[p] : r) : r + sh->slots[p];
    if (int rc = direct_barrier(c, who)) return rc;          // (nobody overwrites a slot before everyone has read it)
    *v = r;
    return IG_OK;
}

// in place: buf <- sum over ranks.  Host-synchronous (see the struct's comment); window-sized pieces.
int direct_allreduce_impl(ig_comm* c, float* buf, int64_t n);
int direct_allreduce(ig_comm* c, float* buf, int64_t n) {
    const int rc = direct_allreduce_impl(c, buf, n);
    if (rc != IG_OK) c->shm->error.store(1);              // (the peers fail at their next barrier instead of waiting out the timeout)
    return rc;
}
int direct_allreduce_impl(ig_comm* c, float* buf, int64_t n) {
    ig_ctx* ctx = c->ctx;
    const int64_t cap = (int64_t)(c->win_bytes / 4);
    const unsigned grid = (unsigned)ctx->num_cu * 8;
    IG_HIP(ctx, hipEventRecord(c->ev_work, ctx->stream));
    IG_HIP(ctx, hipStreamWaitEvent(c->side, c->ev_work, 0));
    for (int64_t off = 0; off < n; off += cap) {
        const int64_t m = n - off < cap ? n - off : cap;
        hipLaunchKernelGGL(k_direct_put, dim3(grid), dim3(256), 0, c->side, (float*)c->win, buf + off, m);
        IG_HIP(ctx, hipStreamSynchronize(c->side));
        if (int rc = direct_barrier(c, "ig_allreduce_sum_f32 (direct)")) return rc;          // every window holds its rank's piece
        const int64_t per = ((m + c->nranks - 1) / c->nranks + 63) & ~(int64_t)63;
        const int64_t lo = per * c->rank < m ? per * c->rank : m, hi = lo + per < m ? lo + per : m;
        if (hi > lo) hipLaunchKernelGGL(k_direct_reduce, dim3(grid), dim3(256), 0, c->side, (float* const*)c->d_peers, c->nranks, lo, hi);
        IG_HIP(ctx, hipStreamSynchronize(c->side));
        if (int rc = direct_barrier(c, "ig_allreduce_sum_f32 (direct)")) return rc;          // every slab of every window is reduced
        hipLaunchKernelGGL(k_direct_get, dim3(grid), dim3(256), 0, c->side, buf + off, (const float*)c->win, m);
        if (off + cap < n) {            // the window is reused: nobody refills it while a peer still reads or reduces the piece before
            IG_HIP(ctx, hipStreamSynchronize(c->side));
            if (int rc = direct_barrier(c, "ig_allreduce_sum_f32 (direct)")) return rc;
        }
    }
    IG_LAUNCH_CHECK(ctx, "k_direct_*");
    IG_HIP(ctx, hipEventRecord(c->ev_side, c->side));
    IG_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_side, 0));
    c->side_busy = false;
    return IG_OK;
}

}  // namespace

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

// The direct communicator (no RCCL): `name` = a POSIX shared-memory name all ranks of ONE node agree on ("/..."), unique to this
// communicator; window_bytes = the device window every rank exposes to the others (messages larger than it go in pieces).
// Collective: returns when every rank has mapped every window, or fails on all ranks within timeout_s.
int ig_comm_init_direct(ig_ctx* ctx, int nranks, int rank, const char* name, size_t window_bytes, double timeout_s, ig_comm** out) {
    IG_REQUIRE(ctx, ctx && name && out && name[0] == '/', "ig_comm_init_direct: bad arguments (the name starts with '/')");
    IG_REQUIRE(ctx, nranks >= 1 && nranks <= IG_DIRECT_MAX_RANKS && rank >= 0 && rank < nranks, "ig_comm_init_direct: rank %d of %d (at most %d)", rank, nranks, IG_DIRECT_MAX_RANKS);
    IG_REQUIRE(ctx, window_bytes >= 4096 && window_bytes % 4096 == 0, "ig_comm_init_direct: the window is a multiple of 4096 bytes");
    *out = nullptr;
    if (int rc = ig_set_device(ctx)) return rc;
    ig_comm* c = new ig_comm();
    c->ctx = ctx; c->rank = rank; c->nranks = nranks; c->direct = true; c->shm_name = name; c->win_bytes = window_bytes;
    c->timeout_s = timeout_s > 0 ? timeout_s : 60.0;
    auto bail = [&](int rc) { ig_comm_destroy(c); return rc; };
    // the segment: rank 0 creates it fresh (a left-over of an earlier run under this name is removed), the others wait for its magic
    int fd = -1;
    const auto t0 = std::chrono::steady_clock::now();
    if (rank == 0) {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(DirectShm)) != 0) { if (fd >= 0) close(fd); return bail(ig_fail(ctx, IG_ERR_HIP, "ig_comm_init_direct: cannot create shared memory %s", name)); }
    } else {
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s)
                return bail(ig_fail(ctx, IG_ERR_HIP, "ig_comm_init_direct: rank %d never saw rank 0's shared memory %s", rank, name));
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    void* m = MAP_FAILED;
    for (;;) {          // (a reader may open the segment before rank 0 has sized it)
        m = mmap(nullptr, sizeof(DirectShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {
            off_t len = lseek(fd, 0, SEEK_END);
            if (len >= (off_t)sizeof(DirectShm)) break;
            munmap(m, sizeof(DirectShm)); m = MAP_FAILED;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    close(fd);
    if (m == MAP_FAILED) return bail(ig_fail(ctx, IG_ERR_HIP, "ig_comm_init_direct: cannot map shared memory %s", name));
    c->shm = (DirectShm*)m;
    if (rank == 0) {
        c->shm->nranks = (uint32_t)nranks;
        c->shm->arrived.store(0); c->shm->generation.store(0); c->shm->error.store(0);
        c->shm->magic.store(IG_DIRECT_MAGIC);
    } else {
        while (c->shm->magic.load() != IG_DIRECT_MAGIC) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s)
                return bail(ig_fail(ctx, IG_ERR_HIP, "ig_comm_init_direct: rank 0 never initialised %s", name));
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if ((int)c->shm->nranks != nranks) return bail(ig_fail(ctx, IG_ERR_ARG, "ig_comm_init_direct: %s belongs to a communicator of %u ranks", name, c->shm->nranks));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_work, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&c->win, window_bytes);
    if (e == hipSuccess) e = hipMemset(c->win, 0, window_bytes);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&c->shm->win[rank], c->win);
    if (e != hipSuccess) { c->shm->error.store(1); return bail(ig_fail(ctx, IG_ERR_HIP, "ig_comm_init_direct: window / IPC handle: %s", hipGetErrorString(e))); }
    if (int rc = direct_barrier(c, "ig_comm_init_direct")) return bail(rc);                  // every handle is published
    c->peer[rank] = c->win;
    for (int p = 0; p < nranks && e == hipSuccess; ++p)
        if (p != rank) e = hipIpcOpenMemHandle(&c->peer[p], c->shm->win[p], hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_peers, sizeof(float*) * IG_DIRECT_MAX_RANKS);
    if (e == hipSuccess) e = hipMemcpy(c->d_peers, c->peer, sizeof(float*) * IG_DIRECT_MAX_RANKS, hipMemcpyHostToDevice);
    if (e != hipSuccess) { c->shm->error.store(1); return bail(ig_fail(ctx, IG_ERR_HIP, "ig_comm_init_direct: mapping a peer's window: %s", hipGetErrorString(e))); }
    if (int rc = direct_barrier(c, "ig_comm_init_direct")) return bail(rc);                  // every window is mapped everywhere
    if (rank == 0) { (void)shm_unlink(name); c->shm_name.clear(); }                           // (the mappings live on; the name is gone)
    *out = c;
    return IG_OK;
}

int ig_comm_info(ig_comm* c, int* rank, int* nranks, char* lib, size_t len) {
    if (!c) return ig_fail(nullptr, IG_ERR_ARG, "ig_comm_info: comm is NULL");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (lib && len && c->direct) { snprintf(lib, len, "direct reduce-scatter + all-gather over IPC windows of %zu MB", c->win_bytes >> 20); return IG_OK; }
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
    if (c->direct) return direct_allreduce(c, (float*)buf, nfloats);
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
    if (c->direct) return direct_allreduce(c, (float*)buf, nfloats);          // (host-synchronous: nothing is left pending)
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
    if (c->direct) {
        IG_HIP(ctx, hipStreamSynchronize(c->side));
        IG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return direct_host_scalar(c, v, op == RCCL_MAX, who);
    }
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
    if (c->direct) {
        for (int p = 0; p < c->nranks; ++p)
            if (p != c->rank && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
        if (c->d_peers) (void)hipFree(c->d_peers);
        if (c->win) (void)hipFree(c->win);
        if (c->shm) (void)munmap(c->shm, sizeof(DirectShm));
        if (c->rank == 0 && !c->shm_name.empty()) (void)shm_unlink(c->shm_name.c_str());
    }
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    if (c->ev_work) (void)hipEventDestroy(c->ev_work);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return IG_OK;
}

}  // extern "C"
