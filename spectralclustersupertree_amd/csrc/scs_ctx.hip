// Context, communicator and table upload for libscs_hip.so.
//
// One context = one process-side handle on one MI355X: a HIP stream, an
// optional communicator (RCCL over xGMI for world > 1, or the in-process test
// group) and the scratch policy of the build.  RCCL is bound with dlopen so a
// single-GPU run never pays for loading it.

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <mutex>

#include "scs_internal.h"
#include "scs_arena.h"

#include <chrono>
#include <pthread.h>

#include <algorithm>
#include <cmath>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

void scs_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char *scs_last_error(void) { return g_last_error.c_str(); }
// 101 (round 5): scs_build_stats grew by tree_parallel_batches / spec_batches (round 4), scs_tables_split added
// 102 (round 5): scs_stats grew by the mixed-precision loop's fields
// 103: ... and by event_pair_ms
extern "C" int scs_version(void) { return 105; }

extern "C" int scs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------
// RCCL binding (dlopen)
// ---------------------------------------------------------------------------
namespace {

struct rccl_uid {
    char internal[SCS_UNIQUE_ID_BYTES];
};
typedef int (*fn_get_uid)(rccl_uid *);
typedef int (*fn_init_rank)(void **, int, rccl_uid, int);
typedef int (*fn_destroy)(void *);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_send)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_errstr)(int);
typedef int (*fn_count)(void *, int *);

struct rccl_api {
    void *handle = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_errstr errstr = nullptr;
    fn_count comm_count = nullptr, comm_user_rank = nullptr;
};

rccl_api g_rccl;
std::mutex g_rccl_mutex;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return SCS_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        scs_set_error("cannot load librccl: %s", dlerror());
        return SCS_ECOMM;
    }
    g_rccl.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    g_rccl.send = (fn_send)dlsym(h, "ncclSend");
    g_rccl.recv = (fn_recv)dlsym(h, "ncclRecv");
    g_rccl.group_start = (fn_group)dlsym(h, "ncclGroupStart");
    g_rccl.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    g_rccl.comm_count = (fn_count)dlsym(h, "ncclCommCount");
    g_rccl.comm_user_rank = (fn_count)dlsym(h, "ncclCommUserRank");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allgather) {
        scs_set_error("librccl lacks an expected symbol");
        dlclose(h);
        return SCS_ECOMM;
    }
    g_rccl.handle = h;
    return SCS_OK;
}

const char *rccl_err(int rc) { return g_rccl.errstr ? g_rccl.errstr(rc) : "rccl error"; }

}  // namespace

extern "C" int scs_comm_unique_id(void *out128) {
    SCS_REQUIRE(out128 != nullptr, "scs_comm_unique_id: null output");
    SCS_TRY(load_rccl());
    rccl_uid uid;
    int rc = g_rccl.get_uid(&uid);
    if (rc != 0) {
        scs_set_error("ncclGetUniqueId failed: %s", rccl_err(rc));
        return SCS_ECOMM;
    }
    memcpy(out128, uid.internal, SCS_UNIQUE_ID_BYTES);
    return SCS_OK;
}

int scs_comm_init_rccl(scs_comm *comm, int rank, int world, const void *uid_bytes) {
    SCS_TRY(load_rccl());
    rccl_uid uid;
    memcpy(uid.internal, uid_bytes, SCS_UNIQUE_ID_BYTES);
    void *c = nullptr;
    int rc = g_rccl.init_rank(&c, world, uid, rank);
    if (rc != 0) {
        scs_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl_err(rc));
        return SCS_ECOMM;
    }
    comm->rank = rank;
    comm->world = world;
    comm->kind = 1;
    comm->rccl_comm = c;
    return SCS_OK;
}

// What the communicator itself says: kind (0 none, 1 RCCL, 2 in-process team), and -- RCCL -- the
// rank count and this rank's number as ncclCommCount / ncclCommUserRank report them (-1 when the
// library has no such entry point).  bench.py prints it with every multi-GPU line.
extern "C" int scs_ctx_comm_info(scs_ctx *ctx, int32_t *kind, int32_t *world, int32_t *rank,
                                 int32_t *reported_world, int32_t *reported_rank) {
    SCS_REQUIRE(ctx && kind && world && rank && reported_world && reported_rank, "scs_ctx_comm_info: null argument");
    *kind = ctx->comm.kind;
    *world = ctx->comm.world;
    *rank = ctx->comm.rank;
    *reported_world = -1;
    *reported_rank = -1;
    if (ctx->comm.kind == 1 && ctx->comm.rccl_comm) {
        int v = -1;
        if (g_rccl.comm_count && g_rccl.comm_count(ctx->comm.rccl_comm, &v) == 0) *reported_world = v;
        v = -1;
        if (g_rccl.comm_user_rank && g_rccl.comm_user_rank(ctx->comm.rccl_comm, &v) == 0) *reported_rank = v;
    } else if (ctx->comm.kind == 2) {
        *reported_world = ctx->comm.world;
        *reported_rank = ctx->comm.rank;
    }
    return SCS_OK;
}

// ---------------------------------------------------------------------------
// in-process group (test communicator)
// ---------------------------------------------------------------------------
struct scs_local_group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const double *> send;
    std::vector<const int64_t *> send_off;  // all-to-all-v: every rank's offsets into its send buffer
    std::atomic<int> failed{0};  // a rank hit an error inside a collective: everybody returns SCS_ECOMM
    void barrier() {
        std::unique_lock<std::mutex> lock(m);
        uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != gen; });
        }
    }
};

extern "C" int scs_local_group_create(int world, scs_local_group **out) {
    SCS_REQUIRE(out != nullptr && world >= 1 && world <= 64, "scs_local_group_create: bad world %d",
                world);
    auto *g = new scs_local_group();
    g->world = world;
    g->send.assign(world, nullptr);
    g->send_off.assign(world, nullptr);
    *out = g;
    return SCS_OK;
}

extern "C" int scs_local_group_destroy(scs_local_group *group) {
    delete group;
    return SCS_OK;
}

int scs_comm_allgather_f64(scs_comm *comm, const double *sendbuf, double *recvbuf, size_t count,
                           hipStream_t stream) {
    if (comm->world == 1 || comm->kind == 0) {
        if (recvbuf != sendbuf)
            SCS_HIP_CHECK(hipMemcpyAsync(recvbuf, sendbuf, count * sizeof(double),
                                         hipMemcpyDeviceToDevice, stream));
        return SCS_OK;
    }
    if (comm->kind == 1) {
        int rc = g_rccl.allgather(sendbuf, recvbuf, count, /*ncclFloat64*/ 8, comm->rccl_comm,
                                  stream);
        if (rc != 0) {
            scs_set_error("ncclAllGather failed: %s", rccl_err(rc));
            return SCS_ECOMM;
        }
        return SCS_OK;
    }
    // local group: publish, meet, copy every slice, meet again.  A HIP failure on one rank
    // must not leave its peers at the second meeting point: the error is kept, the barrier is
    // reached, and the failure is published so that EVERY rank returns SCS_ECOMM.
    scs_local_group *g = comm->group;
    hipError_t err = hipStreamSynchronize(stream);
    g->send[comm->rank] = sendbuf;
    g->barrier();
    for (int r = 0; r < g->world && err == hipSuccess; ++r)
        err = hipMemcpyAsync(recvbuf + (size_t)r * count, g->send[r], count * sizeof(double),
                             hipMemcpyDeviceToDevice, stream);
    if (err == hipSuccess) err = hipStreamSynchronize(stream);
    if (err != hipSuccess) g->failed.store(1);
    g->barrier();
    if (err != hipSuccess) {
        scs_set_error("local all-gather failed: %s", hipGetErrorString(err));
        return SCS_EHIP;
    }
    if (g->failed.load()) {
        scs_set_error("local all-gather: a peer rank failed");
        return SCS_ECOMM;
    }
    return SCS_OK;
}

// all-to-all-v of fp64: this rank sends doubles [send_off[p], send_off[p+1]) of sendbuf to
// rank p and receives rank p's share for it at recvbuf + recv_off[p] (recv_off[p+1] -
// recv_off[p] doubles, which must be what p sends).  Offsets are host arrays of world + 1.
int scs_comm_alltoallv_f64(scs_comm *comm, const double *sendbuf, const int64_t *send_off,
                           double *recvbuf, const int64_t *recv_off, hipStream_t stream) {
    const int world = comm->world, rank = comm->rank;
    if (world == 1 || comm->kind == 0) {
        const int64_t cnt = send_off[1] - send_off[0];
        if (cnt > 0)
            SCS_HIP_CHECK(hipMemcpyAsync(recvbuf + recv_off[0], sendbuf + send_off[0], (size_t)cnt * 8,
                                         hipMemcpyDeviceToDevice, stream));
        return SCS_OK;
    }
    if (comm->kind == 1) {
        if (!g_rccl.send || !g_rccl.recv || !g_rccl.group_start || !g_rccl.group_end) {
            scs_set_error("librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
            return SCS_ECOMM;
        }
        // own share: a device copy; the peers: one grouped round of point-to-point transfers
        // (xGMI is point to point: every pair has its own link)
        const int64_t own = send_off[rank + 1] - send_off[rank];
        if (own > 0)
            SCS_HIP_CHECK(hipMemcpyAsync(recvbuf + recv_off[rank], sendbuf + send_off[rank],
                                         (size_t)own * 8, hipMemcpyDeviceToDevice, stream));
        int rc = g_rccl.group_start();
        for (int p = 0; p < world && rc == 0; ++p) {
            if (p == rank) continue;
            const int64_t ns = send_off[p + 1] - send_off[p], nr = recv_off[p + 1] - recv_off[p];
            if (ns > 0) rc = g_rccl.send(sendbuf + send_off[p], (size_t)ns, /*ncclFloat64*/ 8, p,
                                         comm->rccl_comm, stream);
            if (rc == 0 && nr > 0)
                rc = g_rccl.recv(recvbuf + recv_off[p], (size_t)nr, /*ncclFloat64*/ 8, p,
                                 comm->rccl_comm, stream);
        }
        const int rc_end = g_rccl.group_end();
        if (rc == 0) rc = rc_end;
        if (rc != 0) {
            scs_set_error("RCCL send/recv exchange failed: %s", rccl_err(rc));
            return SCS_ECOMM;
        }
        return SCS_OK;
    }
    // local group: publish, meet, pull every peer's share, meet again (errors: as in the
    // all-gather above -- always reach the second meeting point)
    scs_local_group *g = comm->group;
    hipError_t err = hipStreamSynchronize(stream);
    g->send[rank] = sendbuf;
    g->send_off[rank] = send_off;
    g->barrier();
    int bad = 0;
    for (int p = 0; p < world && err == hipSuccess; ++p) {
        const int64_t *po = g->send_off[p];
        const int64_t cnt = po[rank + 1] - po[rank];
        if (cnt != recv_off[p + 1] - recv_off[p]) {
            bad = 1;
            continue;
        }
        if (cnt > 0)
            err = hipMemcpyAsync(recvbuf + recv_off[p], g->send[p] + po[rank], (size_t)cnt * 8,
                                 hipMemcpyDeviceToDevice, stream);
    }
    if (err == hipSuccess) err = hipStreamSynchronize(stream);
    if (err != hipSuccess || bad) g->failed.store(1);
    g->barrier();
    if (err != hipSuccess) {
        scs_set_error("local all-to-all-v failed: %s", hipGetErrorString(err));
        return SCS_EHIP;
    }
    if (bad) {
        scs_set_error("all-to-all-v: a peer sends a different count than this rank expects");
        return SCS_ECOMM;
    }
    if (g->failed.load()) {
        scs_set_error("local all-to-all-v: a peer rank failed");
        return SCS_ECOMM;
    }
    return SCS_OK;
}

// One grouped round of ncclSend / ncclRecv to THIS rank itself through the wrapper the tile
// exchange uses (argument order, ncclFloat64 = 8, stream), plus an all-gather: lets a single
// device exercise the RCCL bindings before the first multi-GPU run.  count doubles at
// sendbuf -> recvbuf (device pointers).
extern "C" int scs_debug_comm_selftest(scs_ctx *ctx, int32_t count, const double *host_in,
                                       double *host_out) {
    SCS_REQUIRE(ctx && host_in && host_out && count >= 1, "scs_debug_comm_selftest: bad argument");
    SCS_REQUIRE(ctx->comm.kind == 1, "scs_debug_comm_selftest: the context has no RCCL communicator");
    if (!g_rccl.send || !g_rccl.recv || !g_rccl.group_start || !g_rccl.group_end) {
        scs_set_error("librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
        return SCS_ECOMM;
    }
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    double *d_in = nullptr, *d_out = nullptr;
    SCS_TRY(scs_block_alloc(ctx, (size_t)count * 8, (void **)&d_in));
    int rc_all = scs_block_alloc(ctx, (size_t)count * 8, (void **)&d_out);
    if (rc_all != SCS_OK) {
        scs_block_release(ctx, d_in);
        return rc_all;
    }
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemcpyAsync(d_in, host_in, (size_t)count * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, (size_t)count * 8, s);
    int rc = e == hipSuccess ? 0 : -1;
    if (rc == 0) {
        rc = g_rccl.group_start();
        if (rc == 0) rc = g_rccl.send(d_in, (size_t)count, /*ncclFloat64*/ 8, ctx->comm.rank, ctx->comm.rccl_comm, s);
        if (rc == 0) rc = g_rccl.recv(d_out, (size_t)count, /*ncclFloat64*/ 8, ctx->comm.rank, ctx->comm.rccl_comm, s);
        const int rc_end = g_rccl.group_end();
        if (rc == 0) rc = rc_end;
        // ... and ncclAllGather itself (the solver's per-iteration collective short-cuts a
        // world of one to a device copy): rank 0's slice of the result is d_out again
        if (rc == 0 && ctx->comm.world == 1)
            rc = g_rccl.allgather(d_out, d_in, (size_t)count, /*ncclFloat64*/ 8, ctx->comm.rccl_comm, s);
        if (rc == 0 && ctx->comm.world == 1)
            rc = g_rccl.allgather(d_in, d_out, (size_t)count, /*ncclFloat64*/ 8, ctx->comm.rccl_comm, s);
    }
    if (rc == 0) {
        e = hipMemcpyAsync(host_out, d_out, (size_t)count * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    scs_block_release(ctx, d_in);
    scs_block_release(ctx, d_out);
    if (rc != 0) {
        scs_set_error("RCCL send/recv self-test failed: %s", rc > 0 ? rccl_err(rc) : "HIP copy failed");
        return SCS_ECOMM;
    }
    if (e != hipSuccess) {
        scs_set_error("RCCL send/recv self-test: %s", hipGetErrorString(e));
        return SCS_EHIP;
    }
    return SCS_OK;
}

// the measured copy rate of this device (bench.py prints it beside the nominal HBM peak)
extern "C" int scs_debug_copy_bandwidth(scs_ctx *ctx, int64_t bytes, int32_t reps, double *gbs_out) {
    SCS_REQUIRE(ctx && gbs_out && bytes >= 4096 && reps >= 1, "scs_debug_copy_bandwidth: bad argument");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    void *a = nullptr, *b = nullptr;
    SCS_TRY(scs_block_alloc(ctx, (size_t)bytes, &a));
    const int rc_b = scs_block_alloc(ctx, (size_t)bytes, &b);
    if (rc_b != SCS_OK) {
        scs_block_release(ctx, a);
        return rc_b;
    }
    hipStream_t s = ctx->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0x5A, (size_t)bytes, s);
    if (e == hipSuccess) e = hipMemcpyAsync(b, a, (size_t)bytes, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipEventRecord(e0, s);
    for (int32_t r = 0; r < reps && e == hipSuccess; ++r)
        e = hipMemcpyAsync((r & 1) ? b : a, (r & 1) ? a : b, (size_t)bytes, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    scs_block_release(ctx, a);
    scs_block_release(ctx, b);
    if (e != hipSuccess) {
        scs_set_error("scs_debug_copy_bandwidth: %s", hipGetErrorString(e));
        return SCS_EHIP;
    }
    *gbs_out = ms > 0.f ? 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
    return SCS_OK;
}

int scs_comm_destroy(scs_comm *comm) {
    if (comm->kind == 1 && comm->rccl_comm) {
        g_rccl.destroy(comm->rccl_comm);
        comm->rccl_comm = nullptr;
    }
    comm->kind = 0;
    return SCS_OK;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
static int ctx_common(int device, scs_ctx **out) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        scs_set_error("no HIP device available (%s); this library has no CPU path",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return SCS_EHIP;
    }
    SCS_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
    SCS_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    SCS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    auto *ctx = new scs_ctx();
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
            ctx->max_lds_bytes = lds;
    }
    hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete ctx;
        scs_set_error("hipStreamCreate failed: %s", hipGetErrorString(se));
        return SCS_EHIP;
    }
    // scratch budget per tree batch: a quarter of what is free now, at most 64 GiB
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)16 << 30;
    free_b += scs_arena_free_bytes(device);  // (what the arena holds free is as good as free)
    size_t lim = free_b / 4;
    const size_t cap = (size_t)64 << 30;
    ctx->ws_limit = lim < cap ? lim : cap;
    if (const char *s = getenv("SCS_WS_LIMIT_MB")) {
        long mb = atol(s);
        if (mb > 0) ctx->ws_limit = (size_t)mb << 20;
    }
    *out = ctx;
    return SCS_OK;
}

extern "C" int scs_ctx_create(int device, int rank, int world, const void *unique_id128,
                              scs_ctx **out) {
    SCS_REQUIRE(out != nullptr, "scs_ctx_create: null output");
    SCS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "scs_ctx_create: bad rank %d / world %d",
                rank, world);
    scs_ctx *ctx = nullptr;
    SCS_TRY(ctx_common(device, &ctx));
    ctx->comm.rank = rank;
    ctx->comm.world = world;
    if (world > 1 || unique_id128 != nullptr) {
        if (unique_id128 == nullptr) {
            scs_ctx_destroy(ctx);
            scs_set_error("scs_ctx_create: world > 1 needs a unique id");
            return SCS_EINVAL;
        }
        int rc = scs_comm_init_rccl(&ctx->comm, rank, world, unique_id128);
        if (rc != SCS_OK) {
            std::string keep = g_last_error;
            scs_ctx_destroy(ctx);
            g_last_error = keep;
            return rc;
        }
    }
    *out = ctx;
    return SCS_OK;
}

extern "C" int scs_ctx_create_local(int device, int rank, scs_local_group *group, scs_ctx **out) {
    SCS_REQUIRE(out != nullptr && group != nullptr, "scs_ctx_create_local: null argument");
    SCS_REQUIRE(rank >= 0 && rank < group->world, "scs_ctx_create_local: bad rank %d", rank);
    scs_ctx *ctx = nullptr;
    SCS_TRY(ctx_common(device, &ctx));
    ctx->comm.rank = rank;
    ctx->comm.world = group->world;
    ctx->comm.kind = group->world > 1 ? 2 : 0;
    ctx->comm.group = group;
    *out = ctx;
    return SCS_OK;
}

static const auto g_t_start = std::chrono::steady_clock::now();
static void alloc_trace(const char *what, size_t bytes, std::chrono::steady_clock::time_point t0) {
    const auto t1 = std::chrono::steady_clock::now();
    const double dt = std::chrono::duration<double>(t1 - t0).count();
    if (dt > 0.005)
        fprintf(stderr, "[%9.3f s] thread %lx: %s %.3f GB took %.3f s\n",
                std::chrono::duration<double>(t0 - g_t_start).count(), (unsigned long)pthread_self(), what, bytes / 1e9, dt);
}

static hipError_t driver_malloc(void **p, size_t bytes) {
    if (!scs_dbg("SCS_ALLOC_TRACE")) return hipMalloc(p, bytes);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipMalloc(p, bytes);
    alloc_trace("hipMalloc", bytes, t0);
    return e;
}

static hipError_t driver_free(void *p) {
    if (!scs_dbg("SCS_ALLOC_TRACE")) return hipFree(p);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipFree(p);
    alloc_trace("hipFree", 0, t0);
    return e;
}

// ---------------------------------------------------------------------------
// the arena of every device (scs_arena.h has the logic and the reasons)
// ---------------------------------------------------------------------------
namespace {
constexpr int ARENA_DEVICES = 16;
struct device_arena {
    std::mutex mu;
    scs_arena_core core;
    std::vector<hipEvent_t> event_pool;
    std::atomic<bool> live{false};
};
device_arena g_arena[ARENA_DEVICES];

bool arena_on() {
    static const bool on = [] {
        const char *e = scs_dbg("SCS_ARENA");  // (probe: 0 = every request straight to the driver, as before round 6)
        return !(e && atoi(e) == 0);
    }();
    return on;
}

// (called with a.mu held)
void arena_init(device_arena &a) {
    if (a.live.load()) return;
    a.core.back_alloc = [](size_t bytes) -> void * {
        void *p = nullptr;
        if (driver_malloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return p;
    };
    a.core.back_free = [](void *p) { driver_free(p); };
    a.core.owner_mark = [&a](const void *owner) {
        std::vector<void *> events;
        const scs_ctx *ctx = (const scs_ctx *)owner;
        for (hipStream_t s : {ctx->stream, ctx->small_stream, ctx->copy_stream}) {
            if (!s) continue;
            if (hipStreamQuery(s) == hipSuccess) continue;
            (void)hipGetLastError();
            hipEvent_t ev = nullptr;
            if (!a.event_pool.empty()) {
                ev = a.event_pool.back();
                a.event_pool.pop_back();
            } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                hipStreamSynchronize(s);  // (no event to be had: wait here instead)
                continue;
            }
            hipEventRecord(ev, s);
            events.push_back((void *)ev);
        }
        return events;
    };
    a.core.event_done = [](void *e) {
        if (hipEventQuery((hipEvent_t)e) == hipSuccess) return true;
        (void)hipGetLastError();
        return false;
    };
    a.core.event_wait = [](void *e) { hipEventSynchronize((hipEvent_t)e); };
    a.core.event_recycle = [&a](void *e) { a.event_pool.push_back((hipEvent_t)e); };
    a.live.store(true);
}
}  // namespace

hipError_t scs_dev_malloc_impl(scs_ctx *ctx, void **p, size_t bytes) {
    if (!ctx || !arena_on() || ctx->device < 0 || ctx->device >= ARENA_DEVICES) return driver_malloc(p, bytes);
    device_arena &a = g_arena[ctx->device];
    std::lock_guard<std::mutex> lock(a.mu);
    arena_init(a);
    *p = a.core.alloc(bytes, ctx);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}

hipError_t scs_dev_free(void *p) {
    if (!p) return hipSuccess;
    for (auto &a : g_arena) {
        if (!a.live.load()) continue;
        std::lock_guard<std::mutex> lock(a.mu);
        if (a.core.release(p)) return hipSuccess;
        for (auto &s : a.core.slabs)
            if (s.base && (char *)p >= s.base && (char *)p < s.base + s.bytes) {
                // (inside a slab, but no chunk in use starts here: released twice, or a pointer into a chunk)
                fprintf(stderr, "scs_dev_free: %p is not an allocation of the arena (ignored)\n", p);
                return hipErrorInvalidValue;
            }
    }
    return driver_free(p);
}

size_t scs_arena_free_bytes(int device) {
    if (device < 0 || device >= ARENA_DEVICES || !g_arena[device].live.load()) return 0;
    std::lock_guard<std::mutex> lock(g_arena[device].mu);
    return g_arena[device].core.free_bytes();
}

// a context's last act: what it released is everybody's, what objects made on it still hold is orphaned
static void arena_ctx_gone(scs_ctx *ctx) {
    if (ctx->device < 0 || ctx->device >= ARENA_DEVICES || !g_arena[ctx->device].live.load()) return;
    std::lock_guard<std::mutex> lock(g_arena[ctx->device].mu);
    g_arena[ctx->device].core.owner_gone(ctx);
}

static size_t arena_trim(int device, size_t keep) {
    if (device < 0 || device >= ARENA_DEVICES || !g_arena[device].live.load()) return 0;
    std::lock_guard<std::mutex> lock(g_arena[device].mu);
    return g_arena[device].core.trim(keep);
}

// {bytes of slabs, bytes in use, slabs, chunks, pending chunks, driver allocations, driver releases, requests}
extern "C" int scs_debug_arena_stats(int device, int64_t *out8) {
    SCS_REQUIRE(out8 != nullptr && device >= 0 && device < ARENA_DEVICES, "scs_debug_arena_stats: bad argument");
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    device_arena &a = g_arena[device];
    if (!a.live.load()) return SCS_OK;
    std::lock_guard<std::mutex> lock(a.mu);
    int64_t n_slabs = 0;
    for (auto &s : a.core.slabs) n_slabs += s.base != nullptr;
    out8[0] = (int64_t)a.core.slab_bytes;
    out8[1] = (int64_t)a.core.used_bytes;
    out8[2] = n_slabs;
    out8[3] = (int64_t)a.core.chunks.size();
    out8[4] = (int64_t)a.core.n_pending;
    out8[5] = (int64_t)a.core.n_driver_allocs;
    out8[6] = (int64_t)a.core.n_driver_frees;
    out8[7] = (int64_t)a.core.n_allocs;
    return SCS_OK;
}

// A device block of a context: carved out of the device's arena (until round 6 a per-context cache of whole
// hipMalloc blocks in size classes -- what one context had released no other could use, and every miss and
// every release above the cache's budget was a driver call).
int scs_block_alloc(scs_ctx *ctx, size_t bytes, void **out) {
    if (bytes < 256) bytes = 256;
    if (scs_dev_malloc_impl(ctx, out, bytes) == hipSuccess) return SCS_OK;
    {
        // make room: the W buffer and the image this context keeps for a next graph of the same size
        std::lock_guard<std::mutex> lock(ctx->cache_mu);
        if (ctx->w_cache) {
            scs_dev_free(ctx->w_cache);
            ctx->w_cache = nullptr;
            ctx->w_cache_bytes = 0;
        }
        if (ctx->w32_cache) {
            scs_dev_free(ctx->w32_cache);
            ctx->w32_cache = nullptr;
            ctx->w32_cache_bytes = 0;
        }
    }
    if (scs_dev_malloc_impl(ctx, out, bytes) == hipSuccess) return SCS_OK;
    (void)hipGetLastError();
    scs_set_error("cannot allocate %zu bytes of device memory", bytes);
    return SCS_ENOMEM;
}

int scs_pinned_get(scs_ctx *ctx, size_t bytes, void **out) {
    std::lock_guard<std::mutex> lock(ctx->cache_mu);
    if (bytes < 4096) bytes = 4096;
    for (auto &b : ctx->pinned)
        if (!b.in_use && b.bytes >= bytes && b.bytes <= 2 * bytes + 65536) {
            b.in_use = true;
            *out = b.p;
            return SCS_OK;
        }
    // (sizes in steps of 1/4 octave: the deep recursion asks for a slightly different size every time)
    size_t cap = 4096;
    while (cap < bytes) cap += cap / 4 >= 4096 ? cap / 4 : 4096;
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, cap, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        scs_set_error("cannot pin %zu bytes of host memory: %s", cap, hipGetErrorString(e));
        return SCS_ENOMEM;
    }
    ctx->pinned.push_back({p, cap, true});
    *out = p;
    return SCS_OK;
}

void scs_pinned_release(scs_ctx *ctx, void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(ctx->cache_mu);
    size_t free_bytes = 0;
    for (auto &b : ctx->pinned) {
        if (b.p == p) b.in_use = false;
        if (!b.in_use) free_bytes += b.bytes;
    }
    while (free_bytes > SCS_PINNED_KEEP) {
        size_t pick = ctx->pinned.size();
        for (size_t i = 0; i < ctx->pinned.size(); ++i)
            if (!ctx->pinned[i].in_use && (pick == ctx->pinned.size() || ctx->pinned[i].bytes > ctx->pinned[pick].bytes))
                pick = i;
        if (pick == ctx->pinned.size()) break;
        free_bytes -= ctx->pinned[pick].bytes;
        hipHostFree(ctx->pinned[pick].p);
        ctx->pinned.erase(ctx->pinned.begin() + pick);
    }
}

void scs_block_drop_free(scs_ctx *ctx) {
    // (the arena makes room by itself when the driver refuses a slab: scs_arena_core::alloc)
    (void)ctx;
}

void scs_block_release(scs_ctx *ctx, void *p) {
    (void)ctx;
    if (p) scs_dev_free(p);
}

extern "C" int scs_ctx_destroy(scs_ctx *ctx) {
    if (!ctx) return SCS_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) {
        hipStreamSynchronize(ctx->stream);
    }
    scs_comm_destroy(&ctx->comm);
    if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
    if (ctx->small_stream) hipStreamSynchronize(ctx->small_stream);
    // (the streams themselves go LAST: until the arena has forgotten this context -- arena_ctx_gone below --
    // another context's request may ask whether they have passed a release)
    if (ctx->h_report) hipHostFree(ctx->h_report);
    for (auto &sl : ctx->scratch)
        if (sl.p) scs_dev_free(sl.p);
    for (auto &e : ctx->build_events)
        if (e) hipEventDestroy(e);
    for (auto &e : ctx->solve_events)
        if (e) hipEventDestroy(e);
    if (ctx->w_cache) scs_dev_free(ctx->w_cache);
    if (ctx->w32_cache) scs_dev_free(ctx->w32_cache);
    // (a page-locked block still lent to a forest's host-side tables stays: arrays may still view it)
    for (auto &b : ctx->pinned)
        if (!b.in_use) hipHostFree(b.p);
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    for (auto &sl : ctx->small_slots) {
        if (sl.done) hipEventDestroy(sl.done);
        if (sl.dev) scs_dev_free(sl.dev);
        if (sl.host) hipHostFree(sl.host);
        if (sl.scratch) scs_dev_free(sl.scratch);
    }
    ctx->small_slots.clear();
    if (ctx->h_flags) hipHostFree(ctx->h_flags);
    arena_ctx_gone(ctx);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->small_stream) hipStreamDestroy(ctx->small_stream);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return SCS_OK;
}

// Give back what the context only keeps for a next call of the same size: the cached W buffer and its
// single-precision image, every free cached block above `keep_bytes` in all (largest first), the free
// page-locked blocks.  The recursion calls it behind its largest nodes: the root's 80 GB buffer serves no
// later node (sizes only shrink), and memory held back here is memory the level forests, the look-ahead
// workers' contexts and the runtime's own scratch cannot have.
extern "C" int scs_ctx_trim(scs_ctx *ctx, int64_t keep_bytes) {
    SCS_REQUIRE(ctx != nullptr, "scs_ctx_trim: null context");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    SCS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->small_stream) SCS_HIP_CHECK(hipStreamSynchronize(ctx->small_stream));
    std::lock_guard<std::mutex> lock(ctx->cache_mu);
    if (ctx->w_cache) {
        scs_dev_free(ctx->w_cache);
        ctx->w_cache = nullptr;
        ctx->w_cache_bytes = 0;
    }
    if (ctx->w32_cache) {
        scs_dev_free(ctx->w32_cache);
        ctx->w32_cache = nullptr;
        ctx->w32_cache_bytes = 0;
    }
    const size_t keep = keep_bytes > 0 ? (size_t)keep_bytes : 0;
    if (keep == 0) {
        // the build's scratch slots are kept between calls (up to SCS_SCRATCH_KEEP each); they were carved out of
        // whatever slab had room -- and one live chunk keeps a whole slab from going back
        for (auto &sl : ctx->scratch) {
            if (sl.p) scs_dev_free(sl.p);
            sl.p = nullptr;
            sl.cap = 0;
        }
        // ... and so are the staging and scratch blocks of the small-solve slots no ticket holds (they used to stay
        // until the context went away -- a level's batch of small nodes takes up to 8 GB of addends)
        for (auto &sl : ctx->small_slots) {
            if (sl.busy) continue;
            if (sl.dev) scs_dev_free(sl.dev);
            if (sl.host) hipHostFree(sl.host);
            if (sl.scratch) scs_dev_free(sl.scratch);
            sl.dev = nullptr;
            sl.host = nullptr;
            sl.cap = 0;
            sl.scratch = nullptr;
            sl.scratch_cap = 0;
        }
    }
    arena_trim(ctx->device, keep);
    for (size_t i = 0; i < ctx->pinned.size();) {
        if (!ctx->pinned[i].in_use && keep == 0) {
            hipHostFree(ctx->pinned[i].p);
            ctx->pinned.erase(ctx->pinned.begin() + i);
        } else {
            ++i;
        }
    }
    return SCS_OK;
}

// Make sure the device's arena holds `bytes` of free memory in one piece (a request of that size will not go to
// the driver): what a warm-up call of the same size would leave behind, without the call.
extern "C" int scs_ctx_reserve(scs_ctx *ctx, int64_t bytes) {
    SCS_REQUIRE(ctx != nullptr && bytes >= 0, "scs_ctx_reserve: bad argument");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    if (bytes == 0) return SCS_OK;
    void *p = nullptr;
    SCS_TRY(scs_block_alloc(ctx, (size_t)bytes, &p));
    scs_block_release(ctx, p);
    return SCS_OK;
}

extern "C" int scs_ctx_synchronize(scs_ctx *ctx) {
    SCS_REQUIRE(ctx != nullptr, "scs_ctx_synchronize: null context");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    SCS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SCS_OK;
}

// ---------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------
// Range checks of the leaf arrays, on the device (the arrays are there anyway; a host loop over
// 5 * 10^6 leaves cost as much as the copy): bit 0 of *flags: a taxon id outside [0, n_taxa),
// bit 1: a negative LCA depth.
__global__ __launch_bounds__(256) void k_validate_tables(const int32_t *__restrict__ leaf_taxon,
                                                          const int32_t *__restrict__ adj_depth,
                                                          int64_t n_leaves, int32_t n_taxa,
                                                          unsigned *__restrict__ flags) {
    unsigned bad = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_leaves;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t tx = leaf_taxon[p];
        if (tx < 0 || tx >= n_taxa) bad |= 1u;
        if (adj_depth[p] < 0) bad |= 2u;
    }
    if (bad) atomicOr(flags, bad);
}

// Page-locked host memory for callers that want their tables to travel at the full PCIe rate
// (a pageable source is staged by the runtime at a fraction of it).
extern "C" int scs_host_alloc(size_t bytes, void **out) {
    SCS_REQUIRE(out != nullptr, "scs_host_alloc: null output");
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        scs_set_error("scs_host_alloc: cannot pin %zu bytes: %s", bytes, hipGetErrorString(e));
        return SCS_ENOMEM;
    }
    *out = p;
    return SCS_OK;
}

extern "C" int scs_host_free(void *p) {
    if (p) SCS_HIP_CHECK(hipHostFree(p));
    return SCS_OK;
}

int scs_tables_wait(scs_ctx *ctx, const scs_tables *t, int32_t t_end, hipStream_t stream) {
    for (size_t c = 0; c < t->late_ev.size(); ++c)
        if (t->late_start[c] < t_end) SCS_HIP_CHECK(hipStreamWaitEvent(stream, t->late_ev[c], 0));
    return SCS_OK;
}

int scs_tables_finish(scs_ctx *ctx, const scs_tables *t) {
    if (t->late_ev.empty()) return SCS_OK;
    hipError_t e = hipSuccess;
    for (hipEvent_t ev : t->late_ev) {
        const hipError_t w = hipEventSynchronize(ev);
        if (e == hipSuccess) e = w;
        hipEventDestroy(ev);
    }
    t->late_ev.clear();
    t->late_start.clear();
    // (the range checks of the late chunks run behind their events on the copy stream)
    if (e == hipSuccess && ctx->copy_stream) e = hipStreamSynchronize(ctx->copy_stream);
    unsigned bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, t->d_flags, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        scs_set_error("table upload failed: %s", hipGetErrorString(e));
        return SCS_EHIP;
    }
    if (bad) {
        scs_set_error("scs_tables_upload: %s", (bad & 1u) ? "a leaf_taxon entry is out of range [0, n_taxa)"
                                                          : "an adj_depth entry is negative");
        return SCS_EINVAL;
    }
    return SCS_OK;
}

// the memory a pointer names is page-locked host memory (hipHostMalloc / hipHostRegister)
static bool scs_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary pageable pointer: not an error of ours
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

extern "C" int scs_tables_free(scs_ctx *ctx, scs_tables *t) {
    if (!t) return SCS_OK;
    if (ctx) {
        hipSetDevice(ctx->device);
        (void)scs_tables_finish(ctx, t);  // nothing may still be writing into the block
        scs_block_release(ctx, t->d_block);
    } else if (t->d_block) {
        scs_dev_free(t->d_block);
    }
    delete t;
    return SCS_OK;
}

// leaf_taxon of a resident forest's tables through the node's renumbering (null: identity)
__global__ void k_relabel_taxa(const int32_t *__restrict__ src, const int32_t *__restrict__ relabel, int64_t n,
                               int32_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = relabel ? relabel[src[i]] : src[i];
}

// scs_tables of a node whose tables are already on the device: a child of scs_forest_split.  Device-to-
// device copies and the taxon renumbering (present taxa only, contraction groups consecutive) instead of
// the host -> HBM copy of scs_tables_upload; nothing to range-check (the split produced the ids and the
// map comes from the same host code that would have renumbered the host arrays).
extern "C" int scs_tables_from_forest(scs_ctx *ctx, const scs_forest *f, const int32_t *relabel, int32_t n_taxa,
                                      scs_tables **out) {
    SCS_REQUIRE(ctx && f && out, "scs_tables_from_forest: null argument");
    SCS_REQUIRE(f->has_tables && f->h_tree_off, "scs_tables_from_forest: the forest carries no tables (not a child of scs_forest_split)");
    SCS_REQUIRE(n_taxa >= 1 && f->n_trees >= 1, "scs_tables_from_forest: need >= 1 taxon and >= 1 tree");
    const int32_t n_trees = f->n_trees;
    const int64_t L = f->n_leaves;
    int64_t max_leaves = 0;
    for (int32_t t = 0; t < n_trees; ++t) max_leaves = std::max(max_leaves, f->h_tree_off[t + 1] - f->h_tree_off[t]);
    SCS_REQUIRE(max_leaves <= n_taxa, "scs_tables_from_forest: a tree has more leaves (%lld) than the node has taxa (%d)",
                (long long)max_leaves, n_taxa);
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_off = 0;
    const size_t o_tax = o_off + up256(((size_t)n_trees + 1) * 8);
    const size_t o_dep = o_tax + up256((size_t)L * 4);
    const size_t o_val = o_dep + up256((size_t)L * 4);
    const size_t o_w = o_val + up256((size_t)L * 8);
    const size_t o_flag = o_w + up256((size_t)n_trees * 8);
    const size_t o_rl = o_flag + 256;
    void *block = nullptr;
    SCS_TRY(scs_block_alloc(ctx, o_rl + (relabel ? up256((size_t)f->n_taxa * 4) : 0), &block));
    auto *t = new scs_tables();
    t->n_taxa = n_taxa;
    t->n_trees = n_trees;
    t->n_leaves = L;
    t->max_leaves = (int32_t)max_leaves;
    t->h_tree_off.assign(f->h_tree_off, f->h_tree_off + n_trees + 1);
    t->d_block = block;
    char *base = (char *)block;
    t->d_tree_off = (int64_t *)(base + o_off);
    t->d_leaf_taxon = (int32_t *)(base + o_tax);
    t->d_adj_depth = (int32_t *)(base + o_dep);
    t->d_adj_val = (double *)(base + o_val);
    t->d_tree_w = (double *)(base + o_w);
    t->d_flags = (unsigned *)(base + o_flag);
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemsetAsync(t->d_flags, 0, 4, s);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_tree_off, f->tree_off, ((size_t)n_trees + 1) * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && L) e = hipMemcpyAsync(t->d_adj_depth, f->adj_depth, (size_t)L * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && L) e = hipMemcpyAsync(t->d_adj_val, f->adj_val, (size_t)L * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_tree_w, f->weights, (size_t)n_trees * 8, hipMemcpyDeviceToDevice, s);
    const int32_t *d_rl = nullptr;
    if (e == hipSuccess && relabel) {
        // (a pageable source: staged by the runtime before the call returns)
        e = hipMemcpyAsync(base + o_rl, relabel, (size_t)f->n_taxa * 4, hipMemcpyHostToDevice, s);
        d_rl = (const int32_t *)(base + o_rl);
    }
    if (e == hipSuccess && L) {
        k_relabel_taxa<<<(unsigned)((L + 255) / 256), 256, 0, s>>>(f->leaf_taxon, d_rl, L, t->d_leaf_taxon);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // (the forest may be freed by the caller right away)
    if (e != hipSuccess) {
        scs_tables_free(ctx, t);
        scs_set_error("scs_tables_from_forest failed: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? SCS_ENOMEM : SCS_EHIP;
    }
    *out = t;
    return SCS_OK;
}

// leaf_taxon of a tree range of a level forest through the node's renumbering of its id range
__global__ void k_relabel_taxa_range(const int32_t *__restrict__ src, const int32_t *__restrict__ relabel, int32_t u_base,
                                     int64_t n, int32_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = relabel[src[i] - u_base];
}

__global__ void k_rebase_offsets(const int64_t *__restrict__ src, int64_t n, int64_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] - src[0];
}

// scs_tables_from_forest for ONE node of a level forest (scs_forest_split_level): the trees [t_begin, t_end),
// whose taxon ids lie in [u_base, u_base + u_size); relabel[x - u_base] = the node's own id of taxon x.
extern "C" int scs_tables_from_forest_range(scs_ctx *ctx, const scs_forest *f, int32_t t_begin, int32_t t_end,
                                            int32_t u_base, int32_t u_size, const int32_t *relabel, int32_t n_taxa,
                                            scs_tables **out) {
    SCS_REQUIRE(ctx && f && relabel && out, "scs_tables_from_forest_range: null argument");
    SCS_REQUIRE(f->has_tables, "scs_tables_from_forest_range: the forest carries no tables");
    SCS_REQUIRE(t_begin >= 0 && t_begin < t_end && t_end <= f->n_trees, "scs_tables_from_forest_range: bad tree range");
    SCS_REQUIRE(u_base >= 0 && u_size >= 1 && (int64_t)u_base + u_size <= f->n_taxa && n_taxa >= 1,
                "scs_tables_from_forest_range: bad taxon range");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int32_t n_trees = t_end - t_begin;
    auto *t = new scs_tables();
    t->h_tree_off.resize((size_t)n_trees + 1);
    hipError_t e = hipMemcpyAsync(t->h_tree_off.data(), f->tree_off + t_begin, ((size_t)n_trees + 1) * 8,
                                  hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        delete t;
        scs_set_error("scs_tables_from_forest_range failed: %s", hipGetErrorString(e));
        return SCS_EHIP;
    }
    const int64_t lo = t->h_tree_off[0], L = t->h_tree_off[n_trees] - lo;
    int64_t max_leaves = 0;
    for (int32_t i = 0; i <= n_trees; ++i) t->h_tree_off[i] -= lo;
    for (int32_t i = 0; i < n_trees; ++i) max_leaves = std::max(max_leaves, t->h_tree_off[i + 1] - t->h_tree_off[i]);
    if (max_leaves > n_taxa) {
        delete t;
        scs_set_error("scs_tables_from_forest_range: a tree has more leaves (%lld) than the node has taxa (%d)",
                      (long long)max_leaves, n_taxa);
        return SCS_EINVAL;
    }
    auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_off = 0;
    const size_t o_tax = o_off + up256(((size_t)n_trees + 1) * 8);
    const size_t o_dep = o_tax + up256((size_t)L * 4);
    const size_t o_val = o_dep + up256((size_t)L * 4);
    const size_t o_w = o_val + up256((size_t)L * 8);
    const size_t o_flag = o_w + up256((size_t)n_trees * 8);
    const size_t o_rl = o_flag + 256;
    void *block = nullptr;
    const int rc = scs_block_alloc(ctx, o_rl + up256((size_t)u_size * 4), &block);
    if (rc != SCS_OK) {
        delete t;
        return rc;
    }
    t->n_taxa = n_taxa;
    t->n_trees = n_trees;
    t->n_leaves = L;
    t->max_leaves = (int32_t)max_leaves;
    t->d_block = block;
    char *base = (char *)block;
    t->d_tree_off = (int64_t *)(base + o_off);
    t->d_leaf_taxon = (int32_t *)(base + o_tax);
    t->d_adj_depth = (int32_t *)(base + o_dep);
    t->d_adj_val = (double *)(base + o_val);
    t->d_tree_w = (double *)(base + o_w);
    t->d_flags = (unsigned *)(base + o_flag);
    e = hipMemsetAsync(t->d_flags, 0, 4, s);
    if (e == hipSuccess) {
        k_rebase_offsets<<<(unsigned)((n_trees + 1 + 255) / 256), 256, 0, s>>>(f->tree_off + t_begin, (int64_t)n_trees + 1,
                                                                              t->d_tree_off);
        e = hipGetLastError();
    }
    if (e == hipSuccess && L) e = hipMemcpyAsync(t->d_adj_depth, f->adj_depth + lo, (size_t)L * 4, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && L) e = hipMemcpyAsync(t->d_adj_val, f->adj_val + lo, (size_t)L * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_tree_w, f->weights + t_begin, (size_t)n_trees * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(base + o_rl, relabel, (size_t)u_size * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && L) {
        k_relabel_taxa_range<<<(unsigned)((L + 255) / 256), 256, 0, s>>>(f->leaf_taxon + lo, (const int32_t *)(base + o_rl),
                                                                          u_base, L, t->d_leaf_taxon);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        scs_tables_free(ctx, t);
        scs_set_error("scs_tables_from_forest_range failed: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? SCS_ENOMEM : SCS_EHIP;
    }
    *out = t;
    return SCS_OK;
}

extern "C" int scs_tables_upload(scs_ctx *ctx, int32_t n_taxa, int32_t n_trees,
                                 const int64_t *tree_off, const int32_t *leaf_taxon,
                                 const int32_t *adj_depth, const double *adj_val,
                                 const double *tree_w, scs_tables **out) {
    SCS_REQUIRE(ctx && out, "scs_tables_upload: null context or output");
    SCS_REQUIRE(n_taxa >= 1 && n_trees >= 1, "scs_tables_upload: need >= 1 taxon and >= 1 tree");
    SCS_REQUIRE(tree_off && leaf_taxon && adj_depth && adj_val && tree_w,
                "scs_tables_upload: null table pointer");
    SCS_REQUIRE(tree_off[0] == 0, "scs_tables_upload: tree_off[0] must be 0");
    int64_t max_leaves = 0;
    for (int32_t t = 0; t < n_trees; ++t) {
        int64_t n = tree_off[t + 1] - tree_off[t];
        SCS_REQUIRE(n >= 1, "scs_tables_upload: tree %d has %lld leaves", t, (long long)n);
        SCS_REQUIRE(n <= n_taxa, "scs_tables_upload: tree %d has more leaves (%lld) than taxa (%d)",
                    t, (long long)n, n_taxa);
        if (n > max_leaves) max_leaves = n;
    }
    const int64_t L = tree_off[n_trees];
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->h_flags) SCS_HIP_CHECK(hipHostMalloc((void **)&ctx->h_flags, 64, hipHostMallocDefault));
    // one device block for the five arrays (from the context's cache: the recursion uploads
    // thousands of table sets), 256-byte aligned pieces, 4 trailing bytes for the check flags
    auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_off = 0;
    const size_t o_tax = o_off + up256(((size_t)n_trees + 1) * 8);
    const size_t o_dep = o_tax + up256((size_t)L * 4);
    const size_t o_val = o_dep + up256((size_t)L * 4);
    const size_t o_w = o_val + up256((size_t)L * 8);
    const size_t o_flag = o_w + up256((size_t)n_trees * 8);
    void *block = nullptr;
    SCS_TRY(scs_block_alloc(ctx, o_flag + 256, &block));
    auto *t = new scs_tables();
    t->n_taxa = n_taxa;
    t->n_trees = n_trees;
    t->n_leaves = L;
    t->max_leaves = (int32_t)max_leaves;
    t->h_tree_off.assign(tree_off, tree_off + n_trees + 1);
    t->d_block = block;
    char *base = (char *)block;
    t->d_tree_off = (int64_t *)(base + o_off);
    t->d_leaf_taxon = (int32_t *)(base + o_tax);
    t->d_adj_depth = (int32_t *)(base + o_dep);
    t->d_adj_val = (double *)(base + o_val);
    t->d_tree_w = (double *)(base + o_w);
    unsigned *d_flags = (unsigned *)(base + o_flag);
    t->d_flags = d_flags;
    hipStream_t s = ctx->stream;
    // How much has to be there before the call returns.  Pageable arrays: everything (the runtime
    // stages such a copy and blocks anyway).  Page-locked arrays of a forest that scs_pcg_build will
    // walk in several tree batches: the first batch's worth -- the same rule of thumb as the build's
    // (about 600 MB of range-minimum tables, 64 to 256 trees) -- and the rest in chunks of that many
    // trees on the copy stream, overlapping the build's first batches.
    int32_t first = n_trees, chunk = n_trees;
    if (n_taxa > 2048 && (size_t)L * 16 >= ((size_t)8 << 20) && scs_is_pinned(leaf_taxon) &&
        scs_is_pinned(adj_depth) && scs_is_pinned(adj_val)) {
        const double avg = std::max((double)L / n_trees - 1.0, 1.0);
        const double table_bytes = (std::floor(std::log2(avg)) + 1.0) * avg * 8.0;
        const int32_t per = (int32_t)std::min(256.0, std::max(64.0, 600e6 / table_bytes));
        if (n_trees >= 2 * per || n_trees - per >= 32) {
            // the call waits for 64 trees only (scs_pcg_build then makes them a first, short batch:
            // an extra launch is cheaper than waiting for the other three quarters of a batch).
            // (Balancing the first batch's build against the copy of the rest -- a tree of L leaves
            // takes ~1.45e-13 L^2 s to accumulate and 16 L bytes at ~50 GB/s to arrive: F / M =
            // 1 / (1 + L / 2200), ~90 trees at 10 000 x 500 -- was measured (a switch since removed):
            // 64 / 80 / 96 / 112 trees first give 17.17 / 17.07 / 17.11 / 17.16 ms a step: within the
            // run-to-run spread, the rule stays.)
            first = std::min(per, 64);
            chunk = per;
        }
    }
    const int64_t L0 = tree_off[first];
    hipError_t e = hipMemsetAsync(d_flags, 0, 4, s);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_tree_off, tree_off, ((size_t)n_trees + 1) * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && L0) e = hipMemcpyAsync(t->d_leaf_taxon, leaf_taxon, (size_t)L0 * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && L0) e = hipMemcpyAsync(t->d_adj_depth, adj_depth, (size_t)L0 * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && L0) e = hipMemcpyAsync(t->d_adj_val, adj_val, (size_t)L0 * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_tree_w, tree_w, (size_t)n_trees * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        const int grid = (int)std::min<int64_t>((L0 + 255) / 256 + 1, 4096);
        k_validate_tables<<<grid, 256, 0, s>>>(t->d_leaf_taxon, t->d_adj_depth, L0, n_taxa, d_flags);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_flags, d_flags, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && first < n_trees) {
        if (!ctx->copy_stream) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
        hipStream_t cs = ctx->copy_stream;
        for (int32_t a = first; a < n_trees && e == hipSuccess; a += chunk) {
            const int32_t b = std::min(n_trees, a + chunk);
            const int64_t p0 = tree_off[a], cnt = tree_off[b] - tree_off[a];
            e = hipMemcpyAsync(t->d_leaf_taxon + p0, leaf_taxon + p0, (size_t)cnt * 4, hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) e = hipMemcpyAsync(t->d_adj_depth + p0, adj_depth + p0, (size_t)cnt * 4, hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) e = hipMemcpyAsync(t->d_adj_val + p0, adj_val + p0, (size_t)cnt * 8, hipMemcpyHostToDevice, cs);
            // the chunk's event fires when its BYTES have arrived; the range check follows it on the copy
            // stream and is waited for by scs_tables_finish only.  (Round 5: with the check in front of the
            // event a build whose tile kernel fills every CU -- one twelve-wave workgroup each, all vector
            // registers taken, milliseconds per workgroup -- kept the check, and with it the next tree
            // batch, waiting for a free slot: 210 ms of a configs[4] step.  The build's own kernels guard
            // against ids out of range; the verdict is collected at the end of the build as before.)
            hipEvent_t ev = nullptr;
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(ev, cs);
            if (e == hipSuccess) {
                t->late_start.push_back(a);
                t->late_ev.push_back(ev);
            } else if (ev) {
                hipEventDestroy(ev);
            }
            if (e == hipSuccess) {
                const int grid = (int)std::min<int64_t>((cnt + 255) / 256 + 1, 4096);
                k_validate_tables<<<grid, 256, 0, cs>>>(t->d_leaf_taxon + p0, t->d_adj_depth + p0, cnt, n_taxa, d_flags);
                e = hipGetLastError();
            }
        }
    }
    if (e != hipSuccess) {
        scs_tables_free(ctx, t);
        scs_set_error("table upload failed: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? SCS_ENOMEM : SCS_EHIP;
    }
    const unsigned bad = *ctx->h_flags;
    if (bad) {
        scs_tables_free(ctx, t);
        scs_set_error("scs_tables_upload: %s", (bad & 1u) ? "a leaf_taxon entry is out of range [0, n_taxa)"
                                                          : "an adj_depth entry is negative");
        return SCS_EINVAL;
    }
    *out = t;
    return SCS_OK;
}
