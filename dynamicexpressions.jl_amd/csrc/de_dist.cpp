// de_dist.cpp — the multi-GPU exchange of the path behind the C ABI (include/de_hip.h, "multi-GPU"): a population is
// tree-sharded round-robin over one process per GPU, X is replicated, every rank keeps its output slab,
// and the ONLY data exchanged per evaluation are the per-tree completion flags — one ncclAllGather of ceil(n_trees /
// world) bytes per rank over RCCL / xGMI, latency-bound.  A Julia (or C) caller gets the same three calls bench.py makes
// through torch.distributed: replicate X once (broadcast), evaluate the local shard (de_eval), gather the flags.
//
// RCCL is loaded at run time (dlopen): libde_hip.so has no link dependency on it and single-GPU users never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <string>
#include <thread>

#include "../../include/de_hip.h"
#include "de_kernels.h"

namespace {
struct NcclId { char internal[128]; }; // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *NcclComm;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(NcclComm *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommCount)(NcclComm, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommAbort)(NcclComm) = nullptr;              // optional: a timed-out communicator is aborted, not destroyed (destroy would wait for the stuck collective)
    int (*CommGetAsyncError)(NcclComm, int *) = nullptr; // optional
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
#define SYM(F, N)                                                           \
    F = reinterpret_cast<decltype(F)>(dlsym(lib, N));                       \
    if (!F) { err = std::string("librccl.so lacks ") + N; return false; }
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(CommCount, "ncclCommCount")
        SYM(AllGather, "ncclAllGather")
        SYM(Broadcast, "ncclBroadcast")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
        CommGetAsyncError = reinterpret_cast<decltype(CommGetAsyncError)>(dlsym(lib, "ncclCommGetAsyncError"));
        return true;
    }
};
Rccl g_rccl;
bool is_dev(const void *p) {
    hipPointerAttribute_t at;
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}
constexpr int kNcclUint8 = 1; // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1
} // namespace

struct de_comm {
    de_ctx_t *ctx = nullptr;
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
    uint8_t *send = nullptr, *recv = nullptr; // device staging: ceil(n / world) and world * ceil(n / world) bytes
    uint8_t *stage = nullptr;                 // device staging for HOST flag arrays: n bytes (local flags in, global flags out)
    size_t cap = 0, stage_cap = 0;
    // Round 6: every collective is BOUNDED.  timeout_ms > 0: the call that queued a collective waits for it (polling the stream and the
    // communicator's asynchronous error state) and gives up after timeout_ms with DE_ERR_RCCL and a message that says which rank waited
    // for what — a peer that died or never entered the collective otherwise hangs every other rank for ever.  0 = the calls stay
    // asynchronous (round 5's behaviour; the caller bounds its own synchronisation).  Default: DE_DIST_TIMEOUT_MS, else 0.
    int64_t timeout_ms = 0;
    bool dead = false; // a collective timed out: the communicator was aborted, every later call fails at once
    std::string err;
};

static int64_t env_ms(const char *name, int64_t dflt) {
    const char *v = getenv(name);
    return v && *v ? atoll(v) : dflt;
}

static int dfail(de_comm *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_rccl.err = buf;
    return code;
}

// Wait (bounded) for what was just queued on the context's stream; on timeout abort the communicator.
static int bounded_wait(de_comm *c, hipStream_t stream, const char *what) {
    if (c->timeout_ms <= 0) return DE_OK;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(c->timeout_ms);
    for (int spins = 0;; spins++) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return DE_OK;
        if (q != hipErrorNotReady) return dfail(c, DE_ERR_HIP, "%s: %s", what, hipGetErrorString(q));
        if (c->comm && g_rccl.CommGetAsyncError) {
            int ae = 0;
            if (g_rccl.CommGetAsyncError(c->comm, &ae) == 0 && ae != 0) {
                c->dead = true;
                if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
                c->comm = nullptr;
                return dfail(c, DE_ERR_RCCL, "%s: RCCL reported an asynchronous error on rank %d of %d: %s", what, c->rank, c->world, g_rccl.GetErrorString(ae));
            }
        }
        if (std::chrono::steady_clock::now() > deadline) {
            c->dead = true;
            if (c->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
            c->comm = nullptr;
            return dfail(c, DE_ERR_RCCL, "%s timed out after %lld ms on rank %d of %d: a peer is down or did not enter the same collective "
                         "(every rank must call it with the same sizes); the communicator was aborted", what, (long long)c->timeout_ms, c->rank, c->world);
        }
        if (spins < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

extern "C" {

const char *de_dist_last_error(de_comm_t *c) { return c ? c->err.c_str() : g_rccl.err.c_str(); }

int de_dist_unique_id(void *id) {
    if (!id) return DE_ERR_INVALID_ARG;
    if (!g_rccl.load()) return DE_ERR_RCCL;
    NcclId u;
    const int rc = g_rccl.GetUniqueId(&u);
    if (rc != 0) return dfail(nullptr, DE_ERR_RCCL, "ncclGetUniqueId: %s", g_rccl.GetErrorString(rc));
    std::memcpy(id, u.internal, sizeof u.internal);
    return DE_OK;
}

int de_dist_init(de_ctx_t *ctx, int rank, int world, const void *id, de_comm_t **out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return DE_ERR_INVALID_ARG;
    de_comm *c = new de_comm;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (world > 1) {
        if (!g_rccl.load()) { delete c; return DE_ERR_RCCL; }
        NcclId u;
        std::memcpy(u.internal, id, sizeof u.internal);
        // ncclCommInitRank binds the communicator to the CURRENT device of the calling thread: make that the context's (a process may hold
        // contexts on several GPUs, ABI version 2 (d)), and put the caller's device back afterwards
        int prev = -1;
        (void)hipGetDevice(&prev);
        const int dev = de_ctx_device(ctx);
        if (dev < 0 || hipSetDevice(dev) != hipSuccess) {
            dfail(nullptr, DE_ERR_HIP, "de_dist_init: cannot select the context's device %d", dev);
            delete c;
            return DE_ERR_HIP;
        }
        // ncclCommInitRank blocks until ALL `world` ranks have called it with this id: bounded too (DE_DIST_INIT_TIMEOUT_MS, default 120 s;
        // 0 = wait for ever).  It runs on a helper thread (with the context's device current there); on timeout the thread is left behind —
        // it is blocked inside RCCL and there is no call that cancels it — and the caller gets a status instead of a hang.
        const int64_t init_ms = env_ms("DE_DIST_INIT_TIMEOUT_MS", 120000);
        int rc = 0;
        if (init_ms <= 0) rc = g_rccl.CommInitRank(&c->comm, world, u, rank);
        else {
            struct Slot { NcclComm comm = nullptr; std::promise<int> done; };
            auto slot = std::make_shared<Slot>();
            std::future<int> fut = slot->done.get_future();
            std::thread([slot, world, u, rank, dev] {
                (void)hipSetDevice(dev);
                slot->done.set_value(g_rccl.CommInitRank(&slot->comm, world, u, rank));
            }).detach();
            if (fut.wait_for(std::chrono::milliseconds(init_ms)) != std::future_status::ready) {
                if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
                dfail(nullptr, DE_ERR_RCCL, "ncclCommInitRank(rank %d of %d) timed out after %lld ms: not all %d ranks called de_dist_init with this "
                      "unique id (DE_DIST_INIT_TIMEOUT_MS)", rank, world, (long long)init_ms, world);
                delete c;
                return DE_ERR_RCCL;
            }
            rc = fut.get();
            c->comm = slot->comm;
        }
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
        if (rc != 0) {
            dfail(nullptr, DE_ERR_RCCL, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(rc));
            delete c;
            return DE_ERR_RCCL;
        }
    }
    c->timeout_ms = env_ms("DE_DIST_TIMEOUT_MS", 0);
    *out = c;
    return DE_OK;
}

int de_dist_set_timeout(de_comm_t *c, int64_t timeout_ms) {
    if (!c || timeout_ms < 0) return DE_ERR_INVALID_ARG;
    c->timeout_ms = timeout_ms;
    return DE_OK;
}

int de_dist_destroy(de_comm_t *c) {
    if (!c) return DE_OK;
    if (c->comm) (void)g_rccl.CommDestroy(c->comm); // (an aborted communicator is already gone: comm == nullptr)
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->stage) (void)hipFree(c->stage);
    delete c;
    return DE_OK;
}

int de_dist_world_size(de_comm_t *c) { // the communicator's own count (ncclCommCount), not what the caller passed to de_dist_init
    if (!c) return -1;
    if (c->world == 1 || !c->comm) return 1;
    int n = -1;
    const int rc = g_rccl.CommCount(c->comm, &n);
    if (rc != 0) { dfail(c, DE_ERR_RCCL, "ncclCommCount: %s", g_rccl.GetErrorString(rc)); return -1; }
    return n;
}

int64_t de_dist_shard_size(int64_t n_trees, int rank, int world) { // trees {t : t mod world == rank}
    return n_trees > rank ? (n_trees - rank + world - 1) / world : 0;
}

int de_dist_broadcast(de_comm_t *c, void *buf, size_t bytes, int root) {
    if (!c || (!buf && bytes) || root < 0 || root >= c->world) return DE_ERR_INVALID_ARG;
    if (c->world == 1 || bytes == 0) return DE_OK;
    if (c->dead) return dfail(c, DE_ERR_RCCL, "de_dist_broadcast: the communicator was aborted by an earlier time-out");
    hipStream_t stream = static_cast<hipStream_t>(de_ctx_stream(c->ctx));
    if (hipSetDevice(de_ctx_device(c->ctx)) != hipSuccess) return dfail(c, DE_ERR_HIP, "de_dist_broadcast: cannot select the context's device");
    const int rc = g_rccl.Broadcast(buf, buf, bytes, kNcclUint8, root, c->comm, stream);
    if (rc != 0) return dfail(c, DE_ERR_RCCL, "ncclBroadcast: %s", g_rccl.GetErrorString(rc));
    return bounded_wait(c, stream, "de_dist_broadcast (ncclBroadcast)");
}

int de_dist_gather_flags(de_comm_t *c, const uint8_t *ok_local, int64_t n_trees, uint8_t *ok_global) {
    if (!c || n_trees < 0 || (n_trees > 0 && (!ok_local || !ok_global))) return DE_ERR_INVALID_ARG;
    if (n_trees == 0) return DE_OK;
    if (c->dead) return dfail(c, DE_ERR_RCCL, "de_dist_gather_flags: the communicator was aborted by an earlier time-out");
    hipStream_t stream = static_cast<hipStream_t>(de_ctx_stream(c->ctx));
    const int64_t mine = de_dist_shard_size(n_trees, c->rank, c->world);
    const size_t per = (size_t)((n_trees + c->world - 1) / c->world);
#define HIPD(expr)                                                                                      \
    do {                                                                                                \
        hipError_t st_ = (expr);                                                                        \
        if (st_ != hipSuccess) return dfail(c, DE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(st_));  \
    } while (0)
    HIPD(hipSetDevice(de_ctx_device(c->ctx))); // (the launches below go to the CURRENT device: make it the context's, as every de_eval* does)
    if (c->world == 1) {
        HIPD(hipMemcpyAsync(ok_global, ok_local, (size_t)n_trees, hipMemcpyDefault, stream));
        return bounded_wait(c, stream, "de_dist_gather_flags (one rank: a copy)");
    }
    if (c->cap < per) {
        if (c->send) (void)hipFree(c->send);
        if (c->recv) (void)hipFree(c->recv);
        c->send = c->recv = nullptr;
        c->cap = 0;
        HIPD(hipMalloc(reinterpret_cast<void **>(&c->send), per));
        HIPD(hipMalloc(reinterpret_cast<void **>(&c->recv), per * (size_t)c->world));
        c->cap = per;
    }
    // THREE stream operations per exchange when the flags live on the device (round 5; eleven before: memset + copy + all_gather + one
    // strided 1-byte-row copy per rank): pack (pads the ranks with one tree fewer with 1 = complete), all_gather, unpack (entry [r][i]
    // of the gathered block is tree r + i * world).  Host arrays go through a device staging buffer: one copy in, one copy out.
    const bool loc_dev = is_dev(ok_local), glob_dev = is_dev(ok_global);
    if ((!loc_dev || !glob_dev) && c->stage_cap < (size_t)n_trees) {
        if (c->stage) (void)hipFree(c->stage);
        c->stage = nullptr;
        c->stage_cap = 0;
        HIPD(hipMalloc(reinterpret_cast<void **>(&c->stage), (size_t)n_trees));
        c->stage_cap = (size_t)n_trees;
    }
    const uint8_t *src = ok_local;
    if (!loc_dev) {
        if (mine > 0) HIPD(hipMemcpyAsync(c->stage, ok_local, (size_t)mine, hipMemcpyHostToDevice, stream));
        src = c->stage;
    }
    HIPD(de::launch_dist_pack(c->send, src, mine, (int64_t)per, stream));
    const int rc = g_rccl.AllGather(c->send, c->recv, per, kNcclUint8, c->comm, stream);
    if (rc != 0) return dfail(c, DE_ERR_RCCL, "ncclAllGather: %s", g_rccl.GetErrorString(rc));
    HIPD(de::launch_dist_unpack(glob_dev ? ok_global : c->stage, c->recv, (int64_t)per, c->world, n_trees, stream));
    if (!glob_dev) HIPD(hipMemcpyAsync(ok_global, c->stage, (size_t)n_trees, hipMemcpyDeviceToHost, stream));
#undef HIPD
    return bounded_wait(c, stream, "de_dist_gather_flags (ncclAllGather)");
}

// Test / measurement hook (no RCCL needed): the pack and unpack launches of de_dist_gather_flags for a SIMULATED world on one GPU — every
// rank's shard of `flags_global` (host, n_trees bytes) is packed into its block of the gathered buffer (what the all_gather would deliver),
// then unpacked into `out` (host, n_trees bytes: must equal flags_global).  *ms (may be null) = device time of ONE rank's share of an
// exchange: one pack + one unpack launch (hipEvents).
int de_dist_reorder_selftest(de_ctx_t *ctx, const uint8_t *flags_global, int64_t n_trees, int world, uint8_t *out, float *ms) {
    if (!ctx || !flags_global || !out || n_trees < 1 || world < 1) return DE_ERR_INVALID_ARG;
    hipStream_t stream = static_cast<hipStream_t>(de_ctx_stream(ctx));
    const int64_t per = (n_trees + world - 1) / world;
    uint8_t *loc = nullptr, *send = nullptr, *recv = nullptr, *glob = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = DE_OK;
#define HIPS(expr) do { if (rc == DE_OK && (expr) != hipSuccess) { (void)hipGetLastError(); rc = DE_ERR_HIP; } } while (0)
    HIPS(hipSetDevice(de_ctx_device(ctx)));
    HIPS(hipMalloc(reinterpret_cast<void **>(&loc), (size_t)per));
    HIPS(hipMalloc(reinterpret_cast<void **>(&send), (size_t)per));
    HIPS(hipMalloc(reinterpret_cast<void **>(&recv), (size_t)per * world));
    HIPS(hipMalloc(reinterpret_cast<void **>(&glob), (size_t)n_trees));
    HIPS(hipEventCreate(&e0));
    HIPS(hipEventCreate(&e1));
    std::string shard((size_t)per, '\0');
    for (int r = 0; r < world && rc == DE_OK; r++) {
        const int64_t mine = de_dist_shard_size(n_trees, r, world);
        for (int64_t i = 0; i < mine; i++) shard[(size_t)i] = (char)flags_global[r + i * world];
        if (mine > 0) HIPS(hipMemcpyAsync(loc, shard.data(), (size_t)mine, hipMemcpyHostToDevice, stream));
        HIPS(hipStreamSynchronize(stream));
        if (r == 0) HIPS(hipEventRecord(e0, stream));
        HIPS(de::launch_dist_pack(send, loc, mine, per, stream));
        if (r == 0) { // rank 0's own exchange: pack, (all_gather), unpack — timed without the collective; its unpack result is overwritten below
            HIPS(de::launch_dist_unpack(glob, recv, per, world, n_trees, stream));
            HIPS(hipEventRecord(e1, stream));
        }
        HIPS(hipMemcpyAsync(recv + (size_t)r * per, send, (size_t)per, hipMemcpyDeviceToDevice, stream));
    }
    HIPS(de::launch_dist_unpack(glob, recv, per, world, n_trees, stream));
    HIPS(hipMemcpyAsync(out, glob, (size_t)n_trees, hipMemcpyDeviceToHost, stream));
    HIPS(hipStreamSynchronize(stream));
    if (ms && rc == DE_OK) HIPS(hipEventElapsedTime(ms, e0, e1));
#undef HIPS
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (uint8_t *q : {loc, send, recv, glob}) if (q) (void)hipFree(q);
    return rc;
}

} // extern "C"
