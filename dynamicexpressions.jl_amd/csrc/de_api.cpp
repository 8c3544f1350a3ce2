// de_api.cpp — C ABI (include/de_hip.h): registry, contexts, the pool of host threads, recycled device buffers and parked programs.
// The rest of the ABI: de_api_program.cpp, de_api_eval.cpp, de_api_grad.cpp (de_api_internal.h says what lives where).
#include "de_api_internal.h"

extern "C" {
int fail(de_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}
// the pair of events that brackets the launches of a call: the context's own pair, or the next slot of the timing ring
hipError_t time_begin(de_ctx *c) {
    return hipEventRecord(c->ring.empty() ? c->ev0 : c->ring[(size_t)(c->ring_at % (c->ring.size() / 2)) * 2], c->stream);
}
hipError_t time_end(de_ctx *c) {
    hipEvent_t e = c->ev1;
    if (!c->ring.empty()) { e = c->ring[(size_t)(c->ring_at % (c->ring.size() / 2)) * 2 + 1]; c->ring_at++; }
    c->timed = true;
    return hipEventRecord(e, c->stream);
}
bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    hipError_t st = hipPointerGetAttributes(&at, p);
    if (st != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged
#if defined(hipMemoryTypeArray)
           || at.type == hipMemoryTypeArray
#endif
        ;
}

// ---------------------------------------------------------------------------
}
// registry
struct OpName { const char *name; int degree; int code; };
static const OpName kOps[] = {
    {"neg", 1, DE_U_NEG}, {"-", 1, DE_U_NEG}, {"abs", 1, DE_U_ABS}, {"square", 1, DE_U_SQUARE},
    {"cube", 1, DE_U_CUBE}, {"relu", 1, DE_U_RELU}, {"sign", 1, DE_U_SIGN}, {"round", 1, DE_U_ROUND},
    {"floor", 1, DE_U_FLOOR}, {"ceil", 1, DE_U_CEIL}, {"inv", 1, DE_U_INV}, {"sqrt", 1, DE_U_SQRT},
    {"cbrt", 1, DE_U_CBRT}, {"exp", 1, DE_U_EXP}, {"exp2", 1, DE_U_EXP2}, {"log", 1, DE_U_LOG},
    {"log2", 1, DE_U_LOG2}, {"log10", 1, DE_U_LOG10}, {"log1p", 1, DE_U_LOG1P}, {"sin", 1, DE_U_SIN},
    {"cos", 1, DE_U_COS}, {"tan", 1, DE_U_TAN}, {"sinh", 1, DE_U_SINH}, {"cosh", 1, DE_U_COSH},
    {"tanh", 1, DE_U_TANH}, {"asin", 1, DE_U_ASIN}, {"acos", 1, DE_U_ACOS}, {"atan", 1, DE_U_ATAN},
    {"asinh", 1, DE_U_ASINH}, {"acosh", 1, DE_U_ACOSH}, {"atanh", 1, DE_U_ATANH},
    {"safe_log", 1, DE_U_SAFE_LOG}, {"safe_log2", 1, DE_U_SAFE_LOG2}, {"safe_log10", 1, DE_U_SAFE_LOG10},
    {"safe_log1p", 1, DE_U_SAFE_LOG1P}, {"safe_sqrt", 1, DE_U_SAFE_SQRT},
    {"safe_acosh", 1, DE_U_SAFE_ACOSH}, {"custom_cos", 1, DE_U_COS2}, {"gamma", 1, DE_U_GAMMA},
    {"+", 2, DE_B_ADD}, {"add", 2, DE_B_ADD}, {"-", 2, DE_B_SUB}, {"sub", 2, DE_B_SUB},
    {"*", 2, DE_B_MUL}, {"mult", 2, DE_B_MUL}, {"/", 2, DE_B_DIV}, {"div", 2, DE_B_DIV},
    {"^", 2, DE_B_POW}, {"pow", 2, DE_B_POW}, {"max", 2, DE_B_MAX}, {"min", 2, DE_B_MIN},
    {"mod", 2, DE_B_MOD}, {"rem", 2, DE_B_REM}, {"greater", 2, DE_B_GREATER},
    {"pow_abs2", 2, DE_B_POW_ABS2},
    {"fma", 3, DE_T_FMA}, {"clamp", 3, DE_T_CLAMP}, {"+", 3, DE_T_ADD3}, {"max", 3, DE_T_MAX3},
};

// Host-side lowering is ~3 us per tree and per pass; populations of 10^4..10^5 trees are re-created every
// generation by a search loop, so EVERY per-tree pass of de_program_create runs on a few host threads (round 5: the serial passes behind
// the lowering — merge, bind, superinstructions, record chaining — were 80 % of a creation).  One persistent pool per process (spawning
// 16 threads costs ~0.3 ms, a creation has ~10 parallel regions); one region at a time — a second context asking meanwhile runs its
// ranges inline, in the same partition, so the result never depends on who ran it.  DE_HOST_THREADS=n caps the workers (1 = serial).
namespace {
thread_local bool in_job = false;      // this thread is running a job of a parallel region
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
}
// A creation is a BURST of ~12 short regions; a thread that sleeps on a condition variable between them wakes in 50 - 90 us on these
// (shared, 256-core) hosts — the regions of a 10^3-tree creation take less.  Workers therefore spin for ~100 us after a region before
// they go to sleep, and the caller spins for the stragglers (DE_HOST_SPIN=n: iterations; 0 = sleep at once).
struct HostPool {
    std::mutex region;                 // held for the duration of a parallel region
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *job = nullptr; // (job, n_jobs: written under m before `gen` moves, read after it was seen to move)
    int n_jobs = 0;
    std::atomic<int> pending{0};
    std::atomic<uint64_t> gen{0};
    std::atomic<bool> failed{false};
    std::exception_ptr first_error;    // (under m) what the first failing job of the region threw: rethrown on the caller's thread
    std::atomic<int> sleepers{0};      // workers blocked in cv_work (the publisher only notifies when there are any)
    int n_workers = 0;
    const int spin = [] { const char *v = getenv("DE_HOST_SPIN"); const int n = v && *v ? atoi(v) : 4000; return n < 0 ? 0 : n; }();
    void worker(int id) {
        uint64_t seen = 0;
        for (;;) {
            int spins = 0;
            while (gen.load(std::memory_order_acquire) == seen) {
                if (++spins <= spin) { cpu_relax(); continue; }
                std::unique_lock<std::mutex> lk(m);
                sleepers.fetch_add(1);
                cv_work.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen; });
                sleepers.fetch_sub(1);
            }
            const std::function<void(int)> *j = nullptr;
            {
                // (the publisher moves `gen` under m after writing job / n_jobs: taking m orders this thread behind it, and a worker that
                // overslept a whole region sees n_jobs == 0 or the NEXT region's job — never a dangling one)
                const std::lock_guard<std::mutex> lk(m);
                seen = gen.load(std::memory_order_acquire);
                if (id + 1 < n_jobs) j = job;
                if (j && claimed[id] == seen) j = nullptr; // (already ran this region's job)
                if (j) claimed[id] = seen;
            }
            if (!j) continue;
            bool bad = false;
            in_job = true;
            std::exception_ptr err;
            try { (*j)(id + 1); } catch (...) { bad = true; err = std::current_exception(); }
            in_job = false;
            if (bad) {
                const std::lock_guard<std::mutex> lk(m);
                if (!first_error) first_error = err;
                failed.store(true);
            }
            if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                const std::lock_guard<std::mutex> lk(m);
                cv_done.notify_one();
            }
        }
    }
    uint64_t claimed[HOST_RANGES_MAX] = {0};
    void ensure(int want) { // (under `region`)
        while (n_workers < want && n_workers < HOST_RANGES_MAX) {
            const int id = n_workers;
            try { std::thread([this, id] { worker(id); }).detach(); } catch (...) { return; }
            n_workers++;
        }
    }
    // job(k) for k = 0 .. n - 1, job(0) on the calling thread; false = the pool is busy or could not start threads: nothing was run
    bool run(int n, const std::function<void(int)> &f) {
        if (in_job) return false; // a pass started from inside a job (never done today) runs inline instead of locking `region` twice
        std::unique_lock<std::mutex> rl(region, std::try_to_lock);
        if (!rl.owns_lock()) return false;
        ensure(n - 1);
        if (n_workers < n - 1) return false;
        {
            const std::lock_guard<std::mutex> lk(m);
            job = &f;
            n_jobs = n;
            pending.store(n - 1);
            failed.store(false);
            first_error = nullptr;
            gen.fetch_add(1, std::memory_order_release);
        }
        if (sleepers.load() > 0) cv_work.notify_all();
        std::exception_ptr mine;
        in_job = true;
        try { f(0); } catch (...) { mine = std::current_exception(); }
        in_job = false;
        for (int spins = 0; pending.load(std::memory_order_acquire) != 0 && spins < spin; spins++) cpu_relax();
        std::exception_ptr theirs;
        {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return pending.load(std::memory_order_acquire) == 0; });
            job = nullptr;
            n_jobs = 0;
            if (failed.load()) theirs = first_error;
            first_error = nullptr;
        }
        // the caller sees what was actually thrown (everything used to become std::bad_alloc = "out of host memory")
        if (mine) std::rethrow_exception(mine);
        if (theirs) std::rethrow_exception(theirs);
        return true;
    }
};
// (never destroyed: its threads are detached.  A fork()ed child has none of them: it starts with a pool of its own.)
HostPool *g_host_pool = nullptr;
HostPool &host_pool() {
    static const bool once = [] {
        g_host_pool = new HostPool();
        (void)pthread_atfork(nullptr, nullptr, [] { g_host_pool = new HostPool(); });
        return true;
    }();
    (void)once;
    return *g_host_pool;
}
unsigned host_threads_for_impl(int64_t n, int64_t grain) { // grain > 0: at least that many items per range (loops of a few ns per item)
    const unsigned hw = std::thread::hardware_concurrency();
    const char *env = getenv("DE_HOST_THREADS");
    // 10^4 / 10^5 trees on a 256-core box: 8 threads 8.3 / 88 ms, 16: 7.0 / 68, 24: 5.2 / 53, 32: 5.3 / 50 (best of 6, shared host)
    unsigned nt = env && *env ? (unsigned)atoi(env) : (hw >= 48 ? 24u : std::min(hw ? hw : 1u, 16u));
    nt = std::min(nt, (unsigned)HOST_RANGES_MAX);
    static const int64_t min_trees = [] { const char *v = getenv("DE_HOST_MIN_TREES"); const int64_t m = v && *v ? atoll(v) : 32; return m < 1 ? 1 : m; }();
    const int64_t per_range = grain > min_trees ? grain : min_trees;
    if ((int64_t)nt > n / per_range) nt = (unsigned)(n / per_range); // (a woken pool thread costs ~10 us, 32 trees are ~50 us of a pass: 10^3 trees 2.3 -> 1.4 ms against a floor of 256)
    return nt;
}
} // namespace
extern "C" {
bool host_pool_run(int n, const std::function<void(int)> &job) { return host_pool().run(n, job); }
unsigned host_threads_for(int64_t n, int64_t grain) { return host_threads_for_impl(n, grain); }
}
extern "C" {
// ---- program buffers: recycled across de_program_destroy / de_program_create (see de_ctx::recycled) ----
static constexpr size_t PROG_RECYCLE_MIN = 256u << 10, PROG_RECYCLE_BYTES = 256u << 20, PROG_RECYCLE_MAX = 12, SMALL_RECYCLE_MAX = 64;
static bool prog_recycle_enabled() {
    static const bool on = [] { const char *v = getenv("DE_NO_PROG_RECYCLE"); return !(v && *v == '1'); }();
    return on;
}
// Instruction streams must lie inside ONE 4 GiB window (the handlers bump record pointers without a carry; the early-exit walk rebuilds
// record addresses from their low 32 bits).  The property is established HERE, on the granule-rounded size, when a buffer is first
// allocated, so a recycled buffer is safe for every request it can serve (the callers used to test the requested byte count — a
// parked buffer that was fine for a smaller program could straddle for the next one, be rejected, re-parked and picked again forever).
// A fresh allocation that straddles (once in ~10^4 for a 400 KB stream) is set aside, redone, and FREED — never parked.
hipError_t prog_malloc(de_ctx *c, void **out, size_t bytes) {
    *out = nullptr;
    const bool small = bytes < PROG_RECYCLE_MIN;
    const bool pooled = prog_recycle_enabled();
    size_t need = bytes;
    if (pooled && small) { need = 512; while (need < bytes) need <<= 1; }
    else if (pooled) need = (bytes + 0xFFFFFu) & ~(size_t)0xFFFFFu;
    if (pooled && small) {
        for (size_t i = c->small_free.size(); i-- > 0;)
            if (c->small_free[i].second == need) {
                *out = c->small_free[i].first;
                c->big_live[*out] = need;
                c->small_free.erase(c->small_free.begin() + (long)i);
                return hipSuccess;
            }
    } else if (pooled) {
        int best = -1;
        for (int i = 0; i < (int)c->recycled.size(); i++) {
            const size_t sz = c->recycled[(size_t)i].second;
            if (sz >= need && sz <= 2 * need && (best < 0 || sz < c->recycled[(size_t)best].second)) best = i;
        }
        if (best >= 0) { // (every parked buffer passed the window test on its full size when it was allocated)
            *out = c->recycled[(size_t)best].first;
            c->big_live[*out] = c->recycled[(size_t)best].second;
            c->recycled.erase(c->recycled.begin() + best);
            return hipSuccess;
        }
    }
    void *rejected[4] = {nullptr, nullptr, nullptr, nullptr};
    int n_rej = 0;
    hipError_t st = hipSuccess;
    for (;;) {
        st = hipMalloc(out, need);
        if (st != hipSuccess) { *out = nullptr; break; }
        if (need > 0xFFFFFFFFull || in_one_window(*out, need)) break;
        if (n_rej == 4) { (void)hipFree(*out); *out = nullptr; st = hipErrorOutOfMemory; break; }
        rejected[n_rej++] = *out;
        *out = nullptr;
    }
    for (int k = 0; k < n_rej; k++) (void)hipFree(rejected[k]);
    if (st == hipSuccess && pooled) c->big_live[*out] = need;
    return st;
}
// (the caller has synchronised the context's stream: nothing queued reads the buffer any more)
void prog_free(de_ctx *c, void *ptr) {
    if (!ptr) return;
    auto it = c->big_live.find(ptr);
    if (it == c->big_live.end()) { (void)hipFree(ptr); return; }
    const size_t sz = it->second;
    c->big_live.erase(it);
    if (sz < PROG_RECYCLE_MIN) {
        if (c->small_free.size() >= SMALL_RECYCLE_MAX) { (void)hipFree(c->small_free.front().first); c->small_free.erase(c->small_free.begin()); }
        c->small_free.emplace_back(ptr, sz);
        return;
    }
    size_t held = 0;
    for (const auto &r : c->recycled) held += r.second;
    if (c->recycled.size() >= PROG_RECYCLE_MAX || held + sz > PROG_RECYCLE_BYTES) {
        // make room by dropping the oldest entry; a buffer larger than the whole budget is simply freed
        if (sz > PROG_RECYCLE_BYTES) { (void)hipFree(ptr); return; }
        while (!c->recycled.empty() && (c->recycled.size() >= PROG_RECYCLE_MAX || held + sz > PROG_RECYCLE_BYTES)) {
            held -= c->recycled.front().second;
            (void)hipFree(c->recycled.front().first);
            c->recycled.erase(c->recycled.begin());
        }
    }
    c->recycled.emplace_back(ptr, sz);
}

// The host vectors of a program that keep their capacity across a destroy / create pair.  A member missing here is merely allocated afresh.
#define DE_PROGRAM_VECTORS(X) \
    X(code) X(code_off) X(const_off) X(const_instr) X(const_checks) X(n_consts_tree) X(host_ok_eval) X(host_ok_grad) X(consts) X(fcode) \
    X(fcode_off) X(fconst_instr) X(folds) X(aux_const_src) X(fold_ok) X(bcode) X(tcode) X(fbcode) X(tcode_off) X(ccode) X(ccode_off) \
    X(bcode_off) X(gbcode) X(gbcode_off) X(gtcode) X(gtcode_off) X(bsite) X(tsite) X(gbsite) X(gtsite_of_gb) X(rtcode) X(rtcode_off) \
    X(rtcode_mid) X(rtsite_of_gb) X(fold_host) X(fold_nodes) X(fold_noff) X(fold_coff) X(aux_fold) X(aux_csrc) X(kfold) X(kf_csrc)
static constexpr size_t PARKED_MAX = 4, PARKED_BYTES = 512u << 20;
static size_t program_host_bytes(const de_program *p) {
    size_t b = 0;
#define X(v) b += p->v.capacity() * sizeof(p->v[0]);
    DE_PROGRAM_VECTORS(X)
#undef X
    return b;
}
// de_program_destroy's last step (device buffers are gone, `aux` is destroyed): park the shell or delete it
void park_program(de_ctx *c, de_program *p) {
    size_t held = 0;
    for (const de_program *q : c->parked) held += program_host_bytes(q);
    if (!prog_recycle_enabled() || c->parked.size() >= PARKED_MAX || held + program_host_bytes(p) > PARKED_BYTES) { delete p; return; }
    // only the LISTED vectors survive, in a fresh shell: everything else a program holds (site lists, certificate tables, gradient id
    // tables ...) is released with `p`, so a parked shell pins exactly what program_host_bytes counts
    de_program *shell = new (std::nothrow) de_program();
    if (!shell) { delete p; return; }
#define X(v) p->v.clear(); shell->v.swap(p->v);
    DE_PROGRAM_VECTORS(X)
#undef X
    delete p;
    c->parked.push_back(shell);
}
// a fresh (default-constructed) program takes over the vectors of the parked shell whose capacity is the largest
void adopt_parked(de_ctx *c, de_program *fresh) {
    if (c->parked.empty()) return;
    size_t best = 0, best_b = 0;
    for (size_t i = 0; i < c->parked.size(); i++) {
        const size_t b = program_host_bytes(c->parked[i]);
        if (b >= best_b) { best = i; best_b = b; }
    }
    de_program *old = c->parked[best];
    c->parked.erase(c->parked.begin() + (long)best);
#define X(v) fresh->v.swap(old->v);
    DE_PROGRAM_VECTORS(X)
#undef X
    delete old;
}

// A constant subtree of IEEE-exact operators, evaluated in the element type exactly as dispatch_constant_tree does
// DE_DEBUG_TIMING: microseconds since the previous lap of this thread, on stderr
void dbg_lap(const char *what) {
    static const bool on = getenv("DE_DEBUG_TIMING") != nullptr;
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    if (what) fprintf(stderr, "    [lap] %-40s %9.1f us\n", what, std::chrono::duration<double, std::micro>(now - last).count());
    last = std::chrono::steady_clock::now();
}

int de_abi_version(void) { return DE_HIP_ABI_VERSION; }
int de_opcode_table_version(void) { return DE_OPCODE_TABLE_VERSION; }

int de_opcode_by_name(const char *name, int degree) {
    if (!name) return -1;
    for (const OpName &o : kOps)
        if (o.degree == degree && std::strcmp(o.name, name) == 0) return o.code;
    return -1;
}
const char *de_opcode_name(int opcode) {
    for (const OpName &o : kOps)
        if (o.code == opcode && !(o.code == DE_U_NEG && o.name[0] == '-')) return o.name;
    return nullptr;
}
int de_opcode_degree(int opcode) {
    if (opcode >= DE_U_NEG && opcode < DE_U_LAST_) return 1;
    if (opcode >= DE_B_ADD && opcode < DE_B_LAST_) return 2;
    if (opcode >= DE_T_FMA && opcode < DE_T_LAST_) return 3;
    return -1;
}
const char *de_status_string(int s) {
    switch (s) {
    case DE_OK: return "ok";
    case DE_ERR_INVALID_ARG: return "invalid argument";
    case DE_ERR_BAD_TAPE: return "malformed tape";
    case DE_ERR_UNSUPPORTED_OP: return "unsupported operator";
    case DE_ERR_HIP: return "HIP runtime error";
    case DE_ERR_NO_DEVICE: return "no gfx950 device";
    case DE_ERR_OUT_OF_RANGE: return "index out of range";
    case DE_ERR_UNSUPPORTED: return "unsupported request";
    case DE_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
    }
}

// ---------------------------------------------------------------------------
int de_ctx_create(int device, void *stream, de_ctx_t **out_ctx) {
    if (!out_ctx) return DE_ERR_INVALID_ARG;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return DE_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return DE_ERR_INVALID_ARG;
    de_ctx *c = new (std::nothrow) de_ctx();
    if (!c) return DE_ERR_HIP;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return DE_ERR_HIP;
    }
    if (stream == DE_STREAM_NULL) {
        c->stream = nullptr; // HIP null stream
    } else if (stream) {
        c->stream = static_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return DE_ERR_HIP;
        }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return DE_ERR_HIP;
    }
    *out_ctx = c;
    return DE_OK;
}

int de_ctx_destroy(de_ctx_t *c) {
    if (!c) return DE_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->sX, &c->sOut, &c->sGrad, &c->sOk, &c->sParams, &c->sClasses, &c->sOut2, &c->sGoff, &c->sNg, &c->sY, &c->sW, &c->sLoss, &c->sPartial, &c->sSeg, &c->sDloss, &c->sColOff, &c->sDoff, &c->sPrio, &c->sPrioDs, &c->sCert, &c->sBcLoss, &c->sBcDloss, &c->sBcOk, &c->sBcNg, &c->sBcDoff, &c->sBcOut, &c->sBcTiles}) b->release();
    for (const auto &r : c->recycled) (void)hipFree(r.first);
    for (const auto &r : c->small_free) (void)hipFree(r.first);
    c->recycled.clear();
    c->small_free.clear();
    for (de_program *q : c->parked) delete q;
    c->parked.clear();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->ring) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return DE_OK;
}

// Give back what the context keeps for the NEXT program: the parked host vectors of destroyed programs (<= 4 shells / 512 MB), the recycled
// device buffers (<= 12 / 256 MB + 64 small ones) and the staging scratch of host-pointer calls.  A context that is idle for long — a
// Julia task per context multiplies the retention — calls this; the next de_program_create simply allocates afresh.
int de_ctx_trim(de_ctx_t *c) {
    if (!c) return DE_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (DevBuf *b : {&c->sX, &c->sOut, &c->sGrad, &c->sOk, &c->sParams, &c->sClasses, &c->sOut2, &c->sGoff, &c->sNg, &c->sY, &c->sW, &c->sLoss, &c->sPartial, &c->sSeg, &c->sDloss, &c->sColOff, &c->sDoff, &c->sCert, &c->sBcLoss, &c->sBcDloss, &c->sBcOk, &c->sBcNg, &c->sBcDoff, &c->sBcOut, &c->sBcTiles}) b->release();
    for (const auto &r : c->recycled) (void)hipFree(r.first);
    for (const auto &r : c->small_free) (void)hipFree(r.first);
    c->recycled.clear();
    c->small_free.clear();
    for (de_program *q : c->parked) delete q;
    c->parked.clear();
    return DE_OK;
}

int de_ctx_set_stream(de_ctx_t *c, void *stream) {
    if (!c) return DE_ERR_INVALID_ARG;
    if (!stream) return fail(c, DE_ERR_INVALID_ARG, "de_ctx_set_stream: pass a hipStream_t or DE_STREAM_NULL");
    hipStream_t ns = stream == DE_STREAM_NULL ? nullptr : static_cast<hipStream_t>(stream);
    if (ns == c->stream && !c->own_stream) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // the context's scratch buffers may still be in use by work queued on the old stream: order the new stream behind it
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(ns, c->ev1, 0));
    if (c->own_stream && c->stream) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        (void)hipStreamDestroy(c->stream);
        c->own_stream = false;
    }
    c->stream = ns;
    c->timed = false;
    return DE_OK;
}

int de_ctx_synchronize(de_ctx_t *c) {
    if (!c) return DE_ERR_INVALID_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DE_OK;
}
void *de_ctx_stream(de_ctx_t *c) { return c ? c->stream : nullptr; }
int de_ctx_device(de_ctx_t *c) { return c ? c->device : -1; }
// A dataset that stays as it is between calls (X of a symbolic-regression search: thousands of de_eval* calls on the same matrix):
// statistics of it — today the 3 F priority-tile keys, a pass over X of 0.11 ms at 10^7 samples — are computed here, once, and every
// later call on this context whose (X, N, ldX) are the declared ones skips its own pass.  X == NULL withdraws the declaration.
int de_ctx_declare_dataset(de_ctx_t *c, int dtype, const void *X, int64_t N, int64_t ldX, int32_t n_features) {
    if (!c) return DE_ERR_INVALID_ARG;
    c->ds_X = nullptr;
    if (!X) return DE_OK;
    if ((dtype != DE_F32 && dtype != DE_F64) || N < 1 || n_features < 1 || ldX < n_features) return fail(c, DE_ERR_INVALID_ARG, "bad dataset shape");
    if (!is_device_ptr(X)) return fail(c, DE_ERR_INVALID_ARG, "de_ctx_declare_dataset needs a device-resident X (host buffers are staged on every call)");
    if (n_features > DE_PRIO_MAX_F) return DE_OK; // nothing to cache for such an X (no priority tiles): not an error
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->sPrioDs.reserve((size_t)3 * DE_PRIO_MAX_F * sizeof(unsigned long long)));
    // (the pass runs on the context's CURRENT stream; a later de_ctx_set_stream orders the new stream behind everything queued on this one)
    HIP_TRY(c, launch_tile_extremes(dtype, X, N, ldX, n_features, c->sPrioDs.p, c->stream));
    c->ds_X = X;
    c->ds_N = N;
    c->ds_ldX = ldX;
    c->ds_F = n_features;
    c->ds_dtype = dtype;
    return DE_OK;
}
bool dataset_keys(const de_ctx *c, int dtype, const void *X, int64_t N, int64_t ldX, int32_t F, void **keys) {
    if (!c->ds_X || c->ds_X != X || c->ds_N != N || c->ds_ldX != ldX || c->ds_F != F || c->ds_dtype != dtype) return false;
    *keys = c->sPrioDs.p;
    return true;
}

const char *de_last_error(de_ctx_t *c) { return c ? c->err.c_str() : "null context"; }

int de_ctx_last_kernel_ms(de_ctx_t *c, float *ms) {
    if (!c || !ms) return DE_ERR_INVALID_ARG;
    if (!c->timed) return fail(c, DE_ERR_INVALID_ARG, "no timed launch on this context yet");
    hipEvent_t e0 = c->ev0, e1 = c->ev1;
    if (!c->ring.empty()) {
        if (c->ring_at == 0) return fail(c, DE_ERR_INVALID_ARG, "no timed launch since de_ctx_timing_ring");
        const size_t k = (size_t)((c->ring_at - 1) % (c->ring.size() / 2));
        e0 = c->ring[2 * k];
        e1 = c->ring[2 * k + 1];
    }
    HIP_TRY(c, hipEventSynchronize(e1));
    HIP_TRY(c, hipEventElapsedTime(ms, e0, e1));
    return DE_OK;
}
// n > 0: keep the event pairs of the last n timed calls (de_eval*, one pair per call); n == 0: back to the single pair.
int de_ctx_timing_ring(de_ctx_t *c, int32_t n) {
    if (!c || n < 0 || n > 65536) return DE_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (hipEvent_t e : c->ring) (void)hipEventDestroy(e);
    c->ring.clear();
    c->ring_at = 0;
    c->timed = false;
    for (int32_t i = 0; i < 2 * n; i++) {
        hipEvent_t e = nullptr;
        HIP_TRY(c, hipEventCreate(&e));
        c->ring.push_back(e);
    }
    return DE_OK;
}
// Device time (ms) of the timed calls since de_ctx_timing_ring / the last read, oldest first, at most `cap` and at most the ring's size
// (older ones are overwritten); waits for the last one.  *n_out = how many were written.  The ring restarts.
int de_ctx_timing_read(de_ctx_t *c, float *ms, int32_t cap, int32_t *n_out) {
    if (!c || !ms || !n_out || cap < 0) return DE_ERR_INVALID_ARG;
    *n_out = 0;
    if (c->ring.empty()) return fail(c, DE_ERR_INVALID_ARG, "de_ctx_timing_read without de_ctx_timing_ring");
    const uint64_t slots = c->ring.size() / 2, have = c->ring_at < slots ? c->ring_at : slots;
    const uint64_t take = have < (uint64_t)cap ? have : (uint64_t)cap;
    for (uint64_t i = 0; i < take; i++) {
        const size_t k = (size_t)((c->ring_at - take + i) % slots);
        HIP_TRY(c, hipEventSynchronize(c->ring[2 * k + 1]));
        HIP_TRY(c, hipEventElapsedTime(ms + i, c->ring[2 * k], c->ring[2 * k + 1]));
    }
    *n_out = (int32_t)take;
    c->ring_at = 0;
    return DE_OK;
}
const char *de_ctx_last_kernel_name(de_ctx_t *c) { return c ? c->last_kernel : ""; }

int64_t de_host_pool_selftest(int64_t n, int32_t *n_ranges) {
    if (n < 0) return -1;
    std::vector<uint8_t> hit((size_t)n, 0);
    std::atomic<int32_t> ranges{0};
    try {
        parallel_tree_ranges(n, [&](int, int64_t b, int64_t e) {
            ranges.fetch_add(1);
            for (int64_t i = b; i < e; i++) hit[(size_t)i]++;
        });
    } catch (...) { return -1; }
    if (n_ranges) *n_ranges = ranges.load();
    int64_t once = 0;
    for (uint8_t h : hit) once += h == 1;
    return once;
}

// Test hook: one number over every HOST-side stream and table de_program_create built (generic, folded, bound, fused, threaded and chained
// records, offsets, constant sites, fold list, flags, and the auxiliary program's) — the per-tree passes run on a pool of host threads
// and must build what one thread builds (tests/test_gpu_round5.py: DE_HOST_THREADS=1 against the default).  FNV-1a, 64 bits.
} // extern "C"
