// de_api.cpp — the C ABI of libde_hip.so (include/de_hip.h): contexts, population
// programs, evaluation entry points.  No exception leaves this file.
#include <hip/hip_runtime.h>
#include <pthread.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <memory>
#include <new>
#include <string>
#include <algorithm>
#include <array>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/de_hip.h"
#include "de_kernels.h"
#include "de_lower.h"

using namespace de;

// ---------------------------------------------------------------------------
struct DevBuf { // grow-only device scratch
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t st = hipMalloc(&p, n);
        if (st == hipSuccess) cap = n;
        return st;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct de_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // de_ctx_timing_ring: pairs of events for the last `ring.size() / 2` timed calls, so that a caller can read the device time of EVERY
    // call of a free-running loop afterwards (de_ctx_last_kernel_ms blocks until the call is done)
    std::vector<hipEvent_t> ring;
    uint64_t ring_at = 0;
    std::string err;
    const char *last_kernel = "";
    // Device buffers of destroyed programs, recycled (round 5): a search loop creates and destroys a program per generation, and hipFree
    // of a multi-megabyte buffer takes 0.3 - 0.5 ms (10^4 trees: de_program_destroy 1.9 ms of a 10 ms generation).  Instruction streams
    // of >= PROG_RECYCLE_MIN bytes are allocated in 1 MiB granules through prog_malloc and parked here by prog_free; at most
    // PROG_RECYCLE_MAX of them / PROG_RECYCLE_BYTES in total, the rest is freed.  DE_NO_PROG_RECYCLE=1: plain hipMalloc / hipFree.
    std::vector<std::pair<void *, size_t>> recycled;
    // ... and the SMALL ones (round 6): a one-tree program (de_eval_tree_array: the reference's own call shape) is a few hundred bytes, and
    // its hipMalloc / hipFree pairs were a third of the call.  Power-of-two size classes from 512 B up to PROG_RECYCLE_MIN, at most
    // SMALL_RECYCLE_MAX buffers parked per context.
    std::vector<std::pair<void *, size_t>> small_free;
    std::map<void *, size_t> big_live; // granule-sized allocations in use (their sizes)
    // ... and the HOST side of destroyed programs: `delete` of a 10^4-tree program is 1.5 ms of munmap (its ~40 vectors are tens of
    // megabytes), and the next creation faults the same pages in again.  Up to four destroyed programs are parked with their vectors
    // cleared; a creation takes the vectors' capacity over (park_program / adopt_parked).
    std::vector<struct de_program *> parked;
    DevBuf sX, sOut, sGrad, sOk, sParams, sClasses, sOut2, sGoff, sNg, sY, sW, sLoss, sPartial, sSeg, sDloss, sColOff, sDoff, sPrio;
    DevBuf sCert; // de_eval_sum_certificate: per-tree maxima
    DevBuf sBcLoss, sBcDloss, sBcOk, sBcNg, sBcDoff, sBcOut, sBcTiles; // de_eval_loss_grad_by_class
    int nested = 0; // > 0 inside a call made of several inner calls: those do not touch the timing events
    // de_ctx_declare_dataset: a device-resident X the caller promises not to modify — its priority-tile keys are computed once
    const void *ds_X = nullptr;
    int64_t ds_N = 0, ds_ldX = 0;
    int32_t ds_F = 0;
    int ds_dtype = -1;
    DevBuf sPrioDs;
};

struct de_program {
    de_ctx *ctx = nullptr;
    int dtype = DE_F32;
    uint32_t options = 0;
    int32_t n_features = 0, n_params = 0;
    int64_t n_trees = 0, n_nodes = 0;
    int n_slots = 0;
    bool prows = false; // eval kernels: the parameters are staged per tile as LDS rows F + n_slots + p (rebind), operands like features; false: BOP_GEN_PARAM gathers
    bool uses_params = false;
    std::vector<Instr> code;            // host copy (patched by set_consts)
    std::vector<int32_t> code_off;      // n_trees + 1
    std::vector<int64_t> const_off;     // n_trees + 1
    std::vector<int32_t> const_instr;   // per const (global index): global instr index
    std::vector<uint8_t> const_checks;  // per const: CONST_CHECK_* bits
    std::vector<int32_t> n_consts_tree; // per tree
    bool cse_generic = false;           // some tree's GENERIC (gradient) program is the CSE lowering: a persistent row has several consumers (no reverse accumulation)
    std::vector<uint8_t> host_ok_eval;  // per tree: constant part of the eval flag
    std::vector<uint8_t> host_ok_grad;  // per tree: all constants finite
    std::vector<double> consts;         // current constants as double
    // Constant folding on the device (LowerOptions.fold): the eval path runs `fcode`, in which every
    // maximal constant subtree is one constant operand; the subtrees themselves form the `aux`
    // population, evaluated once per constant update by the same kernels (N = 1).
    bool folded = false;
    std::vector<Instr> fcode;
    std::vector<int32_t> fcode_off;
    std::vector<int32_t> fconst_instr;  // per constant: index into fcode, or < 0 if folded away
    struct Fold { int32_t tree, instr; bool tested_always; };
    std::vector<Fold> folds;            // aux tree j -> (owning tree, fcode instruction holding its value)
    std::vector<int64_t> aux_const_src; // constant k of the fold spans (all folds, concatenated) = consts[aux_const_src[k]]
    de_program *aux = nullptr;          // the folds that are evaluated ON THE DEVICE (fold_host[j] == 0), as a population of their own
    std::vector<uint8_t> fold_ok;
    // Round 6: a constant subtree made of IEEE-exact operators only (+ - * /) is folded ON THE HOST — the same bits by construction
    // (the device's + - * / are correctly rounded, tests/test_gpu_eval.py::test_ieee_exact_operators_are_bit_identical, and every
    // translation unit is built with -ffp-contract=off) — and never enters the auxiliary program: about half of the constant subtrees of the
    // benchmark's operator set.  fold_nodes / fold_noff / fold_coff: every fold's tape slice (constant leaves numbered from the span's first
    // slot) and its range in aux_const_src; aux_fold: auxiliary tree -> fold; aux_csrc: the auxiliary program's constants -> consts.
    std::vector<uint8_t> fold_host;     // per fold: 1 = folded on the host, 2 = by de_fold_kernel (one thread per subtree), 0 = through `aux`
    // the subtrees de_fold_kernel evaluates: kfold[k] = fold index; device image [tape slices | node offsets | constant offsets | constant
    // values | values out | flags out] in ONE pooled allocation (uploaded once; the constant values again at every de_program_set_consts)
    std::vector<int32_t> kfold;
    std::vector<int64_t> kf_csrc;       // constant k of the kernel folds = consts[kf_csrc[k]]
    char *d_kf = nullptr;
    size_t kf_o_noff = 0, kf_o_coff = 0, kf_o_cvals = 0, kf_o_out = 0, kf_o_ok = 0, kf_bytes = 0;
    std::vector<de_tape_node_t> fold_nodes;
    std::vector<int64_t> fold_noff, fold_coff;
    std::vector<int32_t> aux_fold;
    std::vector<int64_t> aux_csrc;
    std::vector<BoundInstr> bcode;      // bound form of the eval program (handler ids)
    std::vector<BoundInstr> tcode;      // threaded form: handler address offsets + LDS byte offsets
    std::vector<BoundInstr> fbcode;     // fused (superinstruction) form the threaded code is made from
    std::vector<int32_t> tcode_off;     // n_trees + 1 offsets into tcode / fbcode
    // what the threaded kernel reads (de_kernels.hip "direct-threaded dispatch"): one 16-byte record per instruction
    // {operand word, immediate, address of its handler} and an end record per tree; made from tcode
    std::vector<BoundInstr> ccode;
    std::vector<int32_t> ccode_off;     // n_trees + 1: first record of each tree
    uint64_t end_handler = 0;
    uint64_t endv_handler[TOPX_ENDV_COUNT] = {0}; // "last instruction + end of tree" variants (de_bind.h topx_endv_of)
    bool threaded = false;
    bool direct = false;                // X too wide for the LDS tile (decided at creation)
    uint64_t handler_base = 0;
    std::vector<int32_t> bcode_off;     // n_trees + 1
    BoundInstr *d_code = nullptr;
    int32_t *d_code_off = nullptr;
    bool eval_arena = false;            // d_code_off / d_compact_ints / d_ok_eval point into d_code's allocation
    // compaction of the live trees (de_kernels.hip de_compact_live_kernel): the second half of the d_code allocation (same 4 GiB window) and
    // (n_trees + 1) + n_trees + 4 ints; null when the program is not threaded
    BoundInstr *d_compact_code = nullptr;
    int32_t *d_compact_ints = nullptr;
    bool last_compacted = false; // the most recent eval launch compacted its live trees (de_program_last_live_trees)
    // de_eval_sum_certificate: the eval program with EVERY operator result validity-tested (no exact elision), bound for the flat-switch kernel
    BoundInstr *d_cert_code = nullptr;
    int32_t *d_cert_off = nullptr;
    size_t cert_cap = 0;
    uint64_t consts_gen = 0;       // bumped by every de_program_set_consts
    uint64_t cert_gen = ~0ull;     // consts_gen the uploaded certificate program was built for (~0: none)
    std::vector<double> cert_cmax; // per tree: the largest |constant operand| (an array of N copies of it is summed by the reference)
    BoundInstr *d_gcode = nullptr;      // bound UNFOLDED program on the device (gradient kernels), lazily uploaded
    int32_t *d_gcode_off = nullptr;
    std::vector<BoundInstr> gbcode;
    std::vector<int32_t> gbcode_off;
    bool gcode_stale = true;
    // threaded form of the gradient program for one (mode, window width): de_grad_threaded.hip
    std::vector<BoundInstr> gtcode;
    std::vector<int32_t> gtcode_off;
    BoundInstr *d_gtcode = nullptr;
    int32_t *d_gtcode_off = nullptr;
    int32_t *d_gt_ids = nullptr;        // tree indices grouped by bucket
    uint8_t *d_ok_eval = nullptr;       // device copy of host_ok_eval (initial value of the flags of every eval call)
    // device-resident per-call tables of de_eval_grad, so that a call copies nothing from pageable host memory and never
    // blocks the stream: initial flags, gradient widths of the last mode and packed offsets of the last (mode, N)
    uint8_t *d_ok_grad = nullptr;
    int32_t *d_ng = nullptr;
    int64_t *d_goff = nullptr;
    int tab_mode = -1;
    int64_t tab_N = -1;
    bool tab_ok_stale = true;
    // immediate sites (set_consts patches constants in place): for a generic instruction with a constant
    // operand, the index of the instruction carrying its bits in bcode / tcode (eval source program) and in
    // gbcode / gtcode (unfolded program); -1 elsewhere.  Empty = not available (full rebuild instead).
    std::vector<int32_t> bsite, tsite, gbsite, gtsite_of_gb;
    // compact forms for de_program_set_consts (rebuilt when site_gen moves): only the instructions that carry an immediate
    struct EvalSite { int32_t src, b, t, c; };       // source instruction (fcode/code), bcode index, tcode index, ccode index
    struct GradSite { int32_t src, gb, gt, rt; };   // code instruction, gbcode index, gtcode / rtcode index or -1
    std::vector<EvalSite> eval_sites;
    std::vector<GradSite> grad_sites;
    uint64_t site_gen = 1, lists_gen = 0;
    // reverse-accumulation form (de_rev_threaded.hip) for one gradient mode
    std::vector<BoundInstr> rtcode;
    std::vector<int32_t> rtcode_off, rtcode_mid, rtsite_of_gb;
    BoundInstr *d_rtcode = nullptr;
    int32_t *d_rtcode_off = nullptr, *d_rtcode_mid = nullptr, *d_rt_ids = nullptr;
    int rt_mode = -1, rt_stage_cols = 0, rt_n_groups = 0;
    GradArgs::RevGroup rt_groups[8];
    bool rt_valid = false;
    uint64_t rt_handler_base = 0;
    uint32_t rt_param_off = 0;
    int gt_mode = -1;
    bool gt_valid = false, gt_wide = false;
    int gt_n_buckets = 0;
    GradArgs::Bucket gt_buckets[24];
};

static int fail(de_ctx *c, int code, const char *fmt, ...) {
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
#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t st_ = (expr);                                                             \
        if (st_ != hipSuccess) {                                                             \
            (void)hipGetLastError();                                                         \
            return fail((ctx), DE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(st_)); \
        }                                                                                    \
    } while (0)

// the pair of events that brackets the launches of a call: the context's own pair, or the next slot of the timing ring
static hipError_t time_begin(de_ctx *c) {
    return hipEventRecord(c->ring.empty() ? c->ev0 : c->ring[(size_t)(c->ring_at % (c->ring.size() / 2)) * 2], c->stream);
}
static hipError_t time_end(de_ctx *c) {
    hipEvent_t e = c->ev1;
    if (!c->ring.empty()) { e = c->ring[(size_t)(c->ring_at % (c->ring.size() / 2)) * 2 + 1]; c->ring_at++; }
    c->timed = true;
    return hipEventRecord(e, c->stream);
}
static bool is_device_ptr(const void *p) {
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
constexpr int HOST_RANGES_MAX = 32; // ranges of one parallel pass (per-worker vectors are arrays of this size)
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
unsigned host_threads_for(int64_t n, int64_t grain = 0) { // grain > 0: at least that many items per range (loops of a few ns per item)
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

// The trees in contiguous ranges, one per worker: f(k, b, e) with k < HOST_RANGES_MAX — for passes that append to a per-worker vector which is
// concatenated afterwards, or that write disjoint slices of pre-sized vectors.  The partition depends on n and the thread count only.
template <class F> static void parallel_tree_ranges(int64_t n, F f, int64_t grain = 0) {
    const unsigned nt = host_threads_for(n, grain);
    if (nt <= 1) {
        f(0, (int64_t)0, n);
        return;
    }
    const int64_t per = (n + nt - 1) / nt;
    const int n_ranges = (int)((n + per - 1) / per);
    const std::function<void(int)> job = [&](int k) {
        const int64_t b = (int64_t)k * per, e = std::min<int64_t>(n, b + per);
        if (b < e) f(k, b, e);
    };
    if (!host_pool().run(n_ranges, job))
        for (int k = 0; k < n_ranges; k++) job(k);
}
template <class F> static void parallel_for_trees(int64_t n, F f, int64_t grain = 0) {
    parallel_tree_ranges(n, [&](int, int64_t b, int64_t e) { for (int64_t i = b; i < e; i++) f(i); }, grain);
}

// Pair the constant-carrying instructions of a generic program with those of a derived (bound / fused)
// stream, tree by tree, in program order.  Returns false if the counts disagree (never expected).
template <class Derived, class Pred>
static bool match_const_sites(const std::vector<Instr> &src, const std::vector<int32_t> &src_off, const std::vector<Derived> &dst,
                              const std::vector<int32_t> &dst_off, int64_t n_trees, Pred carries, std::vector<int32_t> *site) {
    site->assign(src.size(), -1);
    std::atomic<bool> ok{true};
    parallel_tree_ranges(n_trees, [&](int, int64_t tb, int64_t te) { // a tree writes its own instructions' entries only
        for (int64_t t = tb; t < te && ok; t++) {
            int32_t j = dst_off[(size_t)t];
            const int32_t j1 = dst_off[(size_t)t + 1];
            for (int32_t i = src_off[(size_t)t]; i < src_off[(size_t)t + 1]; i++) {
                if (((src[(size_t)i].hdr >> H_SRC_SHIFT) & H_SRC_MASK) != SRC_CONST) continue;
                while (j < j1 && !carries(dst[(size_t)j])) j++;
                if (j >= j1) { ok = false; break; }
                (*site)[(size_t)i] = j++;
            }
            while (j < j1 && !carries(dst[(size_t)j])) j++;
            if (j != j1) ok = false;
        }
    });
    if (!ok) { site->clear(); return false; }
    return true;
}

// A per-tree pass that APPENDS records: every worker fills a vector of its own over its range of trees (emit(t, &out)), the pieces are
// concatenated in tree order and off[t] .. off[t + 1] names tree t's records — the stream a serial loop over the trees would have built.
template <class Rec, class Emit>
static void build_stream_by_trees(int64_t n_trees, std::vector<Rec> *stream, std::vector<int32_t> *off, Emit emit) {
    std::vector<Rec> parts[HOST_RANGES_MAX];
    int64_t first[HOST_RANGES_MAX], last[HOST_RANGES_MAX];
    for (int k = 0; k < HOST_RANGES_MAX; k++) first[k] = last[k] = 0;
    std::vector<int32_t> cnt((size_t)n_trees, 0);
    parallel_tree_ranges(n_trees, [&](int k, int64_t tb, int64_t te) {
        std::vector<Rec> &out = parts[k];
        first[k] = tb;
        last[k] = te;
        for (int64_t t = tb; t < te; t++) {
            const size_t before = out.size();
            emit(t, &out);
            cnt[(size_t)t] = (int32_t)(out.size() - before);
        }
    });
    off->assign((size_t)n_trees + 1, 0);
    for (int64_t t = 0; t < n_trees; t++) (*off)[(size_t)t + 1] = (*off)[(size_t)t] + cnt[(size_t)t];
    stream->clear();
    stream->resize((size_t)(*off)[(size_t)n_trees]);
    for (int k = 0; k < HOST_RANGES_MAX; k++) // (a few MB: memcpy-bound, kept serial)
        if (last[k] > first[k] && !parts[k].empty())
            std::memcpy(static_cast<void *>(stream->data() + (*off)[(size_t)first[k]]), parts[k].data(), parts[k].size() * sizeof(Rec));
}

static void dbg_lap(const char *what);
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
static inline bool in_one_window(const void *ptr, size_t bytes) {
    const uint64_t a0 = (uint64_t)(uintptr_t)ptr;
    return bytes == 0 || (a0 >> 32) == ((a0 + bytes - 1) >> 32);
}
static hipError_t prog_malloc(de_ctx *c, void **out, size_t bytes) {
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
static void prog_free(de_ctx *c, void *ptr) {
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
static void park_program(de_ctx *c, de_program *p) {
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
static void adopt_parked(de_ctx *c, de_program *fresh) {
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
// (src/Evaluate.jl:1002-1067: every node's output is validity-tested; the arithmetic goes on, IEEE propagates what it must).
template <typename T>
static bool host_fold_eval(const de_tape_node_t *nd, int64_t n, const double *consts, const int64_t *csrc, T *value) {
    T stack_small[32];
    std::vector<T> stack_big;
    T *st = stack_small;
    if (n > 32) { stack_big.resize((size_t)n); st = stack_big.data(); }
    int sp = 0;
    bool ok = true;
    for (int64_t i = 0; i < n; i++) {
        T v;
        if (nd[i].degree == 0) v = (T)consts[csrc[nd[i].arg]];
        else {
            const T b = st[--sp], a = st[--sp];
            switch (nd[i].op) {
            case DE_B_ADD: v = a + b; break;
            case DE_B_SUB: v = a - b; break;
            case DE_B_MUL: v = a * b; break;
            default: v = a / b; break; // DE_B_DIV (host_foldable admits nothing else)
            }
        }
        ok = ok && std::isfinite(v);
        st[sp++] = v;
    }
    *value = st[0];
    return ok;
}
static bool host_foldable(const de_tape_node_t *nd, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        if (nd[i].degree == 0) { if (nd[i].op != DE_LEAF_CONST) return false; }
        else if (nd[i].degree != 2 || nd[i].op < DE_B_ADD || nd[i].op > DE_B_DIV) return false;
    }
    return n > 0;
}

// No exception leaves this file: the gradient entry points build host vectors (and run passes on the host pool, which reports a worker's
// failure as std::bad_alloc) — an allocation failure becomes a status like everywhere else.
#define DE_NOTHROW(CTX, CALL)                                                                   \
    do {                                                                                        \
        try { return (CALL); }                                                                  \
        catch (const std::bad_alloc &) { return fail((CTX), DE_ERR_HIP, "out of host memory"); } \
        catch (const std::exception &e) { return fail((CTX), DE_ERR_HIP, "internal error: %s", e.what()); } \
        catch (...) { return fail((CTX), DE_ERR_HIP, "internal error (unknown exception)"); }  \
    } while (0)

extern "C" {

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
    for (de_program *q : c->parked) delete q;
    c->parked.clear();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t e : c->ring) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
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
static bool dataset_keys(const de_ctx *c, int dtype, const void *X, int64_t N, int64_t ldX, int32_t F, void **keys) {
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

// ---------------------------------------------------------------------------
static void write_imm(Instr &ins, int dtype, double v) {
    if (dtype == DE_F32) {
        ins.imm.u32[1] = 0;
        ins.imm.f32 = (float)v;
    } else ins.imm.f64 = v;
}
// Early exit at tree granularity (kernels: a workgroup does not evaluate the trees whose flag is already 0).  DE_NO_TREE_SKIP=1
// restores the evaluate-everything behaviour for A/B measurements (the option bit DE_OPT_FULL_EVAL does the same per program).
static bool tree_skip_enabled() {
    static const bool on = [] { const char *v = getenv("DE_NO_TREE_SKIP"); return !(v && *v == '1'); }();
    return on;
}
static bool finite_in(int dtype, double v) { return dtype == DE_F32 ? std::isfinite((float)v) : std::isfinite(v); }

// Parameters as staged rows (round 3): every use of a parameter was a gather of its samples' values through the vector cache (h_param:
// 4 loads per lane and use; per-sample parameters, C = N, ran at 27 % VALU utilisation).  With <= 16 parameters the eval kernels
// instead stage the tile's parameter values once per workgroup, transposed like X, into P more LDS rows behind the spill slots and
// the binder treats a parameter operand as a row operand (every fused form applies).  DE_NO_PARAM_ROWS=1: the gathers.
static bool param_rows_enabled() { // (read at every de_program_create: the tests switch it inside one process)
    const char *v = getenv("DE_NO_PARAM_ROWS");
    return !(v && *v == '1');
}
static int64_t eval_rows(const de_program *p) { return (int64_t)p->n_features + p->n_slots + (p->prows ? p->n_params : 0); }
static void rebind(de_program *p) {
    const bool ee = (p->options & DE_OPT_EARLY_EXIT) != 0;
    p->prows = p->uses_params && p->n_params > 0 && p->n_params <= 16 && param_rows_enabled() &&
               ((size_t)p->n_features + (size_t)p->n_slots + (size_t)p->n_params) * 257 * 16 <= 150 * 1024;
    const int prb = p->prows ? p->n_features + p->n_slots : -1;
    const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
    const std::vector<int32_t> &off = p->folded ? p->fcode_off : p->code_off;
    const int nf = p->n_features;
    build_stream_by_trees<BoundInstr>(p->n_trees, &p->bcode, &p->bcode_off, [&](int64_t t, std::vector<BoundInstr> *out) {
        const int32_t i0 = off[(size_t)t], i1 = off[(size_t)t + 1];
        bind_tree(src.data() + i0, (size_t)(i1 - i0), ee, nf, out, prb);
    });
    match_const_sites(src, off, p->bcode, p->bcode_off, p->n_trees, [](const BoundInstr &b) { return bop_is_const_source(b.bop); }, &p->bsite);
    p->tsite.clear();
    p->site_gen++;
}
// (bind_tree / fuse_tree are ~0.3 us per tree — 3 ms each for 10^4 trees on one thread: build_stream_by_trees)

// Threaded-code form of the bound program (de_kernels.hip, de_eval_threaded_kernel): word 0 =
// handler address - handler_base, word 1 = LDS byte offset of the operand row | aux << 24.
// DE_DEBUG_TIMING: microseconds since the previous lap of this thread, on stderr
static void dbg_lap(const char *what);
static void dbg_lap(const char *what) {
    static const bool on = getenv("DE_DEBUG_TIMING") != nullptr;
    if (!on) return;
    static thread_local std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    if (what) fprintf(stderr, "    [lap] %-40s %9.1f us\n", what, std::chrono::duration<double, std::micro>(now - last).count());
    last = std::chrono::steady_clock::now();
}
static void make_chained(de_program *p);
static int make_threaded(de_ctx *c, de_program *p) {
    p->threaded = false;
    dbg_lap(nullptr);
    // the LDS-staged kernels need (n_features + n_slots) rows of 4112 B; wider X uses the direct variant
    p->direct = (size_t)eval_rows(p) * 257 * 16 > 150 * 1024; // (the flat-switch geometry decides)
    if (p->direct || !eval_uses_threaded()) return DE_OK;
    if (eval_rows(p) > 4000) return DE_OK; // row offsets must fit 24 bits
    uint64_t table[TOPX_TABLE];
    hipError_t st = eval_handler_table(p->dtype, (p->options & DE_OPT_TURBO) != 0, table);
    if (st != hipSuccess) return fail(c, DE_ERR_HIP, "handler table: %s", hipGetErrorString(st));
    dbg_lap("handler table");
    uint64_t base = table[0];
    for (int i = 0; i < (int)TOPX_TABLE; i++) base = std::min<uint64_t>(base, table[i]);
    for (int i = 0; i < (int)TOPX_TABLE; i++) {
        if (table[i] - base > 0xFFFFFFFFull) return DE_OK; // cannot encode: keep the switch kernel
        // Float64 records carry 32 bits of the next handler's address (the high half is the current pc's)
        if (p->dtype != DE_F32 && (table[i] >> 32) != (table[0] >> 32)) return DE_OK;
    }
    const bool hot_unary = !getenv("DE_NO_CONST_UNARY_HOT"); // unary operators outside the binder's hot set: their own handlers
    const uint32_t row_bytes = (uint32_t)trow_bytes(p->dtype);
    // superinstructions (de_bind.h): fewer dispatches for the same arithmetic
    const char *nf = getenv("DE_NO_FUSE");
    const bool fuse = !(nf && *nf == '1');
    build_stream_by_trees<BoundInstr>(p->n_trees, &p->fbcode, &p->tcode_off, [&](int64_t t, std::vector<BoundInstr> *out) {
        const int32_t b0 = p->bcode_off[(size_t)t], b1 = p->bcode_off[(size_t)t + 1];
        if (fuse) fuse_tree(p->bcode.data() + b0, (size_t)(b1 - b0), out);
        else out->insert(out->end(), p->bcode.begin() + b0, p->bcode.begin() + b1);
    });
    dbg_lap("fuse_tree");
    p->tcode.resize(p->fbcode.size());
    parallel_tree_ranges(p->n_trees, [&](int, int64_t tb, int64_t te) {
    for (size_t i = (size_t)p->tcode_off[(size_t)tb]; i < (size_t)p->tcode_off[(size_t)te]; i++) {
        const BoundInstr &b = p->fbcode[i];
        BoundInstr t = b;
        t.bop = (uint32_t)(table[b.bop] - base);
        if (hot_unary && (b.bop == BOP_GEN_ROW || b.bop == BOP_GEN_ACC)) {
            const int k = gun_index((int)(b.arg >> 24), DE_U_COS, DE_U_EXP, DE_U_SIN, DE_U_NEG, DE_U_SQUARE, DE_U_CUBE, DE_U_ABS, DE_U_LOG, DE_U_SAFE_LOG,
                                    DE_U_SQRT, DE_U_SAFE_SQRT, DE_U_TANH, DE_U_RELU);
            if (k >= 3) t.bop = (uint32_t)(table[TOPX_UN_BASE + (uint32_t)(k - 3) * 2 + (b.bop == BOP_GEN_ACC ? 1 : 0)] - base);
        }
        if (hot_unary && (b.bop == BOP_GEN_ROW || b.bop == BOP_GEN_CONST) && ((b.arg >> 24) == (uint32_t)DE_B_MAX || (b.arg >> 24) == (uint32_t)DE_B_MIN))
            t.bop = (uint32_t)(table[TOPX_BIN_BASE + ((b.arg >> 24) == (uint32_t)DE_B_MAX ? 0u : 2u) + (b.bop == BOP_GEN_CONST ? 1u : 0u)] - base);
        if (b.bop < BOP_COUNT && bop_is_const_source(b.bop)) {
            t.arg = b.arg & 0xFF000000u; // constant ordinal is only for the gradient kernel
        } else if (b.bop == BOP_GEN_PARAM) { // immediate = LDS byte offset of the class row (behind X and the spill slots)
            t.lo = (uint32_t)(p->n_features + p->n_slots) * row_bytes;
            t.hi = 0;
        } else {
            const uint32_t row = b.arg & 0xFFFFFFu, aux = b.arg >> 24;
            t.arg = (row * row_bytes) | (aux << 24);
            if (b.bop == BOP_TERN) t.lo = (b.lo - row) * row_bytes; // byte distance row B -> row C (mod 2^32)
            if (b.bop >= TOP_BIN2_BASE && b.bop < TOP_COUNT && !(((b.bop - TOP_BIN2_BASE) >> 2) & 1))
                t.lo = (uint32_t)((int32_t)b.lo * (int32_t)row_bytes); // row-row: byte distance row A -> row B
        }
        p->tcode[i] = t;
    }
    });
    dbg_lap("threaded words");
    {
        const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
        const std::vector<int32_t> &off = p->folded ? p->fcode_off : p->code_off;
        match_const_sites(src, off, p->fbcode, p->tcode_off, p->n_trees, [](const BoundInstr &b) { return top_carries_const(b.bop); }, &p->tsite);
        p->site_gen++;
    }
    dbg_lap("constant sites");
    p->handler_base = base;
    p->end_handler = table[TOPX_END];
    for (uint32_t k = 0; k < TOPX_ENDV_COUNT; k++) p->endv_handler[k] = table[TOPX_ENDV_BASE + k];
    make_chained(p);
    dbg_lap("chained records");
    p->threaded = true;
    return DE_OK;
}

// The device layout of the threaded program (see de_kernels.hip): one head record, then per tree one record per instruction
// and an end record.  A record = {its operand word, its immediate, the address of the NEXT record's handler}: the head record
// names the first handler of tree 0, a tree's last instruction names h_tree_end (whose operand word is the tree's index), an
// end record names the first handler of the next tree — a chunk of consecutive trees is ONE chain and a handler knows where
// to jump before the record it has to fetch arrives.  BoundInstr fields by word: Float32 {bop: operand word, arg: imm, lo/hi:
// next handler}; Float64 {bop: operand word, arg: next handler (low half), lo/hi: imm}.
static void make_chained(de_program *p) {
    const bool f32 = p->dtype == DE_F32;
    p->ccode.assign(p->tcode.size() + (size_t)p->n_trees + 1, BoundInstr{0u, 0u, 0u, 0u});
    p->ccode_off.assign((size_t)p->n_trees + 1, 0);
    auto put = [&](BoundInstr &r, uint32_t la, uint32_t lo, uint32_t hi) { // operand words; the handler word is set by the predecessor
        r.bop = la;
        if (f32) r.arg = lo;
        else { r.lo = lo; r.hi = hi; }
    };
    auto name_next = [&](BoundInstr &r, uint64_t handler) {
        if (f32) { r.lo = (uint32_t)handler; r.hi = (uint32_t)(handler >> 32); }
        else r.arg = (uint32_t)handler;
    };
    // (the DE_NO_END_FUSE switch of round 2 measured <= 1 % and is gone)
    // a tree that finishes in a validity-tested hot operator runs that instruction and its end as ONE dispatch (h_chain_end);
    // a one-instruction tree keeps the plain form (the kernel's first call cannot tell the two apart)
    auto ev_of = [&](int64_t t) -> int {
        const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1];
        return i1 - i0 >= 2 ? topx_endv_of(p->fbcode[(size_t)i1 - 1].bop) : -1;
    };
    auto handler_of = [&](int32_t i, int32_t i1, int ev) -> uint64_t {
        return (i == i1 - 1 && ev >= 0) ? p->endv_handler[ev] : p->handler_base + p->tcode[(size_t)i].bop;
    };
    // Pass A, on the host threads: the records a tree OWNS — its instruction records and its end record (h = one end record per
    // preceding tree + the head record).  Pass B, serial (three writes per tree): what a tree writes into its PREDECESSOR's last two
    // records — the handler of its first instruction, and the header words over the end record's.
    parallel_tree_ranges(p->n_trees, [&](int, int64_t tb, int64_t te) {
        for (int64_t t = tb; t < te; t++) {
            const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1];
            const size_t h = (size_t)i0 + (size_t)t + 1;
            p->ccode_off[(size_t)t] = (int32_t)h;
            const int ev = ev_of(t);
            for (int32_t i = i0; i < i1; i++) {
                const BoundInstr &s = p->tcode[(size_t)i];
                put(p->ccode[h + (size_t)(i - i0)], s.arg, s.lo, s.hi);
                if (i > i0) name_next(p->ccode[h + (size_t)(i - i0) - 1], handler_of(i, i1, ev)); // in the record in front
            }
            put(p->ccode[h + (size_t)(i1 - i0)], (uint32_t)t, 0u, 0u); // end record (operand word: the tree's index, informational)
            if (i1 > i0) name_next(p->ccode[h + (size_t)(i1 - i0) - 1], p->end_handler);
        }
    });
    for (int64_t t = 0; t < p->n_trees; t++) {
        const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1];
        const size_t h = (size_t)p->ccode_off[(size_t)t];
        const int ev = ev_of(t);
        if (i1 > i0) {
            const uint64_t first = handler_of(i0, i1, ev);
            name_next(p->ccode[h - 1], first); // in the head record / the previous tree's end record
            // the previous tree finishes in an end-fused handler: ITS last instruction names this tree's first handler, stepping over its end record
            if (t > 0 && ev_of(t - 1) >= 0) name_next(p->ccode[h - 2], first);
        } else name_next(p->ccode[h - 1], p->end_handler);
        // the record in front of the tree (head record / previous tree's end record) is its HEADER: its immediate = the number of
        // instruction records of the tree, which is what h_tree_skip needs to step over a tree that is not evaluated
        // bit 31 = the tree finishes in an end-fused handler (its last instruction record names the next tree's first handler too):
        // what de_compact_live_kernel (de_kernels.hip) needs to re-link a tree behind another one
        put(p->ccode[h - 1], t == 0 ? 0u : (uint32_t)(t - 1), (uint32_t)(i1 - i0) | (ev >= 0 ? DE_HDR_FUSED_END : 0u), 0u);
    }
    if (p->n_trees > 0) name_next(p->ccode.back(), p->end_handler); // never followed: the last tree's end returns (left == 1)
    p->ccode_off[(size_t)p->n_trees] = (int32_t)p->ccode.size();
}
static inline void patch_chained_imm(de_program *p, int32_t c, uint32_t lo, uint32_t hi) {
    if (p->dtype == DE_F32) p->ccode[(size_t)c].arg = lo;
    else { p->ccode[(size_t)c].lo = lo; p->ccode[(size_t)c].hi = hi; }
}

static void recompute_host_ok(de_program *p) {
    const bool ee = (p->options & DE_OPT_EARLY_EXIT) != 0;
    parallel_for_trees(p->n_trees, [&](int64_t t) {
        bool ok_eval = true, ok_grad = true;
        for (int64_t k = p->const_off[t]; k < p->const_off[t + 1]; k++) {
            const bool fin = finite_in(p->dtype, p->consts[k]);
            ok_grad = ok_grad && fin;
            const uint8_t ch = p->const_checks[k];
            if (!fin && ((ch & CONST_CHECK_ALWAYS) || (ee && (ch & CONST_CHECK_EE)))) ok_eval = false;
        }
        p->host_ok_eval[t] = ok_eval;
        p->host_ok_grad[t] = ok_grad;
    }, 1024);
    // a constant subtree that evaluates to a non-finite value clears the flag — with the flag
    // semantics of the program's own options (dispatch_constant_tree tests unconditionally,
    // the Bumper path only under early_exit): that is exactly what `aux` was lowered with
    // — except for a subtree the reference never hands to dispatch_constant_tree (inner branch of a fused
    // 3-node kernel): its non-finite value is only noticed by the early-exit tests
    for (size_t j = 0; j < p->folds.size() && j < p->fold_ok.size(); j++)
        if (!p->fold_ok[j] && (p->folds[j].tested_always || ee)) p->host_ok_eval[(size_t)p->folds[j].tree] = 0;
}

static int create_impl(de_ctx_t *ctx, int dtype, const de_tape_node_t *nodes, const int64_t *node_offsets,
                       int64_t n_trees, const void *consts, const int64_t *const_offsets, int32_t n_features,
                       int32_t n_params, uint32_t options, bool allow_fold, de_program_t **out_program,
                       const de_tape_node_t *cse_nodes = nullptr, const int64_t *cse_offsets = nullptr);

// Device copy of the host part of the eval flag: every de_eval starts from it with one device-to-device copy
// (a pageable host-to-device copy per call costs ~10 us, a fifth of a small-population call).
static int upload_ok_eval(de_ctx *c, de_program *p) {
    p->tab_ok_stale = true; // host_ok_grad moves with the constants too
    if (p->n_trees == 0) return DE_OK;
    if (!p->d_ok_eval) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_ok_eval), (size_t)std::max<int64_t>(p->n_trees, 1))); // (never taken since round 6: the arena holds it)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(p->d_ok_eval, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice));
    return DE_OK;
}

// (Re-)evaluate the folded constant subtrees — the IEEE-exact ones on the host, the others on the device — and patch their
// values into fcode.
// `aux_current`: the auxiliary program was created with the present constants this very moment (de_program_create: setting them
// again cost 0.9 of the 1.1 ms this step took for 10^4 trees).
static int refresh_folds(de_ctx *c, de_program *p, bool aux_current = false) {
    if (!p->folded || p->folds.empty()) return DE_OK; // (a CSE-only eval program has no constant subtrees to evaluate)
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    const size_t nf = p->folds.size();
    p->fold_ok.assign(nf, 0);
    parallel_for_trees((int64_t)nf, [&](int64_t j) {
        if (p->fold_host[(size_t)j] != 1) return;
        const de_tape_node_t *nd = p->fold_nodes.data() + p->fold_noff[(size_t)j];
        const int64_t n = p->fold_noff[(size_t)j + 1] - p->fold_noff[(size_t)j];
        const int64_t *csrc = p->aux_const_src.data() + p->fold_coff[(size_t)j];
        double v;
        bool ok;
        if (p->dtype == DE_F32) { float f; ok = host_fold_eval<float>(nd, n, p->consts.data(), csrc, &f); v = (double)f; }
        else ok = host_fold_eval<double>(nd, n, p->consts.data(), csrc, &v);
        p->fold_ok[(size_t)j] = ok ? 1 : 0;
        write_imm(p->fcode[(size_t)p->folds[(size_t)j].instr], p->dtype, v);
    }, 256);
    if (!p->kfold.empty()) {
        // the subtrees with other operators: one thread each on the device (de_fold_kernel), the operators' own device code.
        // `aux_current`: the image uploaded at creation already holds these constants.
        const size_t nk = p->kfold.size();
        HIP_TRY(c, hipSetDevice(c->device));
        if (!aux_current) {
            std::vector<unsigned char> cv(std::max<size_t>(p->kf_csrc.size(), 1) * es);
            for (size_t k = 0; k < p->kf_csrc.size(); k++) {
                const double v = p->consts[(size_t)p->kf_csrc[k]];
                if (p->dtype == DE_F32) reinterpret_cast<float *>(cv.data())[k] = (float)v;
                else reinterpret_cast<double *>(cv.data())[k] = v;
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream)); // (an earlier launch may still read the values)
            if (!p->kf_csrc.empty()) HIP_TRY(c, hipMemcpy(p->d_kf + p->kf_o_cvals, cv.data(), p->kf_csrc.size() * es, hipMemcpyHostToDevice));
        }
        HIP_TRY(c, launch_fold(p->dtype, p->d_kf, reinterpret_cast<const int64_t *>(p->d_kf + p->kf_o_noff), reinterpret_cast<const int64_t *>(p->d_kf + p->kf_o_coff),
                               p->d_kf + p->kf_o_cvals, (int64_t)nk, p->d_kf + p->kf_o_out, reinterpret_cast<uint8_t *>(p->d_kf + p->kf_o_ok), c->stream));
        std::vector<unsigned char> res(p->kf_bytes - p->kf_o_out);
        HIP_TRY(c, hipMemcpyAsync(res.data(), p->d_kf + p->kf_o_out, res.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const unsigned char *okb = res.data() + (p->kf_o_ok - p->kf_o_out);
        for (size_t k = 0; k < nk; k++) {
            const size_t j = (size_t)p->kfold[k];
            const double v = p->dtype == DE_F32 ? (double)reinterpret_cast<const float *>(res.data())[k] : reinterpret_cast<const double *>(res.data())[k];
            p->fold_ok[j] = okb[k];
            write_imm(p->fcode[(size_t)p->folds[j].instr], p->dtype, v);
        }
    }
    if (!p->aux) return DE_OK;
    const size_t na = p->aux_fold.size();
    int rc = DE_OK;
    if (!aux_current) {
        std::vector<unsigned char> ac(std::max<size_t>(p->aux_csrc.size(), 1) * es);
        for (size_t k = 0; k < p->aux_csrc.size(); k++) {
            const double v = p->consts[(size_t)p->aux_csrc[k]];
            if (p->dtype == DE_F32) reinterpret_cast<float *>(ac.data())[k] = (float)v;
            else reinterpret_cast<double *>(ac.data())[k] = v;
        }
        rc = de_program_set_consts(p->aux, ac.data());
        if (rc != DE_OK) return fail(c, rc, "constant folding: %s", p->aux->ctx->err.c_str());
    }
    std::vector<unsigned char> X(std::max<size_t>((size_t)p->n_features, 1) * es, 0), out(na * es);
    std::vector<uint8_t> aok(na, 0);
    rc = de_eval(c, p->aux, X.data(), 1, std::max<int64_t>(p->n_features, 1), nullptr, out.data(), 1, aok.data());
    if (rc != DE_OK) return rc;
    for (size_t a = 0; a < na; a++) {
        const size_t j = (size_t)p->aux_fold[a];
        const double v = p->dtype == DE_F32 ? (double)reinterpret_cast<float *>(out.data())[a]
                                            : reinterpret_cast<double *>(out.data())[a];
        p->fold_ok[j] = aok[a];
        write_imm(p->fcode[(size_t)p->folds[j].instr], p->dtype, v);
    }
    return DE_OK;
}


int de_program_create(de_ctx_t *ctx, int dtype, const de_tape_node_t *nodes, const int64_t *node_offsets,
                      int64_t n_trees, const void *consts, const int64_t *const_offsets,
                      int32_t n_features, int32_t n_params, uint32_t options, de_program_t **out_program) {
    const char *nf = getenv("DE_NO_FOLD");
    if (!ctx) return DE_ERR_INVALID_ARG;
    DE_NOTHROW(ctx, create_impl(ctx, dtype, nodes, node_offsets, n_trees, consts, const_offsets, n_features, n_params, options,
                                !(nf && *nf == '1'), out_program));
}

int de_program_create_cse(de_ctx_t *ctx, int dtype, const de_tape_node_t *nodes, const int64_t *node_offsets,
                          const de_tape_node_t *cse_nodes, const int64_t *cse_offsets, int64_t n_trees, const void *consts,
                          const int64_t *const_offsets, int32_t n_features, int32_t n_params, uint32_t options,
                          de_program_t **out_program) {
    const char *nf = getenv("DE_NO_FOLD"), *nc = getenv("DE_NO_CSE");
    const bool fold = !(nf && *nf == '1'), cse = !(nc && *nc == '1');
    if (!ctx) return DE_ERR_INVALID_ARG;
    if (n_trees > 0 && cse_nodes && !cse_offsets) return fail(ctx, DE_ERR_INVALID_ARG, "cse_offsets is null");
    DE_NOTHROW(ctx, create_impl(ctx, dtype, nodes, node_offsets, n_trees, consts, const_offsets, n_features, n_params, options, fold, out_program,
                                fold && cse ? cse_nodes : nullptr, cse_offsets));
}

// The eval program of tree t is lowered from its CSE tape when the caller supplied one (a GraphNode tree: shared subtrees
// appear once, de_program_create_cse); everything else — gradients, constant bookkeeping, flags — follows the expanded tape.
static int create_impl(de_ctx_t *ctx, int dtype, const de_tape_node_t *nodes, const int64_t *node_offsets,
                       int64_t n_trees, const void *consts, const int64_t *const_offsets, int32_t n_features,
                       int32_t n_params, uint32_t options, bool allow_fold, de_program_t **out_program,
                       const de_tape_node_t *cse_nodes, const int64_t *cse_offsets) {
    if (!ctx) return DE_ERR_INVALID_ARG;
    if (!out_program) return fail(ctx, DE_ERR_INVALID_ARG, "out_program is null");
    *out_program = nullptr;
    if (dtype != DE_F32 && dtype != DE_F64) return fail(ctx, DE_ERR_INVALID_ARG, "dtype must be DE_F32 or DE_F64");
    if (n_trees < 0 || n_features < 0 || n_params < 0 || n_features > 65535 || n_params > 65535)
        return fail(ctx, DE_ERR_INVALID_ARG, "bad sizes");
    if (n_trees > 0 && (!nodes || !node_offsets || !const_offsets))
        return fail(ctx, DE_ERR_INVALID_ARG, "null tape pointers");
    if (n_trees > 0x7fffffff) return fail(ctx, DE_ERR_UNSUPPORTED, "too many trees");
    std::unique_ptr<de_program> p(new (std::nothrow) de_program());
    if (!p) return fail(ctx, DE_ERR_HIP, "out of host memory");
    adopt_parked(ctx, p.get());
    // DE_DEBUG_TIMING: microseconds per phase of the creation on stderr (tools/bench_create.py)
    const bool timing = getenv("DE_DEBUG_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[de_program_create %lld trees%s] %-28s %9.1f us\n", (long long)n_trees, allow_fold ? "" : " (aux)", what,
                std::chrono::duration<double, std::micro>(now - t_last).count());
        t_last = std::chrono::steady_clock::now();
    };
    try {
        p->ctx = ctx;
        p->dtype = dtype;
        p->options = options;
        p->n_features = n_features;
        p->n_params = n_params;
        p->n_trees = n_trees;
        LowerOptions lo;
        lo.early_exit = (options & DE_OPT_EARLY_EXIT) != 0;
        lo.fuse1 = (options & DE_OPT_FUSE_DEG1) != 0;
        lo.fuse2 = (options & DE_OPT_FUSE_DEG2) != 0;
        lo.bumper = (options & DE_OPT_BUMPER_CHECKS) != 0;
        lo.n_features = n_features;
        lo.n_params = n_params;
        lo.dtype = dtype;
        p->code_off.assign((size_t)n_trees + 1, 0);
        p->const_off.assign((size_t)n_trees + 1, 0);
        p->n_consts_tree.assign((size_t)n_trees, 0);
        p->host_ok_eval.assign((size_t)n_trees, 1);
        p->host_ok_grad.assign((size_t)n_trees, 1);
        const int64_t total_consts = n_trees ? const_offsets[n_trees] - const_offsets[0] : 0;
        if (total_consts < 0) return fail(ctx, DE_ERR_INVALID_ARG, "const_offsets not monotone");
        if (total_consts > 0 && !consts) return fail(ctx, DE_ERR_INVALID_ARG, "consts is null");
        p->consts.resize((size_t)total_consts);
        p->const_instr.assign((size_t)total_consts, -1);
        p->const_checks.assign((size_t)total_consts, 0);
        for (int64_t t = 0; t < n_trees; t++)
            if (node_offsets[t + 1] < node_offsets[t] || const_offsets[t + 1] < const_offsets[t])
                return fail(ctx, DE_ERR_INVALID_ARG, "offsets not monotone at tree %lld", (long long)t);
        // both lowerings of every tree (plain, and with constant subtrees folded), on host threads
        struct Lowered { TreeProgram plain, folded; int rc = DE_OK, rcf = DE_OK; bool cse = false, cse_plain = false; std::string why; };
        std::vector<Lowered> low((size_t)n_trees);
        {
            LowerOptions lof = lo;
            lof.fold = true;
            std::atomic<bool> oom{false};
            parallel_for_trees(n_trees, [&](int64_t t) {
                Lowered &L = low[(size_t)t];
                const int64_t n0 = node_offsets[t], c0 = const_offsets[t];
                try {
                    L.rc = lower_tree(nodes + n0, node_offsets[t + 1] - n0, const_offsets[t + 1] - c0, lo, &L.plain, &L.why);
                    if (L.rc == DE_OK && cse_nodes && cse_offsets[t + 1] > cse_offsets[t] && !getenv("DE_NO_GRAD_CSE")) {
                        // GraphNode sharing in the GENERIC program too (round 3; the gradient kernels, eval_diff and the unfolded
                        // eval run it): a shared subtree's dual number is computed once into a persistent slot and read by every
                        // consumer — the reference evaluates it once per parent with the same arithmetic, so values, Jacobian rows
                        // of features / parameters and flags are those of the expansion.  A constant inside a shared subtree keeps
                        // the gradient row of its FIRST occurrence, which receives every consumer's contribution; the rows of its
                        // later occurrences stay zero (callers sum the occurrence rows: the reference's shared NodeIndex row).
                        TreeProgram pc;
                        std::string why2;
                        LowerOptions loc = lo;
                        loc.cse = true;
                        if (lower_tree(cse_nodes + cse_offsets[t], cse_offsets[t + 1] - cse_offsets[t], const_offsets[t + 1] - c0, loc, &pc, &why2) == DE_OK) {
                            L.plain = std::move(pc);
                            L.cse_plain = true;
                        }
                    }
                    if (L.rc == DE_OK && allow_fold) {
                        if (cse_nodes && cse_offsets[t + 1] > cse_offsets[t]) {
                            LowerOptions loc = lof;
                            loc.cse = true;
                            L.rcf = lower_tree(cse_nodes + cse_offsets[t], cse_offsets[t + 1] - cse_offsets[t], const_offsets[t + 1] - c0, loc, &L.folded, &L.why);
                            L.cse = L.rcf == DE_OK;
                            if (L.rcf == DE_ERR_UNSUPPORTED) {
                                // the CSE form does not fit (spill slots + shared rows > 16, a share in an unsupported position): the
                                // expanded tape has the same values and flags (the reference evaluates a shared node once per
                                // parent), so this tree alone runs expanded instead of failing the whole population
                                L.folded = TreeProgram();
                                L.why.clear();
                                L.rcf = lower_tree(nodes + n0, node_offsets[t + 1] - n0, const_offsets[t + 1] - c0, lof, &L.folded, &L.why);
                            }
                        } else L.rcf = lower_tree(nodes + n0, node_offsets[t + 1] - n0, const_offsets[t + 1] - c0, lof, &L.folded, &L.why);
                    }
                } catch (const std::bad_alloc &) { oom = true; }
            });
            if (oom) return fail(ctx, DE_ERR_HIP, "out of host memory");
        }
        lap("lower (host threads)");
        // merge: offsets by a serial prefix sum, the copies on the host threads (every tree writes slices of its own)
        {
            uint64_t total = 0;
            for (int64_t t = 0; t < n_trees; t++) {
                if (low[(size_t)t].rc != DE_OK) return fail(ctx, low[(size_t)t].rc, "tree %lld: %s", (long long)t, low[(size_t)t].why.c_str());
                total += low[(size_t)t].plain.code.size();
                if (total > 0x7fff0000u) return fail(ctx, DE_ERR_UNSUPPORTED, "program too large");
                p->code_off[(size_t)t + 1] = (int32_t)total;
            }
            p->code.resize((size_t)total);
            struct Part { int32_t n_slots = 0; bool cse = false, params = false; int64_t nodes = 0; } part[HOST_RANGES_MAX];
            parallel_tree_ranges(n_trees, [&](int wk, int64_t tb, int64_t te) {
                Part &pt = part[wk];
                for (int64_t t = tb; t < te; t++) {
                    const int64_t n0 = node_offsets[t], n1 = node_offsets[t + 1];
                    const int64_t c0 = const_offsets[t], c1 = const_offsets[t + 1];
                    TreeProgram &tp = low[(size_t)t].plain;
                    const int64_t cb = c0 - const_offsets[0];
                    p->const_off[(size_t)t + 1] = cb + (c1 - c0); // (entry t is tree t - 1's, entry 0 stays 0)
                    p->n_consts_tree[(size_t)t] = (int32_t)(c1 - c0);
                    const int32_t ib = p->code_off[(size_t)t];
                    for (int64_t k = 0; k < c1 - c0; k++) {
                        const double v = dtype == DE_F32 ? (double)static_cast<const float *>(consts)[c0 + k]
                                                         : static_cast<const double *>(consts)[c0 + k];
                        p->consts[(size_t)(cb + k)] = v;
                        // (a CSE lowering has no instruction for the later occurrences of a constant inside a shared subtree: -1)
                        p->const_instr[(size_t)(cb + k)] = tp.const_instr[(size_t)k] >= 0 ? ib + tp.const_instr[(size_t)k] : -1;
                        p->const_checks[(size_t)(cb + k)] = tp.const_checks[(size_t)k];
                        if (tp.const_instr[(size_t)k] >= 0) write_imm(tp.code[(size_t)tp.const_instr[(size_t)k]], dtype, v);
                    }
                    std::copy(tp.code.begin(), tp.code.end(), p->code.begin() + ib);
                    pt.cse = pt.cse || low[(size_t)t].cse_plain;
                    pt.n_slots = std::max(pt.n_slots, tp.n_slots);
                    pt.params = pt.params || tp.uses_params;
                    pt.nodes += n1 - n0;
                }
            });
            for (const Part &pt : part) {
                p->cse_generic = p->cse_generic || pt.cse;
                p->n_slots = std::max(p->n_slots, pt.n_slots);
                p->uses_params = p->uses_params || pt.params;
                p->n_nodes += pt.nodes;
            }
        }
        lap("merge plain");
        // ---- folded lowering of the eval program + the auxiliary population of constant subtrees
        if (allow_fold) {
            lo.fold = true;
            std::vector<de_tape_node_t> &anodes = p->fold_nodes; // (retained: the host-folded subtrees are re-evaluated from them)
            std::vector<int64_t> &anoff = p->fold_noff, &acoff = p->fold_coff;
            bool any_cse = false;
            p->fcode_off.assign((size_t)n_trees + 1, 0);
            p->fconst_instr.assign((size_t)total_consts, -1);
            // offsets of every tree's instructions, folds, auxiliary tape nodes and auxiliary constants by a serial prefix sum ...
            std::vector<int64_t> fold0((size_t)n_trees + 1, 0), anode0((size_t)n_trees + 1, 0), acs0((size_t)n_trees + 1, 0);
            {
                uint64_t total = 0;
                for (int64_t t = 0; t < n_trees; t++) {
                    const TreeProgram &tp = low[(size_t)t].folded;
                    if (low[(size_t)t].rcf != DE_OK)
                        return fail(ctx, low[(size_t)t].rcf, "tree %lld (folded): %s", (long long)t, low[(size_t)t].why.c_str());
                    total += tp.code.size();
                    if (total > 0x7fff0000u) return fail(ctx, DE_ERR_UNSUPPORTED, "program too large");
                    p->fcode_off[(size_t)t + 1] = (int32_t)total;
                    int64_t nn = 0, nc = 0;
                    for (const FoldSpan &sp : tp.folds) { nn += sp.node_end - sp.node_begin; nc += sp.const_end - sp.const_begin; }
                    fold0[(size_t)t + 1] = fold0[(size_t)t] + (int64_t)tp.folds.size();
                    anode0[(size_t)t + 1] = anode0[(size_t)t] + nn;
                    acs0[(size_t)t + 1] = acs0[(size_t)t] + nc;
                }
                p->fcode.resize((size_t)total);
            }
            const size_t n_folds = (size_t)fold0[(size_t)n_trees];
            p->folds.resize(n_folds);
            anodes.resize((size_t)anode0[(size_t)n_trees]);
            p->aux_const_src.resize((size_t)acs0[(size_t)n_trees]);
            anoff.assign(n_folds + 1, 0);
            acoff.assign(n_folds + 1, 0);
            // ... the copies on the host threads
            struct PartF { int32_t n_slots = 0; bool cse = false; } partf[HOST_RANGES_MAX];
            parallel_tree_ranges(n_trees, [&](int wk, int64_t tb, int64_t te) {
                PartF &pt = partf[wk];
                for (int64_t t = tb; t < te; t++) {
                    const int64_t n0 = node_offsets[t];
                    const int64_t c0 = const_offsets[t], c1 = const_offsets[t + 1];
                    TreeProgram &tp = low[(size_t)t].folded;
                    const bool is_cse = low[(size_t)t].cse;
                    const de_tape_node_t *src_nodes = is_cse ? cse_nodes + cse_offsets[t] : nodes + n0; // the tape the fold spans index
                    pt.cse = pt.cse || is_cse;
                    pt.n_slots = std::max(pt.n_slots, tp.n_slots); // a CSE program keeps one persistent row per shared subtree
                    const int64_t cb = c0 - const_offsets[0];
                    const int32_t ib = p->fcode_off[(size_t)t];
                    for (int64_t k = 0; k < c1 - c0; k++) {
                        const int32_t ci = tp.const_instr[(size_t)k];
                        if (ci < 0) continue; // constant lives inside a folded subtree
                        p->fconst_instr[(size_t)(cb + k)] = ib + ci;
                        write_imm(tp.code[(size_t)ci], dtype, p->consts[(size_t)(cb + k)]);
                    }
                    size_t an = (size_t)anode0[(size_t)t], ac = (size_t)acs0[(size_t)t];
                    for (size_t f = 0; f < tp.folds.size(); f++) {
                        const FoldSpan &sp = tp.folds[f];
                        const size_t fi = (size_t)fold0[(size_t)t] + f;
                        p->folds[fi] = {(int32_t)t, ib + tp.const_instr[(size_t)(c1 - c0) + f], sp.tested_always};
                        for (int32_t q = sp.node_begin; q < sp.node_end; q++) {
                            de_tape_node_t nd = src_nodes[q];
                            if (nd.degree == 0 && nd.op == DE_LEAF_CONST) nd.arg = (uint16_t)(nd.arg - sp.const_begin);
                            anodes[an++] = nd;
                        }
                        for (int32_t q = sp.const_begin; q < sp.const_end; q++) p->aux_const_src[ac++] = cb + q;
                        anoff[fi + 1] = (int64_t)an;
                        acoff[fi + 1] = (int64_t)ac;
                    }
                    std::copy(tp.code.begin(), tp.code.end(), p->fcode.begin() + ib);
                }
            });
            for (const PartF &pt : partf) {
                any_cse = any_cse || pt.cse;
                p->n_slots = std::max(p->n_slots, pt.n_slots);
            }
            lap("merge folded");
            if (!p->folds.empty()) {
                const size_t es = dtype == DE_F32 ? 4 : 8;
                // which folds stay on the host: subtrees of + - * / only (the turbo division is not IEEE: such programs fold everything on
                // the device, with the operators they evaluate with; DE_NO_HOST_FOLD=1: everything on the device, for A/B tests)
                const char *nh = getenv("DE_NO_HOST_FOLD");
                const bool host_fold = !(nh && *nh == '1') && !(options & DE_OPT_TURBO);
                // ... and which go to de_fold_kernel (everything else whose evaluation stack fits; a turbo program evaluates with other
                // operators than that kernel has: its subtrees stay with the auxiliary program; DE_NO_KERNEL_FOLD=1 for A/B tests)
                const char *nk_env = getenv("DE_NO_KERNEL_FOLD");
                const bool kernel_fold = !(nk_env && *nk_env == '1') && !(options & DE_OPT_TURBO);
                p->fold_host.assign(n_folds, 0);
                parallel_for_trees((int64_t)n_folds, [&](int64_t j) {
                    const de_tape_node_t *nd = anodes.data() + anoff[(size_t)j];
                    const int64_t n = anoff[(size_t)j + 1] - anoff[(size_t)j];
                    if (host_fold && host_foldable(nd, n)) { p->fold_host[(size_t)j] = 1; return; }
                    if (!kernel_fold) return;
                    int depth = 0, worst = 0;
                    for (int64_t i = 0; i < n; i++) { depth += 1 - (int)nd[i].degree; worst = std::max(worst, depth); }
                    if (worst <= DE_FOLD_STACK) p->fold_host[(size_t)j] = 2;
                }, 512);
                // the kernel's image: tape slices, offsets and constant sources of its subtrees, in fold order
                p->kfold.clear();
                p->kf_csrc.clear();
                {
                    std::vector<de_tape_node_t> knodes;
                    std::vector<int64_t> knoff{0}, kcoff{0};
                    for (size_t j = 0; j < n_folds; j++) {
                        if (p->fold_host[j] != 2) continue;
                        p->kfold.push_back((int32_t)j);
                        knodes.insert(knodes.end(), anodes.begin() + anoff[j], anodes.begin() + anoff[j + 1]);
                        p->kf_csrc.insert(p->kf_csrc.end(), p->aux_const_src.begin() + acoff[j], p->aux_const_src.begin() + acoff[j + 1]);
                        knoff.push_back((int64_t)knodes.size());
                        kcoff.push_back((int64_t)p->kf_csrc.size());
                    }
                    if (!p->kfold.empty()) {
                        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
                        const size_t nk = p->kfold.size();
                        p->kf_o_noff = al(knodes.size() * sizeof(de_tape_node_t));
                        p->kf_o_coff = p->kf_o_noff + al(knoff.size() * sizeof(int64_t));
                        p->kf_o_cvals = p->kf_o_coff + al(kcoff.size() * sizeof(int64_t));
                        p->kf_o_out = p->kf_o_cvals + al(std::max<size_t>(p->kf_csrc.size(), 1) * es);
                        p->kf_o_ok = p->kf_o_out + al(nk * es);
                        p->kf_bytes = p->kf_o_ok + al(nk);
                        std::vector<unsigned char> img(p->kf_o_out, 0);
                        std::memcpy(img.data(), knodes.data(), knodes.size() * sizeof(de_tape_node_t));
                        std::memcpy(img.data() + p->kf_o_noff, knoff.data(), knoff.size() * sizeof(int64_t));
                        std::memcpy(img.data() + p->kf_o_coff, kcoff.data(), kcoff.size() * sizeof(int64_t));
                        for (size_t k = 0; k < p->kf_csrc.size(); k++) {
                            const double v = p->consts[(size_t)p->kf_csrc[k]];
                            if (dtype == DE_F32) reinterpret_cast<float *>(img.data() + p->kf_o_cvals)[k] = (float)v;
                            else reinterpret_cast<double *>(img.data() + p->kf_o_cvals)[k] = v;
                        }
                        HIP_TRY(ctx, hipSetDevice(ctx->device));
                        const hipError_t kst = prog_malloc(ctx, reinterpret_cast<void **>(&p->d_kf), p->kf_bytes);
                        if (kst != hipSuccess) return fail(ctx, DE_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(kst));
                        HIP_TRY(ctx, hipMemcpy(p->d_kf, img.data(), img.size(), hipMemcpyHostToDevice));
                    }
                }
                // the others form the auxiliary population (their tape slices and constants, concatenated in fold order)
                std::vector<de_tape_node_t> xnodes;
                std::vector<int64_t> xnoff{0}, xcoff{0};
                p->aux_fold.clear();
                p->aux_csrc.clear();
                for (size_t j = 0; j < n_folds; j++) {
                    if (p->fold_host[j]) continue;
                    p->aux_fold.push_back((int32_t)j);
                    xnodes.insert(xnodes.end(), anodes.begin() + anoff[j], anodes.begin() + anoff[j + 1]);
                    p->aux_csrc.insert(p->aux_csrc.end(), p->aux_const_src.begin() + acoff[j], p->aux_const_src.begin() + acoff[j + 1]);
                    xnoff.push_back((int64_t)xnodes.size());
                    xcoff.push_back((int64_t)p->aux_csrc.size());
                }
                p->folded = true;
                lap("folds: classify, kernel image, auxiliary tapes");
                if (!p->aux_fold.empty()) {
                    std::vector<unsigned char> ac(std::max<size_t>(p->aux_csrc.size(), 1) * es, 0);
                    for (size_t k = 0; k < p->aux_csrc.size(); k++) {
                        const double v = p->consts[(size_t)p->aux_csrc[k]];
                        if (dtype == DE_F32) reinterpret_cast<float *>(ac.data())[k] = (float)v;
                        else reinterpret_cast<double *>(ac.data())[k] = v;
                    }
                    int rc = create_impl(ctx, dtype, xnodes.data(), xnoff.data(), (int64_t)p->aux_fold.size(), ac.data(),
                                         xcoff.data(), n_features, 0, options, false, &p->aux);
                    if (rc != DE_OK) return rc;
                    lap("aux program (create)");
                }
                int rc = refresh_folds(ctx, p.get(), true);
                if (rc != DE_OK) return rc;
                lap("folds (evaluate: host + kernel + aux)");
            } else if (any_cse) {
                p->folded = true; // the eval program is the CSE lowering even without a constant subtree to fold
            } else {
                p->fcode.clear();
                p->fcode_off.clear();
            }
        }
        // the per-tree lowerings are ~16 small vectors each: released on the threads that allocated them (one thread took 5 ms for 10^4 trees)
        parallel_for_trees(n_trees, [&](int64_t t) { Lowered done; std::swap(done, low[(size_t)t]); });
        lap("release lowerings");
        recompute_host_ok(p.get());
        rebind(p.get());
        lap("bind");
    } catch (const std::bad_alloc &) {
        return fail(ctx, DE_ERR_HIP, "out of host memory");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        int rc = DE_OK;
        try { rc = make_threaded(ctx, p.get()); } catch (const std::bad_alloc &) { rc = fail(ctx, DE_ERR_HIP, "out of host memory"); }
        if (rc != DE_OK) return rc;
    }
    lap("threaded + chained records");
    // one trailing pad instruction: the flat-switch interpreter prefetches code[pc + 1]; the chained form of the
    // threaded kernel has one end record per tree and a head record (and the fused form is never longer than the bound one)
    const size_t cbytes = (p->bcode.size() + (size_t)p->n_trees + 2) * sizeof(BoundInstr); // + head record + one of padding
    {
        // the early-exit walk (h_tree_skip) rebuilds record addresses from their low 32 bits: the stream must lie inside one
        // 4 GiB window.  An allocation that straddles a boundary (once in ~10^4 for a 400 KB stream) is set aside and redone.
        // (a threaded program allocates the stream twice: the second half receives the re-linked stream of the live trees, de_compact_live_kernel)
        // ONE device arena per eval program (round 6): [record stream | its second half for the compacted live trees | tree offsets |
        // compaction control ints | initial flags], one allocation from the context's pool; a small program (the one-tree call of
        // de_eval_tree_array) goes up in ONE copy from a zero-filled host image, a large one in one memset + three copies.
        const size_t abytes = p->threaded ? 2 * cbytes : cbytes;
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t off_bytes = p->bcode_off.size() * sizeof(int32_t);
        const size_t ints_bytes = p->threaded ? ((size_t)2 * (size_t)p->n_trees + 5) * sizeof(int32_t) : 0;
        const size_t o_off = al(abytes), o_ints = o_off + al(off_bytes), o_ok = o_ints + al(ints_bytes);
        const size_t total = o_ok + al((size_t)std::max<int64_t>(p->n_trees, 1));
        const hipError_t ast = prog_malloc(ctx, reinterpret_cast<void **>(&p->d_code), total); // (one 4 GiB window: prog_malloc's contract)
        if (ast != hipSuccess) return fail(ctx, DE_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(ast));
        if (!in_one_window(p->d_code, abytes)) return fail(ctx, DE_ERR_HIP, "instruction stream straddles a 4 GiB boundary");
        char *base = reinterpret_cast<char *>(p->d_code);
        p->eval_arena = true; // (d_code_off, d_compact_ints, d_ok_eval live inside d_code's allocation: never freed on their own)
        p->d_code_off = reinterpret_cast<int32_t *>(base + o_off);
        p->d_ok_eval = reinterpret_cast<uint8_t *>(base + o_ok);
        if (p->threaded) {
            p->d_compact_code = p->d_code + cbytes / sizeof(BoundInstr);
            p->d_compact_ints = reinterpret_cast<int32_t *>(base + o_ints);
        }
        lap("hipMalloc (arena)");
        const std::vector<BoundInstr> &stream = p->threaded ? p->ccode : p->bcode;
        const std::vector<int32_t> &offs = p->threaded ? p->ccode_off : p->bcode_off;
        hipError_t st = hipSuccess;
        if (total <= (size_t)(128u << 10)) {
            // (the second half of a threaded stream needs no initial content: de_compact_live_kernel writes what the launch proper reads)
            std::vector<unsigned char> img(total, 0);
            if (!stream.empty()) std::memcpy(img.data(), stream.data(), stream.size() * sizeof(BoundInstr));
            std::memcpy(img.data() + o_off, offs.data(), off_bytes);
            if (p->n_trees > 0) std::memcpy(img.data() + o_ok, p->host_ok_eval.data(), (size_t)p->n_trees);
            st = hipMemcpy(base, img.data(), total, hipMemcpyHostToDevice);
        } else {
            st = hipMemset(p->d_code, 0, cbytes);
            if (st == hipSuccess && !stream.empty()) st = hipMemcpy(p->d_code, stream.data(), stream.size() * sizeof(BoundInstr), hipMemcpyHostToDevice);
            if (st == hipSuccess) st = hipMemcpy(p->d_code_off, offs.data(), off_bytes, hipMemcpyHostToDevice);
            if (st == hipSuccess && p->n_trees > 0) st = hipMemcpy(p->d_ok_eval, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice);
        }
        if (st != hipSuccess) {
            prog_free(ctx, p->d_code);
            p->d_code = nullptr;
            return fail(ctx, DE_ERR_HIP, "program upload failed: %s", hipGetErrorString(st));
        }
    }
    lap("memset + upload (stream, offsets, flags)");
    if (getenv("DE_VERIFY") && *getenv("DE_VERIFY") == '1') {
        const int rc = de_program_verify(p.get());
        if (rc != DE_OK) return rc;
    }
    *out_program = p.release();
    return DE_OK;
}

static int set_consts_impl(de_program_t *p, const void *consts);
static int set_consts_nothrow(de_program_t *p, const void *consts) {
    if (!p) return DE_ERR_INVALID_ARG;
    DE_NOTHROW(p->ctx, set_consts_impl(p, consts));
}
int de_program_set_consts(de_program_t *p, const void *consts) {
    const int rc = set_consts_nothrow(p, consts);
    if (rc == DE_OK && p && getenv("DE_VERIFY") && *getenv("DE_VERIFY") == '1') return de_program_verify(p);
    return rc;
}
static int set_consts_impl(de_program_t *p, const void *consts) {
    if (!p) return DE_ERR_INVALID_ARG;
    de_ctx *ctx = p->ctx;
    if (!consts && !p->consts.empty()) return fail(ctx, DE_ERR_INVALID_ARG, "consts is null");
    p->consts_gen++; // (the cached certificate program belongs to the old constants)
    const bool timing = getenv("DE_DEBUG_TIMING") != nullptr; // stderr: microseconds per phase
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    const auto t0 = now();
    // (every loop of this function over constants, trees or sites writes slots of its own: on the host pool; 10^4 trees: 0.85 -> see DESIGN 9.1)
    try {
        parallel_tree_ranges((int64_t)p->consts.size(), [&](int, int64_t kb, int64_t ke) {
            for (size_t k = (size_t)kb; k < (size_t)ke; k++) {
                const double v = p->dtype == DE_F32 ? (double)static_cast<const float *>(consts)[k]
                                                    : static_cast<const double *>(consts)[k];
                p->consts[k] = v;
                if (p->const_instr[k] >= 0) write_imm(p->code[(size_t)p->const_instr[k]], p->dtype, v);
                if (p->folded && p->fconst_instr[k] >= 0) write_imm(p->fcode[(size_t)p->fconst_instr[k]], p->dtype, v);
            }
        }, 4096);
    } catch (const std::bad_alloc &) { return fail(ctx, DE_ERR_HIP, "out of host memory"); }
    const auto t1 = now();
    if (p->folded) {
        int rc = DE_OK;
        try { rc = refresh_folds(ctx, p); } catch (const std::bad_alloc &) { rc = fail(ctx, DE_ERR_HIP, "out of host memory"); }
        if (rc != DE_OK) return rc;
    }
    const auto t2 = now();
    recompute_host_ok(p);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        const int rc = upload_ok_eval(ctx, p);
        if (rc != DE_OK) return rc;
    }
    const auto t3 = now();
    // Same tree shapes, new immediates: patch the bits where they live (the optimiser calls this once per
    // step — re-binding 10^4 trees costs milliseconds, the kernel it feeds a few hundred microseconds).
    const char *nopatch = getenv("DE_NO_CONST_PATCH");
    if (!(nopatch && *nopatch == '1') && p->threaded && !p->tsite.empty() && !p->bsite.empty()) {
        const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
        const bool gpatch = p->d_gcode && !p->gcode_stale && !p->gbsite.empty();
        const bool tpatch = gpatch && p->gt_valid && !p->gtsite_of_gb.empty();
        const bool rpatch = gpatch && p->rt_valid && !p->rtsite_of_gb.empty();
        if (p->lists_gen != p->site_gen) { // one pass over all instructions, then only the immediates are visited
            p->eval_sites.clear();
            p->grad_sites.clear();
            for (size_t i = 0; i < src.size(); i++)
                if (p->bsite[i] >= 0) {
                    // record of tcode[j] in the chained stream: one end record per preceding tree, behind the head record
                    const int32_t j = p->tsite[i];
                    const int64_t tree = (std::upper_bound(p->tcode_off.begin(), p->tcode_off.end(), j) - p->tcode_off.begin()) - 1;
                    p->eval_sites.push_back({(int32_t)i, p->bsite[i], j, (int32_t)(j + tree + 1)}); // + the head record
                }
            if (!p->gbsite.empty())
                for (size_t i = 0; i < p->code.size(); i++) {
                    const int32_t gj = p->gbsite[i];
                    if (gj < 0) continue;
                    p->grad_sites.push_back({(int32_t)i, gj, p->gtsite_of_gb.empty() ? -1 : p->gtsite_of_gb[(size_t)gj],
                                             p->rtsite_of_gb.empty() ? -1 : p->rtsite_of_gb[(size_t)gj]});
                }
            p->lists_gen = p->site_gen;
        }
        try {
        parallel_tree_ranges((int64_t)p->eval_sites.size(), [&](int, int64_t sb, int64_t se) {
            for (size_t q = (size_t)sb; q < (size_t)se; q++) {
                const de_program::EvalSite &e = p->eval_sites[q];
                const uint32_t lo = src[(size_t)e.src].imm.u32[0], hi = src[(size_t)e.src].imm.u32[1];
                p->bcode[(size_t)e.b].lo = lo;
                p->bcode[(size_t)e.b].hi = hi;
                p->tcode[(size_t)e.t].lo = lo;
                p->tcode[(size_t)e.t].hi = hi;
                patch_chained_imm(p, e.c, lo, hi);
            }
        }, 4096);
        if (gpatch)
            parallel_tree_ranges((int64_t)p->grad_sites.size(), [&](int, int64_t sb, int64_t se) {
                for (size_t q = (size_t)sb; q < (size_t)se; q++) {
                    const de_program::GradSite &g = p->grad_sites[q];
                    const uint32_t lo = p->code[(size_t)g.src].imm.u32[0], hi = p->code[(size_t)g.src].imm.u32[1];
                    p->gbcode[(size_t)g.gb].lo = lo;
                    p->gbcode[(size_t)g.gb].hi = hi;
                    if (tpatch && g.gt >= 0) {
                        p->gtcode[(size_t)g.gt].lo = lo;
                        p->gtcode[(size_t)g.gt].hi = hi;
                    }
                    if (rpatch && g.rt >= 0) {
                        p->rtcode[(size_t)g.rt].lo = lo;
                        p->rtcode[(size_t)g.rt].hi = hi;
                    }
                }
            }, 4096);
        } catch (const std::bad_alloc &) { return fail(ctx, DE_ERR_HIP, "out of host memory"); }
        if (gpatch) {
            if (!tpatch) p->gt_valid = false;
            if (!rpatch) p->rt_valid = false;
        } else {
            p->gcode_stale = true;
        }
        const auto t4 = now();
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // the program may be in use by work already queued
        if (!p->ccode.empty())
            HIP_TRY(ctx, hipMemcpy(p->d_code, p->ccode.data(), p->ccode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        if (gpatch) {
            if (!p->gbcode.empty())
                HIP_TRY(ctx, hipMemcpy(p->d_gcode, p->gbcode.data(), p->gbcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
            if (p->gt_valid && !p->gtcode.empty())
                HIP_TRY(ctx, hipMemcpy(p->d_gtcode, p->gtcode.data(), p->gtcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
            if (p->rt_valid && !p->rtcode.empty())
                HIP_TRY(ctx, hipMemcpy(p->d_rtcode, p->rtcode.data(), p->rtcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        }
        if (timing)
            fprintf(stderr, "set_consts us: write %ld, refresh_folds %ld, flags %ld, patch %ld, upload %ld\n", us(t0, t1), us(t1, t2), us(t2, t3),
                    us(t3, t4), us(t4, now()));
        return DE_OK;
    }
    p->gcode_stale = true;
    try {
        rebind(p); // same shape: only immediates change
    } catch (const std::bad_alloc &) {
        return fail(ctx, DE_ERR_HIP, "out of host memory");
    }
    {
        int rc = DE_OK;
        try { rc = make_threaded(ctx, p); } catch (const std::bad_alloc &) { rc = fail(ctx, DE_ERR_HIP, "out of host memory"); }
        if (rc != DE_OK) return rc;
    }
    // the program may be in use by work already queued on the stream
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!p->bcode.empty())
        HIP_TRY(ctx, hipMemcpy(p->d_code, (p->threaded ? p->ccode : p->bcode).data(),
                               (p->threaded ? p->ccode : p->bcode).size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
    return DE_OK;
}

int de_program_destroy(de_program_t *p) {
    if (!p) return DE_OK;
    (void)hipSetDevice(p->ctx->device);
    dbg_lap(nullptr);
    (void)hipStreamSynchronize(p->ctx->stream);
    de_ctx *c = p->ctx;
    dbg_lap("destroy: stream sync");
    prog_free(c, p->d_code);
    if (!p->eval_arena) {
        if (p->d_code_off) (void)hipFree(p->d_code_off);
        if (p->d_compact_ints) (void)hipFree(p->d_compact_ints);
    }
    if (p->d_cert_code) (void)hipFree(p->d_cert_code);
    if (p->d_cert_off) (void)hipFree(p->d_cert_off);
    dbg_lap("destroy: eval streams");
    if (p->aux) de_program_destroy(p->aux);
    prog_free(c, p->d_kf);
    p->d_kf = nullptr;
    dbg_lap(nullptr);
    prog_free(c, p->d_gcode);
    if (p->d_gcode_off) (void)hipFree(p->d_gcode_off);
    prog_free(c, p->d_gtcode);
    if (p->d_gtcode_off) (void)hipFree(p->d_gtcode_off);
    if (p->d_gt_ids) (void)hipFree(p->d_gt_ids);
    prog_free(c, p->d_rtcode);
    for (void *q : {(void *)p->d_rtcode_off, (void *)p->d_rtcode_mid, (void *)p->d_rt_ids})
        if (q) (void)hipFree(q);
    if (p->d_ok_eval && !p->eval_arena) (void)hipFree(p->d_ok_eval);
    for (void *q : {(void *)p->d_ok_grad, (void *)p->d_ng, (void *)p->d_goff})
        if (q) (void)hipFree(q);
    dbg_lap("destroy: gradient streams, flags");
    p->aux = nullptr;
    park_program(c, p);
    dbg_lap("destroy: host vectors");
    return DE_OK;
}

int64_t de_program_n_trees(const de_program_t *p) { return p ? p->n_trees : -1; }
int64_t de_program_n_nodes(const de_program_t *p) { return p ? p->n_nodes : -1; }
int64_t de_program_n_grad(const de_program_t *p, int64_t tree, int mode) {
    if (!p || tree < 0 || tree >= p->n_trees) return -1;
    const int64_t nc = p->n_consts_tree[(size_t)tree];
    const int64_t nv = (int64_t)p->n_features + p->n_params;
    switch (mode) {
    case DE_GRAD_VARIABLE: return nv;
    case DE_GRAD_CONSTANT: return nc;
    case DE_GRAD_BOTH: return nv + nc;
    default: return -1;
    }
}

int de_program_last_live_trees(de_program_t *p, int64_t *n_live) {
    if (!p || !n_live) return DE_ERR_INVALID_ARG;
    *n_live = -1;
    if (!p->last_compacted || !p->d_compact_ints) return DE_OK;
    de_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    int32_t v = -1;
    HIP_TRY(c, hipMemcpy(&v, p->d_compact_ints + (2 * (size_t)p->n_trees + 1), sizeof v, hipMemcpyDeviceToHost));
    *n_live = v;
    return DE_OK;
}

int de_prio_tiles_wanted(int64_t N, int32_t n_features, int64_t n_trees) { return prio_tiles_wanted(N, n_features, n_trees) ? 1 : 0; }

int de_eval_plan(const de_program_t *p, int64_t N, int32_t *plan) {
    if (!p || !plan || N < 0) return DE_ERR_INVALID_ARG;
    eval_plan(p->dtype, p->n_trees, N, &plan[0], &plan[1], &plan[2]);
    return DE_OK;
}

// Host-only test hook (no HIP call): n items over the pool of host threads that de_program_create's per-tree passes run on; returns how
// many items were visited exactly once (n when all is well), *n_ranges = the ranges the items were split into (1 = ran inline).
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
uint64_t de_program_stream_hash(const de_program_t *p) {
    if (!p) return 0;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *data, size_t bytes) {
        const unsigned char *b = static_cast<const unsigned char *>(data);
        for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211ull; }
        const uint64_t n = bytes; // (the length too: an empty vector and a missing one differ from a shifted boundary)
        for (int i = 0; i < 8; i++) { h ^= (n >> (8 * i)) & 0xFF; h *= 1099511628211ull; }
    };
    auto vec = [&](const auto &v) { mix(v.data(), v.size() * sizeof(v[0])); };
    vec(p->code); vec(p->code_off); vec(p->fcode); vec(p->fcode_off); vec(p->bcode); vec(p->bcode_off);
    vec(p->fbcode); vec(p->tcode); vec(p->tcode_off); vec(p->ccode); vec(p->ccode_off); vec(p->bsite); vec(p->tsite);
    vec(p->consts); vec(p->const_off); vec(p->const_instr); vec(p->fconst_instr); vec(p->const_checks); vec(p->n_consts_tree);
    vec(p->aux_const_src); vec(p->host_ok_eval); vec(p->host_ok_grad); vec(p->fold_ok);
    vec(p->fold_host); vec(p->fold_noff); vec(p->fold_coff); vec(p->aux_fold); vec(p->aux_csrc); vec(p->kfold); vec(p->kf_csrc);
    mix(p->fold_nodes.data(), p->fold_nodes.size() * sizeof(de_tape_node_t));
    for (const auto &f : p->folds) { const int32_t w[3] = {f.tree, f.instr, f.tested_always ? 1 : 0}; mix(w, sizeof w); }
    const int64_t scal[6] = {p->n_trees, p->n_nodes, p->n_slots, p->uses_params ? 1 : 0, p->folded ? 1 : 0, p->threaded ? 1 : 0};
    mix(scal, sizeof scal);
    if (p->aux) { const uint64_t a = de_program_stream_hash(p->aux); mix(&a, sizeof a); }
    return h;
}

// Program sanitizer (SURVEY.md §5 "sanitizer / bounds-checked debug"): the kernels trust the instruction streams —
// an LDS offset, a spill slot, a handler address are used as they come.  This walks every stream of the program on the
// host and checks each field against the bounds the launch will allocate: generic code (opcodes, operand rows <
// n_features + n_slots, push / pop slots, constant slots), bound and fused code (handler ids, rows, the int8 push
// distance of the superinstructions), and the chained stream the threaded kernel executes (every handler address is an
// entry of the device handler table, LDS byte offsets lie inside the launch's allocation, every tree ends in the end
// record).  DE_VERIFY=1 runs it after every de_program_create / de_program_set_consts.
int de_program_verify(const de_program_t *p) {
    if (!p) return DE_ERR_INVALID_ARG;
    de_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device)); // the handler tables are cached per device: the program's addresses are its OWN device's
    const int64_t rows = eval_rows(p), spill_end = (int64_t)p->n_features + p->n_slots;
    auto bad = [&](const char *what, int64_t tree, int64_t i, uint64_t v) {
        return fail(c, DE_ERR_BAD_TAPE, "program verify: %s (tree %lld, instruction %lld, value 0x%llx)", what, (long long)tree, (long long)i,
                    (unsigned long long)v);
    };
    const std::vector<Instr> *gens[2] = {&p->code, p->folded ? &p->fcode : nullptr};
    const std::vector<int32_t> *goffs[2] = {&p->code_off, p->folded ? &p->fcode_off : nullptr};
    for (int g = 0; g < 2; g++) {
        if (!gens[g]) continue;
        const auto &code = *gens[g];
        const auto &off = *goffs[g];
        if ((int64_t)off.size() != p->n_trees + 1 || off[0] != 0 || off.back() != (int32_t)code.size()) return bad("generic offsets", -1, g, off.size());
        for (int64_t t = 0; t < p->n_trees; t++) {
            if (off[(size_t)t + 1] <= off[(size_t)t]) return bad("empty tree", t, 0, 0);
            for (int32_t i = off[(size_t)t]; i < off[(size_t)t + 1]; i++) {
                const Instr &ins = code[(size_t)i];
                const uint32_t op = ins.hdr & H_OP_MASK, src = (ins.hdr >> H_SRC_SHIFT) & H_SRC_MASK;
                const bool known = op == DOP_LOAD || (op >= DE_U_NEG && op < DE_U_LAST_) || (op >= DE_B_ADD && op < DE_B_LAST_) ||
                                   (op >= DE_T_FMA && op < DE_T_LAST_) || (op >= DOP_RSUB && op <= DOP_RPOW_ABS2);
                if (!known) return bad("unknown opcode", t, i, op);
                if (src == SRC_ROW && (int64_t)(ins.feat & 0xFFFFu) >= rows) return bad("operand row outside X + spill slots", t, i, ins.feat);
                if (src == SRC_PARAM && (int64_t)(ins.feat & 0xFFFFu) >= p->n_params) return bad("parameter row out of range", t, i, ins.feat);
                if (src != SRC_ACC && src != SRC_ROW && src != SRC_CONST && src != SRC_PARAM) return bad("operand kind", t, i, src);
                if ((ins.hdr & H_PUSH) && (int)((ins.hdr >> H_PUSH_SHIFT) & H_SLOT_MASK) >= p->n_slots) return bad("push slot", t, i, ins.hdr);
                if (op >= DE_T_FMA && op < DE_T_LAST_ && (int)((ins.hdr >> H_POPC_SHIFT) & H_SLOT_MASK) >= p->n_slots) return bad("ternary slot", t, i, ins.hdr);
            }
        }
    }
    for (size_t i = 0; i < p->bcode.size(); i++) {
        const BoundInstr &b = p->bcode[i];
        if (b.bop >= BOP_COUNT) return bad("bound handler id", -1, (int64_t)i, b.bop);
        if (!bop_is_const_source(b.bop) && b.bop != BOP_GEN_PARAM && b.bop != BOP_LOAD_CONST && b.bop != BOP_CHECK_ACC && b.bop != BOP_GEN_ACC &&
            b.bop != BOP_INJ_ACC && !(b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END && !((b.bop - BOP_UN_BASE) & 2)) &&
            (int64_t)(b.arg & 0xFFFFFFu) >= rows)
            return bad("bound operand row", -1, (int64_t)i, b.arg);
    }
    if (p->threaded) {
        uint64_t table[TOPX_TABLE];
        if (eval_handler_table(p->dtype, (p->options & DE_OPT_TURBO) != 0, table) != hipSuccess) return fail(c, DE_ERR_HIP, "handler table");
        std::vector<uint64_t> valid(table, table + TOPX_TABLE); // (the end-fused variants included)
        std::sort(valid.begin(), valid.end());
        const uint64_t lds_bytes = (uint64_t)(rows + (p->uses_params ? 2 : 0)) * trow_bytes(p->dtype);
        if ((int64_t)p->ccode_off.size() != p->n_trees + 1 || p->ccode.size() != p->tcode.size() + (size_t)p->n_trees + 1) return bad("chained layout", -1, 0, p->ccode.size());
        const bool f32 = p->dtype == DE_F32;
        for (int64_t t = 0; t < p->n_trees; t++) {
            const int32_t i0 = p->tcode_off[(size_t)t], i1 = p->tcode_off[(size_t)t + 1], h = p->ccode_off[(size_t)t];
            if (h != i0 + (int32_t)t + 1) return bad("chained offset", t, h, (uint64_t)i0);
            if (i1 > i0) { // a tree never starts by reading the accumulator: the end of the previous tree leaves it as it is
                const BoundInstr &f0 = p->fbcode[(size_t)i0];
                const uint32_t aux0 = f0.arg >> 24;
                const int deg0 = aux0 == (uint32_t)DOP_LOAD ? 0 : de_opcode_degree((int)aux0);
                if (top_reads_acc(f0.bop, deg0)) return bad("first instruction of a tree reads the accumulator", t, 0, f0.bop);
            }
            { // the header record (in front of the tree) carries the tree's record count: h_tree_skip steps over the tree with it
                const BoundInstr &hd = p->ccode[(size_t)h - 1];
                if (((f32 ? hd.arg : hd.lo) & ~DE_HDR_FUSED_END) != (uint32_t)(i1 - i0)) return bad("tree header does not carry the tree's length", t, 0, f32 ? hd.arg : hd.lo);
            }
            for (int32_t i = i0; i <= i1; i++) {
                const BoundInstr &r = p->ccode[(size_t)(h + (i - i0))];
                const BoundInstr &q = p->ccode[(size_t)(h + (i - i0) - 1)]; // a record's handler is named by the record in front of it
                const uint64_t addr = f32 ? (((uint64_t)q.hi << 32) | q.lo) : ((table[0] & 0xFFFFFFFF00000000ull) | q.arg);
                if (!std::binary_search(valid.begin(), valid.end(), addr)) return bad("handler address not in the device table", t, i - i0, addr);
                // an end-fused last instruction (h_chain_end) steps over the end record: its own record names what the end record names
                const int ev = i1 - i0 >= 2 ? topx_endv_of(p->fbcode[(size_t)i1 - 1].bop) : -1;
                const uint64_t last_plain = p->handler_base + p->tcode[(size_t)i1 - 1].bop;
                const BoundInstr &lastq = p->ccode[(size_t)(h + (i1 - 1 - i0) - 1)];
                const uint64_t last_addr = f32 ? (((uint64_t)lastq.hi << 32) | lastq.lo) : ((table[0] & 0xFFFFFFFF00000000ull) | lastq.arg);
                const bool fused_end = ev >= 0 && last_addr == p->endv_handler[ev] && last_addr != last_plain;
                if (i == i0 && (((f32 ? p->ccode[(size_t)h - 1].arg : p->ccode[(size_t)h - 1].lo) & DE_HDR_FUSED_END) != 0) != fused_end)
                    return bad("tree header's end-fused bit", t, 0, (uint64_t)fused_end);
                if (i == i1) {
                    const BoundInstr &e = p->ccode[(size_t)(h + (i1 - i0))]; // the end record itself names the next tree's first handler
                    const uint64_t after = f32 ? (((uint64_t)e.hi << 32) | e.lo) : ((table[0] & 0xFFFFFFFF00000000ull) | e.arg);
                    if (fused_end ? addr != after : addr != p->end_handler) return bad("tree does not end in the end record", t, i - i0, addr);
                    if (r.bop != (uint32_t)t) return bad("end record does not name its tree", t, i - i0, r.bop);
                    continue;
                }
                const BoundInstr &fb = p->fbcode[(size_t)i];
                if (i == i1 - 1 ? (addr != last_plain && !fused_end) : addr != p->handler_base + p->tcode[(size_t)i].bop)
                    return bad("record / threaded code mismatch", t, i - i0, addr);
                if (fb.bop >= TOPX_COUNT) return bad("fused handler id", t, i - i0, fb.bop);
                const bool no_row = top_is_const_source(fb.bop) || fb.bop == BOP_CHECK_ACC || fb.bop == BOP_GEN_ACC || fb.bop == BOP_INJ_ACC ||
                                    fb.bop == BOP_GEN_PARAM || (fb.bop >= BOP_UN_BASE && fb.bop < BOP_UN_END && !((fb.bop - BOP_UN_BASE) & 2)) ||
                                    (fb.bop >= TOPX_UN_BASE && fb.bop < TOPX_BIN_BASE && ((fb.bop - TOPX_UN_BASE) & 1)) ||
                                    (fb.bop >= TOPX_BIN_BASE && ((fb.bop - TOPX_BIN_BASE) & 1));
                if (!no_row) {
                    const uint64_t off = r.bop & 0xFFFFFFu;
                    if (off % trow_bytes(p->dtype) != 0 || off + trow_bytes(p->dtype) > lds_bytes) return bad("LDS operand offset outside the launch's allocation", t, i - i0, r.bop);
                    const bool pushes = (fb.bop >= TOP_LOADROW_BASE && fb.bop < TOP_LOADCONST_PUSH && ((fb.bop - TOP_LOADROW_BASE) & 2)) ||
                                        (fb.bop >= TOP_UNROW_BASE && fb.bop < TOP_BINROWC_BASE && ((fb.bop - TOP_UNROW_BASE) & 2)) ||
                                        (fb.bop >= TOP_BIN2_BASE && fb.bop < TOP_COUNT && ((fb.bop - TOP_BIN2_BASE) & 1));
                    if (pushes) {
                        const int64_t prow = (int64_t)(off / trow_bytes(p->dtype)) + (int8_t)(r.bop >> 24);
                        if (prow < p->n_features || prow >= spill_end) return bad("push row of a superinstruction outside the spill slots", t, i - i0, r.bop);
                    }
                    if (fb.bop >= TOP_BIN2_BASE && fb.bop < TOP_COUNT && !(((fb.bop - TOP_BIN2_BASE) >> 2) & 1)) { // row-row: second row by distance
                        const int64_t second = (int64_t)off + (int32_t)(f32 ? r.arg : r.lo);
                        if (second < 0 || second % (int64_t)trow_bytes(p->dtype) != 0 || (uint64_t)second + trow_bytes(p->dtype) > lds_bytes) return bad("second operand row of a two-operand form", t, i - i0, (uint64_t)second);
                    }
                }
                if (fb.bop == BOP_GEN_PARAM && (f32 ? r.arg : r.lo) != (uint32_t)rows * (uint32_t)trow_bytes(p->dtype)) return bad("class-row offset of a parameter operand", t, i - i0, r.arg);
            }
        }
    }
    return DE_OK;
}

int64_t de_program_dump(const de_program_t *p, int64_t tree, uint32_t *words, int64_t cap, int which) {
    if (!p || tree < 0 || tree >= p->n_trees) return -DE_ERR_INVALID_ARG;
    if (which == 1) { // metadata: n_slots, host_ok_eval, host_ok_grad, uses_params
        if (cap < 4) return -DE_ERR_INVALID_ARG;
        words[0] = (uint32_t)p->n_slots;
        words[1] = p->host_ok_eval[(size_t)tree];
        words[2] = p->host_ok_grad[(size_t)tree];
        words[3] = p->uses_params;
        return 4;
    }
    if (which == 2) { // bound instructions (de_bind.h)
        const int32_t b0 = p->bcode_off[(size_t)tree], b1 = p->bcode_off[(size_t)tree + 1];
        const int64_t nb = (int64_t)(b1 - b0) * 4;
        if (!words || cap < nb) return nb;
        std::memcpy(words, p->bcode.data() + b0, (size_t)nb * 4);
        return nb;
    }
    if (which == 3) { // fused (superinstruction) form of the threaded kernel; empty when that kernel is not in use
        if (!p->threaded) return 0;
        const int32_t b0 = p->tcode_off[(size_t)tree], b1 = p->tcode_off[(size_t)tree + 1];
        const int64_t nb = (int64_t)(b1 - b0) * 4;
        if (!words || cap < nb) return nb;
        std::memcpy(words, p->fbcode.data() + b0, (size_t)nb * 4);
        return nb;
    }
    const int32_t i0 = p->code_off[(size_t)tree], i1 = p->code_off[(size_t)tree + 1];
    const int64_t nw = (int64_t)(i1 - i0) * 4;
    if (!words || cap < nw) return nw;
    std::memcpy(words, p->code.data() + i0, (size_t)nw * 4);
    return nw;
}

int64_t de_lower_tape(int dtype, const de_tape_node_t *nodes, int64_t n_nodes, const void *consts,
                      int64_t n_consts, int32_t n_features, int32_t n_params, uint32_t options, uint32_t *words,
                      int64_t cap, int32_t *meta) {
    if (!nodes || (dtype != DE_F32 && dtype != DE_F64) || (n_consts > 0 && !consts)) return -DE_ERR_INVALID_ARG;
    try {
        LowerOptions lo;
        lo.early_exit = (options & DE_OPT_EARLY_EXIT) != 0;
        lo.fuse1 = (options & DE_OPT_FUSE_DEG1) != 0;
        lo.fuse2 = (options & DE_OPT_FUSE_DEG2) != 0;
        lo.bumper = (options & DE_OPT_BUMPER_CHECKS) != 0;
        lo.n_features = n_features;
        lo.n_params = n_params;
        lo.dtype = dtype;
        for (int64_t i = 0; i < n_nodes; i++) // a CSE tape (GraphNode sharing) announces itself by its markers
            if ((nodes[i].degree == 1 && nodes[i].op == DE_OP_SHARE) || (nodes[i].degree == 0 && nodes[i].op == DE_LEAF_SHARED)) lo.cse = true;
        TreeProgram tp;
        std::string why;
        int rc = lower_tree(nodes, n_nodes, n_consts, lo, &tp, &why);
        if (rc != DE_OK) return -rc;
        bool ok_eval = true, ok_grad = true;
        for (int64_t k = 0; k < n_consts; k++) {
            const double v = dtype == DE_F32 ? (double)static_cast<const float *>(consts)[k]
                                             : static_cast<const double *>(consts)[k];
            if (tp.const_instr[(size_t)k] >= 0) write_imm(tp.code[(size_t)tp.const_instr[(size_t)k]], dtype, v);
            const bool fin = finite_in(dtype, v);
            ok_grad = ok_grad && fin;
            const uint8_t ch = tp.const_checks[(size_t)k];
            if (!fin && ((ch & CONST_CHECK_ALWAYS) || (lo.early_exit && (ch & CONST_CHECK_EE)))) ok_eval = false;
        }
        if (meta) {
            meta[0] = tp.n_slots;
            meta[1] = ok_eval;
            meta[2] = ok_grad;
            meta[3] = tp.uses_params;
        }
        const int64_t nw = (int64_t)tp.code.size() * 4;
        if (!words || cap < nw) return nw;
        std::memcpy(words, tp.code.data(), (size_t)nw * 4);
        return nw;
    } catch (...) {
        return -DE_ERR_HIP;
    }
}

int64_t de_lower_tape_stage(int dtype, const de_tape_node_t *nodes, int64_t n_nodes, const void *consts,
                            int64_t n_consts, int32_t n_features, int32_t n_params, uint32_t options, int stage,
                            uint32_t *words, int64_t cap) {
    if (stage != 2 && stage != 3) return -DE_ERR_INVALID_ARG;
    std::vector<uint32_t> g;
    int64_t nw = de_lower_tape(dtype, nodes, n_nodes, consts, n_consts, n_features, n_params, options, nullptr, 0, nullptr);
    if (nw < 0) return nw;
    try {
        g.resize((size_t)nw);
        nw = de_lower_tape(dtype, nodes, n_nodes, consts, n_consts, n_features, n_params, options, g.data(), nw, nullptr);
        if (nw < 0) return nw;
        std::vector<BoundInstr> b, f;
        bind_tree(reinterpret_cast<const Instr *>(g.data()), (size_t)nw / 4, (options & DE_OPT_EARLY_EXIT) != 0, n_features, &b);
        if (stage == 3) fuse_tree(b.data(), b.size(), &f);
        const std::vector<BoundInstr> &o = stage == 3 ? f : b;
        const int64_t n = (int64_t)o.size() * 4;
        if (!words || cap < n) return n;
        std::memcpy(words, o.data(), (size_t)n * 4);
        return n;
    } catch (...) {
        return -DE_ERR_HIP;
    }
}

// ---------------------------------------------------------------------------
// Stage a caller buffer: device pointers are used in place; host pointers are
// copied into context scratch (and copied back by the caller of this helper).
struct Staged {
    void *dev = nullptr;
    bool staged = false;
};
static int stage_in(de_ctx *c, DevBuf &buf, const void *user, size_t bytes, Staged *s) {
    s->dev = const_cast<void *>(user);
    s->staged = false;
    if (!user || bytes == 0 || is_device_ptr(user)) return DE_OK;
    HIP_TRY(c, buf.reserve(bytes));
    HIP_TRY(c, hipMemcpyAsync(buf.p, user, bytes, hipMemcpyHostToDevice, c->stream));
    s->dev = buf.p;
    s->staged = true;
    return DE_OK;
}
static int stage_out(de_ctx *c, DevBuf &buf, void *user, size_t bytes, Staged *s) {
    s->dev = user;
    s->staged = false;
    if (!user || bytes == 0 || is_device_ptr(user)) return DE_OK;
    HIP_TRY(c, buf.reserve(bytes));
    s->dev = buf.p;
    s->staged = true;
    return DE_OK;
}

static int check_param_args(de_ctx *c, const de_program *p, const de_param_args_t *pa, int64_t N) {
    if (!p->uses_params) return DE_OK;
    if (!pa || !pa->params || !pa->classes)
        return fail(c, DE_ERR_INVALID_ARG, "program has parameter leaves: params/classes required "
                                           "(reference: \"You must pass the `classes::Vector` argument\")");
    if (pa->ld_params < p->n_params || pa->n_classes <= 0) return fail(c, DE_ERR_INVALID_ARG, "bad parameter matrix shape");
    // `@assert maximum(classes) <= size(parameters, 2)` (src/ParametricExpression.jl:378-379): checked here when the ids are
    // host memory; ids already on the device are the caller's to check (the kernels clamp them, so a bad id cannot fault)
    if (N > 0 && !is_device_ptr(pa->classes)) {
        int64_t lo = pa->class_base, hi = pa->class_base;
        if (pa->classes_is_i64) {
            const int64_t *q = static_cast<const int64_t *>(pa->classes);
            for (int64_t j = 0; j < N; j++) { lo = std::min(lo, q[j]); hi = std::max(hi, q[j]); }
        } else {
            const int32_t *q = static_cast<const int32_t *>(pa->classes);
            for (int64_t j = 0; j < N; j++) { lo = std::min<int64_t>(lo, q[j]); hi = std::max<int64_t>(hi, q[j]); }
        }
        if (lo < pa->class_base || hi - pa->class_base >= pa->n_classes)
            return fail(c, DE_ERR_OUT_OF_RANGE, "class id outside [%d, %lld): maximum(classes) <= size(parameters, 2) violated",
                        (int)pa->class_base, (long long)(pa->class_base + pa->n_classes));
    }
    return DE_OK;
}

struct LossReq {
    const void *y, *w;
    int32_t kind;
    void *loss;
};

struct CertReq {
    uint8_t *certified; // host, n_trees
    double *max_abs;    // host, n_trees, or null
};
static int eval_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                     void *out, int64_t ld_out, uint8_t *ok, const LossReq *lr, const CertReq *cr = nullptr);
static int ensure_cert_program(de_ctx *c, de_program *p);

// The certificate program (de_eval_sum_certificate): the eval program's generic form with the result of EVERY operator validity-tested —
// the exact elision of de_lower.cpp (a test is dropped when the value's consumer maps a non-finite input onto a tested non-finite output)
// keeps the FLAG exact but drops values the reference still sums — bound for the flat-switch kernel; plus, per tree, the largest
// |constant operand| (deg0_eval of a constant is an array of N copies: the reference sums that too).  A superset of the arrays the
// reference sums (the inner values of its fused 2/3-node kernels are never materialised there): sound, slightly conservative.
// Cached per constants generation (it used to be rebuilt serially, with a stream synchronisation and an upload, at EVERY call);
// built on the pool of host threads like every other per-tree pass.
static int ensure_cert_program(de_ctx *c, de_program *p) {
    if (p->cert_gen == p->consts_gen && p->d_cert_code && p->d_cert_off) return DE_OK;
    const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
    const std::vector<int32_t> &off = p->folded ? p->fcode_off : p->code_off;
    const int prb = p->prows ? p->n_features + p->n_slots : -1;
    std::vector<BoundInstr> bc;
    std::vector<int32_t> boff;
    p->cert_cmax.assign((size_t)p->n_trees, 0.0);
    build_stream_by_trees(p->n_trees, &bc, &boff, [&](int64_t t, std::vector<BoundInstr> *out) {
        std::vector<Instr> tmp(src.begin() + off[(size_t)t], src.begin() + off[(size_t)t + 1]);
        double cm = 0.0;
        for (Instr &ins : tmp) {
            if ((ins.hdr & H_OP_MASK) != DOP_LOAD) ins.hdr |= H_CHECK_OUT;
            if (((ins.hdr >> H_SRC_SHIFT) & H_SRC_MASK) == SRC_CONST) {
                const double v = p->dtype == DE_F32 ? (double)ins.imm.f32 : ins.imm.f64;
                if (v == v) cm = std::max(cm, std::fabs(v));
            }
        }
        p->cert_cmax[(size_t)t] = cm;
        bind_tree(tmp.data(), tmp.size(), true, p->n_features, out, prb);
    });
    bc.push_back(BoundInstr{0u, 0u, 0u, 0u}); // (the kernel prefetches pc + 1)
    HIP_TRY(c, hipSetDevice(c->device));
    if (p->cert_cap < bc.size()) {
        if (p->d_cert_code) (void)hipFree(p->d_cert_code);
        p->d_cert_code = nullptr;
        p->cert_cap = 0;
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_cert_code), bc.size() * sizeof(BoundInstr)));
        p->cert_cap = bc.size();
    }
    if (!p->d_cert_off) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_cert_off), boff.size() * sizeof(int32_t)));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // (an earlier certificate launch may still read the buffers)
    HIP_TRY(c, hipMemcpy(p->d_cert_code, bc.data(), bc.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(p->d_cert_off, boff.data(), boff.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    p->cert_gen = p->consts_gen;
    return DE_OK;
}

int de_eval_sum_certificate(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                            uint8_t *ok, uint8_t *certified, double *max_abs) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (N < 0 || !ok || !certified || (p->n_trees > 0 && N > 0 && !X)) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (is_device_ptr(certified) || (max_abs && is_device_ptr(max_abs))) return fail(c, DE_ERR_INVALID_ARG, "certified / max_abs are host arrays");
    if (p->direct) return fail(c, DE_ERR_UNSUPPORTED, "de_eval_sum_certificate needs the LDS-tiled kernel (feature matrix too wide)");
    if (!(p->options & DE_OPT_EARLY_EXIT) || N == 0) {
        // early_exit = false: the reference sums nothing (src/Evaluate.jl:16-32 are no-ops), the flag is the constant part alone; N = 0: sum(empty) = 0
        for (int64_t t = 0; t < p->n_trees; t++) { certified[t] = 1; if (max_abs) max_abs[t] = 0.0; }
        if (is_device_ptr(ok)) { HIP_TRY(c, hipSetDevice(c->device)); HIP_TRY(c, hipMemcpy(ok, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice)); }
        else std::memcpy(ok, p->host_ok_eval.data(), (size_t)p->n_trees);
        return DE_OK;
    }
    const CertReq cr{certified, max_abs};
    DE_NOTHROW(c, eval_impl(c, p, X, N, ldX, pa, nullptr, N, ok, nullptr, &cr)); // (builds host vectors: no exception may leave the C ABI)
}

int de_eval(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
            void *out, int64_t ld_out, uint8_t *ok) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (N < 0 || !ok || (p->n_trees > 0 && N > 0 && (!X || !out))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (ld_out < N) return fail(c, DE_ERR_INVALID_ARG, "ld_out < N");
    DE_NOTHROW(c, eval_impl(c, p, X, N, ldX, pa, out, ld_out, ok, nullptr));
}

int de_eval_loss(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                 const void *y, const void *w, int32_t loss_kind, void *loss, uint8_t *ok) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (N < 0 || !ok || (p->n_trees > 0 && (!loss || (N > 0 && (!X || !y))))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (loss_kind != DE_LOSS_L2 && loss_kind != DE_LOSS_L1) return fail(c, DE_ERR_INVALID_ARG, "unknown loss_kind %d", loss_kind);
    if (p->direct || !p->threaded)
        return fail(c, DE_ERR_UNSUPPORTED, "de_eval_loss needs the LDS-tiled kernel (feature matrix too wide for this build)");
    const LossReq lr{y, w, loss_kind, loss};
    DE_NOTHROW(c, eval_impl(c, p, X, N, ldX, pa, nullptr, N, ok, &lr));
}

static int eval_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                     void *out, int64_t ld_out, uint8_t *ok, const LossReq *lr, const CertReq *cr) {
    if (p->ctx != c) return fail(c, DE_ERR_INVALID_ARG, "program belongs to another context");
    if (ldX < p->n_features) return fail(c, DE_ERR_INVALID_ARG, "ldX < n_features");
    int rc = check_param_args(c, p, pa, N);
    if (rc != DE_OK) return rc;
    if (p->n_trees == 0) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    const bool ok_dev = is_device_ptr(ok);
    if (N == 0) { // nothing to evaluate: only the constant part of the flag (sum(empty) is finite)
        if (ok_dev) HIP_TRY(c, hipMemcpyAsync(ok, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
        else std::memcpy(ok, p->host_ok_eval.data(), (size_t)p->n_trees);
        if (lr) { // empty sum = 0; NaN where a constant already failed the flag
            std::vector<unsigned char> z((size_t)p->n_trees * es);
            for (int64_t t = 0; t < p->n_trees; t++) {
                const double v = p->host_ok_eval[(size_t)t] ? 0.0 : std::nan("");
                if (p->dtype == DE_F32) reinterpret_cast<float *>(z.data())[t] = (float)v;
                else reinterpret_cast<double *>(z.data())[t] = v;
            }
            if (is_device_ptr(lr->loss)) {
                HIP_TRY(c, hipMemcpyAsync(lr->loss, z.data(), z.size(), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            } else std::memcpy(lr->loss, z.data(), z.size());
        }
        return DE_OK;
    }
    const bool direct = p->direct;

    Staged sX, sOut, sOk, sPar, sCls, sY, sW, sLoss;
    LossArgs la;
    std::memset(&la, 0, sizeof la);
    rc = stage_in(c, c->sX, X, (size_t)ldX * (size_t)N * es, &sX);
    if (rc) return rc;
    if (lr) {
        rc = stage_in(c, c->sY, lr->y, (size_t)N * es, &sY);
        if (rc) return rc;
        if (lr->w) {
            rc = stage_in(c, c->sW, lr->w, (size_t)N * es, &sW);
            if (rc) return rc;
        }
        rc = stage_out(c, c->sLoss, lr->loss, (size_t)p->n_trees * es, &sLoss);
        if (rc) return rc;
        size_t pb = 0, sb = 0;
        loss_scratch_bytes(p->dtype, p->n_trees, N, &pb, &sb);
        HIP_TRY(c, c->sPartial.reserve(pb));
        HIP_TRY(c, c->sSeg.reserve(sb));
        la.y = sY.dev;
        la.w = lr->w ? sW.dev : nullptr;
        la.kind = lr->kind;
        la.partial = c->sPartial.p;
        la.seg_sum = c->sSeg.p;
        la.loss = sLoss.dev;
    } else if (!cr) {
        rc = stage_out(c, c->sOut, out, ((size_t)(p->n_trees - 1) * (size_t)ld_out + (size_t)N) * es, &sOut);
        if (rc) return rc;
    }
    if (cr) {
        rc = ensure_cert_program(c, p);
        if (rc) return rc;
        HIP_TRY(c, c->sCert.reserve((size_t)p->n_trees * es));
        HIP_TRY(c, hipMemsetAsync(c->sCert.p, 0, (size_t)p->n_trees * es, c->stream));
    }
    // ok[] starts as the host-side (constant) part of the flag; the kernel only clears bytes
    if (ok_dev) {
        sOk.dev = ok;
    } else {
        HIP_TRY(c, c->sOk.reserve((size_t)p->n_trees));
        sOk.dev = c->sOk.p;
        sOk.staged = true;
    }
    if (p->d_ok_eval) HIP_TRY(c, hipMemcpyAsync(sOk.dev, p->d_ok_eval, (size_t)p->n_trees, hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(c, hipMemcpyAsync(sOk.dev, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
    if (p->uses_params) {
        rc = stage_in(c, c->sParams, pa->params, (size_t)pa->ld_params * (size_t)pa->n_classes * es, &sPar);
        if (rc) return rc;
        rc = stage_in(c, c->sClasses, pa->classes, (size_t)N * (pa->classes_is_i64 ? 8 : 4), &sCls);
        if (rc) return rc;
    }
    EvalArgs a;
    std::memset(&a, 0, sizeof a);
    a.code = p->d_code;
    a.code_off = p->d_code_off;
    a.n_trees = (int32_t)p->n_trees;
    a.n_slots = p->n_slots + (p->prows ? p->n_params : 0); // (LDS rows behind X: spill slots, then the staged parameter rows)
    a.prow_base = p->prows ? p->n_features + p->n_slots : 0;
    a.n_prows = p->prows ? p->n_params : 0;
    a.uses_params = p->uses_params;
    a.X = sX.dev;
    a.N = N;
    a.ldX = ldX;
    a.F = p->n_features;
    a.out = sOut.dev;
    a.ld_out = ld_out;
    a.ok = static_cast<uint8_t *>(sOk.dev);
    if (p->uses_params) {
        a.params = sPar.dev;
        a.ld_params = pa->ld_params;
        a.n_classes = pa->n_classes;
        a.classes = sCls.dev;
        a.classes_is_i64 = pa->classes_is_i64;
        a.class_base = pa->class_base;
    }
    a.early_exit = (p->options & DE_OPT_EARLY_EXIT) != 0;
    a.skip_flagged = a.early_exit && !(p->options & DE_OPT_FULL_EVAL) && tree_skip_enabled();
    a.turbo = (p->options & DE_OPT_TURBO) != 0;
    a.threaded = p->threaded && !direct;
    a.direct = direct;
    a.loss = lr ? &la : nullptr;
    HIP_TRY(c, c->sPrio.reserve((size_t)3 * DE_PRIO_MAX_F * sizeof(unsigned long long)));
    a.prio_keys = c->sPrio.p;
    a.prio_keys_ready = !sX.staged && dataset_keys(c, p->dtype, X, N, ldX, p->n_features, &a.prio_keys);
    a.compact_code = p->d_compact_code;
    a.compact_ints = p->d_compact_ints;
    if (cr) { // the certificate pass: the un-elided program through the flat-switch kernel's CERT variant, nothing stored
        a.code = p->d_cert_code;
        a.code_off = p->d_cert_off;
        a.threaded = false;
        a.cert_max = c->sCert.p;
        a.out = nullptr;
        a.prio_keys = nullptr;
        a.compact_code = nullptr;
        a.compact_ints = nullptr;
    }
    HIP_TRY(c, time_begin(c));
    a.compacted = &p->last_compacted;
    p->last_compacted = false;
    HIP_TRY(c, launch_eval(p->dtype, a, c->stream, &c->last_kernel));
    HIP_TRY(c, time_end(c));
    if (sLoss.staged) HIP_TRY(c, hipMemcpyAsync(lr->loss, sLoss.dev, (size_t)p->n_trees * es, hipMemcpyDeviceToHost, c->stream));
    if (sOut.staged) {
        if (ld_out == N) // one block (the constant-folding population is 10^3..10^5 one-sample rows)
            HIP_TRY(c, hipMemcpyAsync(out, sOut.dev, (size_t)p->n_trees * (size_t)N * es, hipMemcpyDeviceToHost, c->stream));
        else
            for (int64_t t = 0; t < p->n_trees; t++) // rows are strided in the caller's buffer
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(out) + (size_t)t * (size_t)ld_out * es,
                                          static_cast<char *>(sOut.dev) + (size_t)t * (size_t)ld_out * es,
                                          (size_t)N * es, hipMemcpyDeviceToHost, c->stream));
    }
    if (sOk.staged) HIP_TRY(c, hipMemcpyAsync(ok, sOk.dev, (size_t)p->n_trees, hipMemcpyDeviceToHost, c->stream));
    if (cr) {
        // certified[t]: the reference's `complete` provably equals ok[t].  It tests isfinite(sum(x)) over N values (src/ValueInterface.jl:9)
        // where the device tests every element: the two differ only when all elements are finite and a (partial) sum overflows — impossible
        // while N * max|x| stays below the largest finite value.  ok[t] == 0 means some element is non-finite: its sum is too.
        std::vector<unsigned char> mx((size_t)p->n_trees * es);
        std::vector<uint8_t> okh((size_t)p->n_trees);
        HIP_TRY(c, hipMemcpyAsync(mx.data(), c->sCert.p, mx.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(okh.data(), sOk.dev, okh.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const double top = p->dtype == DE_F32 ? (double)std::numeric_limits<float>::max() : std::numeric_limits<double>::max();
        for (int64_t t = 0; t < p->n_trees; t++) {
            double m = p->dtype == DE_F32 ? (double)reinterpret_cast<float *>(mx.data())[t] : reinterpret_cast<double *>(mx.data())[t];
            m = std::max(m, p->cert_cmax[(size_t)t]);
            if (cr->max_abs) cr->max_abs[t] = m;
            cr->certified[t] = (!okh[(size_t)t] || m * (double)N * 1.001 < top || !(m == m)) ? 1 : 0; // (the margin: the summation's own roundings)
            if (!std::isfinite(m) && okh[(size_t)t]) cr->certified[t] = 0;
        }
    }
    if (sX.staged || sOut.staged || sOk.staged || sPar.staged || sCls.staged || sY.staged || sW.staged || sLoss.staged)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (sOut.staged && a.skip_flagged) {
        // host output buffer: the rows of incomplete trees were only partly written on the device, and the staging buffer is shared by
        // every program of the context — they would carry an earlier call's data.  NaN-fill them (what the callable sugar does anyway,
        // src/EvaluationHelpers.jl:29-33); the flags come from the caller's host array or, for a device `ok`, from a copy.
        std::vector<uint8_t> okh;
        const uint8_t *okp = ok;
        if (ok_dev) {
            okh.resize((size_t)p->n_trees);
            HIP_TRY(c, hipMemcpy(okh.data(), ok, (size_t)p->n_trees, hipMemcpyDeviceToHost));
            okp = okh.data();
        }
        for (int64_t t = 0; t < p->n_trees; t++) {
            if (okp[t]) continue;
            if (p->dtype == DE_F32) std::fill_n(static_cast<float *>(out) + (size_t)t * (size_t)ld_out, (size_t)N, std::nanf(""));
            else std::fill_n(static_cast<double *>(out) + (size_t)t * (size_t)ld_out, (size_t)N, std::nan(""));
        }
    }
    return DE_OK;
}

int de_eval_tree_array(de_ctx_t *c, int dtype, const de_tape_node_t *nodes, int64_t n_nodes, const void *consts,
                       int64_t n_consts, const void *X, int32_t n_features, int64_t N, uint32_t options, void *out,
                       uint8_t *ok) {
    if (!c) return DE_ERR_INVALID_ARG;
    const int64_t noff[2] = {0, n_nodes}, coff[2] = {0, n_consts};
    de_program_t *p = nullptr;
    int rc = de_program_create(c, dtype, nodes, noff, 1, consts, coff, n_features, 0, options, &p);
    if (rc != DE_OK) return rc;
    rc = de_eval(c, p, X, N, n_features, nullptr, out, N, ok);
    if (rc == DE_OK) rc = de_ctx_synchronize(c);
    de_program_destroy(p);
    return rc;
}

static int ensure_generic_code(de_ctx *c, de_program *p) {
    if (p->gcode_stale || !p->d_gcode) {
        // gradients flow through constant subtrees, so this is the UNFOLDED program; every value the
        // reference tests is tested (ee binding) whatever the eval options were
        p->gt_valid = false;
        p->rt_valid = false;
        // bound per worker into a vector of its own, then concatenated (10^4 trees: 3 ms on one thread)
        build_stream_by_trees<BoundInstr>(p->n_trees, &p->gbcode, &p->gbcode_off, [&](int64_t t, std::vector<BoundInstr> *out) {
            const int32_t i0 = p->code_off[(size_t)t], i1 = p->code_off[(size_t)t + 1];
            bind_tree(p->code.data() + i0, (size_t)(i1 - i0), true, p->n_features, out);
        });
        match_const_sites(p->code, p->code_off, p->gbcode, p->gbcode_off, p->n_trees,
                          [](const BoundInstr &b) { return bop_is_const_source(b.bop); }, &p->gbsite);
        p->gtsite_of_gb.clear();
        p->site_gen++;
    }
    if (!p->d_gcode) {
        HIP_TRY(c, prog_malloc(c, reinterpret_cast<void **>(&p->d_gcode), (p->gbcode.size() + 1) * sizeof(BoundInstr)));
        HIP_TRY(c, hipMemset(p->d_gcode, 0, (p->gbcode.size() + 1) * sizeof(BoundInstr)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_gcode_off), p->gbcode_off.size() * sizeof(int32_t)));
        HIP_TRY(c, hipMemcpy(p->d_gcode_off, p->gbcode_off.data(), p->gbcode_off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        p->gcode_stale = true;
    }
    if (p->gcode_stale) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!p->gbcode.empty())
            HIP_TRY(c, hipMemcpy(p->d_gcode, p->gbcode.data(), p->gbcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        p->gcode_stale = false;
    }
    return DE_OK;
}

// Threaded form of the gradient program (de_grad_threaded.hip) for `mode`.  Trees are grouped into
// buckets by gradient width n_grad(t, mode): widths 1..6 and 7-8 each run in the module built for that
// window (every seed is known here and compiled into the handler choice), wider trees in windows of 8
// with run-time seeds.  Fills g->threaded_code & co. when the program can be expressed this way; otherwise
// leaves them null and the flat-switch kernel runs.  Call after ensure_generic_code().
static int ensure_grad_threaded(de_ctx *c, de_program *p, int mode, const std::vector<int32_t> &ng, int64_t N, GradArgs *g) {
    g->threaded_code = nullptr;
    g->n_buckets = 0;
    const char *env = getenv("DE_GRAD_THREADED");
    if (env && *env == '0') return DE_OK;
    const int F = p->n_features, P = p->n_params;
    // Parameter leaves are LDS rows of their own: the kernel gathers params[:, class] into P rows behind the X rows when it
    // stages a tile (the reference's formulation, src/ParametricExpression.jl:381-389), so every hot handler serves them.
    const int FE = F + (p->uses_params ? P : 0);
    const bool hot_const_unary = !getenv("DE_NO_CONST_UNARY_HOT"); // unary operators outside the binder's hot set through hot handlers
    const bool fuse_push = true;                                   // PUSH + LOAD pairs as one instruction
    auto gun_of = [&](uint32_t op) { // hot unary index of a de_opcode (de_bind.h), or -1
        return hot_const_unary ? gun_index((int)op, DE_U_COS, DE_U_EXP, DE_U_SIN, DE_U_NEG, DE_U_SQUARE, DE_U_CUBE, DE_U_ABS, DE_U_LOG, DE_U_SAFE_LOG,
                                           DE_U_SQRT, DE_U_SAFE_SQRT, DE_U_TANH, DE_U_RELU) : -1;
    };
    // Two samples per lane double the buckets (launches) and the tile: they pay from ~10^5 samples on (10^4 trees x
    // 10^3 rows: 0.55 ms with them, 0.35 ms without; 10^3 trees x 10^6 rows: 12.1 against 13.4 ms)
    const char *envn = getenv("DE_GRAD_VS2_MIN_N");
    const bool wide = N >= (envn ? atoll(envn) : 65536);
    if (!(p->gt_valid && p->gt_mode == mode && p->gt_wide == wide)) {
        dbg_lap(nullptr);
        // bucket of a tree: (width index 0..6 = single window of width 1,2,3,4,5,6,8; 7 = several windows of 8)
        // x (samples per lane - 1).  The two-sample modules exist for Float32 windows <= 6; their rows are
        // twice as long, so they only pay while a workgroup's LDS stays small: at most DE_GRAD_VS2_ROWS (15) rows per wave.
        // width index 0..6 = single window of width 1,2,3,4,5,6,8; 7,8,9 = several windows of 8,5,6 (the
        // narrowest module that covers the gradient in ceil(G/8) windows: 9-10 rows -> 2x5, 11-12 -> 2x6, 17-18 -> 3x6)
        static const int WIDTH[10] = {1, 2, 3, 4, 5, 6, 8, 8, 5, 6};
        constexpr int NW = 10, NB = 2 * NW;
        const char *env2 = getenv("DE_GRAD_VS2_ROWS"); // most LDS rows per wave (X + parameters + slots) that still run two samples per lane
        const int vs2_rows = env2 ? atoi(env2) : 15;    // 15 rows x 512 B x 4 waves = 30.7 KB: 5 workgroups per CU
        std::vector<int32_t> tslots((size_t)p->n_trees, 0); // spill slots of each tree (rows >= F the code names)
        parallel_for_trees(p->n_trees, [&](int64_t t) {
            int32_t need = 0;
            for (int32_t i = p->gbcode_off[(size_t)t]; i < p->gbcode_off[(size_t)t + 1]; i++) {
                const BoundInstr &b = p->gbcode[(size_t)i];
                const uint32_t row = b.arg & 0xFFFFFFu;
                const bool names_row = b.bop == BOP_PUSH || b.bop == BOP_LOAD_ROW || b.bop == BOP_GEN_ROW || b.bop == BOP_TERN ||
                                       (b.bop >= BOP_BIN_BASE && b.bop < BOP_BIN_END && !((b.bop - BOP_BIN_BASE) & 2)) ||
                                       (b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END && ((b.bop - BOP_UN_BASE) & 2));
                if (names_row && row >= (uint32_t)F) need = std::max(need, (int32_t)(row - (uint32_t)F) + 1);
                if (b.bop == BOP_TERN && b.lo >= (uint32_t)F) need = std::max(need, (int32_t)(b.lo - (uint32_t)F) + 1);
            }
            tslots[(size_t)t] = need;
        });
        dbg_lap("grad threaded: spill slots per tree");
        auto bucket_of = [&](int64_t t) {
            const int32_t G = ng[(size_t)t];
            int w;
            if (G <= 6) w = G < 1 ? 0 : G - 1;
            else if (G <= 8) w = 6;
            else {
                const int windows = (G + 7) / 8, per = (G + windows - 1) / windows;
                w = per <= 5 ? 8 : (per <= 6 ? 9 : 7);
                if (!grad_threaded_has(p->dtype, WIDTH[w], 1)) w = 7;
            }
            const int rows2 = FE + std::max(tslots[(size_t)t] * (1 + WIDTH[w]), WIDTH[w]);
            const bool two = wide && p->dtype == DE_F32 && WIDTH[w] <= 6 && rows2 <= vs2_rows && grad_threaded_has(p->dtype, WIDTH[w], 2);
            return w + (two ? NW : 0);
        };
        int32_t count[NB] = {0}, maxg[NB] = {0}, slots[NB] = {0};
        for (int64_t t = 0; t < p->n_trees; t++) {
            const int32_t G = ng[(size_t)t];
            if (G > 240) return DE_OK; // gradient rows travel in 8 bits
            const int b = bucket_of(t);
            // Float64 states wider than 16 dwords are passed through scratch memory by the calling convention
            if (!grad_threaded_has(p->dtype, WIDTH[b % NW], 1 + b / NW)) return DE_OK;
            count[b]++;
            maxg[b] = std::max(maxg[b], G);
            slots[b] = std::max(slots[b], tslots[(size_t)t]);
        }
        const uint32_t es32 = p->dtype == DE_F32 ? 4u : 8u;
        std::vector<std::array<uint64_t, GOP_MAX>> tables(NB);
        uint64_t bases[NB] = {0};
        for (int b = 0; b < NB; b++) {
            if (!count[b]) continue;
            const int GC = WIDTH[b % NW], VS = 1 + b / NW;
            const uint64_t RBb = 64ull * VS * es32; // one wave's row
            const uint64_t rows = (uint64_t)FE + std::max<uint64_t>((uint64_t)slots[b] * (1 + GC), (uint64_t)GC);
            if (4 * rows * RBb > 160 * 1024) return DE_OK; // four waves' rows must fit the CU's LDS
            hipError_t st = grad_handler_table(p->dtype, GC, VS, tables[b].data());
            if (st != hipSuccess) return fail(c, DE_ERR_HIP, "gradient handler table: %s", hipGetErrorString(st));
            uint64_t base = tables[b][0];
            for (int i = 0; i < (int)gop_count(GC); i++) base = std::min<uint64_t>(base, tables[b][i]);
            for (int i = 0; i < (int)gop_count(GC); i++)
                if (tables[b][i] - base > 0xFFFFFFFFull) return DE_OK;
            bases[b] = base;
        }
        dbg_lap("grad threaded: buckets, handler tables");
        auto leaf_seed = [&](uint32_t f) -> uint32_t { return mode != DE_GRAD_CONSTANT ? (uint32_t)P + f : 0xFFu; };
        auto const_seed = [&](uint32_t ord) -> uint32_t {
            return mode == DE_GRAD_CONSTANT ? ord : (mode == DE_GRAD_BOTH ? (uint32_t)(P + F) + ord : 0xFFu);
        };
        p->gtcode.clear();
        p->gtcode_off.assign((size_t)p->n_trees + 1, 0);
        p->gtsite_of_gb.assign(p->gbcode.size(), -1);
        p->site_gen++;
        std::atomic<bool> ok{true};
        // encoded per worker into a vector of its own (sites = positions in that vector), concatenated afterwards
        std::vector<BoundInstr> parts[HOST_RANGES_MAX];
        std::vector<int32_t> tree_cnt((size_t)p->n_trees, 0);
        int64_t part_first[HOST_RANGES_MAX], part_last[HOST_RANGES_MAX];
        for (int k = 0; k < HOST_RANGES_MAX; k++) part_first[k] = part_last[k] = 0;
        parallel_tree_ranges(p->n_trees, [&](int wk, int64_t tb, int64_t te) {
        std::vector<BoundInstr> &out = parts[wk];
        part_first[wk] = tb;
        part_last[wk] = te;
        for (int64_t t = tb; t < te && ok; t++) {
            const size_t out_before = out.size();
            const int bkt = bucket_of(t);
            const int GC = WIDTH[bkt % NW];
            const uint32_t RB = 64u * (uint32_t)(1 + bkt / NW) * es32; // bytes of one wave's row
            const bool one_window = bkt % NW < 7; // then g0 = 0 and every seed is known here
            const uint64_t *table = tables[bkt].data();
            const uint64_t base = bases[bkt];
            auto slot_off = [&](uint32_t row) { return (uint32_t)((FE + (row - (uint32_t)F) * (1 + GC)) * RB); };
            // seed variant of a handler (de_bind.h): 0 run-time, 1 none, 2 + k
            auto seed_variant = [&](uint32_t sd) -> int { return !one_window ? 0 : (sd == 0xFFu ? 1 : (sd < (uint32_t)GC ? 2 + (int)sd : 0)); };
            for (int32_t i = p->gbcode_off[(size_t)t]; i < p->gbcode_off[(size_t)t + 1] && ok; i++) {
                const BoundInstr &b = p->gbcode[(size_t)i];
                const uint32_t row = b.arg & 0xFFFFFFu, aux = b.arg >> 24;
                BoundInstr o = b;
                int src = GSRC_ACC, sv = 0;
                auto row_operand = [&](bool rt = false) { // sets src, sv and o.arg for a row operand (rt: handler reads the seed at run time)
                    if (row < (uint32_t)F) {
                        const uint32_t sd = leaf_seed(row);
                        if (sd != 0xFFu && sd >= 0xF0u) ok = false;
                        src = GSRC_LEAF;
                        sv = rt ? 0 : seed_variant(sd);
                        o.arg = (row * RB) | (sv == 0 ? sd << 24 : 0u); // known seeds are compiled into the handler
                    } else {
                        src = GSRC_SLOT;
                        o.arg = slot_off(row);
                    }
                };
                auto const_operand = [&](uint32_t ord, uint32_t low, bool rt = false) {
                    const uint32_t sd = const_seed(ord);
                    if (sd != 0xFFu && sd >= 0xF0u) ok = false;
                    src = GSRC_CONST;
                    sv = rt ? 0 : seed_variant(sd);
                    o.arg = low | (sv == 0 ? sd << 24 : 0u);
                };
                auto param_operand = [&](uint32_t prm, bool rt = false) { // parameter row prm = LDS row F + prm, seed = its gradient row
                    const uint32_t sd = mode != DE_GRAD_CONSTANT ? prm : 0xFFu;
                    if (sd != 0xFFu && sd >= 0xF0u) ok = false;
                    src = GSRC_LEAF;
                    sv = rt ? 0 : seed_variant(sd);
                    o.arg = (((uint32_t)F + prm) * RB) | (sv == 0 ? sd << 24 : 0u);
                };
                uint32_t gop = 0;
                if (b.bop == BOP_CHECK_ROW) continue; // leaf operands are tested where they are read
                if (b.bop == BOP_LOAD_ROW) { row_operand(); gop = gop_load(GC, src, sv); }
                else if (b.bop == BOP_LOAD_CONST) { const_operand(b.arg & 0xFFFFu, 0); gop = gop_load(GC, src, sv); }
                else if (b.bop == BOP_PUSH && fuse_push && i + 1 < p->gbcode_off[(size_t)t + 1] &&
                         (p->gbcode[(size_t)i + 1].bop == BOP_LOAD_CONST ||
                          (p->gbcode[(size_t)i + 1].bop == BOP_LOAD_ROW && (p->gbcode[(size_t)i + 1].arg & 0xFFFFFFu) < (uint32_t)F))) {
                    // PUSH followed by the LOAD that starts the next subtree: one dispatch (g_pushload)
                    const BoundInstr &b2 = p->gbcode[(size_t)i + 1];
                    const uint32_t slot = slot_off(row);
                    if (b2.bop == BOP_LOAD_CONST) {
                        const_operand(b2.arg & 0xFFFFu, slot);
                        o.lo = b2.lo;
                        o.hi = b2.hi;
                        p->gtsite_of_gb[(size_t)i + 1] = (int32_t)out.size(); // the constant lives in the fused instruction
                    } else {
                        const uint32_t row2 = b2.arg & 0xFFFFFFu, sd = leaf_seed(row2);
                        if (sd != 0xFFu && sd >= 0xF0u) ok = false;
                        src = GSRC_LEAF;
                        sv = seed_variant(sd);
                        o.arg = (row2 * RB) | (sv == 0 ? sd << 24 : 0u);
                        o.lo = slot - row2 * RB; // byte distance row -> slot
                        o.hi = 0;
                    }
                    if (!ok) break;
                    o.bop = (uint32_t)(table[gop_pushload(GC, src, sv)] - base);
                    out.push_back(o);
                    i++; // the LOAD is part of this instruction
                    continue;
                }
                else if (b.bop == BOP_PUSH) { gop = gop_push(GC); o.arg = slot_off(row); }
                else if (b.bop == BOP_CHECK_ACC) { gop = gop_check_acc(GC); o.arg = 0; }
                else if (b.bop >= BOP_BIN_BASE && b.bop < BOP_BIN_END) {
                    const uint32_t v = b.bop - BOP_BIN_BASE;
                    if (v & 2) const_operand(b.arg & 0xFFFFu, 0);
                    else row_operand();
                    gop = gop_bin(GC, (int)(v >> 2), src, sv, (v & 1) != 0);
                } else if (b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END) {
                    const uint32_t v = b.bop - BOP_UN_BASE;
                    if (v & 2) row_operand();
                    else o.arg = 0;
                    gop = gop_un(GC, (int)(v >> 2), src, sv, (v & 1) != 0);
                } else if (b.bop == BOP_GEN_ROW && hot_const_unary && (aux == (uint32_t)DE_B_MAX || aux == (uint32_t)DE_B_MIN)) {
                    row_operand(); gop = gop_bin(GC, aux == (uint32_t)DE_B_MAX ? 6 : 7, src, sv, false); o.lo = o.hi = 0;
                } else if (b.bop == BOP_GEN_ROW && gun_of(aux) >= 0) { row_operand(); gop = gop_un(GC, gun_of(aux), src, sv, false); o.lo = o.hi = 0; }
                else if (b.bop == BOP_GEN_ROW) { row_operand(true); gop = gop_gen(GC, src); o.lo = aux; o.hi = 0; }
                else if (b.bop == BOP_GEN_CONST && hot_const_unary && (aux == (uint32_t)DE_B_MAX || aux == (uint32_t)DE_B_MIN)) {
                    const_operand(b.arg & 0xFFFFu, 0); gop = gop_bin(GC, aux == (uint32_t)DE_B_MAX ? 6 : 7, src, sv, false);
                }
                else if (b.bop == BOP_GEN_CONST && gun_of(aux) >= 0) {
                    // cos / exp / sin of a constant leaf (common: half the leaves are constants and the gradient program
                    // is not folded): load the constant, then the hot unary handler on the accumulator — not the generic
                    // handler (out-of-line operator switch, OCML functions, scratch traffic of its spills)
                    const_operand(b.arg & 0xFFFFu, 0);
                    o.bop = (uint32_t)(table[gop_load(GC, src, sv)] - base);
                    p->gtsite_of_gb[(size_t)i] = (int32_t)out.size();
                    out.push_back(o);
                    BoundInstr u = b;
                    u.arg = 0;
                    u.lo = u.hi = 0;
                    u.bop = (uint32_t)(table[gop_un(GC, gun_of(aux), GSRC_ACC, 0, false)] - base);
                    out.push_back(u);
                    continue;
                }
                else if (b.bop == BOP_GEN_CONST) { const_operand(b.arg & 0xFFFFu, aux << 16, true); gop = gop_gen(GC, GSRC_CONST); }
                else if (b.bop == BOP_GEN_ACC && gun_of(aux) >= 0) { gop = gop_un(GC, gun_of(aux), GSRC_ACC, 0, false); o.arg = 0; o.lo = o.hi = 0; }
                else if (b.bop == BOP_GEN_ACC) { gop = gop_gen(GC, GSRC_ACC); o.arg = 0; o.lo = aux; o.hi = 0; }
                else if (b.bop == BOP_GEN_PARAM) { // operand = parameter row (b.arg & 0xFFFF), operator aux: the leaf-operand handlers
                    const uint32_t prm = b.arg & 0xFFFFu;
                    int k = -1, ku = -1;
                    switch (aux) {
                    case DE_B_ADD: k = 0; break;
                    case DE_B_SUB: k = 1; break;
                    case DOP_RSUB: k = 2; break;
                    case DE_B_MUL: k = 3; break;
                    case DE_B_DIV: k = 4; break;
                    case DOP_RDIV: k = 5; break;
                    case DE_B_MAX: k = hot_const_unary ? 6 : -1; break;
                    case DE_B_MIN: k = hot_const_unary ? 7 : -1; break;
                    default: ku = gun_of(aux); break;
                    }
                    o.lo = o.hi = 0;
                    if (aux == (uint32_t)DOP_LOAD) { param_operand(prm); gop = gop_load(GC, src, sv); }
                    else if (k >= 0) { param_operand(prm); gop = gop_bin(GC, k, src, sv, false); }
                    else if (ku >= 0) { param_operand(prm); gop = gop_un(GC, ku, src, sv, false); }
                    else { param_operand(prm, true); gop = gop_gen(GC, GSRC_LEAF); o.lo = aux; }
                }
                else if (b.bop == BOP_TERN) {
                    if (row < (uint32_t)F || b.lo < (uint32_t)F) ok = false; // both operands are spilled duals
                    else { gop = gop_tern(GC); o.arg = slot_off(row) | (aux << 24); o.lo = slot_off(b.lo) - slot_off(row); o.hi = 0; }
                } else ok = false; // INJ_*: only bound with early_exit=false, never for gradients
                if (!ok) break;
                o.bop = (uint32_t)(table[gop] - base);
                p->gtsite_of_gb[(size_t)i] = (int32_t)out.size();
                out.push_back(o);
            }
            // the end record: every tree's chain finishes in g_end (the table slot of round 1's parameter handler)
            out.push_back(BoundInstr{(uint32_t)(table[gop_param(GC)] - base), 0u, 0u, 0u});
            tree_cnt[(size_t)t] = (int32_t)(out.size() - out_before);
        }
        });
        dbg_lap("grad threaded: encode (host threads)");
        if (!ok) { p->gtsite_of_gb.clear(); p->site_gen++; return DE_OK; }
        for (int64_t t = 0; t < p->n_trees; t++) p->gtcode_off[(size_t)t + 1] = p->gtcode_off[(size_t)t] + tree_cnt[(size_t)t];
        p->gtcode.resize((size_t)p->gtcode_off[(size_t)p->n_trees]);
        // ranges are in tree order; sites move from worker-local to global positions (the same partition as the encoding pass: worker k
        // copies the piece it encoded)
        parallel_tree_ranges(p->n_trees, [&](int k, int64_t tb, int64_t te) {
            if (part_first[k] != tb || part_last[k] != te) return; // (never: the partition depends on n_trees only)
            const int32_t base_k = p->gtcode_off[(size_t)tb];
            if (!parts[k].empty()) std::memcpy(static_cast<void *>(p->gtcode.data() + base_k), parts[k].data(), parts[k].size() * sizeof(BoundInstr));
            if (base_k != 0)
                for (int32_t i = p->gbcode_off[(size_t)tb]; i < p->gbcode_off[(size_t)te]; i++)
                    if (p->gtsite_of_gb[(size_t)i] >= 0) p->gtsite_of_gb[(size_t)i] += base_k;
        });
        {
            size_t copied = 0;
            for (int k = 0; k < HOST_RANGES_MAX; k++) copied += parts[k].size();
            if (copied != p->gtcode.size()) { p->gtsite_of_gb.clear(); p->site_gen++; return fail(c, DE_ERR_HIP, "gradient program: the host threads' partitions disagree"); }
        }
        dbg_lap("grad threaded: concatenate + sites");
        // the handler word of a record names the handler of the record BEHIND it, the end record names the tree's first handler
        // (de_grad_threaded.hip: a handler knows its successor at entry and jumps without waiting for the record it loads)
        parallel_for_trees(p->n_trees, [&](int64_t t) {
            const int32_t a0 = p->gtcode_off[(size_t)t], b0 = p->gtcode_off[(size_t)t + 1];
            if (b0 - a0 < 2) return;
            const uint32_t first = p->gtcode[(size_t)a0].bop;
            for (int32_t i = a0; i < b0 - 1; i++) p->gtcode[(size_t)i].bop = p->gtcode[(size_t)i + 1].bop;
            p->gtcode[(size_t)b0 - 1].bop = first;
        });
        dbg_lap("grad threaded: successor words");
        std::vector<int32_t> ids((size_t)p->n_trees);
        int32_t start[NB], run = 0;
        for (int b = 0; b < NB; b++) { start[b] = run; run += count[b]; }
        {
            int32_t fill[NB];
            for (int b = 0; b < NB; b++) fill[b] = start[b];
            for (int64_t t = 0; t < p->n_trees; t++) ids[(size_t)fill[bucket_of(t)]++] = (int32_t)t;
        }
        if (!p->d_gtcode) {
            // a unary operator on a constant leaf becomes two instructions: at most twice the bound program
            // ... plus one end record per tree and one of padding (every handler reads the record behind its own)
            const size_t gt_cap = 2 * p->gbcode.size() + (size_t)p->n_trees + 1;
            // (inside one 4 GiB window: the handlers bump the record pointer without a carry; a straddling allocation is set aside and redone)
            {
                const hipError_t ast = prog_malloc(c, reinterpret_cast<void **>(&p->d_gtcode), gt_cap * sizeof(BoundInstr));
                if (ast != hipSuccess) return fail(c, DE_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(ast));
                if (!in_one_window(p->d_gtcode, gt_cap * sizeof(BoundInstr))) return fail(c, DE_ERR_HIP, "gradient instruction stream straddles a 4 GiB boundary");
            }
            HIP_TRY(c, hipMemset(p->d_gtcode, 0, gt_cap * sizeof(BoundInstr)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_gtcode_off), p->gtcode_off.size() * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_gt_ids), std::max<size_t>(ids.size(), 1) * sizeof(int32_t)));
        }
        dbg_lap("grad threaded: ids, hipMalloc, memset");
        HIP_TRY(c, hipStreamSynchronize(c->stream)); // the previous form may be in use by queued work
        if (!p->gtcode.empty())
            HIP_TRY(c, hipMemcpy(p->d_gtcode, p->gtcode.data(), p->gtcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(p->d_gtcode_off, p->gtcode_off.data(), p->gtcode_off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!ids.empty()) HIP_TRY(c, hipMemcpy(p->d_gt_ids, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        p->gt_n_buckets = 0;
        for (int b = 0; b < NB; b++) {
            if (!count[b]) continue;
            GradArgs::Bucket &bk = p->gt_buckets[p->gt_n_buckets++];
            bk.GC = WIDTH[b % NW];
            bk.VS = 1 + b / NW;
            bk.windows = b % NW >= 7 ? (maxg[b] + bk.GC - 1) / bk.GC : 1;
            bk.max_grad = maxg[b];
            bk.n_slots = slots[b];
            bk.ids = p->d_gt_ids + start[b];
            bk.n = count[b];
            bk.handler_base = bases[b];
            bk.param_handler_off = (uint32_t)(tables[b][gop_param(WIDTH[b % NW])] - bases[b]);
        }
        dbg_lap("grad threaded: upload");
        p->gt_mode = mode;
        p->gt_wide = wide;
        p->gt_valid = true;
    }
    g->threaded_code = p->d_gtcode;
    g->e.code_off = p->d_gtcode_off;
    g->n_buckets = p->gt_n_buckets;
    for (int b = 0; b < p->gt_n_buckets; b++) g->buckets[b] = p->gt_buckets[b];
    return DE_OK;
}

// Reverse-accumulation form of the gradient program (de_rev_threaded.hip) for `mode`: per tree the forward
// instructions (every operator also stores its partials in LDS rows of its own), then the backward instructions
// in execution order.  Fills g->rev_* when the program can be expressed this way (otherwise leaves rev_code
// null and the forward-dual kernels run).  Call after ensure_generic_code().
static int ensure_rev_threaded(de_ctx *c, de_program *p, int mode, GradArgs *g) {
    g->rev_code = nullptr;
    // Reverse accumulation costs two sweeps whatever the number of gradient rows; forward duals cost one sweep
    // of (1 + rows) values (and one sweep per window of 8 rows).  Measured break-even on MI355X: ~8 rows per tree
    // (20-node trees: 3.5 rows 9.6 ms forward / 17.2 ms reverse; 17 rows 44.6 ms / 30.2 ms).  DE_LOSS_GRAD_REVERSE=1|0 forces.
    const char *env = getenv("DE_LOSS_GRAD_REVERSE");
    if (env && *env == '0') return DE_OK;
    // DE_OPT_FORWARD_GRAD: the caller wants the reference's forward-mode flag semantics exactly (a product chain that overflows in one
    // association only flips `ok` in ~0.03 % of Float32 fuzz cases under reverse accumulation): forward duals whatever the width
    if (p->options & DE_OPT_FORWARD_GRAD) return DE_OK;
    // ABI 3 (round 6): reverse accumulation is an OPT-IN (DE_OPT_REVERSE_GRAD, or DE_LOSS_GRAD_REVERSE=1 for the tests / experiments): the
    // default keeps the reference's forward-mode flag semantics
    if (!(p->options & DE_OPT_REVERSE_GRAD) && !(env && *env == '1')) return DE_OK;
    // a CSE program (GraphNode trees, §3.1) reads a persistent row from several consumers: the backward sweep ACCUMULATES their adjoints
    // into that row (round 4: `acc_use` below); DE_REV_NO_SHARED=1 restores round 3's fall-back to forward duals for such populations
    if (p->cse_generic && getenv("DE_REV_NO_SHARED")) return DE_OK;
    if (!(env && *env == '1')) {
        int64_t total = 0;
        for (int64_t t = 0; t < p->n_trees; t++) total += de_program_n_grad(p, t, mode);
        if (total < 8 * p->n_trees) return DE_OK;
    }
    const int F = p->n_features, P = p->n_params;
    if (!(p->rt_valid && p->rt_mode == mode)) {
        const uint32_t es32 = p->dtype == DE_F32 ? 4u : 8u, RB = 64u * es32;
        // parameter leaves are LDS rows F .. F+P (gathered by class when the kernel stages a tile), slots follow
        const uint32_t FE = (uint32_t)F + (p->uses_params ? (uint32_t)P : 0u);
        const bool hot_const_unary = !getenv("DE_NO_CONST_UNARY_HOT");
        auto gun_of = [&](uint32_t op) { // hot unary index of a de_opcode (de_bind.h), or -1
            return hot_const_unary ? gun_index((int)op, DE_U_COS, DE_U_EXP, DE_U_SIN, DE_U_NEG, DE_U_SQUARE, DE_U_CUBE, DE_U_ABS, DE_U_LOG, DE_U_SAFE_LOG,
                                               DE_U_SQRT, DE_U_SAFE_SQRT, DE_U_TANH, DE_U_RELU) : -1;
        };
        const uint32_t PR0 = FE + (uint32_t)p->n_slots; // first partial row
        uint64_t table[ROP_COUNT];
        hipError_t hst = rev_handler_table(p->dtype, table);
        if (hst != hipSuccess) return fail(c, DE_ERR_HIP, "reverse handler table: %s", hipGetErrorString(hst));
        uint64_t base = table[0];
        for (int i = 0; i < (int)ROP_COUNT; i++) base = std::min<uint64_t>(base, table[i]);
        for (int i = 0; i < (int)ROP_COUNT; i++)
            if (table[i] - base > 0xFFFFFFFFull) return DE_OK;
        constexpr uint32_t NONE = 0xFFFFFFFFu, ACC = 0x80000000u;
        auto leaf_col = [&](uint32_t f) -> uint32_t { return mode != DE_GRAD_CONSTANT ? (1u + (uint32_t)P + f) | ACC : NONE; };
        auto param_col = [&](uint32_t r) -> uint32_t { return mode != DE_GRAD_CONSTANT ? (1u + r) | ACC : NONE; };
        auto const_col = [&](uint32_t ord) -> uint32_t {
            return mode == DE_GRAD_CONSTANT ? 1u + ord : (mode == DE_GRAD_BOTH ? 1u + (uint32_t)(P + F) + ord : NONE);
        };
        auto rowb = [&](uint32_t row) { return (row < (uint32_t)F ? row : row + (FE - (uint32_t)F)) * RB; }; // LDS byte offset of a bound row
        p->rtcode.clear();
        p->rtcode_off.assign((size_t)p->n_trees + 1, 0);
        p->rtcode_mid.assign((size_t)p->n_trees, 0);
        p->rtsite_of_gb.assign(p->gbcode.size(), -1);
        p->site_gen++;
        uint32_t max_prows = 0;
        bool ok = true;
        std::vector<uint32_t> need((size_t)p->n_trees, 0);
        std::vector<BoundInstr> rv;
        std::vector<uint8_t> rv_col; // rv[k] carries a gradient column word in .lo
        std::vector<uint32_t> rv_rop; // rop of rv[k]
        std::vector<BoundInstr> bw;   // a tree's backward records in execution order
        std::vector<uint32_t> bw_rop;
        std::vector<uint8_t> acc_use; // per instruction of the tree: reads a shared row and is not its last reader (adds its adjoint)
        std::map<uint32_t, std::pair<uint32_t, std::pair<uint32_t, uint32_t>>> occ; // column -> (leaves, (seen, row))
        std::map<uint32_t, uint32_t> rop_of_off; // handler offset -> rop id (DE_REV_STATS)
        uint32_t mk_rop = 0;                     // rop of the record `mk` made last (the emitters below read it)
        const bool rfuse = !getenv("DE_REV_NO_FUSE"); // fused pairs / triples (de_rev_threaded.hip rh_pushload ...): same bits, fewer dispatches
        auto mk = [&](uint32_t rop, uint32_t y, uint32_t z, uint32_t w) {
            BoundInstr o;
            o.bop = (uint32_t)(table[rop] - base);
            rop_of_off[o.bop] = rop;
            mk_rop = rop;
            o.arg = y;
            o.lo = z;
            o.hi = w;
            return o;
        };
        for (int64_t t = 0; t < p->n_trees && ok; t++) {
            uint32_t n_prows = 0;
            rv.clear();
            rv_col.clear();
            rv_rop.clear();
            auto alloc = [&](uint32_t n) { const uint32_t r = (PR0 + n_prows) * RB; n_prows += n; return r; };
            uint32_t last_f_rop = 0xFFFFFFFFu; // rop of this tree's last forward record
            auto F_ = [&](const BoundInstr &o) -> int32_t {
                const uint32_t rop = mk_rop;
                if (rfuse && last_f_rop == ROP_PUSH) { // PUSH + the load / unary function of a leaf that starts the next subtree: one record
                    const uint32_t push_off = p->rtcode.back().arg;
                    const bool un_leaf = rop >= ROP_UN_BASE && rop < ROP_GEN_BASE && (((rop - ROP_UN_BASE) >> 1) & 1u);
                    BoundInstr f{0u, 0u, 0u, 0u};
                    bool fused = false;
                    if (rop == rop_load(RSRC_LEAF) && push_off < 65536u && o.arg < 65536u) { f = mk(ROP_F_PUSHLOAD_BASE + 0, push_off | (o.arg << 16), 0, 0); fused = true; }
                    else if (rop == rop_load(RSRC_CONST) && push_off < 65536u) { f = mk(ROP_F_PUSHLOAD_BASE + 1, push_off, o.lo, o.hi); fused = true; }
                    else if (un_leaf && push_off < 65536u && o.arg < 65536u) {
                        const uint32_t v = rop - ROP_UN_BASE;
                        f = mk(rop_pushun((int)(v >> 2), (v & 1u) != 0), push_off | (o.arg << 16), o.lo, 0);
                        fused = true;
                    }
                    if (fused) {
                        p->rtcode.back() = f;
                        last_f_rop = 0xFFFFFFFEu;
                        return (int32_t)p->rtcode.size() - 1;
                    }
                }
                p->rtcode.push_back(o);
                last_f_rop = rop;
                return (int32_t)p->rtcode.size() - 1; // (the record that carries o's immediate: de_program_set_consts patches it there)
            };
            auto R_ = [&](const BoundInstr &o, bool has_col = false) { rv.push_back(o); rv_col.push_back(has_col ? 1 : 0); rv_rop.push_back(mk_rop); }; // pushed in forward order, reversed below
            // SHARED ROWS.  A slot row is written by a PUSH and normally read once; a GraphNode program reads a persistent row from several
            // consumers.  Backwards the consumers run in reverse order and the definition's r_pop last: the consumer that runs FIRST in the
            // backward sweep (the last reader in program order) stores its adjoint contribution into the row, every other one adds to it.
            // acc_use[i] = instruction i reads a slot row and is NOT that row's last reader before its next PUSH.
            acc_use.assign((size_t)(p->gbcode_off[(size_t)t + 1] - p->gbcode_off[(size_t)t]), 0);
            {
                std::map<uint32_t, int32_t> last_reader; // slot row -> the last instruction seen reading it since its PUSH
                for (int32_t i = p->gbcode_off[(size_t)t]; i < p->gbcode_off[(size_t)t + 1]; i++) {
                    const BoundInstr &b = p->gbcode[(size_t)i];
                    const uint32_t row = b.arg & 0xFFFFFFu;
                    if (b.bop == BOP_PUSH) { last_reader.erase(row); continue; }
                    const bool reads_row = b.bop == BOP_LOAD_ROW || b.bop == BOP_GEN_ROW || (b.bop >= BOP_BIN_BASE && b.bop < BOP_BIN_END && !((b.bop - BOP_BIN_BASE) & 2)) ||
                                           (b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END && ((b.bop - BOP_UN_BASE) & 2));
                    if (!reads_row || row < (uint32_t)F) continue;
                    auto it = last_reader.find(row);
                    if (it != last_reader.end()) acc_use[(size_t)(it->second - p->gbcode_off[(size_t)t])] = 1; // no longer the last reader: it adds
                    last_reader[row] = i;
                }
            }
            auto accumulates = [&](int32_t i) { return acc_use[(size_t)(i - p->gbcode_off[(size_t)t])] != 0; };
            // backward of "acc' = op(acc, operand)" whose partial rows (d/d acc, d/d operand) start at pr
            auto back_binary = [&](int pk, uint32_t pr, bool slot, uint32_t slot_byte, uint32_t col, bool add = false) {
                if (slot && add) R_(mk(ROP_R_BINACC_BASE + (uint32_t)pk, pk == 0 ? pr : 0, slot_byte, 0));
                else if (slot) R_(mk(rop_rbin(pk, 0), pk == 0 ? pr : 0, slot_byte, 0));
                else if (col != NONE) R_(mk(rop_rbin(pk, 1), pk == 0 ? pr : 0, col, 0), true);
                else if (pk == 0) R_(mk(ROP_R_UN, pr, 0, 0));
                else if (pk == 3) R_(mk(ROP_R_NEG, 0, 0, 0));
            };
            // backward of "acc' = f(leaf)": first the unary partial, then the leaf's row — pushed in reverse
            auto back_unary_leaf = [&](uint32_t pr, uint32_t col) {
                if (col != NONE) R_(mk(ROP_R_LEAF, 0, col, 0), true);
                R_(mk(ROP_R_UN, pr, 0, 0));
            };
            // backward of "acc' = f(shared row)": the unary partial, then the row's adjoint receives the result
            auto back_unary_slot = [&](uint32_t pr, uint32_t slot_byte, bool add) {
                R_(mk(ROP_R_SLOTACC_BASE + (add ? 1u : 0u), slot_byte, 0, 0));
                R_(mk(ROP_R_UN, pr, 0, 0));
            };
            for (int32_t i = p->gbcode_off[(size_t)t]; i < p->gbcode_off[(size_t)t + 1] && ok; i++) {
                const BoundInstr &b = p->gbcode[(size_t)i];
                const uint32_t row = b.arg & 0xFFFFFFu, aux = b.arg >> 24, ord = b.arg & 0xFFFFu;
                const bool is_leaf = row < (uint32_t)F;
                if (b.bop == BOP_CHECK_ROW) continue; // leaf operands are tested where they are read
                if (b.bop == BOP_LOAD_ROW && !is_leaf) { // acc = a shared (persistent) row
                    F_(mk(rop_load(RSRC_SLOT), rowb(row), 0, 0));
                    R_(mk(ROP_R_SLOTACC_BASE + (accumulates(i) ? 1u : 0u), rowb(row), 0, 0));
                } else if (b.bop == BOP_LOAD_ROW) {
                    F_(mk(rop_load(RSRC_LEAF), rowb(row), 0, 0));
                    if (leaf_col(row) != NONE) R_(mk(ROP_R_LEAF, 0, leaf_col(row), 0), true);
                } else if (b.bop == BOP_LOAD_CONST) {
                    p->rtsite_of_gb[(size_t)i] = F_(mk(rop_load(RSRC_CONST), 0, b.lo, b.hi));
                    if (const_col(ord) != NONE) R_(mk(ROP_R_LEAF, 0, const_col(ord), 0), true);
                } else if (b.bop == BOP_PUSH) {
                    // A spill is followed by the load that starts the next subtree (the accumulator's value is dead: the backward sweep
                    // continues with the slot's adjoint).  A SHARED definition that is used at once stays in the accumulator: the next
                    // instruction reads it, and backwards BOTH adjoints — the accumulator's and the row's — flow into the definition.
                    bool acc_live = false;
                    for (int32_t q = i + 1; q < p->gbcode_off[(size_t)t + 1]; q++) {
                        const BoundInstr &nx = p->gbcode[(size_t)q];
                        if (nx.bop == BOP_CHECK_ROW || nx.bop == BOP_CHECK_ACC || nx.bop == BOP_PUSH) continue;
                        const uint32_t nau = nx.arg >> 24;
                        acc_live = top_reads_acc(nx.bop, nau == (uint32_t)DOP_LOAD ? 0 : de_opcode_degree((int)nau));
                        break;
                    }
                    F_(mk(ROP_PUSH, rowb(row), 0, 0));
                    R_(mk(acc_live ? (uint32_t)ROP_R_POPADD : (uint32_t)ROP_R_POP, rowb(row), 0, 0));
                } else if (b.bop == BOP_CHECK_ACC) {
                    F_(mk(ROP_CHECK, 0, 0, 0));
                } else if (b.bop >= BOP_BIN_BASE && b.bop < BOP_BIN_END) {
                    const uint32_t v = b.bop - BOP_BIN_BASE;
                    const int k = (int)(v >> 2);
                    const bool cst = (v & 2) != 0, chk = (v & 1) != 0;
                    const uint32_t pr = k >= 3 ? alloc(2) : 0;
                    const int pk = k == 0 ? 1 : (k == 1 ? 2 : (k == 2 ? 3 : 0));
                    if (cst) {
                                                p->rtsite_of_gb[(size_t)i] = F_(mk(rop_bin(k, RSRC_CONST, chk), pr, b.lo, b.hi));
                        back_binary(pk, pr, false, 0, const_col(ord));
                    } else {
                        F_(mk(rop_bin(k, is_leaf ? RSRC_LEAF : RSRC_SLOT, chk), rowb(row), pr, 0));
                        back_binary(pk, pr, !is_leaf, rowb(row), is_leaf ? leaf_col(row) : NONE, !is_leaf && accumulates(i));
                    }
                } else if (b.bop >= BOP_UN_BASE && b.bop < BOP_UN_END) {
                    const uint32_t v = b.bop - BOP_UN_BASE;
                    const int k = (int)(v >> 2);
                    const bool from_row = (v & 2) != 0, chk = (v & 1) != 0;
                    const uint32_t pr = alloc(1);
                    if (from_row && !is_leaf) { // unary function of a shared row
                        F_(mk(rop_un_slot(k, chk), rowb(row), pr, 0));
                        back_unary_slot(pr, rowb(row), accumulates(i));
                    } else if (from_row) {
                        F_(mk(rop_un(k, RSRC_LEAF, chk), rowb(row), pr, 0));
                        back_unary_leaf(pr, leaf_col(row));
                    } else {
                        F_(mk(rop_un(k, RSRC_ACC, chk), pr, 0, 0));
                        R_(mk(ROP_R_UN, pr, 0, 0));
                    }
                } else if (b.bop == BOP_GEN_ROW && hot_const_unary && (aux == (uint32_t)DE_B_MAX || aux == (uint32_t)DE_B_MIN)) {
                    const uint32_t pr = alloc(2);
                    F_(mk(rop_bin(aux == (uint32_t)DE_B_MAX ? 6 : 7, is_leaf ? RSRC_LEAF : RSRC_SLOT, false), rowb(row), pr, 0));
                    back_binary(0, pr, !is_leaf, rowb(row), is_leaf ? leaf_col(row) : NONE, !is_leaf && accumulates(i));
                } else if (b.bop == BOP_GEN_ROW && gun_of(aux) >= 0 && is_leaf) {
                    const uint32_t pr = alloc(1);
                    F_(mk(rop_un(gun_of(aux), RSRC_LEAF, false), rowb(row), pr, 0));
                    back_unary_leaf(pr, leaf_col(row));
                } else if (b.bop == BOP_GEN_ROW) {
                    const bool unary = aux < (uint32_t)DE_B_ADD;
                    const uint32_t pr = alloc(unary ? 1 : 2);
                    F_(mk(rop_gen(is_leaf ? RSRC_LEAF : RSRC_SLOT), rowb(row), pr | (aux << 24), 0));
                    if (unary && !is_leaf) back_unary_slot(pr, rowb(row), accumulates(i));
                    else if (unary) back_unary_leaf(pr, leaf_col(row));
                    else back_binary(0, pr, !is_leaf, rowb(row), is_leaf ? leaf_col(row) : NONE, !is_leaf && accumulates(i));
                } else if (b.bop == BOP_GEN_CONST && hot_const_unary && (aux == (uint32_t)DE_B_MAX || aux == (uint32_t)DE_B_MIN)) {
                    const uint32_t pr = alloc(2);
                    p->rtsite_of_gb[(size_t)i] = F_(mk(rop_bin(aux == (uint32_t)DE_B_MAX ? 6 : 7, RSRC_CONST, false), pr, b.lo, b.hi));
                    back_binary(0, pr, false, 0, const_col(ord));
                } else if (b.bop == BOP_GEN_CONST && gun_of(aux) >= 0) {
                    // cos / exp / sin of a constant leaf: load + hot unary handler instead of the generic one
                    const uint32_t pr = alloc(1);
                    p->rtsite_of_gb[(size_t)i] = F_(mk(rop_load(RSRC_CONST), 0, b.lo, b.hi));
                    F_(mk(rop_un(gun_of(aux), RSRC_ACC, false), pr, 0, 0));
                    back_unary_leaf(pr, const_col(ord));
                } else if (b.bop == BOP_GEN_CONST) {
                    const bool unary = aux < (uint32_t)DE_B_ADD;
                    const uint32_t pr = alloc(unary ? 1 : 2);
                    p->rtsite_of_gb[(size_t)i] = F_(mk(rop_gen(RSRC_CONST), pr | (aux << 24), b.lo, b.hi));
                    if (unary) back_unary_leaf(pr, const_col(ord));
                    else back_binary(0, pr, false, 0, const_col(ord));
                } else if (b.bop == BOP_GEN_ACC && gun_of(aux) >= 0) {
                    const uint32_t pr = alloc(1);
                    F_(mk(rop_un(gun_of(aux), RSRC_ACC, false), pr, 0, 0));
                    R_(mk(ROP_R_UN, pr, 0, 0));
                } else if (b.bop == BOP_GEN_ACC) {
                    const uint32_t pr = alloc(1);
                    F_(mk(rop_gen(RSRC_ACC), pr | (aux << 24), 0, 0));
                    R_(mk(ROP_R_UN, pr, 0, 0));
                } else if (b.bop == BOP_GEN_PARAM) { // operand = parameter row prm = LDS leaf row F + prm
                    const uint32_t prm = b.arg & 0xFFFFu, prow = ((uint32_t)F + prm) * RB;
                    int k = -1, ku = -1;
                    switch (aux) {
                    case DE_B_ADD: k = 0; break;
                    case DE_B_SUB: k = 1; break;
                    case DOP_RSUB: k = 2; break;
                    case DE_B_MUL: k = 3; break;
                    case DE_B_DIV: k = 4; break;
                    case DOP_RDIV: k = 5; break;
                    case DE_B_MAX: k = hot_const_unary ? 6 : -1; break;
                    case DE_B_MIN: k = hot_const_unary ? 7 : -1; break;
                    default: ku = gun_of(aux); break;
                    }
                    if (aux == (uint32_t)DOP_LOAD) {
                        F_(mk(rop_load(RSRC_LEAF), prow, 0, 0));
                        if (param_col(prm) != NONE) R_(mk(ROP_R_LEAF, 0, param_col(prm), 0), true);
                    } else if (k >= 0) {
                        const uint32_t pr = k >= 3 ? alloc(2) : 0;
                        F_(mk(rop_bin(k, RSRC_LEAF, false), prow, pr, 0));
                        back_binary(k == 0 ? 1 : (k == 1 ? 2 : (k == 2 ? 3 : 0)), pr, false, 0, param_col(prm));
                    } else if (ku >= 0) {
                        const uint32_t pr = alloc(1);
                        F_(mk(rop_un(ku, RSRC_LEAF, false), prow, pr, 0));
                        back_unary_leaf(pr, param_col(prm));
                    } else {
                        const bool unary = aux < (uint32_t)DE_B_ADD;
                        const uint32_t pr = alloc(unary ? 1 : 2);
                        F_(mk(rop_gen(RSRC_LEAF), prow, pr | (aux << 24), 0));
                        if (unary) back_unary_leaf(pr, param_col(prm));
                        else back_binary(0, pr, false, 0, param_col(prm));
                    }
                } else if (b.bop == BOP_TERN) {
                    if (is_leaf || b.lo < (uint32_t)F || row > 0xFFFFu || b.lo > 0xFFFFu) { ok = false; break; }
                    if (p->cse_generic) { ok = false; break; } // (a ternary operator's slot operands may be shared rows: r_tern stores; such populations keep forward duals)
                    const uint32_t pr = alloc(3);
                    const uint32_t rb_ = row + (FE - (uint32_t)F), rc_ = b.lo + (FE - (uint32_t)F);
                    if (rb_ > 0xFFFFu || rc_ > 0xFFFFu) { ok = false; break; }
                    F_(mk(ROP_TERN, pr | (aux << 24), rb_ | (rc_ << 16), 0));
                    R_(mk(ROP_R_TERN, pr, rb_ | (rc_ << 16), 0));
                } else ok = false; // INJ_*: only bound with early_exit=false, never for gradients
            }
            if (!ok) break;
            // end record of the forward sweep (r_end: the table slot of round 1's parameter handler); the backward sweep's
            // first record follows it
            p->rtcode.push_back(mk(ROP_PARAM, 0, 0, 0));
            p->rtcode_mid[(size_t)t] = (int32_t)p->rtcode.size();
            // Gradient rows several leaves share (features, parameters): the leaves' contributions are added per
            // SAMPLE in an LDS row and reduced once, at the last of them — paths that cancel within a sample then
            // cancel before the reduction, as they do in the forward Jacobian.
            // column word: [15:0] column, [29:16] accumulation row, [31:30] 0 reduce now, 1 first, 2 middle, 3 last
            occ.clear();
            for (size_t k = 0; k < rv.size(); k++)
                if (rv_col[k] && (rv[k].lo & ACC)) occ[rv[k].lo & 0xFFFFu].first++;
            uint32_t n_acc = 0;
            bw.clear();
            bw_rop.clear();
            for (size_t k = rv.size(); k-- > 0;) { // execution order
                BoundInstr o = rv[k];
                if (rv_col[k]) {
                    const uint32_t col = o.lo & 0xFFFFu;
                    if ((o.lo & 0x7FFFFFFFu) > 0xFFFFu) { ok = false; break; }
                    uint32_t word = col;
                    if (o.lo & ACC) {
                        auto &oc = occ[col];
                        if (oc.first > 1) {
                            if (oc.second.first == 0) oc.second.second = n_acc++;
                            const uint32_t nth = ++oc.second.first;
                            const uint32_t md = nth == 1 ? 1u : (nth == oc.first ? 3u : 2u);
                            word = col | ((PR0 + n_prows + oc.second.second) << 16) | (md << 30);
                        }
                    }
                    o.lo = word;
                }
                bw.push_back(o);
                bw_rop.push_back(rv_rop[k]);
            }
            if (!ok) break;
            for (size_t k = 0; k < bw.size();) { // fused backward sequences: [r_un] r_leaf [r_pop]  and  r_bin<PK, column> r_leaf [r_pop]
                auto is = [&](size_t q, uint32_t rop) { return q < bw.size() && bw_rop[q] == rop; };
                auto small = [&](size_t q) { return q >= bw.size() || bw[q].arg < 65536u; };
                if (rfuse && is(k, ROP_R_UN) && is(k + 1, ROP_R_LEAF) && small(k) && (!is(k + 2, ROP_R_POP) || small(k + 2))) {
                    const bool pop = is(k + 2, ROP_R_POP);
                    p->rtcode.push_back(mk(rop_leafx(true, pop), bw[k].arg | (pop ? bw[k + 2].arg << 16 : 0u), bw[k + 1].lo, 0));
                    k += pop ? 3 : 2;
                } else if (rfuse && is(k, ROP_R_LEAF) && is(k + 1, ROP_R_POP) && small(k + 1)) {
                    p->rtcode.push_back(mk(rop_leafx(false, true), bw[k + 1].arg << 16, bw[k].lo, 0));
                    k += 2;
                } else if (rfuse && k < bw.size() && bw_rop[k] >= ROP_R_BIN_BASE && bw_rop[k] < ROP_R_TERN && ((bw_rop[k] - ROP_R_BIN_BASE) & 1u) && is(k + 1, ROP_R_LEAF) &&
                           small(k) && (!is(k + 2, ROP_R_POP) || small(k + 2))) {
                    const bool pop = is(k + 2, ROP_R_POP);
                    p->rtcode.push_back(mk(rop_bincolx((int)((bw_rop[k] - ROP_R_BIN_BASE) >> 1), pop), bw[k].arg | (pop ? bw[k + 2].arg << 16 : 0u), bw[k].lo, bw[k + 1].lo));
                    k += pop ? 3 : 2;
                } else {
                    p->rtcode.push_back(bw[k]);
                    k += 1;
                }
            }
            if (PR0 + n_prows + n_acc > 0x3FFFu) { ok = false; break; }
            p->rtcode.push_back(mk(ROP_PARAM, 0, 0, 0)); // end record of the backward sweep
            p->rtcode_off[(size_t)t + 1] = (int32_t)p->rtcode.size();
            max_prows = std::max(max_prows, n_prows + n_acc);
            need[(size_t)t] = n_prows + n_acc;
        }
        if (getenv("DE_REV_STATS") && ok) { // dispatch classes and adjacent pairs of the two sweeps (what a fusion would save)
            auto cls = [&](uint32_t off) -> std::string {
                const uint32_t r = rop_of_off.count(off) ? rop_of_off[off] : 9999u;
                char buf[48];
                if (r < 3) snprintf(buf, sizeof buf, "LOAD%c", "LSC"[r]);
                else if (r == ROP_PUSH) return "PUSH";
                else if (r == ROP_CHECK) return "CHECK";
                else if (r >= ROP_BIN_BASE && r < ROP_UN_BASE) snprintf(buf, sizeof buf, "BIN%c", "LSC"[((r - ROP_BIN_BASE) / 2) % 3]);
                else if (r >= ROP_UN_BASE && r < ROP_GEN_BASE) snprintf(buf, sizeof buf, "UN%c", ((r - ROP_UN_BASE) / 2) % 2 ? 'L' : 'A');
                else if (r >= ROP_GEN_BASE && r < ROP_TERN) return "GEN";
                else if (r == ROP_PARAM) return "END";
                else if (r == ROP_R_UN) return "r_un";
                else if (r == ROP_R_NEG) return "r_neg";
                else if (r == ROP_R_POP) return "r_pop";
                else if (r == ROP_R_LEAF) return "r_leaf";
                else if (r >= ROP_R_BIN_BASE && r < ROP_R_TERN) snprintf(buf, sizeof buf, "r_bin%s", (r - ROP_R_BIN_BASE) % 2 ? "col" : "slot");
                else if (r >= ROP_F_PUSHLOAD_BASE && r < ROP_R_LEAFX_BASE) return "PUSH+";
                else if (r >= ROP_R_LEAFX_BASE && r < ROP_R_BINCOLX_BASE) return "r_leafx";
                else if (r >= ROP_R_BINCOLX_BASE && r < ROP_COUNT) return "r_bincolx";
                else return "other";
                return buf;
            };
            std::map<std::string, int64_t> one, two;
            for (size_t i = 0; i < p->rtcode.size(); i++) {
                const std::string a = cls(p->rtcode[i].bop);
                one[a]++;
                if (i + 1 < p->rtcode.size() && a != "END") two[a + " " + cls(p->rtcode[i + 1].bop)]++;
            }
            fprintf(stderr, "DE_REV_STATS: %zu records, %lld trees: %.2f dispatches per tree\n", p->rtcode.size(), (long long)p->n_trees, (double)p->rtcode.size() / (double)p->n_trees);
            for (auto &kv : one) fprintf(stderr, "  %-10s %8.3f per tree\n", kv.first.c_str(), (double)kv.second / (double)p->n_trees);
            std::vector<std::pair<int64_t, std::string>> v;
            for (auto &kv : two) v.push_back({kv.second, kv.first});
            std::sort(v.rbegin(), v.rend());
            for (size_t i = 0; i < v.size() && i < 24; i++) fprintf(stderr, "  pair %-22s %8.3f per tree\n", v[i].second.c_str(), (double)v[i].first / (double)p->n_trees);
        }
        // per-wave staging of the column sums: one LDS row, or the widest tree's columns
        int64_t stage_cols = 64;
        for (int64_t t = 0; t < p->n_trees; t++) stage_cols = std::max<int64_t>(stage_cols, 1 + de_program_n_grad(p, t, mode));
        const uint64_t stage_rows = ((uint64_t)stage_cols * es32 + RB - 1) / RB;
        const uint64_t rows = (uint64_t)PR0 + max_prows + stage_rows;
        if (!ok || 4 * rows * RB > 160 * 1024 || rows * RB >= (1u << 24)) { p->rtsite_of_gb.clear(); p->site_gen++; return DE_OK; }
        // The kernel is latency-bound and its occupancy is set by the LDS rows of the neediest tree of a launch
        // (5 -> 4 workgroups per CU: +17 % time): trees are grouped by the number of workgroups per CU their own
        // need allows and every group is a launch of its own (small groups join the next needier one).
        auto wgs_of = [&](uint32_t nd) { return (int)std::min<uint64_t>(8, (160 * 1024) / (4 * ((uint64_t)PR0 + nd + stage_rows) * RB)); };
        std::vector<int32_t> ids((size_t)p->n_trees);
        for (int64_t t = 0; t < p->n_trees; t++) ids[(size_t)t] = (int32_t)t;
        std::stable_sort(ids.begin(), ids.end(), [&](int32_t x, int32_t y) { return need[(size_t)x] < need[(size_t)y]; });
        p->rt_n_groups = 0;
        const bool grouping = true;
        for (int64_t k = 0; k < p->n_trees;) {
            int64_t e = k;
            const int w = wgs_of(need[(size_t)ids[(size_t)k]]);
            while (e < p->n_trees && grouping && wgs_of(need[(size_t)ids[(size_t)e]]) == w) e++;
            if (!grouping) e = p->n_trees;
            // a group too small to fill the chip, or the last slot: extend to the end / absorb into the next group
            if (p->rt_n_groups == 7) e = p->n_trees;
            while (e < p->n_trees && e - k < std::max<int64_t>(64, p->n_trees / 16)) e++;
            if (p->n_trees - e < std::max<int64_t>(64, p->n_trees / 16)) e = p->n_trees;
            GradArgs::RevGroup &gr = p->rt_groups[p->rt_n_groups++];
            gr.first = (int32_t)k;
            gr.n = (int32_t)(e - k);
            gr.rows = (int32_t)(PR0 + need[(size_t)ids[(size_t)e - 1]] + stage_rows);
            std::sort(ids.begin() + k, ids.begin() + e); // tree order inside a group: adjacent trees share staging batches
            k = e;
        }
        // the handler word of a record names the handler of the record BEHIND it; the end record of a sweep names the sweep's first
        // handler (de_rev_threaded.hip: a handler knows its successor at entry)
        for (int64_t t = 0; t < p->n_trees; t++) {
            const int32_t lim[3] = {p->rtcode_off[(size_t)t], p->rtcode_mid[(size_t)t], p->rtcode_off[(size_t)t + 1]};
            for (int sw = 0; sw < 2; sw++) {
                const int32_t a0 = lim[sw], b0 = lim[sw + 1];
                if (b0 - a0 < 2) continue;
                const uint32_t first = p->rtcode[(size_t)a0].bop;
                for (int32_t i = a0; i < b0 - 1; i++) p->rtcode[(size_t)i].bop = p->rtcode[(size_t)i + 1].bop;
                p->rtcode[(size_t)b0 - 1].bop = first;
            }
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream)); // the previous form may be in use by queued work
        if (p->d_rtcode) { // sizes depend on the mode
            prog_free(c, p->d_rtcode);
            p->d_rtcode = nullptr;
        }
        { // (inside one 4 GiB window: the handlers bump the record pointer without a carry)
            const size_t rbytes = (p->rtcode.size() + 1) * sizeof(BoundInstr);
            const hipError_t ast = prog_malloc(c, reinterpret_cast<void **>(&p->d_rtcode), rbytes);
            if (ast != hipSuccess) return fail(c, DE_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(ast));
            if (!in_one_window(p->d_rtcode, rbytes)) return fail(c, DE_ERR_HIP, "reverse instruction stream straddles a 4 GiB boundary");
        }
        HIP_TRY(c, hipMemset(p->d_rtcode, 0, (p->rtcode.size() + 1) * sizeof(BoundInstr)));
        if (!p->d_rtcode_off) {
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_rtcode_off), p->rtcode_off.size() * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_rtcode_mid), std::max<size_t>(p->rtcode_mid.size(), 1) * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_rt_ids), std::max<size_t>(ids.size(), 1) * sizeof(int32_t)));
        }
        if (!p->rtcode.empty())
            HIP_TRY(c, hipMemcpy(p->d_rtcode, p->rtcode.data(), p->rtcode.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(p->d_rtcode_off, p->rtcode_off.data(), p->rtcode_off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!ids.empty()) {
            HIP_TRY(c, hipMemcpy(p->d_rtcode_mid, p->rtcode_mid.data(), p->rtcode_mid.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(p->d_rt_ids, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        p->rt_stage_cols = (int)stage_cols;
        p->rt_handler_base = base;
        p->rt_param_off = (uint32_t)(table[ROP_PARAM] - base);
        p->rt_mode = mode;
        p->rt_valid = true;
    }
    g->rev_code = p->d_rtcode;
    g->rev_code_off = p->d_rtcode_off;
    g->rev_code_mid = p->d_rtcode_mid;
    g->rev_ids = p->d_rt_ids;
    g->rev_n_groups = p->rt_n_groups;
    for (int k = 0; k < p->rt_n_groups; k++) g->rev_groups[k] = p->rt_groups[k];
    g->rev_stage_cols = p->rt_stage_cols;
    g->rev_handler_base = p->rt_handler_base;
    g->rev_param_off = p->rt_param_off;
    return DE_OK;
}

// Shared body of de_eval_grad / de_eval_diff.
static int grad_impl(de_ctx *c, de_program *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                     int mode, int diff_direction, void *out, int64_t ld_out, void *grad,
                     const int64_t *grad_offsets, uint8_t *ok, const void *dY = nullptr) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (p->ctx != c) return fail(c, DE_ERR_INVALID_ARG, "program belongs to another context");
    const bool diff = diff_direction >= 0;
    if (N < 0 || !ok || (p->n_trees > 0 && N > 0 && (!X || !grad))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (ldX < p->n_features || ((out || diff) && ld_out < N)) return fail(c, DE_ERR_INVALID_ARG, "ldX < n_features or ld_out < N");
    if (!diff && mode != DE_GRAD_VARIABLE && mode != DE_GRAD_CONSTANT && mode != DE_GRAD_BOTH)
        return fail(c, DE_ERR_INVALID_ARG, "bad gradient mode");
    if (diff && diff_direction >= p->n_features) return fail(c, DE_ERR_OUT_OF_RANGE, "direction >= n_features");
    int rc = check_param_args(c, p, pa, N);
    if (rc != DE_OK) return rc;
    if (p->n_trees == 0) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    const bool ok_dev = is_device_ptr(ok);
    std::vector<uint8_t> ones;
    const uint8_t *ok_init = p->host_ok_grad.data();
    if (diff) { // no validity test on this path: always complete (src/EvaluateDerivative.jl:117)
        ones.assign((size_t)p->n_trees, 1);
        ok_init = ones.data();
    }
    if (N == 0) {
        if (ok_dev) {
            HIP_TRY(c, hipMemcpyAsync(ok, ok_init, (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        } else std::memcpy(ok, ok_init, (size_t)p->n_trees);
        return DE_OK;
    }
    // per-tree gradient geometry
    std::vector<int32_t> ng((size_t)p->n_trees);
    std::vector<int64_t> goff((size_t)p->n_trees);
    int64_t span = 0, run = 0;
    int32_t maxg = 0;
    for (int64_t t = 0; t < p->n_trees; t++) {
        const int32_t g = diff ? 1 : (int32_t)de_program_n_grad(p, t, mode);
        ng[(size_t)t] = g;
        maxg = std::max(maxg, g);
        const int64_t off = diff ? t * ld_out : (grad_offsets ? grad_offsets[t] : run);
        if (off < 0) return fail(c, DE_ERR_INVALID_ARG, "negative gradient offset");
        goff[(size_t)t] = off;
        run += (int64_t)g * N;
        span = std::max(span, off + (int64_t)g * N);
    }
    const size_t lds_need = ((size_t)p->n_features + (size_t)p->n_slots * (1 + (size_t)std::min(maxg, 8))) * 260 * es;
    if (lds_need > 160 * 1024) return fail(c, DE_ERR_UNSUPPORTED, "gradient kernel: LDS footprint too large for this tree shape");
    rc = ensure_generic_code(c, p);
    if (rc) return rc;

    Staged sX, sOut, sGrad, sOk, sPar, sCls;
    rc = stage_in(c, c->sX, X, (size_t)ldX * (size_t)N * es, &sX);
    if (rc) return rc;
    if (out) {
        rc = stage_out(c, c->sOut, out, ((size_t)(p->n_trees - 1) * (size_t)ld_out + (size_t)N) * es, &sOut);
        if (rc) return rc;
    }
    rc = stage_out(c, diff ? c->sOut2 : c->sGrad, grad, (size_t)span * es, &sGrad);
    if (rc) return rc;
    if (ok_dev) sOk.dev = ok;
    else {
        HIP_TRY(c, c->sOk.reserve((size_t)p->n_trees));
        sOk.dev = c->sOk.p;
        sOk.staged = true;
    }
    // The initial flags, the gradient widths and (packed layout) the offsets depend on the program, the mode and N only:
    // they live on the device and are refreshed when one of those changes — the usual call copies nothing from pageable
    // host memory and does not block.  Caller-supplied offsets and eval_diff take the staged path.
    const bool cached = !diff && !grad_offsets;
    const int64_t *d_goff_use = nullptr;
    const int32_t *d_ng_use = nullptr;
    if (cached) {
        const size_t nt = (size_t)p->n_trees;
        if (!p->d_ok_grad) {
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_ok_grad), nt));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_ng), nt * sizeof(int32_t)));
            HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_goff), nt * sizeof(int64_t)));
            p->tab_ok_stale = true;
            p->tab_mode = -1;
        }
        if (p->tab_ok_stale || p->tab_mode != mode || p->tab_N != N) {
            HIP_TRY(c, hipStreamSynchronize(c->stream)); // earlier calls may still read the tables
            HIP_TRY(c, hipMemcpy(p->d_ok_grad, p->host_ok_grad.data(), nt, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(p->d_ng, ng.data(), nt * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(p->d_goff, goff.data(), nt * sizeof(int64_t), hipMemcpyHostToDevice));
            p->tab_ok_stale = false;
            p->tab_mode = mode;
            p->tab_N = N;
        }
        HIP_TRY(c, hipMemcpyAsync(sOk.dev, p->d_ok_grad, nt, hipMemcpyDeviceToDevice, c->stream));
        d_goff_use = p->d_goff;
        d_ng_use = p->d_ng;
    } else {
        HIP_TRY(c, hipMemcpyAsync(sOk.dev, ok_init, (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, c->sGoff.reserve(goff.size() * sizeof(int64_t)));
        HIP_TRY(c, c->sNg.reserve(ng.size() * sizeof(int32_t)));
        HIP_TRY(c, hipMemcpyAsync(c->sGoff.p, goff.data(), goff.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->sNg.p, ng.data(), ng.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        d_goff_use = static_cast<const int64_t *>(c->sGoff.p);
        d_ng_use = static_cast<const int32_t *>(c->sNg.p);
    }
    if (p->uses_params) {
        rc = stage_in(c, c->sParams, pa->params, (size_t)pa->ld_params * (size_t)pa->n_classes * es, &sPar);
        if (rc) return rc;
        rc = stage_in(c, c->sClasses, pa->classes, (size_t)N * (pa->classes_is_i64 ? 8 : 4), &sCls);
        if (rc) return rc;
    }
    // the pageable host vectors of the staged path must outlive their async copies
    if (!cached) HIP_TRY(c, hipStreamSynchronize(c->stream));

    GradArgs g;
    std::memset(&g, 0, sizeof g);
    g.generic_code = p->d_gcode;
    g.e.code_off = nullptr;
    g.e.n_trees = (int32_t)p->n_trees;
    g.e.skip_flagged = !(p->options & DE_OPT_FULL_EVAL) && tree_skip_enabled(); // (the gradient entry points always test validity)
    if (c->sPrio.reserve((size_t)3 * DE_PRIO_MAX_F * sizeof(unsigned long long)) == hipSuccess) g.e.prio_keys = c->sPrio.p; // priority tiles (de_kernels.hip)
    g.prio_ready = false;
    g.e.prio_keys_ready = !sX.staged && g.e.prio_keys && dataset_keys(c, p->dtype, X, N, ldX, p->n_features, &g.e.prio_keys);
    g.e.n_slots = p->n_slots;
    g.e.uses_params = p->uses_params;
    g.e.X = sX.dev;
    g.e.N = N;
    g.e.ldX = ldX;
    g.e.F = p->n_features;
    g.e.out = out ? sOut.dev : nullptr;
    g.e.ld_out = ld_out;
    g.e.ok = static_cast<uint8_t *>(sOk.dev);
    if (p->uses_params) {
        g.e.params = sPar.dev;
        g.e.ld_params = pa->ld_params;
        g.e.n_classes = pa->n_classes;
        g.e.classes = sCls.dev;
        g.e.classes_is_i64 = pa->classes_is_i64;
        g.e.class_base = pa->class_base;
    }
    g.mode = diff ? DE_GRAD_VARIABLE : mode;
    g.P = p->n_params;
    g.grad = sGrad.dev;
    g.grad_off = d_goff_use;
    g.n_grad = d_ng_use;
    g.max_grad = maxg;
    g.diff_direction = diff ? diff_direction : -1;
    g.e.code_off = p->d_gcode_off;
    if (!diff) {
        rc = ensure_grad_threaded(c, p, mode, ng, N, &g);
        if (rc) return rc;
    }
    Staged sDY;
    if (dY) {
        rc = stage_in(c, c->sY, dY, (size_t)N * es, &sDY);
        if (rc) return rc;
    }
    HIP_TRY(c, time_begin(c));
    HIP_TRY(c, launch_grad(p->dtype, g, c->stream, &c->last_kernel));
    if (dY) // the pullback's dX .* dY' (and its NaN fill) on the Jacobians just written
        HIP_TRY(c, launch_pullback_scale(p->dtype, sGrad.dev, g.grad_off, g.n_grad, g.e.ok, sDY.dev, N, p->n_trees, maxg, c->stream));
    HIP_TRY(c, time_end(c));
    if (out && sOut.staged)
        for (int64_t t = 0; t < p->n_trees; t++)
            HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(out) + (size_t)t * (size_t)ld_out * es,
                                      static_cast<char *>(sOut.dev) + (size_t)t * (size_t)ld_out * es, (size_t)N * es,
                                      hipMemcpyDeviceToHost, c->stream));
    if (sGrad.staged) {
        if (diff) {
            for (int64_t t = 0; t < p->n_trees; t++)
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(grad) + (size_t)t * (size_t)ld_out * es,
                                          static_cast<char *>(sGrad.dev) + (size_t)t * (size_t)ld_out * es, (size_t)N * es,
                                          hipMemcpyDeviceToHost, c->stream));
        } else {
            for (int64_t t = 0; t < p->n_trees; t++)
                if (ng[(size_t)t] > 0)
                    HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(grad) + (size_t)goff[(size_t)t] * es,
                                              static_cast<char *>(sGrad.dev) + (size_t)goff[(size_t)t] * es,
                                              (size_t)ng[(size_t)t] * (size_t)N * es, hipMemcpyDeviceToHost, c->stream));
        }
    }
    if (sOk.staged) HIP_TRY(c, hipMemcpyAsync(ok, sOk.dev, (size_t)p->n_trees, hipMemcpyDeviceToHost, c->stream));
    if (sX.staged || sOut.staged || sGrad.staged || sOk.staged || sPar.staged || sCls.staged) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (g.e.skip_flagged && ((out && sOut.staged) || sGrad.staged)) {
        // host buffers: rows / Jacobians of incomplete trees were only partly written into staging buffers every program of the context
        // shares — NaN-fill them (as eval_impl does; src/EvaluationHelpers.jl:56-62 does the same one level up)
        std::vector<uint8_t> okh;
        const uint8_t *okp = ok;
        if (ok_dev) {
            okh.resize((size_t)p->n_trees);
            HIP_TRY(c, hipMemcpy(okh.data(), ok, (size_t)p->n_trees, hipMemcpyDeviceToHost));
            okp = okh.data();
        }
        auto fill = [&](void *base, size_t off, size_t n) {
            if (p->dtype == DE_F32) std::fill_n(static_cast<float *>(base) + off, n, std::nanf(""));
            else std::fill_n(static_cast<double *>(base) + off, n, std::nan(""));
        };
        for (int64_t t = 0; t < p->n_trees; t++) {
            if (okp[t]) continue;
            if (out && sOut.staged) fill(out, (size_t)t * (size_t)ld_out, (size_t)N);
            if (sGrad.staged && diff) fill(grad, (size_t)t * (size_t)ld_out, (size_t)N);
            else if (sGrad.staged && ng[(size_t)t] > 0) fill(grad, (size_t)goff[(size_t)t], (size_t)ng[(size_t)t] * (size_t)N);
        }
    }
    return DE_OK;
}

// By-class reduction in ONE pass (de_eval_loss_grad_by_class): class-aligned tiles, then one pair of finish passes per
// class into loss_c / dloss_c ([C][n_trees] and [C][span], device).  Only the reverse kernel takes a tile table:
// `done` stays false when the population runs forward duals and the caller falls back to one call per class.
struct ByClassPlan {
    const int64_t *class_starts;
    int64_t C, span;
    void *loss_c, *dloss_c;
    bool done;
};
static int loss_grad_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                          int mode, const void *y, const void *w, int32_t loss_kind, void *loss, void *dloss,
                          const int64_t *dloss_offsets, uint8_t *ok, ByClassPlan *plan);
int de_eval_loss_grad(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                      int mode, const void *y, const void *w, int32_t loss_kind, void *loss, void *dloss,
                      const int64_t *dloss_offsets, uint8_t *ok) {
    DE_NOTHROW(c, loss_grad_impl(c, p, X, N, ldX, pa, mode, y, w, loss_kind, loss, dloss, dloss_offsets, ok, nullptr));
}
static int loss_grad_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                          int mode, const void *y, const void *w, int32_t loss_kind, void *loss, void *dloss,
                          const int64_t *dloss_offsets, uint8_t *ok, ByClassPlan *plan) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (p->ctx != c) return fail(c, DE_ERR_INVALID_ARG, "program belongs to another context");
    if (N < 0 || !ok || (p->n_trees > 0 && (!dloss || (N > 0 && (!X || !y))))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (ldX < p->n_features) return fail(c, DE_ERR_INVALID_ARG, "ldX < n_features");
    if (mode != DE_GRAD_VARIABLE && mode != DE_GRAD_CONSTANT && mode != DE_GRAD_BOTH) return fail(c, DE_ERR_INVALID_ARG, "bad gradient mode");
    if (loss_kind != DE_LOSS_L2 && loss_kind != DE_LOSS_L1 && loss_kind != DE_LOSS_PULLBACK)
        return fail(c, DE_ERR_INVALID_ARG, "unknown loss_kind %d", loss_kind);
    int rc = check_param_args(c, p, pa, N);
    if (rc != DE_OK) return rc;
    if (p->n_trees == 0) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    // per-tree geometry: tree t owns reduction columns col_off[t] (loss) .. col_off[t] + n_grad[t]
    std::vector<int32_t> ng((size_t)p->n_trees);
    std::vector<int64_t> coloff((size_t)p->n_trees + 1, 0), doff((size_t)p->n_trees);
    int64_t span = 0, run = 0;
    int32_t maxg = 0;
    for (int64_t t = 0; t < p->n_trees; t++) {
        const int32_t g = (int32_t)de_program_n_grad(p, t, mode);
        ng[(size_t)t] = g;
        maxg = std::max(maxg, g);
        const int64_t off = dloss_offsets ? dloss_offsets[t] : run;
        if (off < 0) return fail(c, DE_ERR_INVALID_ARG, "negative dloss offset");
        doff[(size_t)t] = off;
        run += g;
        span = std::max(span, off + g);
        coloff[(size_t)t + 1] = coloff[(size_t)t] + 1 + g;
    }
    const int64_t n_cols = coloff[(size_t)p->n_trees];
    const bool ok_dev = is_device_ptr(ok);
    if (N == 0) { // empty sums: 0, or NaN where a constant already fails the flag
        std::vector<unsigned char> zl((size_t)p->n_trees * es), zd((size_t)std::max<int64_t>(span, 1) * es);
        auto put = [&](unsigned char *b, int64_t i, double v) {
            if (p->dtype == DE_F32) reinterpret_cast<float *>(b)[i] = (float)v;
            else reinterpret_cast<double *>(b)[i] = v;
        };
        for (int64_t t = 0; t < p->n_trees; t++) {
            const double v = p->host_ok_grad[(size_t)t] ? 0.0 : std::nan("");
            put(zl.data(), t, v);
            for (int32_t k = 0; k < ng[(size_t)t]; k++) put(zd.data(), doff[(size_t)t] + k, v);
        }
        for (int64_t t = 0; t < p->n_trees; t++) // only the entries each tree owns are written
            if (ng[(size_t)t] > 0)
                HIP_TRY(c, hipMemcpy(static_cast<char *>(dloss) + (size_t)doff[(size_t)t] * es, zd.data() + (size_t)doff[(size_t)t] * es,
                                     (size_t)ng[(size_t)t] * es, hipMemcpyDefault));
        if (loss) HIP_TRY(c, hipMemcpy(loss, zl.data(), zl.size(), hipMemcpyDefault));
        HIP_TRY(c, hipMemcpy(ok, p->host_ok_grad.data(), (size_t)p->n_trees, hipMemcpyDefault));
        return DE_OK;
    }
    const bool timing = getenv("DE_DEBUG_TIMING") != nullptr;
    const auto tg0 = std::chrono::steady_clock::now();
    rc = ensure_generic_code(c, p);
    if (rc) return rc;
    const auto tg1 = std::chrono::steady_clock::now();

    Staged sX, sY, sW, sLoss, sDl, sOk, sPar, sCls;
    rc = stage_in(c, c->sX, X, (size_t)ldX * (size_t)N * es, &sX);
    if (rc) return rc;
    rc = stage_in(c, c->sY, y, (size_t)N * es, &sY);
    if (rc) return rc;
    if (w) {
        rc = stage_in(c, c->sW, w, (size_t)N * es, &sW);
        if (rc) return rc;
    }
    if (loss) {
        rc = stage_out(c, c->sLoss, loss, (size_t)p->n_trees * es, &sLoss);
        if (rc) return rc;
    }
    rc = stage_out(c, c->sDloss, dloss, (size_t)std::max<int64_t>(span, 1) * es, &sDl);
    if (rc) return rc;
    if (ok_dev) sOk.dev = ok;
    else {
        HIP_TRY(c, c->sOk.reserve((size_t)p->n_trees));
        sOk.dev = c->sOk.p;
        sOk.staged = true;
    }
    int64_t n_tiles = (N + 255) / 256;
    std::vector<int64_t> tile_range, class_tile0; // by-class: (first, last) sample of every class-aligned tile; first tile of every class
    if (plan) {
        class_tile0.assign((size_t)plan->C + 1, 0);
        for (int64_t k = 0; k < plan->C; k++) {
            const int64_t j0 = plan->class_starts[k], j1 = plan->class_starts[k + 1];
            for (int64_t b = j0; b < j1; b += 256) {
                tile_range.push_back(b);
                tile_range.push_back(j1 - 1);
            }
            class_tile0[(size_t)k + 1] = (int64_t)(tile_range.size() / 2);
        }
        n_tiles = (int64_t)(tile_range.size() / 2);
    }
    HIP_TRY(c, c->sPartial.reserve((size_t)n_tiles * (size_t)n_cols * 4 * es));
    // (by class: three regions — the finish passes of the classes run on the caller's stream and two side streams, launch_loss_grad_finish_ranges)
    const size_t seg_region = (size_t)loss_segments(n_tiles) * (size_t)n_cols * 4 * sizeof(double);
    const int seg_regions = plan ? 3 : 1;
    HIP_TRY(c, c->sSeg.reserve(seg_region * (size_t)seg_regions));
    HIP_TRY(c, c->sNg.reserve(ng.size() * sizeof(int32_t)));
    HIP_TRY(c, c->sColOff.reserve(coloff.size() * sizeof(int64_t)));
    HIP_TRY(c, c->sDoff.reserve(doff.size() * sizeof(int64_t)));
    HIP_TRY(c, hipMemcpyAsync(sOk.dev, p->host_ok_grad.data(), (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sNg.p, ng.data(), ng.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sColOff.p, coloff.data(), coloff.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sDoff.p, doff.data(), doff.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (p->uses_params) {
        rc = stage_in(c, c->sParams, pa->params, (size_t)pa->ld_params * (size_t)pa->n_classes * es, &sPar);
        if (rc) return rc;
        rc = stage_in(c, c->sClasses, pa->classes, (size_t)N * (pa->classes_is_i64 ? 8 : 4), &sCls);
        if (rc) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // the pageable host vectors above must outlive their async copies

    LossArgs la;
    std::memset(&la, 0, sizeof la);
    la.y = sY.dev;
    la.w = w ? sW.dev : nullptr;
    la.kind = loss_kind;
    la.partial = c->sPartial.p;
    la.seg_sum = c->sSeg.p;
    la.loss = loss ? sLoss.dev : nullptr;
    GradArgs g;
    std::memset(&g, 0, sizeof g);
    g.generic_code = p->d_gcode;
    g.e.code_off = p->d_gcode_off;
    g.e.n_trees = (int32_t)p->n_trees;
    g.e.skip_flagged = !(p->options & DE_OPT_FULL_EVAL) && tree_skip_enabled(); // (the gradient entry points always test validity)
    if (c->sPrio.reserve((size_t)3 * DE_PRIO_MAX_F * sizeof(unsigned long long)) == hipSuccess) g.e.prio_keys = c->sPrio.p; // priority tiles (de_kernels.hip)
    g.prio_ready = false;
    g.e.prio_keys_ready = !sX.staged && g.e.prio_keys && dataset_keys(c, p->dtype, X, N, ldX, p->n_features, &g.e.prio_keys);
    g.e.n_slots = p->n_slots;
    g.e.uses_params = p->uses_params;
    g.e.X = sX.dev;
    g.e.N = N;
    g.e.ldX = ldX;
    g.e.F = p->n_features;
    g.e.out = nullptr;
    g.e.ld_out = N;
    g.e.ok = static_cast<uint8_t *>(sOk.dev);
    if (p->uses_params) {
        g.e.params = sPar.dev;
        g.e.ld_params = pa->ld_params;
        g.e.n_classes = pa->n_classes;
        g.e.classes = sCls.dev;
        g.e.classes_is_i64 = pa->classes_is_i64;
        g.e.class_base = pa->class_base;
    }
    g.mode = mode;
    g.P = p->n_params;
    g.grad = nullptr;
    g.grad_off = nullptr;
    g.n_grad = static_cast<const int32_t *>(c->sNg.p);
    g.max_grad = maxg;
    g.diff_direction = -1;
    g.loss = &la;
    g.col_off = static_cast<const int64_t *>(c->sColOff.p);
    g.n_cols = n_cols;
    g.dloss = sDl.dev;
    g.dloss_off = static_cast<const int64_t *>(c->sDoff.p);
    rc = ensure_rev_threaded(c, p, mode, &g);
    if (rc) return rc;
    if (plan && !g.rev_code) return DE_OK; // forward duals: the caller runs one call per class (plan->done stays false)
    if (plan) {
        HIP_TRY(c, c->sBcTiles.reserve(std::max<size_t>(tile_range.size(), 2) * sizeof(int64_t)));
        if (!tile_range.empty())
            HIP_TRY(c, hipMemcpyAsync(c->sBcTiles.p, tile_range.data(), tile_range.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream)); // tile_range is pageable
        g.rev_tile_range = static_cast<const int64_t *>(c->sBcTiles.p);
        g.rev_n_tiles = n_tiles;
    }
    if (!g.rev_code) {
        const size_t lds_need = ((size_t)p->n_features + (size_t)p->n_slots * (1 + (size_t)std::min(maxg, 8))) * 260 * es;
        if (lds_need > 160 * 1024) return fail(c, DE_ERR_UNSUPPORTED, "gradient kernel: LDS footprint too large for this tree shape");
        rc = ensure_grad_threaded(c, p, mode, ng, N, &g);
        if (rc) return rc;
    }
    if (timing) {
        const auto tg2 = std::chrono::steady_clock::now();
        fprintf(stderr, "loss_grad host us: generic code %ld, staging + threaded/reverse code %ld\n",
                (long)std::chrono::duration_cast<std::chrono::microseconds>(tg1 - tg0).count(),
                (long)std::chrono::duration_cast<std::chrono::microseconds>(tg2 - tg1).count());
    }
    if (!c->nested) HIP_TRY(c, time_begin(c));
    if (g.rev_code) HIP_TRY(c, launch_rev_threaded(p->dtype, g, c->stream, &c->last_kernel));
    else HIP_TRY(c, launch_grad(p->dtype, g, c->stream, &c->last_kernel));
    if (plan) { // one pair of finish passes per class over its own tiles
        HIP_TRY(c, launch_loss_grad_finish_ranges(p->dtype, g, plan->C, class_tile0.data(), plan->loss_c, (size_t)p->n_trees * es, plan->dloss_c,
                                                  (size_t)plan->span * es, seg_region, seg_regions, c->stream));
        plan->done = true;
    }
    if (!c->nested) HIP_TRY(c, time_end(c));
    if (sLoss.staged) HIP_TRY(c, hipMemcpyAsync(loss, sLoss.dev, (size_t)p->n_trees * es, hipMemcpyDeviceToHost, c->stream));
    if (sDl.staged)
        for (int64_t t = 0; t < p->n_trees; t++)
            if (ng[(size_t)t] > 0)
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(dloss) + (size_t)doff[(size_t)t] * es,
                                          static_cast<char *>(sDl.dev) + (size_t)doff[(size_t)t] * es, (size_t)ng[(size_t)t] * es,
                                          hipMemcpyDeviceToHost, c->stream));
    if (sOk.staged) HIP_TRY(c, hipMemcpyAsync(ok, sOk.dev, (size_t)p->n_trees, hipMemcpyDeviceToHost, c->stream));
    if (sX.staged || sY.staged || sW.staged || sLoss.staged || sDl.staged || sOk.staged || sPar.staged || sCls.staged)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DE_OK;
}

static int by_class_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX,
                               const de_param_args_t *pa, int mode, const void *y, const void *w, int32_t loss_kind,
                               const int64_t *class_starts, void *loss, void *dloss, const int64_t *dloss_offsets,
                               void *dparams, uint8_t *ok);
int de_eval_loss_grad_by_class(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX,
                               const de_param_args_t *pa, int mode, const void *y, const void *w, int32_t loss_kind,
                               const int64_t *class_starts, void *loss, void *dloss, const int64_t *dloss_offsets,
                               void *dparams, uint8_t *ok) {
    DE_NOTHROW(c, by_class_impl(c, p, X, N, ldX, pa, mode, y, w, loss_kind, class_starts, loss, dloss, dloss_offsets, dparams, ok));
}
static int by_class_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX,
                               const de_param_args_t *pa, int mode, const void *y, const void *w, int32_t loss_kind,
                               const int64_t *class_starts, void *loss, void *dloss, const int64_t *dloss_offsets,
                               void *dparams, uint8_t *ok) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (p->ctx != c) return fail(c, DE_ERR_INVALID_ARG, "program belongs to another context");
    if (p->n_params <= 0 || !pa) return fail(c, DE_ERR_INVALID_ARG, "not a parametric population (n_params = 0 or no parameter arguments)");
    if (mode != DE_GRAD_VARIABLE && mode != DE_GRAD_BOTH)
        return fail(c, DE_ERR_INVALID_ARG, "by-class reduction needs a mode with parameter rows (DE_GRAD_VARIABLE / DE_GRAD_BOTH)");
    if (!pa->params || !pa->classes || pa->ld_params < p->n_params || pa->n_classes <= 0)
        return fail(c, DE_ERR_INVALID_ARG, "bad parameter arguments");
    if (N < 0 || !ok || !class_starts || (p->n_trees > 0 && (!dloss || !dparams))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    const int64_t C = pa->n_classes;
    if (class_starts[0] != 0 || class_starts[C] != N) return fail(c, DE_ERR_INVALID_ARG, "class_starts must run from 0 to N");
    for (int64_t k = 0; k < C; k++)
        if (class_starts[k + 1] < class_starts[k]) return fail(c, DE_ERR_INVALID_ARG, "class_starts must be non-decreasing");
    if (p->n_trees == 0) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    const int P = p->n_params;
    std::vector<int32_t> ng((size_t)p->n_trees);
    std::vector<int64_t> doff((size_t)p->n_trees);
    int64_t span = 0, run = 0;
    for (int64_t t = 0; t < p->n_trees; t++) {
        const int32_t g = (int32_t)de_program_n_grad(p, t, mode);
        ng[(size_t)t] = g;
        const int64_t off = dloss_offsets ? dloss_offsets[t] : run;
        if (off < 0) return fail(c, DE_ERR_INVALID_ARG, "negative dloss offset");
        doff[(size_t)t] = off;
        run += g;
        span = std::max(span, off + g);
    }
    span = std::max<int64_t>(span, 1);
    HIP_TRY(c, c->sBcLoss.reserve((size_t)C * (size_t)p->n_trees * es));
    HIP_TRY(c, c->sBcDloss.reserve((size_t)C * (size_t)span * es));
    HIP_TRY(c, c->sBcOk.reserve((size_t)C * (size_t)p->n_trees));
    HIP_TRY(c, c->sBcNg.reserve(ng.size() * sizeof(int32_t)));
    HIP_TRY(c, c->sBcDoff.reserve(doff.size() * sizeof(int64_t)));
    HIP_TRY(c, hipMemcpyAsync(c->sBcNg.p, ng.data(), ng.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->sBcDoff.p, doff.data(), doff.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    // dloss entries no tree owns (caller-chosen offsets) are never read by the combine pass
    HIP_TRY(c, hipMemsetAsync(c->sBcDloss.p, 0, (size_t)C * (size_t)span * es, c->stream));
    HIP_TRY(c, time_begin(c));
    ByClassPlan plan{class_starts, C, span, c->sBcLoss.p, c->sBcDloss.p, false};
    bool shared_ok = false; // one pass: a single flag array instead of one per class
    {
        const char *env1 = getenv("DE_BY_CLASS_ONE_PASS");
        if (!(env1 && *env1 == '0') && N > 0) {
            c->nested++;
            const int rc1 = loss_grad_impl(c, p, X, N, ldX, pa, mode, y, w, loss_kind, c->sBcLoss.p, c->sBcDloss.p, dloss_offsets,
                                           static_cast<uint8_t *>(c->sBcOk.p), &plan);
            c->nested--;
            if (rc1 != DE_OK) return rc1;
            shared_ok = plan.done;
        }
    }
    struct Nest { // inner calls leave the timing events alone; restored on every exit path
        de_ctx *c;
        explicit Nest(de_ctx *c_) : c(c_) { c->nested++; }
        ~Nest() { c->nested--; }
    };
    int rc = DE_OK;
    const size_t cls_es = pa->classes_is_i64 ? 8 : 4;
    {
    Nest nest(c);
    for (int64_t k = 0; k < C && rc == DE_OK && !plan.done; k++) {
        const int64_t j0 = class_starts[k], n = class_starts[k + 1] - j0;
        de_param_args_t sub = *pa;
        sub.classes = static_cast<const char *>(pa->classes) + (size_t)j0 * cls_es;
        rc = de_eval_loss_grad(c, p, static_cast<const char *>(X) + (size_t)j0 * (size_t)ldX * es, n, ldX, &sub, mode,
                               y ? static_cast<const char *>(y) + (size_t)j0 * es : nullptr,
                               w ? static_cast<const char *>(w) + (size_t)j0 * es : nullptr, loss_kind,
                               static_cast<char *>(c->sBcLoss.p) + (size_t)k * (size_t)p->n_trees * es,
                               static_cast<char *>(c->sBcDloss.p) + (size_t)k * (size_t)span * es, dloss_offsets,
                               static_cast<uint8_t *>(c->sBcOk.p) + (size_t)k * (size_t)p->n_trees);
    }
    }
    if (rc != DE_OK) return rc;
    Staged sLoss, sDl, sDp, sOk;
    if (loss) {
        rc = stage_out(c, c->sLoss, loss, (size_t)p->n_trees * es, &sLoss);
        if (rc) return rc;
    }
    rc = stage_out(c, c->sDloss, dloss, (size_t)span * es, &sDl);
    if (rc) return rc;
    const size_t dp_bytes = (size_t)p->n_trees * (size_t)C * (size_t)P * es;
    rc = stage_out(c, c->sBcOut, dparams, dp_bytes, &sDp);
    if (rc) return rc;
    rc = stage_out(c, c->sOk, ok, (size_t)p->n_trees, &sOk);
    if (rc) return rc;
    ByClassArgs a;
    a.loss_c = c->sBcLoss.p;
    a.dloss_c = c->sBcDloss.p;
    a.ok_c = static_cast<const uint8_t *>(c->sBcOk.p);
    a.n_classes = (int32_t)C;
    a.ok_stride = shared_ok ? 0 : p->n_trees;
    a.n_params = P;
    a.n_trees = p->n_trees;
    a.span = span;
    a.n_grad = static_cast<const int32_t *>(c->sBcNg.p);
    a.dloss_off = static_cast<const int64_t *>(c->sBcDoff.p);
    a.loss = loss ? sLoss.dev : nullptr;
    a.dloss = sDl.dev;
    a.dparams = sDp.dev;
    a.ok = static_cast<uint8_t *>(sOk.dev);
    HIP_TRY(c, launch_by_class_combine(p->dtype, a, c->stream));
    HIP_TRY(c, time_end(c));
    if (sLoss.staged) HIP_TRY(c, hipMemcpyAsync(loss, sLoss.dev, (size_t)p->n_trees * es, hipMemcpyDeviceToHost, c->stream));
    if (sDl.staged)
        for (int64_t t = 0; t < p->n_trees; t++)
            if (ng[(size_t)t] > 0)
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(dloss) + (size_t)doff[(size_t)t] * es,
                                          static_cast<char *>(sDl.dev) + (size_t)doff[(size_t)t] * es, (size_t)ng[(size_t)t] * es,
                                          hipMemcpyDeviceToHost, c->stream));
    if (sDp.staged) HIP_TRY(c, hipMemcpyAsync(dparams, sDp.dev, dp_bytes, hipMemcpyDeviceToHost, c->stream));
    if (sOk.staged) HIP_TRY(c, hipMemcpyAsync(ok, sOk.dev, (size_t)p->n_trees, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // ng/doff (pageable) were copied asynchronously
    return DE_OK;
}

int de_eval_grad(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                 int mode, void *out, int64_t ld_out, void *grad, const int64_t *grad_offsets, uint8_t *ok) {
    DE_NOTHROW(c, grad_impl(c, p, X, N, ldX, pa, mode, -1, out, ld_out, grad, grad_offsets, ok));
}

int de_eval_pullback_dX(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                        const void *dY, void *dX, const int64_t *dX_offsets, uint8_t *ok) {
    if (c && N > 0 && !dY) return fail(c, DE_ERR_INVALID_ARG, "null cotangent dY");
    DE_NOTHROW(c, grad_impl(c, p, X, N, ldX, pa, DE_GRAD_VARIABLE, -1, nullptr, N, dX, dX_offsets, ok, dY));
}

int de_eval_diff(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, int32_t direction, void *out,
                 void *dout, int64_t ld_out, uint8_t *ok) {
    if (direction < 0) return fail(c, DE_ERR_INVALID_ARG, "direction < 0");
    if (p && p->uses_params) return fail(c, DE_ERR_UNSUPPORTED, "eval_diff on parametric trees");
    DE_NOTHROW(c, grad_impl(c, p, X, N, ldX, nullptr, DE_GRAD_VARIABLE, direction, out, ld_out, dout, nullptr, ok));
}

} // extern "C"
