// de_api_internal.h — what the translation units of the C ABI share (round 6: de_api.cpp was one 3 900-line file):
//   de_api.cpp          registry, contexts, the pool of host threads, the pools of device buffers and parked programs
//   de_api_program.cpp  de_program_create / _set_consts / _destroy, constant folding, the threaded and chained streams, verify / dump / hash
//   de_api_eval.cpp     de_eval, de_eval_loss, de_eval_sum_certificate, de_eval_tree_array (staging, launch planning)
//   de_api_grad.cpp     de_eval_grad / _diff / _pullback_dX, de_eval_loss_grad(_by_class): generic, threaded and reverse gradient programs
// No behaviour changed in the split: de_program_stream_hash and the whole test suite are the check.
#ifndef DE_API_INTERNAL_H
#define DE_API_INTERNAL_H
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
    // WAVE GROUPS (round 6; de_kernels.hip KArgs::var_stride; de_api_program.cpp choose_waves): a program whose one-wave workgroup is short of
    // resident waves (staged parameter rows, many features) runs `waves` (2 / 4 / 8) waves per workgroup on one sample tile — X and the
    // parameter rows staged once, every wave a chunk and spill-slot rows of its own.  Slot rows
    // are host data, so the chained stream exists once per wave: ccode_w = variants 1 .. waves - 1 (each ccode.size() records; the same
    // records as ccode but for the operand words that name a slot row), on the device `var_stride` records apart behind variant 0
    // (0: the trees use no slot — one stream serves every wave).  waves == 1: one-wave workgroups, nothing of this exists.
    int waves = 1;
    int waves_choice = 0; // what choose_waves said when the program was created (0: not asked yet): a re-bind (de_program_set_consts under
                          // DE_NO_CONST_PATCH) keeps it — the arena was sized for it
    int64_t var_stride = 0;
    std::vector<BoundInstr> ccode_w;
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
    bool gt_share = false;        // the threaded gradient program exists in four variants (GradArgs::gt_share), gt_stride records apart in gtcode
    int64_t gt_stride = 0;
    size_t gt_cap = 0;            // records d_gtcode has room for
    int gt_n_buckets = 0;
    GradArgs::Bucket gt_buckets[24];
};

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t st_ = (expr);                                                             \
        if (st_ != hipSuccess) {                                                             \
            (void)hipGetLastError();                                                         \
            return fail((ctx), DE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(st_)); \
        }                                                                                    \
    } while (0)


extern "C" {  // (the definitions sit inside the translation units' extern "C" blocks)
int fail(de_ctx *c, int code, const char *fmt, ...);
hipError_t time_begin(de_ctx *c);
hipError_t time_end(de_ctx *c);
bool is_device_ptr(const void *p);
hipError_t prog_malloc(de_ctx *c, void **out, size_t bytes);
void prog_free(de_ctx *c, void *ptr);
void park_program(de_ctx *c, de_program *p);
void adopt_parked(de_ctx *c, de_program *fresh);
bool dataset_keys(const de_ctx *c, int dtype, const void *X, int64_t N, int64_t ldX, int32_t F, void **keys);
bool tree_skip_enabled();
void dbg_lap(const char *what);
struct Staged {
    void *dev = nullptr;
    bool staged = false;
};
int stage_in(de_ctx *c, DevBuf &buf, const void *user, size_t bytes, Staged *s);
int stage_out(de_ctx *c, DevBuf &buf, void *user, size_t bytes, Staged *s);
int check_param_args(de_ctx *c, const de_program *p, const de_param_args_t *pa, int64_t N);
// the pool of host threads (de_api.cpp): job(k) for k = 0 .. n - 1 on the pool (false: busy, nothing was run); threads a pass of n items gets
bool host_pool_run(int n, const std::function<void(int)> &job);
unsigned host_threads_for(int64_t n, int64_t grain);
}

static inline bool in_one_window(const void *ptr, size_t bytes) {
    const uint64_t a0 = (uint64_t)(uintptr_t)ptr;
    return bytes == 0 || (a0 >> 32) == ((a0 + bytes - 1) >> 32);
}
constexpr int HOST_RANGES_MAX = 32; // ranges of one parallel pass (per-worker vectors are arrays of this size)

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
    if (!host_pool_run(n_ranges, job))
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

// failure as std::bad_alloc) — an allocation failure becomes a status like everywhere else.
#define DE_NOTHROW(CTX, CALL)                                                                   \
    do {                                                                                        \
        try { return (CALL); }                                                                  \
        catch (const std::bad_alloc &) { return fail((CTX), DE_ERR_HIP, "out of host memory"); } \
        catch (const std::exception &e) { return fail((CTX), DE_ERR_HIP, "internal error: %s", e.what()); } \
        catch (...) { return fail((CTX), DE_ERR_HIP, "internal error (unknown exception)"); }  \
    } while (0)

#endif
