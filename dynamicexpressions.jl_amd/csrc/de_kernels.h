// de_kernels.h — launch interface between the C ABI (de_api*.cpp) and the gfx950
// kernels (de_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/de_hip.h"
#include "de_bind.h"
#include "de_program.h"

namespace de {

constexpr int BLOCK = 256; // threads per workgroup = 4 wavefronts of 64 (flat-switch and gradient kernels)
// Threaded eval kernel: threads per workgroup.  The kernel is latency-bound: waves per SIMD are what hide
// the LDS / instruction-fetch latency of every dispatch, and LDS — (n_features + slots) rows of DE_TBLK 16-byte vectors per
// workgroup — is what limits them.  ONE wave per workgroup since round 3 (256-sample tiles): the same LDS bytes per wave as
// with two, but a wave no longer holds its partner's LDS and slot while that one finishes its trees, and the staging barrier
// is wave-local (headline 7.73 -> 7.61 ms, C4 9.88 -> 9.63; 4-16 trees x 5e7 samples 3-5 % slower; round 2: 256 -> 128 neutral).
#ifndef DE_TBLK
#define DE_TBLK 64
#endif
constexpr int TBLK = DE_TBLK, TWAVES = DE_TBLK / 64;
// PLANES: 16-byte vectors per lane and row (de_kernels.hip TG<T>).  The handlers are written for any number of planes; the shipped
// build uses ONE for both element types (4 Float32 / 2 Float64 samples per lane, 256- / 128-sample tiles).  Two Float32 planes
// (DE_EXTRA_FLAGS=-DDE_TG=2 bash build.sh: 8 samples per lane, 512-sample tiles, half the dispatches, scalar instructions and tree ends per
// sample, an independent twin for every dependent VALU chain) were built, pass the GPU tests and measured SLOWER in round 4: rows twice as
// long leave 2.75 instead of 5.25 waves per SIMD and the kernel saturates there (complete trees only, same box: 15.3 against 14.6 ms;
// headline 7.17 / 6.95-7.5; turbo 6.02 / 5.42; full evaluation 19.9 / 17.7; occupancy sweeps of both in profiles/r4_planes_occupancy.txt):
// what binds the kernel is VALU work per sample, not the per-dispatch overhead.
#ifndef DE_TG
#define DE_TG 1
#endif
constexpr int TG_F32 = DE_TG, TG_F64 = 1;
constexpr int tg_planes(int dtype) { return dtype == DE_F32 ? TG_F32 : TG_F64; }
constexpr int ttile_samples(int dtype) { return TBLK * tg_planes(dtype) * (dtype == DE_F32 ? 4 : 2); } // samples per workgroup tile
constexpr size_t trow_bytes(int dtype) { return (size_t)(TBLK * tg_planes(dtype) + 1) * 16; } // LDS row stride: the planes + one vector of padding (bank spread)

// Fused loss epilogue of the threaded eval kernel (de_eval_loss): instead of storing out[t][j] the
// kernel reduces sum_j w_j * l(out[t][j] - y[j]) per tree (per-wave partials + two fixed-order passes).
struct LossArgs {
    const void *y;    // device, N
    const void *w;    // device, N, or null (= 1)
    int32_t kind;     // de_loss_kind
    void *partial;    // device scratch, loss_scratch_bytes().partial
    void *seg_sum;    // device scratch (double), loss_scratch_bytes().seg
    void *loss;       // device, n_trees values of the program's dtype
};

struct EvalArgs {
    // program
    const BoundInstr *code;   // device: all trees' BOUND instructions (de_bind.h), +1 pad
    const int32_t *code_off;  // device: n_trees+1 offsets into code
    int32_t n_trees;
    int32_t n_slots;          // LDS rows behind X: spill slots (max over trees) + staged parameter rows
    int32_t prow_base, n_prows; // parameters staged as LDS rows prow_base .. prow_base + n_prows - 1 (0: gathered per use, BOP_GEN_PARAM)
    bool uses_params;
    // data
    const void *X;            // device, [F, N] col-major, ld = ldX
    int64_t N, ldX;
    int32_t F;
    void *out;                // device, [n_trees, ld_out]
    int64_t ld_out;
    uint8_t *ok;              // device, n_trees bytes, pre-set by the caller — or by the launch when ok_init is given
    const uint8_t *ok_init;   // device, n_trees bytes or null: the flags' initial values (the constant part of the flag); launch_eval puts them into
                              // `ok` in front of its first kernel (round 6: inside the priority-tile pre-pass when that runs)
    // parametric
    const void *params;       // device [P, C]
    int64_t ld_params;
    int64_t n_classes;        // columns of params (a small table is staged in LDS)
    const void *classes;      // device, N ids
    int32_t classes_is_i64, class_base;
    // flags
    bool early_exit;
    bool skip_flagged;        // early exit at tree granularity: workgroups do not evaluate trees whose flag is already 0 (off: DE_OPT_FULL_EVAL)
    bool turbo;               // DE_OPT_TURBO (threaded kernel, Float32): relaxed-accuracy cos / exp / sin / division handlers
    // threaded-code variant: `code` holds handler OFFSETS (relative to handler_base) in word 0 and
    // LDS byte offsets in the low 24 bits of word 1
    bool threaded;
    bool direct; // feature matrix too wide for the LDS tile: gather features from global memory (flat-switch kernel)
    const LossArgs *loss; // non-null: fused loss instead of the output store (threaded kernel only)
    void *prio_keys;      // device scratch, 3 * DE_PRIO_MAX_F 64-bit keys: the priority tiles of a large early-exit launch (de_kernels.hip de_tile_extremes_kernel); null: none
    bool prio_keys_ready; // prio_keys already hold the keys of THIS X (de_ctx_declare_dataset: computed once for a dataset that does not change between calls): no pre-pass
    // compaction of the live trees behind the probe launch of the priority tiles (de_kernels.hip de_compact_live_kernel; threaded kernel only):
    // compact_code = room for a second copy of the chained stream INSIDE the same 4 GiB window as `code`, compact_ints = (n_trees + 1) record
    // offsets + n_trees tree indices + 4 control words.  Null: the launch proper walks past flagged trees (round 3).
    void *compact_code;
    int32_t *compact_ints;
    bool *compacted; // out (may be null): this launch compacted its live trees
    // wave groups (de_kernels.hip KArgs::var_stride): `code` (and the room behind compact_code) holds `waves` variants of the chained stream
    // var_stride records apart (0: the trees use no spill slot, one stream serves every wave), each with wave_slots spill-slot rows of its own;
    // 0 / 1 waves: one-wave workgroups
    int32_t waves, wave_slots;
    int64_t var_stride;
    // de_eval_sum_certificate: device array of n_trees zeroed words of the element type's size; non-null selects the CERT variant of the
    // flat-switch kernel (no output): the bits of the largest |validity-tested value| of every tree
    void *cert_max;
};
constexpr int DE_PRIO_MAX_F = 8; // (the pre-pass keeps 6 registers per feature)
constexpr int DE_PRIO_UNIT = 64;  // samples per unit of the keys' position field (every kernel's tile is a multiple)
// multi-GPU flag exchange (de_dist.cpp): pack the local flags into the padded send block / scatter the gathered blocks to global tree order
// constant subtrees (one thread each; de_kernels.hip de_fold_kernel): stack depth a subtree may need
#define DE_FOLD_STACK 16
hipError_t launch_fold(int dtype, const void *nodes, const int64_t *noff, const int64_t *coff, const void *cvals, int64_t n_folds, void *out,
                       uint8_t *ok, hipStream_t stream);
hipError_t launch_dist_pack(uint8_t *send, const uint8_t *ok_local_dev, int64_t mine, int64_t per, hipStream_t stream);
hipError_t launch_dist_unpack(uint8_t *ok_global_dev, const uint8_t *recv, int64_t per, int32_t world, int64_t n_trees, hipStream_t stream);
// (ok_init / ok / n_ok, optional: the kernel also copies the n_ok initial flag bytes into ok — the first kernel of an eval launch)
hipError_t launch_tile_extremes(int dtype, const void *X, int64_t N, int64_t ldX, int F, void *keys, hipStream_t stream, const uint8_t *ok_init = nullptr,
                                uint8_t *ok = nullptr, int32_t n_ok = 0);
bool prio_tiles_wanted(int64_t N, int F, int64_t n_trees);

struct GradArgs {
    EvalArgs e;               // e.code is unused: the gradient kernel runs the bound UNFOLDED program
    const BoundInstr *generic_code; // device, +1 pad (bound form of the unfolded generic program)
    int32_t mode;             // de_grad_mode
    int32_t P;                // n_params (rows before the features in VARIABLE/BOTH)
    void *grad;               // device
    const int64_t *grad_off;  // device, n_trees element offsets of each tree's [G_t, N] matrix
    const int32_t *n_grad;    // device, n_trees: G_t
    int32_t max_grad;         // max G_t
    int32_t diff_direction;   // >=0: eval_diff mode (single direction, output dout rows)
    // fused loss + pullback (de_eval_loss_grad): loss->partial is [n_tiles(256 samples)][n_cols][4]
    const LossArgs *loss;
    const int64_t *col_off;   // device, n_trees + 1: tree t owns columns col_off[t] .. +n_grad[t] (loss first)
    int64_t n_cols;           // = col_off[n_trees]
    void *dloss;              // device: tree t's n_grad[t] reduced gradient entries at dloss_off[t]
    const int64_t *dloss_off; // device, n_trees
    // threaded-code variant (de_grad_threaded.hip): non-null = use it.  Trees are grouped into buckets by
    // gradient width; each bucket is one launch of the module built for its window width.
    const BoundInstr *threaded_code;
    // reverse accumulation (de_rev_threaded.hip; fused loss + gradient only): forward + backward instruction
    // stream per tree, null = not available
    const BoundInstr *rev_code;
    const int32_t *rev_code_off, *rev_code_mid; // n_trees + 1 / n_trees
    const int32_t *rev_ids;                     // device: the trees grouped by LDS need (rev_groups)
    int32_t rev_n_groups;
    const int64_t *rev_tile_range; // by-class reduction: device (first, last) sample of each class-aligned tile; null = regular tiles
    int64_t rev_n_tiles;           // ... and their number (the caller runs the finish passes per class)
    struct RevGroup { int32_t first, n, rows; } rev_groups[8]; // ids[first .. first+n), LDS rows per wave (staging included)
    int32_t rev_stage_cols;                     // column sums a wave stages in LDS between two writes
    uint64_t rev_handler_base;
    uint32_t rev_param_off;
    bool prio_ready;  // e.prio_keys holds this launch's priority tiles (launch_grad_threaded / launch_rev_threaded run the pre-pass)
    // SHARED LEAF ROWS (round 6; de_grad_threaded.hip): the four waves of a workgroup run DIFFERENT trees on the SAME 64 x VS samples — X and
    // the parameter rows are staged once per workgroup instead of once per wave; every wave keeps slot rows of its own, so the stream
    // exists in four variants gt_var_stride records apart (variant w: slot offsets + w x the bucket's slot bytes)
    bool gt_share;
    int64_t gt_var_stride;
    int32_t n_buckets;
    struct Bucket {
        int32_t GC, VS;                // module: window width, samples per lane
        int32_t windows, max_grad;     // windows per tree, widest gradient in the bucket
        int32_t n_slots;               // spill slots the trees of the bucket need (LDS rows of the launch)
        const int32_t *ids;            // device: tree indices of the bucket
        int32_t n;
        uint64_t handler_base;
        uint32_t param_handler_off;
    } buckets[24];
};

// Returns hipSuccess or the failing HIP error.  `kernel_name` receives the symbol
// name of the launched kernel (for matching rocprofv3 kernel-trace rows).
hipError_t launch_eval(int dtype, const EvalArgs &a, hipStream_t stream, const char **kernel_name);
hipError_t launch_grad(int dtype, const GradArgs &a, hipStream_t stream, const char **kernel_name);

// Threaded gradient kernel: window width used for a population whose widest gradient has max_grad rows,
// handler addresses for (dtype, window), launch; pass 2+3 of the fused loss-gradient reduction.
int grad_window(int max_grad);
hipError_t grad_handler_table(int dtype, int GC, int VS, uint64_t *table); // GOP_MAX entries, gop_count(GC) used
bool grad_threaded_has(int dtype, int GC, int VS);
hipError_t rev_handler_table(int dtype, uint64_t *table); // ROP_COUNT entries
hipError_t launch_rev_threaded(int dtype, const GradArgs &a, hipStream_t stream, const char **kernel_name);                            // is there a module for (type, window, samples per lane)?
hipError_t launch_grad_threaded(int dtype, const GradArgs &a, hipStream_t stream, const char **kernel_name);
hipError_t launch_loss_grad_finish(int dtype, const GradArgs &ga, int64_t n_tiles, hipStream_t stream);
// the same over tiles [tile0, tile0 + n_tiles) only, results to loss / dloss (by-class reduction: one call per class)
hipError_t launch_loss_grad_finish_range(int dtype, const GradArgs &ga, int64_t tile0, int64_t n_tiles, void *loss, void *dloss, hipStream_t stream);
// ... of all classes at once, over the caller's stream and its side streams: seg_regions regions of ga.loss->seg_sum, seg_stride bytes apart
hipError_t launch_loss_grad_finish_ranges(int dtype, const GradArgs &ga, int64_t n_classes, const int64_t *tile0, void *loss, size_t loss_stride,
                                          void *dloss, size_t dloss_stride, size_t seg_stride, int seg_regions, hipStream_t stream);
// de_eval_loss_grad_by_class: per-class results ([C][n_trees] losses and flags, [C][span] gradients) -> outputs
struct ByClassArgs {
    const void *loss_c, *dloss_c;
    const uint8_t *ok_c;
    int32_t n_classes, n_params;
    int64_t ok_stride;       // n_trees: one flag array per class; 0: one shared array (single-pass reduction)
    int64_t n_trees, span;
    const int32_t *n_grad;   // device
    const int64_t *dloss_off; // device
    void *loss, *dloss, *dparams; // loss may be null
    uint8_t *ok;
};
hipError_t launch_by_class_combine(int dtype, const ByClassArgs &a, hipStream_t stream);
// de_eval_pullback_dX: grad[t][k, j] *= dY[j] (NaN where !ok[t]) over the Jacobians de_eval_grad wrote
hipError_t launch_pullback_scale(int dtype, void *grad, const int64_t *grad_off, const int32_t *n_grad, const uint8_t *ok,
                                 const void *dY, int64_t N, int64_t n_trees, int32_t max_grad, hipStream_t stream);

// The cached handler addresses are per device (every device loads its own copy of the code object): slot = the current device.
constexpr int DE_MAX_DEVICES = 64;
hipError_t handler_device_slot(int *slot);
// Threaded-code eval kernel: addresses of the TOPX_TABLE device handlers (cached per device).
hipError_t eval_handler_table(int dtype, bool turbo, uint64_t *table);
bool eval_uses_threaded();

// Launch plan of the eval kernel for (n_trees, N): samples per workgroup tile, tree chunks.
void eval_plan(int dtype, int64_t n_trees, int64_t N, int32_t *tile, int32_t *n_chunks, int32_t *trees_per_chunk, int waves = 1); // (waves: de_program::waves — wave groups)

// Scratch the fused-loss reduction needs for (dtype, n_trees, N).
void loss_scratch_bytes(int dtype, int64_t n_trees, int64_t N, size_t *partial_bytes, size_t *seg_bytes);

// Pass 2 of the fused-loss reductions: seg_sum[seg][col] = sum over the tiles of a segment of
// partial[tile][col] (fixed order, double).  n_cols counts (column, wave) pairs.
hipError_t launch_loss_reduce_tiles(int dtype, const void *partial, int64_t n_cols, int64_t n_tiles, void *seg_sum,
                                    int32_t *n_segs, hipStream_t stream);
int32_t loss_segments(int64_t n_tiles);

// LDS bytes the eval kernel needs for (dtype, F, n_slots); 0 if it cannot fit.
size_t eval_lds_bytes(int dtype, int F, int n_slots, int *K_out);

} // namespace de
