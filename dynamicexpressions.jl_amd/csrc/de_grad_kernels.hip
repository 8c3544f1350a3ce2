// de_grad_kernels.hip — forward-mode gradient of a population of trees on gfx950.
// Replaces the reference's src/EvaluateDerivative.jl: eval_grad_tree_array (:193-243),
// grad_degn_eval (:340-365), grad_deg0_eval (:367-404) and eval_diff_tree_array (:40-168).
//
// Every sample carries a dual number (x, d[0..GC)) through the same accumulator program the
// eval kernel runs (de_program.h, generic form).  GC = the window of gradient components a
// launch handles; wider gradients (constant mode with many constants, :both mode) are covered by
// several windows (blockIdx.y), each recomputing x — the alternative, (1+n_grad) live values
// per spill slot, would wreck occupancy.  Semantics follow the reference exactly:
//   * d[k] = g1*d1[k] + g2*d2[k] with dense arithmetic, leaf gradients materialised as 0/1
//     (so an infinite partial times a zero seed is NaN, as in grad_degn_eval);
//   * after EVERY node (leaves included) x and all d[k] are validity-tested (:239-242);
//   * eval_diff (single direction) performs NO validity test (:99-119).
// Partial derivatives restate ChainRules' scalar rules (same table as oracle/de_oracle_ops.h).
#include "de_grad_common.h"
#include <memory>
#include <mutex>
#include <utility>
#include <vector>
#include <algorithm>
#include <cstdlib>

namespace de {


// One sample per thread; LDS rows of GBLK(+4) elements: rows [0,F) = X tile, then each spill
// slot s owns 1+GC rows (x, d[0..GC)).
template <typename T, int GC>
__global__ void __launch_bounds__(GBLK) de_grad_tape_kernel(const GArgs<T> a) {
    constexpr int RS = GBLK + 4; // row stride (elements)
    extern __shared__ __align__(16) unsigned char gsmem[];
    T *__restrict__ rows = reinterpret_cast<T *>(gsmem);

    const GTileMap tm = gmap_block(blockIdx.x, a.n_chunks, a.n_tiles);
    if (!tm.valid) return;
    const int tid = threadIdx.x;
    const int64_t base = tm.tile * GBLK;
    const int64_t last = a.N - 1;
    const int g0 = a.diff_g0 >= 0 ? a.diff_g0 : (int)blockIdx.y * GC; // first gradient component of this window
    const int F = a.F, P = a.P;

    {
        const uint32_t Fu = (uint32_t)F;
        const uint32_t total = (uint32_t)GBLK * Fu;
        for (uint32_t e = tid; e < total; e += GBLK) {
            const uint32_t j = e / Fu, f = e - j * Fu;
            int64_t jj = base + j;
            jj = jj < last ? jj : last;
            rows[f * RS + j] = a.X[f + a.ldX * jj];
        }
    }
    int64_t jj0 = base + tid;
    const bool live = jj0 < a.N;
    jj0 = jj0 < last ? jj0 : last;
    int64_t cls = 0;
    if (a.uses_params)
        cls = clamp_class((a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj0]
                                            : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj0]) - a.class_base, a.n_classes);
    T yv = T(0), wv = T(0);
    if (a.loss_mode) {
        yv = a.y[jj0];
        wv = live ? (a.w ? a.w[jj0] : T(1)) : T(0);
    }
    __syncthreads();

    const ConstU4Ptr code = (ConstU4Ptr)(uintptr_t)a.code;
    const ConstI32Ptr code_off = (ConstI32Ptr)(uintptr_t)a.code_off;
    const ConstI64Ptr col_off = (ConstI64Ptr)(uintptr_t)a.col_off;
    const ConstI32Ptr n_grad = (ConstI32Ptr)(uintptr_t)a.n_grad;
    const ConstI64Ptr grad_off = (ConstI64Ptr)(uintptr_t)a.grad_off;
    const int t0 = tm.chunk * a.trees_per_chunk;
    const int t1 = (t0 + a.trees_per_chunk < a.n_trees) ? t0 + a.trees_per_chunk : a.n_trees;
    T *__restrict__ stk = rows + (size_t)F * RS;

    // window-local seed of a leaf = its gradient row - g0 (or < 0: no gradient in this mode/window)
    const int feat_seed0 = a.mode != DE_GRAD_CONSTANT ? P - g0 : -0x40000000;
    const int param_seed0 = a.mode != DE_GRAD_CONSTANT ? -g0 : -0x40000000;
    const int const_seed0 = a.mode == DE_GRAD_CONSTANT ? -g0 : (a.mode == DE_GRAD_BOTH ? P + F - g0 : -0x40000000);

    const uint64_t skip = gskip_mask(a.ok, (const int32_t *)nullptr, t0, t1, a.check ? a.skip_flagged : 0, (int64_t)tm.tile);
    for (int tree = t0; tree < t1; ++tree) {
        if ((skip >> (tree - t0)) & 1ull) continue; // already incomplete (early exit)
        const int G = a.diff_g0 >= 0 ? 1 : n_grad[tree];
        if (a.diff_g0 < 0 && g0 >= G && g0 > 0) continue; // window without a component of this tree (window 0 always runs: x, flag)
        int pc = code_off[tree];
        const int pe = code_off[tree + 1];
        T x = T(0), d[GC];
        DE_UNROLL for (int k = 0; k < GC; k++) d[k] = T(0);
        T poison = T(0);
        U32x4 nxt = code[pc];
        for (; pc < pe; ++pc) {
            const U32x4 w = nxt;
            nxt = code[pc + 1];
            // One flat wave-uniform switch over the bound handler id (de_bind.h); the operand is a LEAF
            // row (< F: one-hot seed), a spill ROW (>= F: a popped dual number) or a constant (seed).
#define G_SLOT(r) (stk + ((r) - (uint32_t)F) * (1 + GC) * RS + tid)
#define G_LEAF_ROW(r) { xb = rows[(r) * RS + tid]; seed = (int)(r) + feat_seed0; if (a.check) poison = M<T>::fma(xb, T(0), poison); }
#define G_ROW_OPERAND(r)                                                                           \
    T xb, db[GC];                                                                                  \
    if ((r) < (uint32_t)F) {                                                                       \
        int seed;                                                                                  \
        G_LEAF_ROW(r)                                                                              \
        DE_UNROLL for (int k = 0; k < GC; k++) db[k] = (k == seed) ? T(1) : T(0);                  \
    } else {                                                                                       \
        const T *__restrict__ s_ = G_SLOT(r);                                                      \
        xb = s_[0];                                                                                \
        DE_UNROLL for (int k = 0; k < GC; k++) db[k] = s_[(1 + k) * RS];                           \
    }
#define G_CONST_OPERAND()                                                                          \
    const T xb = gimm<T>(w.z, w.w);                                                                \
    T db[GC];                                                                                      \
    { const int seed = (int)(w.y & 0xFFFFu) + const_seed0;                                         \
      DE_UNROLL for (int k = 0; k < GC; k++) db[k] = (k == seed) ? T(1) : T(0); }
#define G_CHK() if (a.check) poison = M<T>::fma(x, T(0), poison);
            // binary hot ops: value v and partials (gl, gr) w.r.t. (left, right); REV: left = operand
#define G_BIN_APPLY(K)                                                                             \
    {                                                                                              \
        constexpr bool REV = (K == 2 || K == 5);                                                   \
        const T lx = REV ? xb : x, ly = REV ? x : xb;                                              \
        T v, gl, gr;                                                                               \
        if (K == 0) { v = lx + ly; gl = T(1); gr = T(1); }                                         \
        else if (K == 1 || K == 2) { v = lx - ly; gl = T(1); gr = T(-1); }                         \
        else if (K == 3) { v = lx * ly; gl = ly; gr = lx; }                                        \
        else { v = lx / ly; gl = T(1) / ly; gr = -(v / ly); }                                      \
        x = v;                                                                                     \
        if (REV) { DE_UNROLL for (int k = 0; k < GC; k++) d[k] = gl * db[k] + gr * d[k]; }         \
        else { DE_UNROLL for (int k = 0; k < GC; k++) d[k] = gl * d[k] + gr * db[k]; }             \
    }
#define G_BIN4(K)                                                                                  \
    case BOP_BIN_BASE + 4 * K + 0: { G_ROW_OPERAND(w.y) G_BIN_APPLY(K) } break;                    \
    case BOP_BIN_BASE + 4 * K + 1: { G_ROW_OPERAND(w.y) G_BIN_APPLY(K) G_CHK() } break;            \
    case BOP_BIN_BASE + 4 * K + 2: { G_CONST_OPERAND() G_BIN_APPLY(K) } break;                     \
    case BOP_BIN_BASE + 4 * K + 3: { G_CONST_OPERAND() G_BIN_APPLY(K) G_CHK() } break;
            // unary hot ops (K: 0 cos, 1 exp, 2 sin) on xin with incoming gradient din[]
#define G_UN_APPLY(K, XIN, DIN)                                                                    \
    {                                                                                              \
        const T xin_ = (XIN);                                                                      \
        UG<T> r;                                                                                   \
        if (sizeof(T) == 4) {                                                                      \
            if (K == 1) { r.y = (T)fast_exp_f32((float)xin_); r.g = r.y; }                         \
            else {                                                                                 \
                float sn, cs;                                                                      \
                fast_sincos_f32((float)xin_, &sn, &cs);                                            \
                if (K == 0) { r.y = (T)cs; r.g = (T)-sn; } else { r.y = (T)sn; r.g = (T)cs; }      \
                /* per ELEMENT: a sample's value must not depend on its wave neighbours */         \
                if (__ballot(M<T>::abs(xin_) > T(DE_TRIG_FAST_BOUND)) != 0ull) {                   \
                    const UG<T> slow = unary_vg<T>(K == 0 ? DE_U_COS : DE_U_SIN, xin_);            \
                    if (M<T>::abs(xin_) > T(DE_TRIG_FAST_BOUND)) r = slow;                         \
                }                                                                                  \
            }                                                                                      \
        } else r = unary_vg<T>(K == 0 ? DE_U_COS : (K == 1 ? DE_U_EXP : DE_U_SIN), xin_);          \
        x = r.y;                                                                                   \
        DE_UNROLL for (int k = 0; k < GC; k++) d[k] = r.g * DIN[k];                                \
    }
#define G_UN4(K)                                                                                   \
    case BOP_UN_BASE + 4 * K + 0: G_UN_APPLY(K, x, d) break;                                       \
    case BOP_UN_BASE + 4 * K + 1: G_UN_APPLY(K, x, d) G_CHK() break;                               \
    case BOP_UN_BASE + 4 * K + 2: { G_ROW_OPERAND(w.y) G_UN_APPLY(K, xb, db) } break;              \
    case BOP_UN_BASE + 4 * K + 3: { G_ROW_OPERAND(w.y) G_UN_APPLY(K, xb, db) G_CHK() } break;
            // generic (cold) operators through the noinline value+partials functions
#define G_GEN_APPLY(OP, SRC_IS_ACC)                                                                \
    {                                                                                              \
        const uint32_t gop_ = (OP);                                                                 \
        if (gop_ < DE_B_ADD) {                                                                      \
            const UG<T> r = unary_vg<T>(gop_, SRC_IS_ACC ? x : xb);                                 \
            x = r.y;                                                                               \
            if (SRC_IS_ACC) { DE_UNROLL for (int k = 0; k < GC; k++) d[k] = r.g * d[k]; }          \
            else { DE_UNROLL for (int k = 0; k < GC; k++) d[k] = r.g * db[k]; }                    \
        } else {                                                                                   \
            uint32_t fop = gop_;                                                                    \
            bool rev = false;                                                                      \
            switch (gop_) {                                                                         \
            case DOP_RSUB: fop = DE_B_SUB; rev = true; break;                                      \
            case DOP_RDIV: fop = DE_B_DIV; rev = true; break;                                      \
            case DOP_RPOW: fop = DE_B_POW; rev = true; break;                                      \
            case DOP_RMOD: fop = DE_B_MOD; rev = true; break;                                      \
            case DOP_RREM: fop = DE_B_REM; rev = true; break;                                      \
            case DOP_RGREATER: fop = DE_B_GREATER; rev = true; break;                              \
            case DOP_RPOW_ABS2: fop = DE_B_POW_ABS2; rev = true; break;                            \
            default: break;                                                                        \
            }                                                                                      \
            const BG<T> r = rev ? binary_vg<T>(fop, xb, x) : binary_vg<T>(fop, x, xb);             \
            x = r.v;                                                                               \
            if (rev) { DE_UNROLL for (int k = 0; k < GC; k++) d[k] = r.gx * db[k] + r.gy * d[k]; } \
            else { DE_UNROLL for (int k = 0; k < GC; k++) d[k] = r.gx * d[k] + r.gy * db[k]; }     \
        }                                                                                          \
    }
            switch (w.x) {
            case BOP_LOAD_ROW: { G_ROW_OPERAND(w.y) x = xb; DE_UNROLL for (int k = 0; k < GC; k++) d[k] = db[k]; } break;
            case BOP_LOAD_CONST: { G_CONST_OPERAND() x = xb; DE_UNROLL for (int k = 0; k < GC; k++) d[k] = db[k]; } break;
            case BOP_PUSH: {
                T *__restrict__ s_ = G_SLOT(w.y);
                s_[0] = x;
                DE_UNROLL for (int k = 0; k < GC; k++) s_[(1 + k) * RS] = d[k];
            } break;
            case BOP_CHECK_ROW: break; // every leaf operand is tested where it is read (G_LEAF_ROW)
            case BOP_CHECK_ACC: G_CHK() break;
            G_BIN4(0) G_BIN4(1) G_BIN4(2) G_BIN4(3) G_BIN4(4) G_BIN4(5)
            G_UN4(0) G_UN4(1) G_UN4(2)
            case BOP_GEN_ROW: { G_ROW_OPERAND(w.y & 0xFFFFFFu) G_GEN_APPLY(w.y >> 24, false) } break;
            case BOP_GEN_CONST: { G_CONST_OPERAND() G_GEN_APPLY(w.y >> 24, false) } break;
            case BOP_GEN_ACC: { const T xb = x; T db[GC]; DE_UNROLL for (int k = 0; k < GC; k++) db[k] = d[k]; G_GEN_APPLY(w.y >> 24, true) } break;
            case BOP_GEN_PARAM: {
                const uint32_t prow = w.y & 0xFFFFu, op_ = w.y >> 24;
                const T xb = a.params[prow + a.ld_params * cls];
                if (a.check) poison = M<T>::fma(xb, T(0), poison);
                T db[GC];
                { const int seed = (int)prow + param_seed0; DE_UNROLL for (int k = 0; k < GC; k++) db[k] = (k == seed) ? T(1) : T(0); }
                if (op_ == DOP_LOAD) { x = xb; DE_UNROLL for (int k = 0; k < GC; k++) d[k] = db[k]; }
                else G_GEN_APPLY(op_, false)
            } break;
            case BOP_TERN: { // acc = op3(row B, row C, acc)
                const T *__restrict__ sb = G_SLOT(w.y & 0xFFFFFFu);
                const T *__restrict__ sc = G_SLOT(w.z);
                const TG<T> r = ternary_vg<T>(w.y >> 24, sb[0], sc[0], x);
                x = r.v;
                DE_UNROLL for (int k = 0; k < GC; k++) d[k] = (r.g0 * sb[(1 + k) * RS] + r.g1 * sc[(1 + k) * RS]) + r.g2 * d[k];
            } break;
            default: break;
            }
        }
        // a non-finite d[k] always survives to the root (every update is linear in it), so the gradient
        // is validity-tested once, here; x was tested where the lowering kept a test (H_CHECK_OUT)
        if (a.check) {
            poison = M<T>::fma(x, T(0), poison);
            DE_UNROLL for (int k = 0; k < GC; k++) poison = M<T>::fma(g0 + k < G ? d[k] : T(0), T(0), poison); // real rows only
        }
        if (a.loss_mode) {
            const T e = x - yv;
            T lp; // w * l'(e)
            T l;  // w * l(e)
            if (a.loss_mode == 1 + DE_LOSS_L2) { l = wv * (e * e); lp = wv * (T(2) * e); }
            else if (a.loss_mode == 1 + DE_LOSS_L1) { l = wv * M<T>::abs(e); lp = wv * jl_sign(e); }
            else { l = wv * (x * yv); lp = wv * yv; } // DE_LOSS_PULLBACK: y holds the cotangent dY
            if (wv == T(0)) { l = T(0); lp = T(0); }  // weight 0 (and samples past N) really excludes the sample
            const int64_t n_cols = col_off[a.n_trees];
            T *__restrict__ pp = a.partial + ((int64_t)tm.tile * n_cols + col_off[tree]) * 4 + (tid >> 6);
            if (g0 == 0) {
                const T s = wave_sum_to_lane63(l);
                if ((tid & 63) == 63) pp[0] = s;
            }
            DE_UNROLL for (int k = 0; k < GC; k++) {
                if (g0 + k < G) { // wave-uniform
                    const T s = wave_sum_to_lane63(wv == T(0) ? T(0) : lp * d[k]);
                    if ((tid & 63) == 63) pp[(int64_t)(1 + g0 + k) * 4] = s;
                }
            }
        } else if (live) {
            if (a.diff_g0 >= 0) {
                if (a.out) a.out[(int64_t)tree * a.ld_out + base + tid] = x;
                a.grad[(int64_t)tree * a.ld_out + base + tid] = d[0];
            } else {
                if (a.out && g0 == 0) a.out[(int64_t)tree * a.ld_out + base + tid] = x;
                T *__restrict__ gp = a.grad + grad_off[tree] + (int64_t)G * (base + tid) + g0;
                DE_UNROLL for (int k = 0; k < GC; k++)
                    if (g0 + k < G) gp[k] = d[k];
            }
        }
        if (a.check && __ballot(poison != poison) != 0ull) gflag_incomplete(a.ok + tree, a.skip_flagged == 1);
    }
}

// Pass 3 of the fused loss+pullback reduction: thread = one reduction column (a tree's loss or one of its
// gradient rows; the owner is found by bisection in col_off); fixed summation order.  NaN where the evaluation
// was incomplete (src/ChainRules.jl:62-64 `dX_constants_dY .= NaN`).
template <typename T>
__global__ void __launch_bounds__(256) de_loss_grad_finish_kernel(const double *__restrict__ seg_sum, int64_t n_trees, int64_t n_cols,
                                                                 int32_t n_segs, const int64_t *__restrict__ col_off,
                                                                 const int32_t *__restrict__ n_grad, const uint8_t *__restrict__ ok,
                                                                 T *__restrict__ loss, T *__restrict__ dloss,
                                                                 const int64_t *__restrict__ dloss_off) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= n_cols) return;
    int64_t lo = 0, hi = n_trees; // col_off[lo] <= col < col_off[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (col_off[mid] <= col) lo = mid;
        else hi = mid;
    }
    const int64_t t = lo;
    const int c = (int)(col - col_off[t]);
    double s = 0.0;
    for (int32_t g = 0; g < n_segs; ++g)
        for (int w = 0; w < 4; ++w) s += seg_sum[((int64_t)g * n_cols + col) * 4 + w];
    const T v = ok[t] != 0 ? (T)s : M<T>::nan();
    if (c == 0) { if (loss) loss[t] = v; }
    else dloss[dloss_off[t] + c - 1] = v;
}

static int g_gcu = 0;

template <typename T, int GC>
static hipError_t launch_grad_t(const GradArgs &ga, int windows, hipStream_t stream) {
    const EvalArgs &e = ga.e;
    GArgs<T> a;
    a.code = nullptr;
    a.code_off = e.code_off;
    a.X = static_cast<const T *>(e.X);
    a.out = static_cast<T *>(e.out);
    a.grad = static_cast<T *>(ga.grad);
    a.grad_off = ga.grad_off;
    a.n_grad = ga.n_grad;
    a.ok = e.ok;
    a.params = static_cast<const T *>(e.params);
    a.classes = e.classes;
    a.N = e.N;
    a.ldX = e.ldX;
    a.ld_out = e.ld_out;
    a.ld_params = e.ld_params;
    a.n_tiles = (e.N + GBLK - 1) / GBLK;
    a.F = e.F;
    a.P = ga.P;
    a.n_trees = e.n_trees;
    a.n_slots = e.n_slots;
    a.mode = ga.mode;
    a.classes_is_i64 = e.classes_is_i64;
    a.class_base = e.class_base;
    a.n_classes = e.n_classes > 0 ? e.n_classes : 1;
    a.uses_params = e.uses_params ? 1 : 0;
    a.check = ga.diff_direction >= 0 ? 0 : 1;
    a.skip_flagged = e.skip_flagged ? 1 : 0;
    a.diff_g0 = ga.diff_direction >= 0 ? ga.P + ga.diff_direction : -1;
    a.code = ga.generic_code;
    a.tree_ids = nullptr;
    a.n_all_trees = e.n_trees;
    a.loss_mode = 0;
    a.y = a.w = nullptr;
    a.partial = nullptr;
    a.col_off = nullptr;
    if (ga.loss) {
        a.loss_mode = 1 + ga.loss->kind;
        a.y = static_cast<const T *>(ga.loss->y);
        a.w = static_cast<const T *>(ga.loss->w);
        a.partial = static_cast<T *>(ga.loss->partial);
        a.col_off = ga.col_off;
    }
    if (g_gcu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_gcu = prop.multiProcessorCount;
        if (g_gcu <= 0) g_gcu = 256;
    }
    int64_t n_chunks = (e.n_trees + 31) / 32;
    const int64_t want_blocks = (int64_t)g_gcu * 4 * 8;
    if (a.n_tiles * n_chunks * windows < want_blocks) n_chunks = (want_blocks + a.n_tiles * windows - 1) / (a.n_tiles * windows);
    const int64_t max_chunks = (e.n_trees + 3) / 4;
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    if (n_chunks < 1) n_chunks = 1;
    a.trees_per_chunk = (int32_t)((e.n_trees + n_chunks - 1) / n_chunks);
    if (a.skip_flagged) a.skip_flagged = a.trees_per_chunk >= 8 ? 1 : 2; // flag protocol (skip_flag_load, de_device_ops.h): these kernels write little, their L1 lines go stale under 2 (reverse kernel 17.0 / 16.0 ms); 2 only for tiny chunks (many tiles on one flag line)
    a.n_chunks = (int32_t)((e.n_trees + a.trees_per_chunk - 1) / a.trees_per_chunk);
    const int64_t blocks = ((a.n_tiles + 7) / 8) * 8 * a.n_chunks;
    if (blocks <= 0 || blocks > 0x7fffffffLL || windows > 65535) return hipErrorInvalidValue;
    const size_t lds = (size_t)(a.F + (size_t)a.n_slots * (1 + GC)) * (GBLK + 4) * sizeof(T);
    auto kern = de_grad_tape_kernel<T, GC>;
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)windows), dim3(GBLK), lds, stream, a);
    hipError_t st = hipGetLastError();
    if (st != hipSuccess || !ga.loss) return st;
    return launch_loss_grad_finish(sizeof(T) == 4 ? DE_F32 : DE_F64, ga, a.n_tiles, stream);
}

// passes 2 and 3 of the deterministic loss-gradient reduction (pass 2 is shared with de_eval_loss)
template <typename T> static hipError_t loss_grad_finish_t(const GradArgs &ga, int64_t n_tiles, hipStream_t stream) {
    int32_t n_segs = 1;
    hipError_t st = launch_loss_reduce_tiles(sizeof(T) == 4 ? DE_F32 : DE_F64, ga.loss->partial, ga.n_cols * 4, n_tiles, ga.loss->seg_sum,
                                             &n_segs, stream);
    if (st != hipSuccess) return st;
    hipLaunchKernelGGL(de_loss_grad_finish_kernel<T>, dim3((unsigned)((ga.n_cols + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const double *>(ga.loss->seg_sum), (int64_t)ga.e.n_trees, ga.n_cols, n_segs, ga.col_off, ga.n_grad,
                       ga.e.ok, static_cast<T *>(ga.loss->loss), static_cast<T *>(ga.dloss), ga.dloss_off);
    return hipGetLastError();
}
hipError_t launch_loss_grad_finish_range(int dtype, const GradArgs &ga, int64_t tile0, int64_t n_tiles, void *loss, void *dloss, hipStream_t stream) {
    LossArgs la = *ga.loss;
    GradArgs g = ga;
    la.partial = static_cast<char *>(la.partial) + (size_t)tile0 * (size_t)ga.n_cols * 4 * (dtype == DE_F32 ? 4 : 8);
    la.loss = loss;
    g.loss = &la;
    g.dloss = dloss;
    return dtype == DE_F32 ? loss_grad_finish_t<float>(g, n_tiles, stream) : loss_grad_finish_t<double>(g, n_tiles, stream);
}
hipError_t launch_loss_grad_finish(int dtype, const GradArgs &ga, int64_t n_tiles, hipStream_t stream) {
    return dtype == DE_F32 ? loss_grad_finish_t<float>(ga, n_tiles, stream) : loss_grad_finish_t<double>(ga, n_tiles, stream);
}

int grad_window(int max_grad) {
    const int m = max_grad < 1 ? 1 : max_grad;
    return m <= 6 ? m : 8; // smallest window that covers the widest gradient in one pass, else windows of 8
}

template <typename T> static hipError_t launch_grad_dt(const GradArgs &ga, hipStream_t stream) {
    const int maxg = ga.max_grad < 1 ? 1 : ga.max_grad;
    // smallest window that covers the widest gradient in one pass, else windows of 8
    if (maxg <= 1) return launch_grad_t<T, 1>(ga, 1, stream);
    if (maxg <= 2) return launch_grad_t<T, 2>(ga, 1, stream);
    if (maxg <= 3) return launch_grad_t<T, 3>(ga, 1, stream);
    if (maxg <= 4) return launch_grad_t<T, 4>(ga, 1, stream);
    if (maxg <= 5) return launch_grad_t<T, 5>(ga, 1, stream);
    if (maxg <= 6) return launch_grad_t<T, 6>(ga, 1, stream);
    return launch_grad_t<T, 8>(ga, (maxg + 7) / 8, stream);
}

// ---- de_eval_loss_grad_by_class: fold the per-class passes into the outputs ---------------------------
// One thread per tree; classes are added in index order, in double: reproducible.
template <typename T>
__global__ void __launch_bounds__(256) de_by_class_combine_kernel(const ByClassArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_trees) return;
    const T *loss_c = static_cast<const T *>(a.loss_c), *dloss_c = static_cast<const T *>(a.dloss_c);
    T *dloss = static_cast<T *>(a.dloss), *dparams = static_cast<T *>(a.dparams);
    const int C = a.n_classes, P = a.n_params, G = a.n_grad[t];
    const int64_t off = a.dloss_off[t];
    bool ok = true;
    for (int c = 0; c < C; c++) ok = ok && a.ok_c[(int64_t)c * a.ok_stride + t] != 0;
    a.ok[t] = ok ? 1 : 0;
    const T nan = T(__builtin_nan(""));
    if (a.loss) {
        double s = 0.0;
        for (int c = 0; c < C; c++) s += (double)loss_c[(int64_t)c * a.n_trees + t];
        static_cast<T *>(a.loss)[t] = ok ? (T)s : nan; // incomplete evaluations are NaN-filled (src/EvaluationHelpers.jl:29-33)
    }
    for (int k = 0; k < G; k++) {
        double s = 0.0;
        for (int c = 0; c < C; c++) {
            const T v = dloss_c[(int64_t)c * a.span + off + k];
            s += (double)v;
            if (k < P) dparams[((int64_t)t * C + c) * P + k] = ok ? v : nan;
        }
        dloss[off + k] = ok ? (T)s : nan;
    }
}
// EvalPullback's `dX = dX_dY .* reshape(dY, 1, :)` (src/ChainRules.jl:74) on the Jacobians de_eval_grad just wrote:
// tree t's [G, N] block (gradient index fastest) is scaled column j by dY[j]; an incomplete tree is NaN-filled
// (`dX_constants_dY .= NaN`, :62-64).  blockIdx.y = tree, 16-byte accesses where the block is aligned.
template <typename T>
__global__ void __launch_bounds__(256) de_pullback_scale_kernel(T *__restrict__ grad, const int64_t *__restrict__ grad_off,
                                                               const int32_t *__restrict__ n_grad, const uint8_t *__restrict__ ok,
                                                               const T *__restrict__ dY, int64_t N, int64_t n_trees) {
    for (int64_t t = blockIdx.y; t < n_trees; t += gridDim.y) { // (gridDim.y <= 65535: larger populations stride)
        const uint32_t G = (uint32_t)n_grad[t];
        if (G == 0) continue;
        T *__restrict__ g = grad + grad_off[t];
        const bool complete = ok[t] != 0;
        const int64_t total = (int64_t)G * N;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
            const int64_t j = e / G;
            g[e] = complete ? g[e] * dY[j] : M<T>::nan();
        }
    }
}
hipError_t launch_pullback_scale(int dtype, void *grad, const int64_t *grad_off, const int32_t *n_grad, const uint8_t *ok,
                                 const void *dY, int64_t N, int64_t n_trees, int32_t max_grad, hipStream_t stream) {
    if (n_trees <= 0 || N <= 0 || max_grad <= 0) return hipSuccess;
    int64_t bx = ((int64_t)max_grad * N + 256 * 8 - 1) / (256 * 8);
    if (bx > 4096) bx = 4096;
    const dim3 grid((unsigned)bx, (unsigned)(n_trees < 65535 ? n_trees : 65535)); // HIP limits gridDim.y: the kernel strides over trees
    if (dtype == DE_F32)
        hipLaunchKernelGGL(de_pullback_scale_kernel<float>, grid, dim3(256), 0, stream, static_cast<float *>(grad), grad_off, n_grad, ok,
                           static_cast<const float *>(dY), N, n_trees);
    else
        hipLaunchKernelGGL(de_pullback_scale_kernel<double>, grid, dim3(256), 0, stream, static_cast<double *>(grad), grad_off, n_grad, ok,
                           static_cast<const double *>(dY), N, n_trees);
    return hipGetLastError();
}

hipError_t launch_by_class_combine(int dtype, const ByClassArgs &a, hipStream_t stream) {
    const dim3 grid((unsigned)((a.n_trees + 255) / 256));
    if (dtype == DE_F32) hipLaunchKernelGGL(de_by_class_combine_kernel<float>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(de_by_class_combine_kernel<double>, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---- threaded variant: one module per (type, window, samples per lane) — de_grad_threaded.hip ----------
#define DE_GT_DECL(TAG, GC, V)                                                       \
    hipError_t grad_thr_fetch_##TAG##GC##v##V(uint64_t *host_table);                  \
    hipError_t grad_thr_launch_##TAG##GC##v##V(const GradArgs &ga, int bucket, hipStream_t stream);
// must agree with build.sh
#define DE_GT_ALL(X)                                                                                     \
    X(f, 1, 1) X(f, 2, 1) X(f, 3, 1) X(f, 4, 1) X(f, 5, 1) X(f, 6, 1) X(f, 8, 1)                          \
    X(f, 1, 2) X(f, 2, 2) X(f, 3, 2) X(f, 4, 2) X(f, 5, 2) X(f, 6, 2)                                     \
    X(d, 1, 1) X(d, 2, 1) X(d, 3, 1) X(d, 4, 1) X(d, 5, 1)
DE_GT_ALL(DE_GT_DECL)

bool grad_threaded_has(int dtype, int GC, int VS) {
    if (GC < 1 || GC > 8 || GC == 7) return false;
    if (dtype == DE_F32) return VS == 1 || (VS == 2 && GC <= 6);
    return VS == 1 && GC <= 5; // Float64 states wider than 16 dwords would be passed through scratch memory
}

hipError_t grad_handler_table(int dtype, int GC, int VS, uint64_t *table) {
    struct Cache { uint64_t t[2][9][3][GOP_MAX]; bool have[2][9][3] = {}; };
    static std::unique_ptr<Cache> caches[DE_MAX_DEVICES]; // per device (de_kernels.hip handler_device_slot)
    static std::mutex mu; // contexts on several host threads may ask at once
    int dev = 0;
    { const hipError_t dst = handler_device_slot(&dev); if (dst != hipSuccess) return dst; }
    const std::lock_guard<std::mutex> lock(mu);
    if (!caches[dev]) caches[dev].reset(new Cache());
    auto &cache = caches[dev]->t;
    auto &have = caches[dev]->have;
    const int k = dtype == DE_F32 ? 0 : 1;
    if (!grad_threaded_has(dtype, GC, VS)) return hipErrorInvalidValue;
    if (!have[k][GC][VS]) {
        hipError_t st = hipErrorInvalidValue;
#define DE_GT_FETCH(TAG, G, V) if (k == (#TAG[0] == 'f' ? 0 : 1) && GC == G && VS == V) st = grad_thr_fetch_##TAG##G##v##V(cache[k][GC][VS]);
        DE_GT_ALL(DE_GT_FETCH)
        if (st != hipSuccess) return st;
        have[k][GC][VS] = true;
    }
    for (int i = 0; i < (int)gop_count(GC); i++) table[i] = cache[k][GC][VS][i];
    return hipSuccess;
}

// Side streams of a caller's stream: the buckets of one gradient call (one launch — with its probe launch in front — per window width and
// samples-per-lane class: 2 ... 9 per call) are independent of each other, and each is short enough (0.2 ... 1.7 ms at 10^6 samples)
// that its ramp and tail are a tenth of it; spread over DE_GRAD_STREAMS streams (default 3; 1 = all on the caller's stream) the tail of
// one bucket runs beside the next.  Fork / join by events (legal under stream capture); one set per caller stream, made on first use.
struct SideStreams {
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
    int n = 0;
};
static SideStreams *side_streams_of(hipStream_t main, int want) {
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, hipStream_t>, std::unique_ptr<SideStreams>>> all;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const std::lock_guard<std::mutex> lock(mu);
    for (auto &e : all) if (e.first.first == dev && e.first.second == main) return e.second->n >= want ? e.second.get() : nullptr;
    if (all.size() >= 32) return nullptr; // a caller that makes a new stream per call: no side streams for the later ones (nothing is ever freed here)
    std::unique_ptr<SideStreams> ss(new SideStreams());
    if (hipEventCreateWithFlags(&ss->fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int i = 0; i < 3; i++) {
        if (hipStreamCreateWithFlags(&ss->s[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ss->join[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        ss->n = i + 1;
    }
    all.emplace_back(std::make_pair(dev, main), std::move(ss));
    SideStreams *r = all.back().second.get();
    return r->n >= want ? r : nullptr;
}

// n_launches independent launch sequences of one call over the caller's stream and its side streams: fork(), next() per sequence, join()
struct ForkJoin {
    hipStream_t main;
    SideStreams *ss = nullptr;
    int n_side = 0, slot = 0;
    ForkJoin(hipStream_t m, int n_launches) : main(m) {
        static const int n_env = [] { const char *v = getenv("DE_GRAD_STREAMS"); const int n = v ? atoi(v) : 3; return n < 1 ? 1 : (n > 4 ? 4 : n); }();
        n_side = n_launches >= 2 ? std::min(n_env, n_launches) - 1 : 0;
        ss = n_side > 0 ? side_streams_of(m, n_side) : nullptr;
        if (!ss) n_side = 0;
    }
    hipError_t fork() {
        if (!ss) return hipSuccess;
        hipError_t st = hipEventRecord(ss->fork, main);
        for (int i = 0; i < n_side && st == hipSuccess; i++) st = hipStreamWaitEvent(ss->s[i], ss->fork, 0);
        return st;
    }
    hipStream_t next() {
        const int k = slot++ % (n_side + 1);
        return k == 0 ? main : ss->s[k - 1];
    }
    hipError_t join() { // (also after an error: the side streams must not run on behind the caller's back)
        hipError_t err = hipSuccess;
        for (int i = 0; i < n_side; i++) {
            hipError_t st = hipEventRecord(ss->join[i], ss->s[i]);
            if (st == hipSuccess) st = hipStreamWaitEvent(main, ss->join[i], 0);
            if (st != hipSuccess && err == hipSuccess) err = st;
        }
        return err;
    }
};

// de_eval_loss_grad_by_class: the pair of finish passes of every class over its own tiles (class k: tiles [tile0[k], tile0[k + 1])) — 2 C
// launches of ~25 us each, independent of each other except for the segment sums they stage: spread over the caller's stream and its side
// streams, each with a region of its own in `seg_sum` (seg_stride bytes apart; classes of one stream run in order).  Same launches, same
// reduction order per class: the same bits as one call per class.
hipError_t launch_loss_grad_finish_ranges(int dtype, const GradArgs &ga, int64_t n_classes, const int64_t *tile0, void *loss, size_t loss_stride,
                                          void *dloss, size_t dloss_stride, size_t seg_stride, int seg_regions, hipStream_t stream) {
    ForkJoin fj(stream, seg_regions > 1 ? (int)std::min<int64_t>(n_classes, seg_regions) : 1);
    hipError_t first_err = fj.fork();
    if (first_err != hipSuccess) return first_err;
    for (int64_t k = 0; k < n_classes; k++) {
        const int slot = fj.slot % (fj.n_side + 1);
        const hipStream_t ks = fj.next();
        LossArgs la = *ga.loss;
        GradArgs g = ga;
        la.partial = static_cast<char *>(la.partial) + (size_t)tile0[k] * (size_t)ga.n_cols * 4 * (dtype == DE_F32 ? 4 : 8);
        la.seg_sum = static_cast<char *>(la.seg_sum) + (size_t)slot * seg_stride;
        la.loss = static_cast<char *>(loss) + (size_t)k * loss_stride;
        g.loss = &la;
        g.dloss = static_cast<char *>(dloss) + (size_t)k * dloss_stride;
        const hipError_t st = dtype == DE_F32 ? loss_grad_finish_t<float>(g, tile0[k + 1] - tile0[k], ks) : loss_grad_finish_t<double>(g, tile0[k + 1] - tile0[k], ks);
        if (st != hipSuccess) { first_err = st; break; }
    }
    { const hipError_t js = fj.join(); if (first_err == hipSuccess) first_err = js; }
    return first_err;
}

static hipError_t grad_prio_prepass(int dtype, const GradArgs &a, hipStream_t stream, GradArgs *with);
hipError_t launch_grad_threaded(int dtype, const GradArgs &a0, hipStream_t stream, const char **kernel_name) {
    if (kernel_name) *kernel_name = "de_grad_threaded_kernel";
    GradArgs a;
    { const hipError_t ps = grad_prio_prepass(dtype, a0, stream, &a); if (ps != hipSuccess) return ps; }
    const int k = dtype == DE_F32 ? 0 : 1;
    if (a.loss) { // two-sample modules use 512-sample tiles: the 256-sample tile slots they never write must read as 0
        bool wide = false;
        for (int b = 0; b < a.n_buckets; b++) wide = wide || (a.buckets[b].n > 0 && a.buckets[b].VS > 1);
        if (wide) {
            const size_t bytes = (size_t)((a.e.N + GBLK - 1) / GBLK) * (size_t)a.n_cols * 4 * (dtype == DE_F32 ? 4 : 8);
            hipError_t st = hipMemsetAsync(a.loss->partial, 0, bytes, stream);
            if (st != hipSuccess) return st;
        }
    }
    int n_active = 0;
    for (int b = 0; b < a.n_buckets; b++) n_active += a.buckets[b].n > 0;
    ForkJoin fj(stream, n_active);
    hipError_t first_err = fj.fork();
    if (first_err != hipSuccess) return first_err;
    for (int b = 0; b < a.n_buckets; b++) {
        const GradArgs::Bucket &bk = a.buckets[b];
        if (bk.n <= 0) continue;
        const hipStream_t bs = fj.next();
        hipError_t st = hipErrorInvalidValue;
#define DE_GT_LAUNCH(TAG, G, V) if (k == (#TAG[0] == 'f' ? 0 : 1) && bk.GC == G && bk.VS == V) st = grad_thr_launch_##TAG##G##v##V(a, b, bs);
        DE_GT_ALL(DE_GT_LAUNCH)
        if (st != hipSuccess) { first_err = st; break; }
    }
    { const hipError_t js = fj.join(); if (first_err == hipSuccess) first_err = js; }
    if (first_err != hipSuccess) return first_err;
    if (!a.loss) return hipSuccess;
    return launch_loss_grad_finish(dtype, a, (a.e.N + GBLK - 1) / GBLK, stream);
}

// ---- reverse accumulation: one module per element type (de_rev_threaded.hip) --------------------------
hipError_t rev_thr_fetch_f(uint64_t *host_table);
hipError_t rev_thr_fetch_d(uint64_t *host_table);
hipError_t rev_thr_launch_f(const GradArgs &ga, int group, hipStream_t stream);
hipError_t rev_thr_launch_d(const GradArgs &ga, int group, hipStream_t stream);
hipError_t rev_handler_table(int dtype, uint64_t *table) {
    struct Cache { uint64_t t[2][ROP_COUNT]; bool have[2] = {false, false}; };
    static std::unique_ptr<Cache> caches[DE_MAX_DEVICES]; // per device
    static std::mutex mu; // contexts on several host threads may ask at once
    int dev = 0;
    { const hipError_t dst = handler_device_slot(&dev); if (dst != hipSuccess) return dst; }
    const std::lock_guard<std::mutex> lock(mu);
    if (!caches[dev]) caches[dev].reset(new Cache());
    auto &cache = caches[dev]->t;
    auto &have = caches[dev]->have;
    const int k = dtype == DE_F32 ? 0 : 1;
    if (!have[k]) {
        const hipError_t st = k == 0 ? rev_thr_fetch_f(cache[k]) : rev_thr_fetch_d(cache[k]);
        if (st != hipSuccess) return st;
        have[k] = true;
    }
    for (int i = 0; i < (int)ROP_COUNT; i++) table[i] = cache[k][i];
    return hipSuccess;
}
// the priority tiles of a gradient / reverse launch: one pre-pass over X for all its buckets (de_kernels.hip de_tile_extremes_kernel)
static hipError_t grad_prio_prepass(int dtype, const GradArgs &a, hipStream_t stream, GradArgs *with) {
    *with = a;
    with->prio_ready = false;
    if (!a.e.skip_flagged || !a.e.prio_keys || !a.e.X || !prio_tiles_wanted(a.e.N, a.e.F, a.e.n_trees)) return hipSuccess;
    const hipError_t st = a.e.prio_keys_ready ? hipSuccess : launch_tile_extremes(dtype, a.e.X, a.e.N, a.e.ldX, a.e.F, a.e.prio_keys, stream);
    if (st == hipSuccess) with->prio_ready = true;
    return st;
}
hipError_t launch_rev_threaded(int dtype, const GradArgs &a0, hipStream_t stream, const char **kernel_name) {
    if (kernel_name) *kernel_name = "de_rev_threaded_kernel";
    GradArgs a;
    { const hipError_t ps = a0.rev_tile_range ? (a = a0, a.prio_ready = false, hipSuccess) : grad_prio_prepass(dtype, a0, stream, &a); if (ps != hipSuccess) return ps; }
    const int64_t n_tiles = (a.e.N + GBLK - 1) / GBLK;
    ForkJoin fj(stream, a.rev_n_groups);
    hipError_t first_err = fj.fork();
    if (first_err != hipSuccess) return first_err;
    for (int k = 0; k < a.rev_n_groups; k++) { // one launch per LDS-need group of trees, spread over the side streams
        const hipError_t st = dtype == DE_F32 ? rev_thr_launch_f(a, k, fj.next()) : rev_thr_launch_d(a, k, fj.next());
        if (st != hipSuccess) { first_err = st; break; }
    }
    { const hipError_t js = fj.join(); if (first_err == hipSuccess) first_err = js; }
    if (first_err != hipSuccess) return first_err;
    if (a.rev_tile_range) return hipSuccess; // by-class: the caller reduces every class's tile range itself
    return launch_loss_grad_finish(dtype, a, n_tiles, stream);
}

hipError_t launch_grad(int dtype, const GradArgs &a, hipStream_t stream, const char **kernel_name) {
    if (a.threaded_code && a.diff_direction < 0) return launch_grad_threaded(dtype, a, stream, kernel_name);
    if (kernel_name) *kernel_name = "de_grad_tape_kernel";
    if (a.diff_direction >= 0) return dtype == DE_F32 ? launch_grad_t<float, 1>(a, 1, stream) : launch_grad_t<double, 1>(a, 1, stream);
    if (dtype == DE_F32) return launch_grad_dt<float>(a, stream);
    return launch_grad_dt<double>(a, stream);
}

} // namespace de
