// de_grad_common.h — device code shared by the two gradient kernels (de_grad_kernels.hip: flat switch,
// eval_diff and wide gradients; de_grad_threaded.hip: threaded code): kernel arguments, value+partial
// functions of every operator (ChainRules scalar rules, same table as oracle/de_oracle_ops.h), tile map.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "de_device_ops.h"
#include "de_kernels.h"

namespace de {

#define DE_CONSTANT __attribute__((address_space(4)))
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
typedef const DE_CONSTANT U32x4 *ConstU4Ptr;
typedef const DE_CONSTANT int32_t *ConstI32Ptr;
typedef const DE_CONSTANT int64_t *ConstI64Ptr;
#define DE_UNROLL _Pragma("unroll")

template <typename T> struct GArgs {
    const BoundInstr *code;  // BOUND form of the UNFOLDED program (de_bind.h, ee binding), +1 pad
    const int32_t *code_off; // n_trees + 1
    const T *X;
    T *out;                  // may be null
    T *grad;
    const int64_t *grad_off; // n_trees element offsets
    const int32_t *n_grad;   // n_trees
    uint8_t *ok;
    const T *params;
    const void *classes;
    int64_t N, ldX, ld_out, ld_params, n_tiles, n_classes;
    int32_t F, P, n_trees, trees_per_chunk, n_chunks, n_slots, mode;
    int32_t FX; // threaded kernels: rows of X; F - FX further leaf rows hold the parameters gathered by class (0 elsewhere)
    int32_t classes_is_i64, class_base, uses_params, check;
    int32_t skip_flagged; // early exit at tree granularity: the trees of a chunk whose flag is already 0 are not evaluated (de_kernels.hip)
    int32_t diff_g0; // >= 0: eval_diff mode — single component diff_g0, dense [n_trees, ld_out] output
    // fused loss + pullback (de_eval_loss_grad): instead of storing x and d[k] the kernel reduces
    //   sum_j w_j l(x_j - y_j)   and   sum_j w_j l'(x_j - y_j) d_k[j]   per wavefront
    int32_t loss_mode;       // 0 = off, 1 + de_loss_kind otherwise
    const T *y;
    const T *w;              // may be null
    T *partial;              // [n_tiles][n_cols][4 waves]; tree t owns columns col_off[t] .. col_off[t] + n_grad[t] (loss first)
    const int64_t *col_off;  // n_trees + 1
    const int32_t *tree_ids; // threaded kernel: the n_trees trees this launch evaluates (null = 0..n_trees-1)
    int32_t n_all_trees;     // trees of the program (col_off has n_all_trees + 1 entries)
    // threaded kernels: a parameter table of <= GPTAB_MAX elements is copied to LDS behind the rows
    int32_t ptab_elems;      // ld_params * n_classes, or 0: read the table from global memory
    uint32_t ptab_offset;    // LDS byte offset of the copy
    const int64_t *tile_range; // de_rev_threaded.hip, by-class reduction: (first sample, last sample) of every tile, or null
    const int32_t *rev_mid;  // de_rev_threaded.hip: first backward instruction of every tree
    int32_t rev_rows;        // ... and LDS rows per wave (X + slots + partial rows + staging)
    int32_t rev_stage_cols, rev_stage_rows; // per-wave staging of the column sums (elements / rows)
    // priority tiles (de_kernels.hip de_tile_extremes_kernel): the first n_prio_blocks workgroups (of every window) run the tiles
    // (prio[k] & 0xFFFFFFFF) >> prio_shift first; null / 0: none
    const unsigned long long *prio;
    uint32_t n_prio, n_prio_blocks, prio_shift;
    // de_grad_threaded.hip, SHARED LEAF ROWS (de_kernels.h GradArgs::gt_share): a workgroup = four waves on the same 64 x VS samples, wave w
    // runs every fourth tree of the chunk through stream variant w (var_stride records apart); tiles are 64 x VS samples
    int32_t share;
    int64_t var_stride;
};

constexpr int GPTAB_MAX = 2048;
// Early exit at tree granularity (src/Evaluate.jl:26-32, src/EvaluateDerivative.jl:230-243: the reference returns at the first
// non-finite array): bit i = the i-th tree of this workgroup's chunk was already flagged incomplete when the workgroup started —
// by a workgroup that ran earlier or by the host (non-finite constant).  Its values / Jacobian rows are unspecified then (SURVEY
// §8a), fused reductions of it are NaN (the finish kernels look at the flag), so the chunk loop steps over it.  Chunks have <= 64
// trees (else: no skipping).  Agent-scope load: past this CU's vector cache.
template <typename IDS> __device__ __forceinline__ uint64_t gskip_mask(uint8_t *ok, IDS tree_ids, int t0, int t1, int enabled, int64_t tile) {
    if (!enabled || t1 - t0 > 64) return 0ull;
    const int i = t0 + (int)(threadIdx.x & 63);
    uint8_t f = 1;
    if (i < t1) f = skip_flag_load(ok + (tree_ids ? tree_ids[i] : i), enabled, tile);
    return __ballot(f == 0);
}
template <typename T> __device__ __forceinline__ T gimm(uint32_t w2, uint32_t w3);
template <> __device__ __forceinline__ float gimm<float>(uint32_t w2, uint32_t) { return __uint_as_float(w2); }
template <> __device__ __forceinline__ double gimm<double>(uint32_t w2, uint32_t w3) {
    return __longlong_as_double((long long)(((unsigned long long)w3 << 32) | w2));
}

// value + partial of a unary operator; Zygote `nothing` -> 0 (ext/DynamicExpressionsZygoteExt.jl:12-15)
template <typename T> struct UG { T y, g; };
template <typename T> __device__ __noinline__ UG<T> unary_vg(uint32_t op, T x) {
    using m = M<T>;
    UG<T> r;
    const T LN2 = T(0.693147180559945309417232121458176568), LN10 = T(2.302585092994045684017991454684364208);
    switch (op) {
    case DE_U_NEG: r.y = -x; r.g = T(-1); break;
    case DE_U_ABS: r.y = m::abs(x); r.g = jl_sign(x); break;
    case DE_U_SQUARE: r.y = x * x; r.g = x + x; break;
    case DE_U_CUBE: r.y = (x * x) * x; r.g = (T(3) * x) * x; break;
    case DE_U_RELU: r.y = x < T(0) ? T(0) : x; r.g = x < T(0) ? T(0) : T(1); break;
    case DE_U_SIGN: r.y = jl_sign(x); r.g = T(0); break;
    case DE_U_ROUND: r.y = m::rint(x); r.g = T(0); break;
    case DE_U_FLOOR: r.y = m::floor(x); r.g = T(0); break;
    case DE_U_CEIL: r.y = m::ceil(x); r.g = T(0); break;
    case DE_U_INV: r.y = T(1) / x; r.g = -(r.y * r.y); break;
    case DE_U_SQRT: r.y = m::sqrt(x); r.g = T(1) / (T(2) * r.y); break;
    case DE_U_CBRT: r.y = m::cbrt(x); r.g = T(1) / (T(3) * (r.y * r.y)); break;
    case DE_U_EXP: r.y = m::exp(x); r.g = r.y; break;
    case DE_U_EXP2: r.y = m::exp2(x); r.g = r.y * LN2; break;
    case DE_U_LOG: r.y = m::log(x); r.g = T(1) / x; break;
    case DE_U_LOG2: r.y = m::log2(x); r.g = (T(1) / x) / LN2; break;
    case DE_U_LOG10: r.y = m::log10(x); r.g = (T(1) / x) / LN10; break;
    case DE_U_LOG1P: r.y = m::log1p(x); r.g = T(1) / (x + T(1)); break;
    case DE_U_SIN: r.y = m::sin(x); r.g = m::cos(x); break;
    case DE_U_COS: r.y = m::cos(x); r.g = -m::sin(x); break;
    case DE_U_TAN: r.y = m::tan(x); r.g = T(1) + r.y * r.y; break;
    case DE_U_SINH: r.y = m::sinh(x); r.g = m::cosh(x); break;
    case DE_U_COSH: r.y = m::cosh(x); r.g = m::sinh(x); break;
    case DE_U_TANH: r.y = m::tanh(x); r.g = T(1) - r.y * r.y; break;
    case DE_U_ASIN: r.y = m::asin(x); r.g = T(1) / m::sqrt(T(1) - x * x); break;
    case DE_U_ACOS: r.y = m::acos(x); r.g = -(T(1) / m::sqrt(T(1) - x * x)); break;
    case DE_U_ATAN: r.y = m::atan(x); r.g = T(1) / (T(1) + x * x); break;
    case DE_U_ASINH: r.y = m::asinh(x); r.g = T(1) / m::sqrt(x * x + T(1)); break;
    case DE_U_ACOSH: r.y = m::acosh(x); r.g = T(1) / (m::sqrt(x - T(1)) * m::sqrt(x + T(1))); break;
    case DE_U_ATANH: r.y = m::atanh(x); r.g = T(1) / (T(1) - x * x); break;
    case DE_U_SAFE_LOG: r.y = x <= T(0) ? m::nan() : m::log(x); r.g = x <= T(0) ? T(0) : T(1) / x; break;
    case DE_U_SAFE_LOG2: r.y = x <= T(0) ? m::nan() : m::log2(x); r.g = x <= T(0) ? T(0) : (T(1) / x) / LN2; break;
    case DE_U_SAFE_LOG10: r.y = x <= T(0) ? m::nan() : m::log10(x); r.g = x <= T(0) ? T(0) : (T(1) / x) / LN10; break;
    case DE_U_SAFE_LOG1P: r.y = x <= T(-1) ? m::nan() : m::log1p(x); r.g = x <= T(-1) ? T(0) : T(1) / (x + T(1)); break;
    case DE_U_SAFE_SQRT:
        if (x < T(0)) { r.y = m::nan(); r.g = T(0); } else { r.y = m::sqrt(x); r.g = T(1) / (T(2) * r.y); }
        break;
    case DE_U_SAFE_ACOSH:
        r.y = x < T(1) ? m::nan() : m::acosh(x);
        r.g = x < T(1) ? T(0) : T(1) / (m::sqrt(x - T(1)) * m::sqrt(x + T(1)));
        break;
    case DE_U_COS2: { const T c = m::cos(x), s = m::sin(x); r.y = c * c; r.g = (T(2) * c) * (-s); } break;
    case DE_U_GAMMA: r.y = m::tgamma(x); r.g = r.y * dev_digamma(x); break;
    default: r.y = m::nan(); r.g = m::nan(); break;
    }
    return r;
}

// Value and derivative of the cheap unary operators, inline: the same expressions as unary_vg (de_grad_common.h), so a tree
// gives the same bits whichever handler serves it.  K: 3 neg 4 square 5 cube 6 abs 7 log 8 safe_log 9 sqrt 10 safe_sqrt 11 tanh 12 relu
template <typename T, int K> __device__ __forceinline__ UG<T> gun_inline(T x) {
    using m = M<T>;
    UG<T> r;
    if constexpr (K == 3) { r.y = -x; r.g = T(-1); }
    else if constexpr (K == 4) { r.y = x * x; r.g = x + x; }
    else if constexpr (K == 5) { r.y = (x * x) * x; r.g = (T(3) * x) * x; }
    else if constexpr (K == 6) { r.y = m::abs(x); r.g = jl_sign(x); }
    else if constexpr (K == 7) { r.y = m::log(x); r.g = T(1) / x; }
    else if constexpr (K == 8) { r.y = x <= T(0) ? m::nan() : m::log(x); r.g = x <= T(0) ? T(0) : T(1) / x; }
    else if constexpr (K == 9) { r.y = m::sqrt(x); r.g = T(1) / (T(2) * r.y); }
    else if constexpr (K == 10) {
        if (x < T(0)) { r.y = m::nan(); r.g = T(0); } else { r.y = m::sqrt(x); r.g = T(1) / (T(2) * r.y); }
    } else if constexpr (K == 11) { r.y = m::tanh(x); r.g = T(1) - r.y * r.y; }
    else { r.y = x < T(0) ? T(0) : x; r.g = x < T(0) ? T(0) : T(1); }
    return r;
}
// value + both partials of op(x, y) (x = first/left argument)
template <typename T> struct BG { T v, gx, gy; };
template <typename T> __device__ __noinline__ BG<T> binary_vg(uint32_t op, T x, T y) {
    using m = M<T>;
    BG<T> r;
    switch (op) {
    case DE_B_ADD: r.v = x + y; r.gx = T(1); r.gy = T(1); break;
    case DE_B_SUB: r.v = x - y; r.gx = T(1); r.gy = T(-1); break;
    case DE_B_MUL: r.v = x * y; r.gx = y; r.gy = x; break;
    case DE_B_DIV: r.v = x / y; r.gx = T(1) / y; r.gy = -(r.v / y); break;
    case DE_B_POW: { // ChainRules _pow_grad_x / _pow_grad_p (real case)
        r.v = m::pow(x, y);
        if (x != T(0) || y < T(0)) r.gx = (!m::isfinite(x) && x == x && y == T(1)) ? T(1) : (r.v * y) / x;
        else if (y == T(1)) r.gx = T(1);
        else if (y == T(0) || y > T(1)) r.gx = T(0);
        else r.gx = m::inf();
        if (x != T(0)) r.gy = r.v * m::log(m::abs(x));
        else if (y > T(0)) r.gy = T(0);
        else r.gy = m::nan();
    } break;
    case DE_B_MAX: { const bool gt = x > y; r.v = jl_max(x, y); r.gx = gt ? T(1) : T(0); r.gy = gt ? T(0) : T(1); } break;
    case DE_B_MIN: { const bool gt = x > y; r.v = jl_min(x, y); r.gx = gt ? T(0) : T(1); r.gy = gt ? T(1) : T(0); } break;
    case DE_B_MOD: {
        r.v = jl_mod(x, y);
        const T u = x / y;
        const bool isint = (u == m::floor(u)) && m::isfinite(u);
        r.gx = isint ? m::nan() : T(1);
        r.gy = isint ? m::nan() : -m::floor(u);
    } break;
    case DE_B_REM: {
        r.v = m::fmod(x, y);
        const T u = x / y;
        const bool isint = (u == m::floor(u)) && m::isfinite(u);
        r.gx = isint ? m::nan() : T(1);
        r.gy = isint ? m::nan() : -m::trunc(u);
    } break;
    case DE_B_GREATER: r.v = x > y ? T(1) : T(0); r.gx = T(0); r.gy = T(0); break;
    case DE_B_POW_ABS2: {
        const T a = m::abs(x), l = m::log(a), mm = y * l;
        r.v = m::exp(mm);
        r.gx = ((r.v * y) * (T(1) / a)) * jl_sign(x);
        r.gy = r.v * l;
    } break;
    default: r.v = r.gx = r.gy = m::nan(); break;
    }
    return r;
}

template <typename T> struct TG { T v, g0, g1, g2; };
template <typename T> __device__ __noinline__ TG<T> ternary_vg(uint32_t op, T x, T y, T z) {
    using m = M<T>;
    TG<T> r;
    switch (op) {
    case DE_T_FMA: r.v = m::fma(x, y, z); r.g0 = y; r.g1 = x; r.g2 = T(1); break;
    case DE_T_CLAMP:
        r.v = x > z ? z : (x < y ? y : x);
        r.g0 = (x > z || x < y) ? T(0) : T(1);
        r.g1 = (x > z) ? T(0) : (x < y ? T(1) : T(0));
        r.g2 = (x > z) ? T(1) : T(0);
        break;
    case DE_T_ADD3: r.v = (x + y) + z; r.g0 = r.g1 = r.g2 = T(1); break;
    default: {
        const T mx = jl_max(x, y);
        const bool gt1 = x > y, gt2 = mx > z;
        r.v = jl_max(mx, z);
        r.g0 = (gt2 && gt1) ? T(1) : T(0);
        r.g1 = (gt2 && !gt1) ? T(1) : T(0);
        r.g2 = gt2 ? T(0) : T(1);
    } break;
    }
    return r;
}

struct GTileMap { int64_t tile; int32_t chunk; bool valid; bool prio = false; };
__device__ __forceinline__ GTileMap gmap_block(uint32_t bid, int32_t n_chunks, int64_t n_tiles) {
    GTileMap m;
    if (n_tiles < 64) {
        // few sample tiles (the many-trees x few-rows shape): X fits in every L2 anyway, and the XCD-aware
        // order below would put all work of tile t on XCD t mod 8 (one eighth of the chip for a single tile)
        m.tile = (int64_t)(bid % (uint32_t)n_tiles);
        m.chunk = (int32_t)(bid / (uint32_t)n_tiles);
        m.valid = m.chunk < n_chunks;
        return m;
    }
    const uint32_t xcd = bid & 7u, idx = bid >> 3;
    m.chunk = (int32_t)(idx % (uint32_t)n_chunks);
    m.tile = (int64_t)(idx / (uint32_t)n_chunks) * 8 + xcd;
    m.valid = m.tile < n_tiles;
    return m;
}

// ... with the launch's priority tiles in front (their flags travel at agent scope whatever the launch's protocol is)
template <typename T> __device__ __forceinline__ GTileMap gmap_block_prio(const GArgs<T> &a, uint32_t bid) {
    if (bid >= a.n_prio_blocks) return gmap_block(bid - a.n_prio_blocks, a.n_chunks, a.n_tiles);
    GTileMap m;
    const uint32_t k = bid / (uint32_t)a.n_chunks;
    m.chunk = (int32_t)(bid % (uint32_t)a.n_chunks);
    m.tile = k < a.n_prio ? (int64_t)((uint32_t)a.prio[k] >> a.prio_shift) : 0;
    m.valid = k < a.n_prio && m.tile < a.n_tiles;
    m.prio = true;
    return m;
}
// (host) fills the priority fields of a launch with n_chunks chunks and tile_samples per tile; returns the blocks to add to the grid
template <typename T> inline int64_t gprio_setup(GArgs<T> &a, const void *keys, int x_features, int tile_samples) {
    a.prio = static_cast<const unsigned long long *>(keys);
    a.n_prio = (uint32_t)(3 * x_features);
    a.prio_shift = 0;
    while ((64 << a.prio_shift) < tile_samples) ++a.prio_shift;
    a.n_prio_blocks = (uint32_t)(((int64_t)a.n_prio * a.n_chunks + 7) / 8 * 8);
    return (int64_t)a.n_prio_blocks;
}

__device__ __noinline__ void gflag_incomplete(uint8_t *ok, int agent) { // agent scope (written through): workgroups that start later skip the tree
    if ((threadIdx.x & 63) == 0) {
        if (agent) __hip_atomic_store(ok, (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *ok = 0;
    }
}

constexpr int GBLK = 256;

} // namespace de
