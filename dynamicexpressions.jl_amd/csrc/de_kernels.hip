// de_kernels.hip — gfx950 (CDNA4 / MI355X) kernels for batched expression-tree
// evaluation.  Replaces the inner loops of the reference's src/Evaluate.jl
// (deg*_eval and the fused deg1_l*/deg2_* kernels, :366-993) and
// src/EvaluateDerivative.jl (grad_degn_eval :340-365, diff_degn_eval :99-119).
//
// Design:
//  * grid  = sample tiles x tree chunks.  The threaded kernel (the default): a workgroup is ONE wave64 (DE_TBLK = 64) that owns
//    TILE = 256 consecutive Float32 samples (4 per lane; 128 Float64) and runs a chunk of <= 63 trees as ONE chain of
//    direct-threaded handlers — or, when LDS rows leave such a workgroup short of resident waves (staged parameter rows, many features),
//    a WAVE GROUP of 2 / 4 / 8 waves on one tile that share the staged rows and run a chunk each (KArgs::var_stride, round 6);
//    the flat-switch fall-back kernel uses 256 threads = 4 wave64 and G vectors per thread.
//    Large early-exit launches are three launches: the priority tiles as a probe, the compaction of the live trees
//    (de_compact_live_kernel), the launch proper over the re-linked stream.
//  * The X tile ([F, TILE], feature-fastest in HBM) is read ONCE per workgroup with
//    fully coalesced loads and transposed into LDS as xs[f][sample], so a leaf
//    read is one conflict-free ds_read_b128 per thread.
//  * Each tree is a wave-uniform accumulator program (de_program.h): instruction
//    records are fetched through the SCALAR cache (constant address space ->
//    s_load_dwordx4, the next record requested while the current handler runs),
//    dispatch runs on the scalar unit, the VALU only sees operator
//    arithmetic on 4 (2) independent samples per lane.  Intermediates live in
//    registers; only the rare both-children-are-subtrees case spills one value
//    per sample to LDS.
//  * NaN/Inf flag: per-lane poison, one wavefront ballot per tree, one byte
//    store per failing wave (no atomics) — and the reference's EARLY EXIT at tree granularity: a workgroup reads the flags of
//    its chunk once and does not evaluate trees that an earlier workgroup already found incomplete (h_tree_skip).
//  * blockIdx -> (tile, chunk) is XCD-aware (block b runs on XCD b % 8).  Narrow X (F <= trees per chunk / 8, the default case): chunk-MAJOR,
//    every workgroup in flight walks the same chunk's records (scalar-cache hits) and X is re-read from HBM once per chunk; wide X:
//    all chunks of one X tile back to back on one XCD, the tile re-served by that XCD's L2 (map_block / map_block_grouped).
//  * No MFMA: this is an elementwise map, not a contraction.
#include <hip/hip_runtime.h>
#include <memory>
#include <mutex>
#include <type_traits>

#include <cstdlib>
#include <cstring>

#include "de_bind.h"
#include "de_device_ops.h"
#include "de_kernels.h"

namespace de {

// Wave-uniform read-only data is addressed through the constant address space so
// the compiler emits scalar loads (s_load_*) for it.
#define DE_CONSTANT __attribute__((address_space(4)))
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
typedef const DE_CONSTANT U32x4 *ConstU4Ptr;
typedef const DE_CONSTANT int32_t *ConstI32Ptr;

template <typename T> struct KArgs {
    const BoundInstr *code;  // bound program, padded with one trailing instruction (prefetch reads pc+1)
    const int32_t *code_off; // n_trees + 1
    const T *X;
    T *out;
    uint8_t *ok;
    const T *params;
    const void *classes;
    int64_t N, ldX, ld_out, ld_params, n_tiles, n_classes;
    int32_t F, n_trees, trees_per_chunk, n_chunks, n_slots, xstride;
    int32_t prow_base, n_prows; // parameters staged as LDS rows (EvalArgs)
    // threaded kernel: the first n_prio_blocks workgroups run the (tile prio[k] & 0xFFFFFFFF, chunk) pairs of the PRIORITY tiles (below)
    const unsigned long long *prio;
    uint32_t n_prio_blocks, n_prio, prio_shift; // (keys hold 64-sample units: tile = unit >> prio_shift)
    int32_t classes_is_i64, class_base, vec_store;
    // fused loss (de_eval_loss): residual target, optional weights, per-wave partial sums
    const T *y;
    const T *w;
    T *partial; // [n_tiles][n_trees][4 waves]
    int32_t loss_kind;
    uint32_t cls_row_off; // parametric populations (threaded kernel): LDS byte offset of the class row (h_param)
    // vectorised staging of the X tile (threaded kernel): X 16-byte aligned with ldX == F
    int32_t x_vec;
    // early exit at tree granularity (threaded kernel): a workgroup reads the flags of its chunk's trees once and does not
    // evaluate the trees already known to be incomplete (h_tree_skip); needs trees_per_chunk <= 64
    int32_t skip_flagged;
    uint32_t f_magic; // ceil(2^32 / F) for F > 1 (e / F == umulhi(e, f_magic) while e * F < 2^32), 0 for F == 1
    // COMPACTED launch (threaded kernel, behind the probe launch of the priority tiles): `code` / `code_off` are the re-linked stream of
    // the trees whose flag was still 1 after the probe (de_compact_live_kernel), live_idx[k] = the population index of compact tree k (for
    // its flag byte; the output row comes from the tree's end record) and ctrl = {live trees, chunks, trees per chunk} as the device
    // planned them — n_trees / n_chunks / trees_per_chunk above are the host's upper bounds (the grid).  Null: a plain launch.
    const int32_t *live_idx;
    const int32_t *ctrl;
    // threaded kernel: sample tiles per XCD that run one chunk before the next chunk starts (map_block_grouped); 0 = chunk-fastest (map_block)
    int32_t map_group;
    // flat-switch kernel, CERT variant (de_eval_sum_certificate): per tree the largest |value| among the values the kernel validity-tests,
    // as the bits of a non-negative T in an unsigned word (atomicMax); nothing is stored to `out`
    void *cert_max;
    // threaded kernel: the LAST chunk of the plan runs as `tail_split` sub-chunks (1 = as it is): the workgroups that finish a launch are
    // short ones — the tail of a launch that fills the chip only a few times (10^6 samples: ~9 times, one 60-tree workgroup = 100 us of 900)
    int32_t tail_split;
    // WAVE GROUPS (threaded kernel, template parameter WW > 1; round 6): a workgroup = WW waves on ONE sample tile — X and the staged parameter
    // rows are shared, every wave runs a chunk of its own (chunk = group * WW + wave) with spill-slot rows of its own.  Slot rows are host
    // data (the records carry LDS offsets), so the stream exists in WW variants, `var_stride` records apart (0: the trees use no slot),
    // variant w = slots behind the parameter rows (de_api_program.cpp make_wave_variants); list_off = LDS byte address of wave 1's live-tree
    // list (wave w: + (w - 1) * DE_SKIPLIST_BYTES; wave 0 keeps the list at 0), behind the rows.
    int64_t var_stride;
    uint32_t list_off;
};

// Chunk plan of a launch over n trees and n_tiles sample tiles (host: plan_chunks; device: de_compact_live_kernel for the live trees):
// chunks of <= tpc_max trees, more of them while the grid would not cover the chip `want_blocks` times, never fewer than 8 trees per chunk.
// nc0 = the chunk count before trees are spread evenly: an upper bound of the final count that is monotone in n.
__host__ __device__ inline void chunk_plan(int64_t n, int64_t n_tiles, int64_t tpc_max, int64_t want_blocks, int32_t *n_chunks_out, int32_t *tpc_out, int32_t *nc0_out) {
    if (tpc_max < 1) tpc_max = 63;
    int64_t n_chunks = (n + tpc_max - 1) / tpc_max;
    if (n_tiles > 0 && n_tiles * n_chunks < want_blocks) n_chunks = (want_blocks + n_tiles - 1) / n_tiles;
    const int64_t max_chunks = (n + 7) / 8; // >= 8 trees per chunk
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    if (n_chunks < 1) n_chunks = 1;
    if (nc0_out) *nc0_out = (int32_t)n_chunks;
    const int64_t tpc = n > 0 ? (n + n_chunks - 1) / n_chunks : 1;
    *tpc_out = (int32_t)tpc;
    *n_chunks_out = (int32_t)(n > 0 ? (n + tpc - 1) / tpc : 0);
}

// A thread owns G groups of VW consecutive samples (VW*sizeof(T) = 16 bytes, one
// ds_read_b128 / global_store_dwordx4 per group): samples base + g*(BLOCK*VW) + tid*VW + i.
template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int W = 4; };
template <> struct VecOf<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int W = 2; };

template <typename T> __device__ __forceinline__ T imm_of(uint32_t w2, uint32_t w3);
template <> __device__ __forceinline__ float imm_of<float>(uint32_t w2, uint32_t) { return __uint_as_float(w2); }
template <> __device__ __forceinline__ double imm_of<double>(uint32_t w2, uint32_t w3) {
    return __longlong_as_double((long long)(((unsigned long long)w3 << 32) | w2));
}

#define DE_UNROLL _Pragma("unroll")
#define FOR_G DE_UNROLL for (int g = 0; g < G; g++)
#define FOR_I DE_UNROLL for (int i = 0; i < VW; i++)

// acc = f(b) for every sample
#define U_CASE(OPC, EXPR)                                    \
    case OPC:                                                \
        FOR_G FOR_I {                                        \
            const T x = b[g][i];                             \
            acc[g][i] = (EXPR);                              \
        }                                                    \
        break;
// acc = f(acc, b)
#define B_CASE(OPC, EXPR)                                    \
    case OPC:                                                \
        FOR_G FOR_I {                                        \
            const T x = acc[g][i], y = b[g][i];              \
            acc[g][i] = (EXPR);                              \
        }                                                    \
        break;

// The interpreter's inner loop must contain ONLY wave-uniform control flow: one divergent
// branch anywhere inside it makes LLVM structurize the whole loop and bury the scalar
// dispatch under "Flow" blocks (the kernel is scalar-issue bound).  Anything
// with lane-divergent branches (OCML pow/fmod/tgamma/Payne-Hanek ...) therefore lives in
// __noinline__ functions that take and return register-resident values.
template <typename T, int G> struct VG { typename VecOf<T>::type v[G]; };

// Everything that is not on the fast path of the interpreter loop.
template <typename T, int G>
__device__ __noinline__ VG<T, G> cold_op(uint32_t op, VG<T, G> accv, VG<T, G> bv) {
    using m = M<T>;
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    V (&acc)[G] = accv.v;
    const V (&b)[G] = bv.v;
    switch (op) {
        U_CASE(DE_U_NEG, -x)
        U_CASE(DE_U_ABS, m::abs(x))
        U_CASE(DE_U_SQUARE, x * x)
        U_CASE(DE_U_CUBE, (x * x) * x)
        U_CASE(DE_U_RELU, x < T(0) ? T(0) : x)
        U_CASE(DE_U_SIGN, jl_sign(x))
        U_CASE(DE_U_ROUND, m::rint(x))
        U_CASE(DE_U_FLOOR, m::floor(x))
        U_CASE(DE_U_CEIL, m::ceil(x))
        U_CASE(DE_U_INV, T(1) / x)
        U_CASE(DE_U_SQRT, m::sqrt(x))
        U_CASE(DE_U_CBRT, m::cbrt(x))
        U_CASE(DE_U_EXP, m::exp(x))
        U_CASE(DE_U_COS, m::cos(x))
        U_CASE(DE_U_EXP2, m::exp2(x))
        U_CASE(DE_U_LOG, m::log(x))
        U_CASE(DE_U_LOG2, m::log2(x))
        U_CASE(DE_U_LOG10, m::log10(x))
        U_CASE(DE_U_LOG1P, m::log1p(x))
        U_CASE(DE_U_SIN, m::sin(x))
        U_CASE(DE_U_TAN, m::tan(x))
        U_CASE(DE_U_SINH, m::sinh(x))
        U_CASE(DE_U_COSH, m::cosh(x))
        U_CASE(DE_U_TANH, m::tanh(x))
        U_CASE(DE_U_ASIN, m::asin(x))
        U_CASE(DE_U_ACOS, m::acos(x))
        U_CASE(DE_U_ATAN, m::atan(x))
        U_CASE(DE_U_ASINH, m::asinh(x))
        U_CASE(DE_U_ACOSH, m::acosh(x))
        U_CASE(DE_U_ATANH, m::atanh(x))
        U_CASE(DE_U_SAFE_LOG, x <= T(0) ? m::nan() : m::log(x))
        U_CASE(DE_U_SAFE_LOG2, x <= T(0) ? m::nan() : m::log2(x))
        U_CASE(DE_U_SAFE_LOG10, x <= T(0) ? m::nan() : m::log10(x))
        U_CASE(DE_U_SAFE_LOG1P, x <= T(-1) ? m::nan() : m::log1p(x))
        U_CASE(DE_U_SAFE_SQRT, x < T(0) ? m::nan() : m::sqrt(x))
        U_CASE(DE_U_SAFE_ACOSH, x < T(1) ? m::nan() : m::acosh(x))
    case DE_U_COS2:
        FOR_G FOR_I {
            const T c = m::cos(b[g][i]);
            acc[g][i] = c * c;
        }
        break;
        U_CASE(DE_U_GAMMA, m::tgamma(x))
        B_CASE(DE_B_ADD, x + y)
        B_CASE(DE_B_SUB, x - y)
        B_CASE(DOP_RSUB, y - x)
        B_CASE(DE_B_MUL, x * y)
        B_CASE(DE_B_DIV, x / y)
        B_CASE(DOP_RDIV, y / x)
        B_CASE(DE_B_POW, m::pow(x, y))
        B_CASE(DOP_RPOW, m::pow(y, x))
        B_CASE(DE_B_MAX, jl_max(x, y))
        B_CASE(DE_B_MIN, jl_min(x, y))
        B_CASE(DE_B_MOD, jl_mod(x, y))
        B_CASE(DOP_RMOD, jl_mod(y, x))
        B_CASE(DE_B_REM, m::fmod(x, y))
        B_CASE(DOP_RREM, m::fmod(y, x))
        B_CASE(DE_B_GREATER, x > y ? T(1) : T(0))
        B_CASE(DOP_RGREATER, y > x ? T(1) : T(0))
        B_CASE(DE_B_POW_ABS2, jl_pow_abs2(x, y))
        B_CASE(DOP_RPOW_ABS2, jl_pow_abs2(y, x))
    default: break;
    }
    return accv;
}

// acc = op3(b, c, acc): b, c from spill slots, acc = third argument
template <typename T, int G>
__device__ __noinline__ VG<T, G> cold_op3(uint32_t op, VG<T, G> accv, VG<T, G> bv, VG<T, G> cv) {
    using m = M<T>;
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    V (&acc)[G] = accv.v;
    const V (&b)[G] = bv.v;
    const V (&c)[G] = cv.v;
    FOR_G FOR_I {
        const T x = b[g][i], y = c[g][i], z = acc[g][i];
        T r;
        switch (op) {
        case DE_T_FMA: r = m::fma(x, y, z); break;
        case DE_T_CLAMP: r = x > z ? z : (x < y ? y : x); break;
        case DE_T_ADD3: r = (x + y) + z; break;
        default: r = jl_max(jl_max(x, y), z); break;
        }
        acc[g][i] = r;
    }
    return accv;
}

// Validity accumulation without touching the scalar unit: poison = fma(v, 0, poison) stays
// +0 while every tested value is finite and turns (and stays) NaN at the first Inf/NaN.
template <typename T, int G, typename V>
__device__ __forceinline__ void poison_with(T &poison, const V (&v)[G]) {
    constexpr int VW = VecOf<T>::W;
    FOR_G FOR_I poison = M<T>::fma(v[g][i], T(0), poison);
}
// ... and, in the CERT variant of the flat-switch kernel, the running maximum of |tested value| (NaN is dropped by fmax: the poison has it)
template <typename T, int G, typename V, bool CERT>
__device__ __forceinline__ void test_with(T &poison, T &vmax, const V (&v)[G]) {
    constexpr int VW = VecOf<T>::W;
    poison_with<T, G, V>(poison, v);
    if constexpr (CERT) { FOR_G FOR_I vmax = M<T>::abs(v[g][i]) > vmax ? M<T>::abs(v[g][i]) : vmax; }
}

// cos/sin/exp over the G*VW samples of a thread.  Float32 uses the fast versions of
// de_device_ops.h with ONE divergent fix-up region for out-of-range arguments.
template <int G, bool SIN>
__device__ __noinline__ VG<float, G> trig_fixup(VG<float, G> r, VG<float, G> x) {
    DE_UNROLL for (int g = 0; g < G; g++) DE_UNROLL for (int i = 0; i < 4; i++)
        if (fabsf(x.v[g][i]) > DE_TRIG_FAST_BOUND) r.v[g][i] = SIN ? sinf(x.v[g][i]) : cosf(x.v[g][i]);
    return r;
}
template <int G, bool SIN>
__device__ __noinline__ VG<double, G> trig_f64(VG<double, G> x) {
    DE_UNROLL for (int g = 0; g < G; g++) DE_UNROLL for (int i = 0; i < 2; i++)
        x.v[g][i] = SIN ? ::sin(x.v[g][i]) : ::cos(x.v[g][i]);
    return x;
}
template <typename T, int G, typename V, bool SIN>
__device__ __forceinline__ void vec_trig(V (&out)[G], const V (&x)[G]) {
    constexpr int VW = VecOf<T>::W;
    if constexpr (sizeof(T) == 4) {
        bool big = false;
        VG<float, G> r, xv;
        FOR_G FOR_I {
            r.v[g][i] = fast_trig_f32<SIN>(x[g][i]);
            big |= M<T>::abs(x[g][i]) > DE_TRIG_FAST_BOUND;
        }
        if (__ballot(big) != 0ull) { // wave-uniform: keeps the interpreter loop free of divergent branches
            FOR_G xv.v[g] = x[g];
            r = trig_fixup<G, SIN>(r, xv);
        }
        FOR_G out[g] = r.v[g];
    } else {
        VG<double, G> xv;
        FOR_G xv.v[g] = x[g];
        xv = trig_f64<G, SIN>(xv);
        FOR_G out[g] = xv.v[g];
    }
}
template <typename T, int G, typename V>
__device__ __forceinline__ void vec_exp(V (&out)[G], const V (&x)[G]) {
    constexpr int VW = VecOf<T>::W;
    V r[G];
    if constexpr (sizeof(T) == 4) { FOR_G FOR_I r[g][i] = fast_exp_f32(x[g][i]); }
    else { FOR_G FOR_I r[g][i] = M<T>::exp(x[g][i]); } // OCML exp (f64) is branch-free
    FOR_G out[g] = r[g];
}

#define COLD_CALL(BV)                                          \
    {                                                          \
        VG<T, G> av_, bv_;                                     \
        FOR_G { av_.v[g] = acc[g]; bv_.v[g] = BV[g]; }         \
        av_ = cold_op<T, G>(op, av_, bv_);                     \
        FOR_G acc[g] = av_.v[g];                               \
    }

// XCD-aware block mapping: hardware dispatches block b to XCD b % 8 (observed, used
// for L2 affinity only — correctness never depends on it).  All chunks of a sample
// tile get block ids with the same residue, i.e. run on one XCD back to back.
struct TileMap {
    int64_t tile;
    int32_t chunk;
    bool valid;
};
__device__ __forceinline__ TileMap map_block(uint32_t bid, int32_t n_chunks, int64_t n_tiles) {
    TileMap m;
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

// The same map with the CHUNK slower than a group of `grp` sample tiles (per XCD): the workgroups resident on a CU at one time then walk
// ONE record stream (~10 KB for 64 trees) instead of all of them (n_chunks x 10 KB against a 16 KB scalar cache), and a group's X tiles
// (grp x 5 KB) are still re-served by the XCD's L2 when the next chunk comes round.  grid = ceil(ceil(n_tiles / 8) / grp) * grp * 8 * n_chunks.
__device__ __forceinline__ TileMap map_block_grouped(uint32_t bid, int32_t n_chunks, int64_t n_tiles, uint32_t grp) {
    TileMap m;
    const uint32_t xcd = bid & 7u, idx = bid >> 3;
    const uint32_t per = grp * (uint32_t)n_chunks;
    const uint32_t g = idx / per, r = idx - g * per;
    m.chunk = (int32_t)(r / grp);
    m.tile = ((int64_t)g * grp + (r - (uint32_t)m.chunk * grp)) * 8 + xcd;
    m.valid = m.tile < n_tiles;
    return m;
}

// Parameters as rows: row prow_base + p of the tile holds params[p, class of the sample] (src/ParametricExpression.jl:381-389), staged
// once per workgroup like the X tile.  row_elems = elements between two rows; samples past N repeat the last one (as X does).
template <typename T>
__device__ __forceinline__ void stage_param_rows(const KArgs<T> &a, T *__restrict__ rows, int row_elems, int64_t base, int tile, int tid, int blk) {
    const int64_t last = a.N - 1;
    for (int j = tid; j < tile; j += blk) {
        int64_t jj = base + j;
        jj = jj < last ? jj : last;
        const int64_t cl = clamp_class((a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj]
                                                         : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj]) - a.class_base, a.n_classes);
        const T *__restrict__ col = a.params + a.ld_params * cl;
        for (int p = 0; p < a.n_prows; p++) rows[(size_t)(a.prow_base + p) * row_elems + j] = col[p];
    }
}
template <typename T, int G>
__device__ __noinline__ void store_ragged(T *o, VG<T, G> v, int64_t remaining, int plane) {
    constexpr int VW = VecOf<T>::W;
    FOR_G FOR_I if ((int64_t)g * plane + i < remaining) o[g * plane + i] = v.v[g][i];
}
__device__ __noinline__ void flag_incomplete(uint8_t *ok, int agent) { // agent scope: workgroups that start later skip the tree (early exit)
    if ((threadIdx.x & 63) == 0) {
        if (agent) __hip_atomic_store(ok, (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *ok = 0;
    }
}

// DIRECT = true: wide feature matrices whose X tile does not fit in LDS — feature operands are
// gathered from global memory (L1/L2 absorb the re-reads), LDS holds only the spill rows.
template <typename T, int G, int BLK, bool EE, bool PARAMS, bool DIRECT = false, bool CERT = false>
__global__ void __launch_bounds__(BLK) de_eval_tape_kernel(const KArgs<T> a) {
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    constexpr int GT = BLK * VW;        // samples per group plane
    constexpr int TILE = GT * G;        // samples per workgroup
    constexpr int ROWV = BLK * G + 1;   // LDS row stride in vectors (+1: bank spread for the staging writes)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *__restrict__ rows = reinterpret_cast<T *>(smem_raw);  // rows 0..F-1: X tile; row F+s: spill slot s
    V *__restrict__ rowsv = reinterpret_cast<V *>(smem_raw);

    const TileMap tm = map_block(blockIdx.x, a.n_chunks, a.n_tiles);
    if (!tm.valid) return;
    const int tid = threadIdx.x;
    const int64_t base = tm.tile * TILE;
    const int64_t last = a.N - 1;

    // ---- stage the X tile: coalesced HBM/L2 read, transposed LDS write ----------
    // sample j of the tile lives at rows[f*ROWV*VW + j]  (plane g = j / GT, lane = (j % GT) / VW)
    if (!DIRECT) {
        const uint32_t F = (uint32_t)a.F;
        const uint32_t total = (uint32_t)TILE * F;
        if (a.ldX == (int64_t)F && base + TILE <= a.N) {
            const T *__restrict__ src = a.X + base * (int64_t)F; // contiguous TILE*F elements
            for (uint32_t e = tid; e < total; e += BLK) {
                const uint32_t j = e / F, f = e - j * F;
                rows[f * (ROWV * VW) + j] = src[e];
            }
        } else { // ragged tail / strided X: clamp to the last real sample
            for (uint32_t e = tid; e < total; e += BLK) {
                const uint32_t j = e / F, f = e - j * F;
                int64_t jj = base + j;
                jj = jj < last ? jj : last;
                rows[f * (ROWV * VW) + j] = a.X[f + a.ldX * jj];
            }
        }
    }
    if (PARAMS && !DIRECT && a.n_prows > 0) stage_param_rows<T>(a, rows, ROWV * VW, base, TILE, tid, BLK);
    int64_t cls[G][VW];
    if (PARAMS) {
        FOR_G FOR_I {
            int64_t jj = base + g * GT + tid * VW + i;
            jj = jj < last ? jj : last;
            cls[g][i] = clamp_class((a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj]
                                                      : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj]) -
                                        a.class_base, a.n_classes);
        }
    }
    __syncthreads();

    const ConstU4Ptr code = (ConstU4Ptr)(uintptr_t)a.code;
    const ConstI32Ptr code_off = (ConstI32Ptr)(uintptr_t)a.code_off;
    const int t0 = tm.chunk * a.trees_per_chunk;
    const int t1 = (t0 + a.trees_per_chunk < a.n_trees) ? t0 + a.trees_per_chunk : a.n_trees;
    const bool full = base + TILE <= a.N;
    uint64_t skip = 0ull; // trees of the chunk already known to be incomplete (see de_eval_threaded_kernel): not evaluated
    if (EE && a.skip_flagged && t1 - t0 <= 64) {
        const int i = t0 + (tid & 63);
        const uint8_t f = i >= t1 ? (uint8_t)1 : skip_flag_load(a.ok + i, a.skip_flagged, tm.tile);
        skip = __ballot(f == 0);
    }

    int pe = code_off[t0];
    for (int tree = t0; tree < t1; ++tree) {
        int pc = pe;
        pe = code_off[tree + 1];
        if ((skip >> (tree - t0)) & 1ull) continue;
        V acc[G];
        FOR_G FOR_I acc[g][i] = T(0);
        T poison = T(0);
        T vmax = T(0); // (CERT) largest |tested value| of this thread's samples
        U32x4 nxt = code[pc]; // scalar load; a tree has at least one instruction
        for (; pc < pe; ++pc) {
            const U32x4 w = nxt;
            nxt = code[pc + 1]; // prefetch (the code buffer carries one trailing pad instruction)
            // One flat, wave-uniform switch over the bound handler id (de_bind.h): every case is
            // straight-line code.  ROW(r) = this thread's vectors of LDS row r.
#define ROWP(r) (rowsv + ((r) - (DIRECT ? (uint32_t)a.F : 0u)) * ROWV + tid)
#define LOAD_ROW(dst, r)                                                                              \
    {                                                                                                 \
        if (DIRECT && (r) < (uint32_t)a.F) {                                                          \
            FOR_G FOR_I {                                                                             \
                int64_t jj_ = base + g * GT + tid * VW + i;                                           \
                jj_ = jj_ < last ? jj_ : last;                                                        \
                dst[g][i] = a.X[(r) + a.ldX * jj_];                                                   \
            }                                                                                         \
        } else {                                                                                      \
            const V *__restrict__ s_ = ROWP(r);                                                       \
            FOR_G dst[g] = s_[g * BLK];                                                               \
        }                                                                                             \
    }
#define BIN4(K, EXPR)                                                                                   \
    case BOP_BIN_BASE + 4 * K + 0: { V b[G]; LOAD_ROW(b, w.y) FOR_G FOR_I { const T x = acc[g][i], y = b[g][i]; acc[g][i] = (EXPR); } } break; \
    case BOP_BIN_BASE + 4 * K + 1: { V b[G]; LOAD_ROW(b, w.y) FOR_G FOR_I { const T x = acc[g][i], y = b[g][i]; acc[g][i] = (EXPR); } test_with<T, G, V, CERT>(poison, vmax, acc); } break; \
    case BOP_BIN_BASE + 4 * K + 2: { const T y = imm_of<T>(w.z, w.w); FOR_G FOR_I { const T x = acc[g][i]; acc[g][i] = (EXPR); } } break; \
    case BOP_BIN_BASE + 4 * K + 3: { const T y = imm_of<T>(w.z, w.w); FOR_G FOR_I { const T x = acc[g][i]; acc[g][i] = (EXPR); } test_with<T, G, V, CERT>(poison, vmax, acc); } break;
#define UN4(K, CALL)                                                                                    \
    case BOP_UN_BASE + 4 * K + 0: { V x_[G]; FOR_G x_[g] = acc[g]; CALL; } break;                      \
    case BOP_UN_BASE + 4 * K + 1: { V x_[G]; FOR_G x_[g] = acc[g]; CALL; test_with<T, G, V, CERT>(poison, vmax, acc); } break; \
    case BOP_UN_BASE + 4 * K + 2: { V x_[G]; LOAD_ROW(x_, w.y) CALL; } break;                          \
    case BOP_UN_BASE + 4 * K + 3: { V x_[G]; LOAD_ROW(x_, w.y) CALL; test_with<T, G, V, CERT>(poison, vmax, acc); } break;
            switch (w.x) {
            case BOP_LOAD_ROW: LOAD_ROW(acc, w.y) break;
            case BOP_LOAD_CONST: { const T c = imm_of<T>(w.z, w.w); FOR_G FOR_I acc[g][i] = c; } break;
            case BOP_PUSH: { V *__restrict__ s_ = ROWP(w.y); FOR_G s_[g * BLK] = acc[g]; } break;
            case BOP_CHECK_ROW: { V b[G]; LOAD_ROW(b, w.y) test_with<T, G, V, CERT>(poison, vmax, b); } break;
            case BOP_CHECK_ACC: test_with<T, G, V, CERT>(poison, vmax, acc); break;
            BIN4(0, x + y)
            BIN4(1, x - y)
            BIN4(2, y - x)
            BIN4(3, x * y)
            BIN4(4, x / y)
            BIN4(5, y / x)
            UN4(0, (vec_trig<T, G, V, false>(acc, x_)))
            UN4(1, (vec_exp<T, G, V>(acc, x_)))
            UN4(2, (vec_trig<T, G, V, true>(acc, x_)))
            case BOP_GEN_ROW: { const uint32_t op = w.y >> 24; V b[G]; LOAD_ROW(b, w.y & 0xFFFFFFu) COLD_CALL(b) } break;
            case BOP_GEN_CONST: { const uint32_t op = w.y >> 24; V b[G]; const T c = imm_of<T>(w.z, w.w); FOR_G FOR_I b[g][i] = c; COLD_CALL(b) } break;
            case BOP_GEN_ACC: { const uint32_t op = w.y >> 24; COLD_CALL(acc) } break;
            case BOP_TERN: {
                const uint32_t op = w.y >> 24;
                VG<T, G> av, bv, cv;
                const V *__restrict__ s1 = ROWP(w.y & 0xFFFFFFu);
                const V *__restrict__ s2 = ROWP(w.z);
                FOR_G { av.v[g] = acc[g]; bv.v[g] = s1[g * BLK]; cv.v[g] = s2[g * BLK]; }
                av = cold_op3<T, G>(op, av, bv, cv);
                FOR_G acc[g] = av.v[g];
            } break;
            case BOP_INJ_ACC: { // is_valid(x_l) ? op(x_l) : Inf   (src/Evaluate.jl:722,787)
                const uint32_t op = w.y >> 24;
                V inj[G];
                FOR_G inj[g] = acc[g];
                COLD_CALL(inj)
                FOR_G FOR_I acc[g][i] = M<T>::isfinite(inj[g][i]) ? acc[g][i] : M<T>::inf();
            } break;
            case BOP_INJ_ROW: {
                const uint32_t op = w.y >> 24;
                V inj[G];
                LOAD_ROW(inj, w.y & 0xFFFFFFu)
                COLD_CALL(inj)
                FOR_G FOR_I acc[g][i] = M<T>::isfinite(inj[g][i]) ? acc[g][i] : M<T>::inf();
            } break;
            case BOP_GEN_PARAM:
                if constexpr (PARAMS) {
                    const uint32_t op = w.y >> 24;
                    V b[G];
                    const T *__restrict__ s_ = a.params + (w.y & 0xFFFFu);
                    FOR_G FOR_I b[g][i] = s_[a.ld_params * cls[g][i]];
                    if (EE && (w.y & (1u << 23))) test_with<T, G, V, CERT>(poison, vmax, b);
                    if (op == DOP_LOAD) { FOR_G acc[g] = b[g]; }
                    else COLD_CALL(b)
                }
                break;
            default: break;
            }
        }
        // ---- store out[tree][...]: one 16-byte store per group, coalesced over the wave
        if constexpr (CERT) {
            // the largest |tested value| of the tree so far: wave maximum, one atomicMax per wave on the value's bits (non-negative
            // floats order like unsigned integers); samples past N repeat the last real one, so they add nothing
            typedef typename std::conditional<sizeof(T) == 4, unsigned int, unsigned long long>::type UB;
            DE_UNROLL for (int m = 32; m >= 1; m >>= 1) {
                const T o2 = __shfl_xor(vmax, m, 64);
                vmax = o2 > vmax ? o2 : vmax;
            }
            if ((tid & 63) == 0 && vmax > T(0)) {
                UB bits;
                __builtin_memcpy(&bits, &vmax, sizeof bits);
                atomicMax(reinterpret_cast<UB *>(a.cert_max) + tree, bits);
            }
            if (__ballot(poison != poison) != 0ull) flag_incomplete(a.ok + tree, a.skip_flagged == 1);
            continue;
        }
        T *__restrict__ o = a.out + (int64_t)tree * a.ld_out + base + tid * VW;
        if (full && a.vec_store) {
            FOR_G *reinterpret_cast<V *>(o + g * GT) = acc[g];
        } else {
            VG<T, G> av;
            FOR_G av.v[g] = acc[g];
            store_ragged<T, G>(o, av, a.N - (base + tid * VW), GT);
        }
        // ---- completion flag: one ballot per wave, one byte store per failing wave
        if (__ballot(poison != poison) != 0ull) flag_incomplete(a.ok + tree, a.skip_flagged == 1);
    }
}

// ===========================================================================
// Threaded-code variant: every bound handler is its own function and the interpreter loop is
// {prefetch, handler address = base + offset, s_swappc}.  The flat switch above costs ~37
// scalar+branch instructions per interpreted instruction (LLVM lowers a switch to a compare
// tree and then structurizes it); an indirect call costs ~20, and each handler is compiled
// as clean straight-line code.  Handler addresses are taken on the device
// (de_fill_handlers), read back once per process, and bound into the instruction stream as
// 32-bit offsets by the host (de_api_program.cpp).  G = 1, 256 threads per workgroup.
// The validity poison is accumulated two lanes wide for Float32 so that one v_pk_fma_f32 tests two
// samples (2 VALU per tested vector instead of 4).
template <typename T> struct PoisonOf { typedef T type; };
template <> struct PoisonOf<float> { typedef float type __attribute__((ext_vector_type(2))); };
// (packed: the 16-byte alignment of `acc` would give the struct 8 bytes of tail padding, which the calling convention passes as
// eight i8 arguments — eight vector registers; they now carry the fused loss's targets and weights, HL_PARAMS below)
// PLANES (round 4).  A lane of the threaded kernel owns TG<T>::G 16-byte vectors per row — "planes": plane g of a row holds the
// samples g * 64 * VW + lane * VW + i of the tile, 1024 bytes behind plane g - 1 both in the LDS row and in the output row — and a
// handler applies its instruction to all of them in one dispatch.  The shipped build has ONE plane; two Float32 planes (-DDE_TG=2: the
// per-dispatch and per-tree costs paid once per 8 samples, an independent twin for every dependent VALU chain) measured slower because
// the longer rows halve the resident waves (de_kernels.h; profiles/r4_planes_occupancy.txt).
template <typename T> struct TG { static constexpr int G = sizeof(T) == 4 ? DE_TG : 1; };
static_assert(DE_TG == TG_F32, "de_kernels.h TG_F32 and DE_TG disagree");
#define DE_PLANE_BYTES (DE_TBLK * 16u) // bytes between two planes of a row (LDS and output alike)
#if DE_TG == 1
// one plane everywhere: NOT a loop (a one-trip loop is unrolled late, and clang's earlier branch-versus-select decisions then differ from
// straight-line code: the exact-extremum branch of cos / sin and the NaN / signed-zero branches of max / min came out if-converted)
#define FOR_PLANES if (constexpr int g = 0; true)
#else
#define FOR_PLANES DE_UNROLL for (int g = 0; g < TG<T>::G; g++)
#endif
// the chain's state: what every handler receives, updates and hands on
template <typename T> struct __attribute__((packed, aligned(8))) HState {
    typename VecOf<T>::type acc[TG<T>::G];
    typename PoisonOf<T>::type poison;
};
// ... and ONE plane of it: what the handler BODIES (b_*) work on
template <typename T> struct __attribute__((packed, aligned(8))) BState {
    typename VecOf<T>::type acc;
    typename PoisonOf<T>::type poison;
};
// this thread's residual targets / weights of the fused loss, per plane: they travel as FOUR vector arguments (ly0, ly1, lw0, lw1: plane 1's
// are unused with one plane) — clang passes an aggregate in registers only while all aggregates of the signature fit 16 registers, and the
// state above takes 10 of them: a struct of the four vectors went through the stack (a scratch pointer per dispatch)
template <typename T> struct HLoss { typename VecOf<T>::type v[2]; };
// ONE call signature for every handler (several call sites with different signatures make the
// compiler shuffle the returned state through a dozen v_movs per call):
//   st  : accumulator + validity poison, in and out, stays in v0..v4 across calls
//   la  : (LDS byte address of this thread's vector of the operand row) | aux << 24, aux = the
//         de_opcode for the generic handlers and 0 for the hot ones (which use `la` unmasked)
//   imm : the immediate's bits (uint32 for Float32, uint64 for Float64); for BOP_TERN the byte
//         distance from operand row B to row C
// = 2 VALU argument moves per call for Float32.
template <typename T> struct ImmBits;
template <> struct ImmBits<float> { typedef uint32_t type; };
template <> struct ImmBits<double> { typedef uint64_t type; };
template <typename T> __device__ __forceinline__ T imm_from(typename ImmBits<T>::type b);
template <> __device__ __forceinline__ float imm_from<float>(uint32_t b) { return __uint_as_float(b); }
template <> __device__ __forceinline__ double imm_from<double>(uint64_t b) { return __longlong_as_double((long long)b); }
// Handler BODIES keep that signature (b_*: forceinline); what the instruction stream points at is h_chain<T, &body>
// below, which fetches the body's operands from the stream and TAIL-CALLS the next instruction's handler.
#define HARGS BState<T> st, uint32_t la, typename ImmBits<T>::type imm
#define LDSP(T, addr) (reinterpret_cast<__attribute__((address_space(3))) typename VecOf<T>::type *>((uintptr_t)(addr)))
template <typename T> using BodyFn = BState<T> (*)(BState<T>, uint32_t, typename ImmBits<T>::type);
// ---- direct-threaded dispatch ------------------------------------------------------------------------------------
// The interpreter has no central loop: every handler ends with a tail call (s_setpc_b64) to the handler of the next
// instruction.  The stream holds one 16-byte record per instruction, plus an end record per tree and one head record in
// front of the first tree; a record carries its own operand words and the address of the NEXT record's handler:
//   Float32 record  { la, imm, next.lo, next.hi }     la = LDS byte offset of the operand row | aux << 24
//   Float64 record  { la, next.lo, imm.lo, imm.hi }   (next.hi = the high half of the current pc: all handlers of a code
//                                                      object lie in one 4 GiB window, checked on the host)
// and it is SOFTWARE-PIPELINED: a handler receives its record's words in SGPRs from its predecessor and, as its first
// instruction, loads the NEXT record (s_load_dwordx4) — the scalar-cache latency overlaps with its own LDS read and
// arithmetic — then tail-calls the next handler, whose address it has known since entry, WITHOUT waiting for that load: the
// record's words travel to the callee still in flight (they are loaded straight into the argument registers) and the
// callee's entry wait completes them, so the instruction fetch of the jump overlaps with the tail of the load as well.
// (Before, the jump target came out of the load itself: every tiny handler sat out the whole scalar-cache latency before it
// could jump, §4.3 of DESIGN.md.)  The end record's handler is h_tree_end; its `next` is the first handler of the next tree.
// Stream pointer and operand words travel in SGPRs (csrc/irpatch.py marks those parameters `inreg` in the optimised IR:
// clang has no source spelling for it on a device function); the argument order puts the record's four words in an aligned
// SGPR quad (s[4:7]) so that the load can target them.
// Behind the operand words every handler hands on five more wave-uniform words untouched (they stay in their SGPRs from
// the kernel's call to the last handler of the chunk): what h_tree_end needs to finish a tree WITHOUT returning to the kernel.
//   outp  : address of out[0, first sample of this tile] minus the LDS base (so that outp + tree * ldo + lds0 is this lane's vector)
//   okp   : the completion flags;   ldo : bytes between two trees' output rows;   left : trees of this chunk still to run
//   flags : HF_* | samples of the tile inside N (slow store)
//   tree  : index of the tree being evaluated (the end handlers count it up)
// argument words of a handler: (la, w1, w23) = the record as loaded { x, y, z:w }
//   Float32: w1 = imm, w23 = next handler      Float64: w1 = next handler (low half), w23 = imm
// ... and, in VECTOR registers, this thread's residual targets and weights of the fused loss (ly, lw: 8 registers every handler passes
// on untouched, undefined outside a fused-loss launch): the end of a tree forms its loss partial from them (h_tree_end_slow, HF_LOSS).
#define HL_T typename VecOf<T>::type
// (the *_C forms carry their trailing comma: a build without the fused loss — -DDE_NO_LOSS=1, the cross-wave module de_kernels_xs — drops
// the arguments altogether: 8 / 16 vector registers every handler would otherwise keep pinned)
#ifndef DE_NO_LOSS
#define DE_NO_LOSS 0
#endif
#if DE_NO_LOSS
#define HL_PARAMS_C
#define HL_PASS_C
#define HL_BOTH(ly, lw) const HLoss<T> ly{}, lw{}
#define HL_TYPES_C(V)
#elif DE_TG == 1
#define HL_PARAMS_C HL_T ly0, HL_T lw0,
#define HL_PASS_C ly0, lw0,
#define HL_BOTH(ly, lw) const HLoss<T> ly{{ly0, ly0}}, lw{{lw0, lw0}}
#define HL_TYPES_C(V) V, V,
#else
#define HL_PARAMS_C HL_T ly0, HL_T ly1, HL_T lw0, HL_T lw1,
#define HL_PASS_C ly0, ly1, lw0, lw1,
#define HL_BOTH(ly, lw) const HLoss<T> ly{{ly0, ly1}}, lw{{lw0, lw1}}
#define HL_TYPES_C(V) V, V, V, V,
#endif
template <typename T> using HandlerFn = HState<T> (*)(HState<T>, HL_TYPES_C(HL_T) uint32_t, ConstU4Ptr, uint64_t, uint32_t, uint32_t, uint64_t, uint64_t, uint64_t, uint64_t, uint32_t, uint32_t);
enum : uint32_t { HF_SLOW_STORE = 1u << 30, HF_NO_STORE = 1u << 29, HF_VALID_MASK = 0x3FFu, // (samples of the tile inside N: <= 512)
                  // bits 10..24: LDS byte address >> 4 of this WAVE's live-tree list (h_tree_skip; 0 = the list in front of row 0 — every
                  // one-wave workgroup, wave 0 of a wave group)
                  HF_LIST_SHIFT = 10, HF_LIST_MASK = 0x7FFFu,
                  HF_SLOW = 1u << 31, // set with any of HF_SLOW_STORE / HF_NO_STORE / HF_LOSS: the out-of-line end of a tree (the sign bit: ONE scalar compare)
                  // plain flag stores (through the caches): always, except under flag protocol 1 (agent scope for every access, an
                  // experiment: skip_flag_load, de_device_ops.h)
                  HF_PLAIN_FLAG = 1u << 28,
                  // fused loss: the end of a tree forms the tree's loss partial of this tile instead of storing the values (h_tree_end_slow);
                  // HF_LOSS_L1 = |e| instead of e^2
                  HF_LOSS = 1u << 27, HF_LOSS_L1 = 1u << 26,
                  // ... of a full tile without weights: sum of e^2 / |e| without the per-sample weight selects (14 instead of ~40 VALU instructions
                  // per tree and wavefront: the fused loss was SLOWER than the eval it replaces, 75 cycles per tree-wave of loss arithmetic
                  // against one store)
                  HF_LOSS_PLAIN = 1u << 25 };
template <typename T> __device__ __forceinline__ HandlerFn<T> arg_next(uint32_t w1, uint64_t w23);
template <> __device__ __forceinline__ HandlerFn<float> arg_next<float>(uint32_t, uint64_t w23) { return reinterpret_cast<HandlerFn<float>>(w23); }
template <> __device__ __forceinline__ HandlerFn<double> arg_next<double>(uint32_t w1, uint64_t) {
    return reinterpret_cast<HandlerFn<double>>((__builtin_amdgcn_s_getpc() & 0xFFFFFFFF00000000ull) | w1);
}
template <typename T> __device__ __forceinline__ typename ImmBits<T>::type arg_imm(uint32_t w1, uint64_t w23);
template <> __device__ __forceinline__ uint32_t arg_imm<float>(uint32_t w1, uint64_t) { return w1; }
template <> __device__ __forceinline__ uint64_t arg_imm<double>(uint32_t, uint64_t w23) { return w23; }
#define DE_SKIPLIST_BYTES 256u // LDS bytes in front of row 0: the live trees of the running (sub-)chunk (h_tree_skip), 64 x 4 bytes
template <typename T> struct RowOf { static constexpr uint32_t BYTES = (uint32_t)(DE_TBLK * TG<T>::G + 1) * 16u; }; // LDS row stride: the planes + one vector of padding (= trow_bytes, de_kernels.h)
// `code` points at the record of the NEXT instruction; (la, w1, w23) are this instruction's record
#define HCHAIN_ARGS HState<T> st, HL_PARAMS_C uint32_t lds0, ConstU4Ptr code, uint64_t outp, uint32_t la, uint32_t w1, uint64_t w23, uint64_t okp, uint64_t ldo, uint64_t skip, uint32_t left, uint32_t flags
#define HCHAIN_NEXT_AT(W, NEXT) [[clang::musttail]] return arg_next<T>(w1, w23)(st, HL_PASS_C lds0, NEXT, outp, (W).x, (W).y, ((uint64_t)(W).w << 32) | (W).z, okp, ldo, skip, left, flags)
// record address + k records WITHOUT a carry into the high half: the stream lies inside one 4 GiB window (checked where it is allocated,
// de_api.cpp prog_malloc), so the bump is ONE scalar instruction (s_add_u32) instead of the add / add-with-carry pair — the cheap handlers are bound
// by their scalar instructions (tools/probe/issue_probe.py: one per ~4 cycles and SIMD; `acc * const` = 6 of them = 24 cycles)
__device__ __forceinline__ ConstU4Ptr code_at(ConstU4Ptr c, int k) {
    const uint64_t a = (uint64_t)(uintptr_t)c;
    return (ConstU4Ptr)(uintptr_t)((a & 0xFFFFFFFF00000000ull) | (uint64_t)((uint32_t)a + (uint32_t)(16 * k)));
}
#define HCHAIN_NEXT(W) HCHAIN_NEXT_AT(W, code_at(code, 1))
// every plane through a body (the planes' instruction sequences are independent: the scheduler interleaves them)
template <typename T, BodyFn<T> BODY> __device__ __forceinline__ void planes_apply(HState<T> &st, uint32_t a, typename ImmBits<T>::type imm) {
    FOR_PLANES {
        BState<T> b{st.acc[g], st.poison};
        b = BODY(b, a + (uint32_t)g * DE_PLANE_BYTES, imm); // (no carry into the aux byte: LDS < 2^18 bytes)
        st.acc[g] = b.acc;
        st.poison = b.poison;
    }
}
template <typename T, BodyFn<T> BODY> __device__ __noinline__ HState<T> h_chain(HCHAIN_ARGS) {
    const U32x4 w = *code;
    planes_apply<T, BODY>(st, lds0 + la, arg_imm<T>(w1, w23)); // la = row byte offset | aux << 24
    HCHAIN_NEXT(w);
}

__device__ __forceinline__ void hpoison_impl(PoisonOf<float>::type &poison, const VecOf<float>::type &v) {
    typedef PoisonOf<float>::type P2;
    const P2 z = {0.0f, 0.0f};
    poison = __builtin_elementwise_fma(P2{v[0], v[1]}, z, poison);
    poison = __builtin_elementwise_fma(P2{v[2], v[3]}, z, poison);
}
__device__ __forceinline__ void hpoison_impl(double &poison, const VecOf<double>::type &v) {
    poison = ::fma(v[0], 0.0, poison);
    poison = ::fma(v[1], 0.0, poison);
}
template <typename T> __device__ __forceinline__ void hpoison(typename PoisonOf<T>::type &poison, const typename VecOf<T>::type &v) {
    hpoison_impl(poison, v);
}
__device__ __forceinline__ bool poison_set(const PoisonOf<float>::type &p) { return (p[0] != p[0]) | (p[1] != p[1]); }
__device__ __forceinline__ bool poison_set(const double &p) { return p != p; }
// The end record of a tree (la = the tree's index).  The trees of a chunk are consecutive in the stream, so the record
// behind it is the first instruction of the next tree: the handler stores the tree's results and its flag, clears the
// state and tail-calls on — the next record was requested at its first instruction, the store is still in flight when
// the next tree starts (the eval handlers do not wait for vector memory at entry: csrc/asmpatch.py).  Measured with
// tools/exp_dispatch_cost.py: returning to a per-tree loop in the kernel (index load, first-record load, two dependent
// scalar-cache round trips per tree) cost ~250 SIMD cycles per tree and wavefront, a quarter of the headline's time.
// Fused loss (HF_LOSS): the same chain; the end of a tree forms its loss partial instead of storing the values (h_tree_end_slow).
// EARLY EXIT (src/Evaluate.jl:26-32: the reference stops evaluating a tree at its first non-finite intermediate array; SURVEY §8a:
// with ok == false only the flag is contractual).  `skip` = the trees from the current one on whose flag was already 0 when this
// workgroup started (bit 0: the current tree; the kernel reads the flags once per chunk, DE_OPT_FULL_EVAL / early_exit = false: 0).
// Such a tree is not evaluated: the end of its predecessor tail-calls h_tree_skip, which walks the HEADER records — the
// record in front of a tree's first instruction (the previous tree's end record / the head record) carries the number of records of
// the tree — to the next tree that still has to run.  One scalar load per skipped tree instead of its evaluation.
// The walk is ONE LDS read and ONE scalar load per run of skipped trees: when a (sub-)chunk starts, the kernel writes the header
// addresses (low 32 bits; the stream lies inside one 4 GiB window: checked on the host) of its LIVE trees to the first
// DE_SKIPLIST_BYTES of LDS, in REVERSE order — entry r = the live tree that has r live trees after it — so that a handler finds
// the next live tree from what it carries anyway: r = (trees left after it) - (skip bits set after it).
// (Round 3's first version followed the headers tree by tree — a dependent scalar-cache round trip per skipped tree, 64 SIMD
// cycles per skipped tree and wavefront, 0.65 ms of the 7.9 ms headline; tools/exp_skip_cost.py.)
template <typename T> __device__ __noinline__ HState<T> h_tree_skip(HCHAIN_ARGS) { // bit 0 of `skip` = the next tree of the stream, which is skipped
    const uint32_t n = (uint32_t)__builtin_ctzll(~skip); // the run of skipped trees (>= 1; bit `left` is the sentinel behind the last tree, nothing above it)
    if (n >= left) return st;
    skip >>= n;
    left -= n;
    const uint32_t r = left - (uint32_t)__builtin_popcountll(skip >> 1); // live trees after the one the chain goes on with (the count holds the sentinel)
    const uint32_t lb = (flags >> (HF_LIST_SHIFT - 4)) & (HF_LIST_MASK << 4); // this wave's list (0 unless the workgroup is a wave group)
    const uint32_t lo = *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)(lb + r * 4u)); // (wave-uniform address)
    const uint64_t hdr = ((uint64_t)(uintptr_t)code & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    const ConstU4Ptr nh = (ConstU4Ptr)(uintptr_t)hdr; // the tree's header record (the record in front of its first instruction)
    const U32x4 hn = nh[0], w = nh[1];
    [[clang::musttail]] return arg_next<T>(hn.y, ((uint64_t)hn.w << 32) | hn.z)(st, HL_PASS_C lds0, code_at(nh, 2), outp, w.x, w.y, ((uint64_t)w.w << 32) | w.z, okp, ldo, skip, left, flags);
}
// the rest of a tree's end: flag byte, last tree of the chunk?, clear the state, on to the next tree (W = its first record,
// HDR = the address of its header record) unless that one is skipped.  `tree` (a local of the caller) = the tree's index, the
// operand word of its END RECORD — not a counter the chain carries: the trees of a chunk need not be consecutive trees of the
// population (the live trees of a compacted launch: de_compact_live_kernel)
#define HTREE_END_TAIL(REC, NEXT, HDR)                                                                       \
    if (__builtin_expect(__ballot(poison_set(st.poison)) != 0ull, 0)) { /* every lane the same byte */         \
        if (flags & HF_PLAIN_FLAG) *reinterpret_cast<__attribute__((address_space(1))) uint8_t *>(okp + tree) = 0; \
        else __hip_atomic_store(reinterpret_cast<__attribute__((address_space(1))) uint8_t *>(okp + tree), (uint8_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* visible to the workgroups that start later */ \
    }                                                                                                        \
    /* (the accumulator is left as it is: no tree starts by reading it — de_bind.h top_reads_acc, checked by de_program_verify) */ \
    st.poison = typename PoisonOf<T>::type{};                                                                \
    {                                                                                                        \
        /* ONE test for "the next tree is skipped" and "this was the last tree": bit 1 of the mask — a skip bit, or the SENTINEL the   \
           kernel puts behind the last tree's bit (a chain runs <= 63 trees).  Written as the two scalar instructions it is: from   \
           `(skip & 2) != 0` the compiler makes a 64-bit and / compare on a copy, and a counter of the trees left costs three more */ \
        __label__ de_end_special;                                                                            \
        asm goto("s_bitcmp1_b32 %0, 1\n\ts_cbranch_scc1 %l[de_end_special]" : : "s"((uint32_t)skip) : "scc" : de_end_special); \
        asm("s_lshr_b64 %0, %0, 1" : "+s"(skip) : : "scc");                                                  \
        HCHAIN_NEXT_AT(REC, NEXT);                                                                           \
    de_end_special:;                                                                                         \
        const uint32_t left_now = 63u - (uint32_t)__builtin_clzll(skip); /* trees left including this one = the sentinel's bit */ \
        if (left_now <= 1u) return st;                                                                       \
        [[clang::musttail]] return h_tree_skip<T>(st, HL_PASS_C lds0, HDR, outp, la, w1, w23, okp, ldo, skip >> 1, left_now - 1u, flags); \
    }
// the output store of a tree's fast end: 16 bytes per lane, 1 KiB contiguous per wavefront.  -DDE_NT_STORE=1 (an A/B switch): non-temporal
// (the 40 GB output stream of the headline is written once and never read by the kernel)
#ifndef DE_NT_STORE
#define DE_NT_STORE 1
#endif
#if DE_NT_STORE
#define DE_OUT_STORE(VT, PTR, VAL) __builtin_nontemporal_store((VAL), reinterpret_cast<__attribute__((address_space(1))) VT *>(PTR))
#else
#define DE_OUT_STORE(VT, PTR, VAL) (*reinterpret_cast<__attribute__((address_space(1))) VT *>(PTR) = (VAL))
#endif
typedef __attribute__((address_space(1))) char *GPtr; // global, not flat: a flat store also ties up lgkmcnt
// The other ends of a tree (flags & HF_SLOW), out of line so that h_tree_end itself is straight-line code: HF_LOSS (fused loss:
// the tree's loss partial of this tile), HF_SLOW_STORE (ragged last tile / output rows that are not 16-byte aligned; LDS base = 0:
// lds0 = DE_SKIPLIST_BYTES + 16 * thread), HF_NO_STORE (DE_DEBUG_NO_STORE, measurement only: keep the value alive, write nothing).
template <typename T> __device__ __noinline__ HState<T> h_tree_end_slow(HCHAIN_ARGS) { // la = the tree's index (both callers)
    constexpr int VW = VecOf<T>::W;
    const U32x4 w = *code;
    const uint32_t tree = la;
    const GPtr row = reinterpret_cast<GPtr>(outp + (uint64_t)tree * ldo);
    if (!DE_NO_LOSS && (flags & HF_LOSS)) {
        // sum_j w_j * l(out_j - y_j) over this wave's 64 * VW samples -> one partial per (tile, tree, wave): outp = &partial[tile, 0, wave],
        // ldo = bytes between two trees' partials (weight 0: samples past N)
        T s = T(0);
        HL_BOTH(ly, lw);
        if (flags & HF_LOSS_PLAIN) { // every sample of the tile counts with weight 1
            FOR_PLANES {
                const typename VecOf<T>::type e = st.acc[g] - ly.v[g];
                if (flags & HF_LOSS_L1) {
                    T a = M<T>::abs(e[0]);
                    DE_UNROLL for (int i = 1; i < VW; i++) a += M<T>::abs(e[i]);
                    s += a;
                } else {
                    T q = e[0] * e[0];
                    DE_UNROLL for (int i = 1; i < VW; i++) q = M<T>::fma(e[i], e[i], q);
                    s += q;
                }
            }
        } else {
            FOR_PLANES DE_UNROLL for (int i = 0; i < VW; i++) {
                const T e = st.acc[g][i] - ly.v[g][i];
                const T l = (flags & HF_LOSS_L1) ? M<T>::abs(e) : e * e;
                s += lw.v[g][i] != T(0) ? lw.v[g][i] * l : T(0); // weight 0 really excludes the sample (0 * Inf would be NaN)
            }
        }
        const int lane = (int)(((lds0 - DE_SKIPLIST_BYTES) >> 4) & 63u); // (no work-item id input in a handler)
        s = wave_sum_to_lane63(s, lane);
        if (lane == 63) *reinterpret_cast<__attribute__((address_space(1))) T *>(row) = s;
        HTREE_END_TAIL(w, code_at(code, 1), code_at(code, -1));
    }
    if (flags & HF_SLOW_STORE) {
        FOR_PLANES {
            const int remaining = (int)(flags & HF_VALID_MASK) - (int)((lds0 - DE_SKIPLIST_BYTES + (uint32_t)g * DE_PLANE_BYTES) / (uint32_t)sizeof(T));
            DE_UNROLL for (int i = 0; i < VW; i++)
                if (i < remaining) reinterpret_cast<__attribute__((address_space(1))) T *>(row + lds0 + (uint32_t)g * DE_PLANE_BYTES)[i] = st.acc[g][i];
        }
    } else {
        FOR_PLANES if (st.acc[g][0] == T(123456.789)) *reinterpret_cast<__attribute__((address_space(1))) T *>(row + lds0) = st.acc[g][0];
    }
    HTREE_END_TAIL(w, code_at(code, 1), code_at(code, -1));
}
template <typename T> __device__ __noinline__ HState<T> h_tree_end(HCHAIN_ARGS) {
    typedef typename VecOf<T>::type V;
    if (__builtin_expect((int32_t)flags < 0, 0)) [[clang::musttail]] return h_tree_end_slow<T>(st, HL_PASS_C lds0, code, outp, la, w1, w23, okp, ldo, skip, left, flags);
    const U32x4 w = *code;
    const uint32_t tree = la; // this IS the end record
    const GPtr row = reinterpret_cast<GPtr>(outp + (uint64_t)tree * (uint64_t)(uint32_t)ldo); // wave-uniform: the store takes it as its scalar base
    FOR_PLANES DE_OUT_STORE(V, row + lds0 + (uint32_t)g * DE_PLANE_BYTES, st.acc[g]); // full tile, aligned rows
    HTREE_END_TAIL(w, code_at(code, 1), code_at(code, -1));
}
// The last instruction of a tree and its end in one dispatch (make_chained picks it when the tree finishes in a validity-tested
// hot operator: ~85 % of the bench population): the body, then what h_tree_end does.  The stream keeps its end record — this
// handler steps over it (code[0]) to the next tree's first record — and this instruction's record names the next tree's first
// handler, as the end record does.
template <typename T, BodyFn<T> BODY> __device__ __noinline__ HState<T> h_chain_end(HCHAIN_ARGS) {
    typedef typename VecOf<T>::type V;
    const uint32_t tree = *reinterpret_cast<const DE_CONSTANT uint32_t *>(code); // the end record this handler steps over: its operand word
    if (__builtin_expect((int32_t)flags < 0, 0)) {
        planes_apply<T, BODY>(st, lds0 + la, arg_imm<T>(w1, w23));
        [[clang::musttail]] return h_tree_end_slow<T>(st, HL_PASS_C lds0, code_at(code, 1), outp, tree, w1, w23, okp, ldo, skip, left, flags);
    }
    const U32x4 w = code[1];
    planes_apply<T, BODY>(st, lds0 + la, arg_imm<T>(w1, w23));
    const GPtr row = reinterpret_cast<GPtr>(outp + (uint64_t)tree * (uint64_t)(uint32_t)ldo);
    FOR_PLANES DE_OUT_STORE(V, row + lds0 + (uint32_t)g * DE_PLANE_BYTES, st.acc[g]);
    HTREE_END_TAIL(w, code_at(code, 2), code);
}

template <typename T> __device__ __forceinline__ BState<T> b_load_row(HARGS) { st.acc = *LDSP(T, la); return st; }
template <typename T> __device__ __forceinline__ BState<T> b_load_const(HARGS) {
    const T c = imm_from<T>(imm);
    DE_UNROLL for (int i = 0; i < VecOf<T>::W; i++) st.acc[i] = c;
    return st;
}
template <typename T> __device__ __forceinline__ BState<T> b_push(HARGS) { *LDSP(T, la) = st.acc; return st; }
template <typename T> __device__ __forceinline__ BState<T> b_check_row(HARGS) {
    const typename VecOf<T>::type v = *LDSP(T, la);
    hpoison<T>(st.poison, v);
    return st;
}
template <typename T> __device__ __forceinline__ BState<T> b_check_acc(HARGS) { hpoison<T>(st.poison, st.acc); return st; }

// Correctly rounded Float32 division, 4 samples.  The compiler's expansion of `/` is
//   v_div_scale x2, v_rcp, 6 dependent FMA/MUL (Newton + two residual corrections), v_div_fmas, v_div_fixup
// per ELEMENT (11 instructions, none packed): division is 1.6 of the ~9.4 dispatches of a bench tree
// and a quarter of the VALU time.  When every numerator and denominator of the wavefront lies in
// [2^-40, 2^40] the scale/fixup steps are the identity (v_div_scale only rescales for exponent
// differences >= 96, denormals, or tiny numerators), so the SAME arithmetic sequence is run two
// elements per v_pk_fma_f32 without them — bit-identical results, 38 issue slots instead of 56.
// Anything else in the wave (zeros, Inf, huge/tiny values; NaN is transparent to max/min and
// propagates through the FMAs) takes the generic expansion, under a wave-uniform branch.
// all eight operands of a wavefront's four divisions in [2^-40, 2^40]?  (all-NaN compares false: generic path)
__device__ __forceinline__ bool div_operands_safe(VecOf<float>::type a, VecOf<float>::type b) {
    // v_max3 / v_min3 on |.|, spelled out (fmaxf(fabsf(.)) costs an extra v_max_f32 |x|, |x| per raw operand: sNaN quieting)
    float hi, lo;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(hi) : "v"(a[0]), "v"(b[0]), "v"(a[1]));
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(hi) : "v"(hi), "v"(b[1]), "v"(a[2]));
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(hi) : "v"(hi), "v"(b[2]), "v"(a[3]));
    asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(hi) : "v"(hi), "v"(b[3]));
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(lo) : "v"(a[0]), "v"(b[0]), "v"(a[1]));
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(lo) : "v"(lo), "v"(b[1]), "v"(a[2]));
    asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(lo) : "v"(lo), "v"(b[2]), "v"(a[3]));
    asm("v_min_f32_e64 %0, %1, |%2|" : "=v"(lo) : "v"(lo), "v"(b[3]));
    return (hi < 0x1p+40f) & (lo > 0x1p-40f);
}
// the correctly rounded quotients of operands that passed div_operands_safe: reciprocal, one Newton step, ONE residual correction.
// (The compiler's expansion runs a second correction; round 3 dropped it.  Markstein's theorem: with y = RN(1/d) and q0 within an
// ulp of n/d, RN(q0 + (n - d q0) y) — the residual is exact in an FMA — IS RN(n/d).  tools/probe/div_probe.hip checked on gfx950 that
// the Newton-refined v_rcp_f32 equals RN(1/d) for ALL 2^23 significands, and the five-operation sequence against the IEEE quotient on
// 1.4e11 random pairs with exponents in [-40, 40] and 1.6e9 quotients placed at rounding boundaries: 0 differences,
// profiles/r3_div_probe.json.  tests/test_gpu_ops.py keeps the bit-identity test against IEEE division.)
__device__ __forceinline__ VecOf<float>::type div_safe(VecOf<float>::type a, VecOf<float>::type b) {
    VecOf<float>::type q;
    DE_UNROLL for (int h = 0; h < 2; h++) {
        const DeF2 n = {a[2 * h], a[2 * h + 1]}, d = {b[2 * h], b[2 * h + 1]};
        DeF2 y = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        // (the negations sit on the fresh temporaries y and t — free source modifiers; on `d` the compiler materialised -d of the second
        // pair with two v_xor_b32: its register had been reused.  -(d) * y == d * (-y) exactly: the same bits)
        const DeF2 e = __builtin_elementwise_fma(d, -y, DE_F2(1.0f));
        y = __builtin_elementwise_fma(e, y, y);
        DeF2 t = n * y;
        const DeF2 r = __builtin_elementwise_fma(d, -t, n);
        t = __builtin_elementwise_fma(r, y, t);
        q[2 * h] = t[0];
        q[2 * h + 1] = t[1];
    }
    return q;
}
// ... with a CONSTANT operand (22 % of the bench population's divisions are x / c, 15 % c / x): the constant is wave-uniform, so
// its range test is three scalar instructions on the exponent field and the vector test covers the four samples only; for x / c
// the reciprocal and its Newton step are computed ONCE per wavefront (one v_rcp_f32 instead of four, no packed Newton step) —
// the refined reciprocal is RN(1/c) for every significand (div_probe, above), the value the packed sequence computes per lane, so
// the quotients are the same bits.
__device__ __forceinline__ bool div_const_in_range(uint32_t bits) { return ((bits >> 23) & 0xFFu) - 88u < 79u; } // |c| in [2^-39, 2^40)
__device__ __forceinline__ bool div_samples_safe(VecOf<float>::type a) {
    float hi, lo;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(hi) : "v"(a[0]), "v"(a[1]), "v"(a[2]));
    asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(hi) : "v"(hi), "v"(a[3]));
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(lo) : "v"(a[0]), "v"(a[1]), "v"(a[2]));
    asm("v_min_f32_e64 %0, %1, |%2|" : "=v"(lo) : "v"(lo), "v"(a[3]));
    return (hi < 0x1p+40f) & (lo > 0x1p-40f);
}
__device__ __forceinline__ VecOf<float>::type div_safe_by_const(VecOf<float>::type a, float c) {
    float y1 = __builtin_amdgcn_rcpf(c);
    y1 = __builtin_fmaf(__builtin_fmaf(-c, y1, 1.0f), y1, y1);
    const DeF2 y = {y1, y1}, d = {c, c};
    VecOf<float>::type q;
    DE_UNROLL for (int h = 0; h < 2; h++) {
        const DeF2 n = {a[2 * h], a[2 * h + 1]};
        DeF2 t = n * y;
        const DeF2 r = __builtin_elementwise_fma(-d, t, n);
        t = __builtin_elementwise_fma(r, y, t);
        q[2 * h] = t[0];
        q[2 * h + 1] = t[1];
    }
    return q;
}
// the divisions of the fast handlers whose operand `b` is the constant `cbits` (K = 4: x / c, 5: c / x): range test, quotient
// "some lane of the wavefront": the ballot builtin on the predicate itself.  (`__ballot(p) != 0` combined with a SCALAR condition in one
// `if` made the compiler materialise the predicate as 0 / 1 in a vector register and compare it again — two more vector and half a
// dozen scalar instructions in every division by a constant; the scalar condition now has a branch of its own.)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
template <int K> __device__ __forceinline__ bool div_const_unsafe(VecOf<float>::type x, uint32_t cbits) { // (one plane: the full handlers' own test)
    return (int)(__ballot(!div_samples_safe(x)) != 0ull) | (int)!div_const_in_range(cbits);
}
template <int K> __device__ __forceinline__ VecOf<float>::type div_const(VecOf<float>::type x, uint32_t cbits) {
    const float c = __builtin_bit_cast(float, cbits);
    if constexpr (K == 4) return div_safe_by_const(x, c);
    else return div_safe(VecOf<float>::type{c, c, c, c}, x);
}
__device__ __forceinline__ VecOf<float>::type div_apply(VecOf<float>::type a, VecOf<float>::type b) {
    const bool safe = div_operands_safe(a, b);
    if (__ballot(!safe) != 0ull) return a / b;
    return div_safe(a, b);
}
__device__ __forceinline__ VecOf<double>::type div_apply(VecOf<double>::type a, VecOf<double>::type b) { return a / b; }
// K: 0 ADD 1 SUB 2 RSUB 3 MUL 4 DIV 5 RDIV  —  x op b (R*: b op x)
// TB = DE_OPT_TURBO: the relaxed-accuracy Float32 operators of de_device_ops.h (Float64 has none: same code as exact)
__device__ __forceinline__ VecOf<float>::type div_turbo(VecOf<float>::type a, VecOf<float>::type b) {
    const DeF2 lo = turbo_div_f32x2(DeF2{a[0], a[1]}, DeF2{b[0], b[1]}), hi = turbo_div_f32x2(DeF2{a[2], a[3]}, DeF2{b[2], b[3]});
    return VecOf<float>::type{lo[0], lo[1], hi[0], hi[1]};
}
template <typename T, int K, bool TB = false> __device__ __forceinline__ typename VecOf<T>::type bin_apply(typename VecOf<T>::type x, typename VecOf<T>::type b) {
    if constexpr (K == 0) return x + b;
    else if constexpr (K == 1) return x - b;
    else if constexpr (K == 2) return b - x;
    else if constexpr (K == 3) return x * b;
    else if constexpr (TB && sizeof(T) == 4) return K == 4 ? div_turbo(x, b) : div_turbo(b, x);
    else if constexpr (K == 4) return div_apply(x, b);
    else return div_apply(b, x);
}
template <typename T> __device__ __forceinline__ typename VecOf<T>::type splat(typename ImmBits<T>::type imm) {
    typename VecOf<T>::type b;
    const T c = imm_from<T>(imm);
    DE_UNROLL for (int i = 0; i < VecOf<T>::W; i++) b[i] = c;
    return b;
}
// K: 0 COS 1 EXP 2 SIN
template <typename T, int K, bool TB = false> __device__ __forceinline__ typename VecOf<T>::type un_apply(typename VecOf<T>::type x) {
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    V r;
    if constexpr (TB && sizeof(T) == 4) {
        const DeF2 xa = {x[0], x[1]}, xb = {x[2], x[3]};
        if constexpr (K == 1) {
            const DeF2 a = turbo_exp_f32x2(xa), b = turbo_exp_f32x2(xb);
            r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
        } else {
            DeF2 ma, mb;
            const DeF2 a = turbo_trig_f32x2<K == 2>(xa, ma), b = turbo_trig_f32x2<K == 2>(xb, mb);
            r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
            if (__ballot(any_abs_exceeds_f32x4(ma[0], ma[1], mb[0], mb[1], DE_TRIG_FAST_BOUND_M)) != 0ull) { // huge arguments, Inf: full range reduction (rare, wave-uniform)
                DE_UNROLL for (int i = 0; i < VW; i++)
                    if (fabsf(x[i]) > DE_TURBO_TRIG_BOUND) r[i] = K == 2 ? sinf(x[i]) : cosf(x[i]);
            }
        }
    } else if constexpr (sizeof(T) == 4) {
        if constexpr (K == 1) {
            // |x log2 e| <= 125.9: the result is a normal number and 2^RN(x log2 e) * (1 + e ln 2) (de_device_ops.h: one v_exp_f32,
            // no rint / clamp / ldexp) is as accurate as the ldexp formulation (1.3 vs 1.2 ulp).  Subnormal results (gradual
            // underflow), overflow and Inf take that one under a wave-uniform branch; NaN flows through either.
            const DeF2 xa = {x[0], x[1]}, xb = {x[2], x[3]};
            const DeF2 ta = xa * DE_F2(0x1.715476p+0f), tb = xb * DE_F2(0x1.715476p+0f);
            DeF2 a = turbo_exp_f32x2(xa, ta), b = turbo_exp_f32x2(xb, tb);
            r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
            if (__ballot(any_abs_exceeds_f32x4(ta[0], ta[1], tb[0], tb[1], DE_EXP_DIRECT_BOUND_T)) != 0ull) {
                // per-element select (not a plain overwrite): keeps the direct results above the branch and in the result registers
                a = fast_exp_f32x2(xa);
                b = fast_exp_f32x2(xb);
                r[0] = __builtin_fabsf(ta[0]) > DE_EXP_DIRECT_BOUND_T ? a[0] : r[0];
                r[1] = __builtin_fabsf(ta[1]) > DE_EXP_DIRECT_BOUND_T ? a[1] : r[1];
                r[2] = __builtin_fabsf(tb[0]) > DE_EXP_DIRECT_BOUND_T ? b[0] : r[2];
                r[3] = __builtin_fabsf(tb[1]) > DE_EXP_DIRECT_BOUND_T ? b[1] : r[3];
            }
        } else {
            const float xi[4] = {x[0], x[1], x[2], x[3]};
            float yo[4];
            const bool slow = fast_trig_f32x4<K == 2>(xi, yo);
            r[0] = yo[0]; r[1] = yo[1]; r[2] = yo[2]; r[3] = yo[3];
            // Inf and |x| > 1e5 take the slow path (the fast path propagates NaN: same result either way)
            if (slow) { // inline: a call here would turn every handler into a non-leaf function
                DE_UNROLL for (int i = 0; i < VW; i++)
                    if (fabsf(x[i]) > DE_TRIG_FAST_BOUND) r[i] = K == 2 ? sinf(x[i]) : cosf(x[i]);
            }
        }
    } else {
        DE_UNROLL for (int i = 0; i < VW; i++) r[i] = K == 0 ? M<T>::cos(x[i]) : (K == 1 ? M<T>::exp(x[i]) : M<T>::sin(x[i]));
    }
    return r;
}
// VAR bit0 = validity-test the result, bit1 = constant operand
template <typename T, int K, int VAR, bool TB = false> __device__ __forceinline__ BState<T> b_bin(HARGS) {
    typedef typename VecOf<T>::type V;
    V b;
    if constexpr (VAR & 2) b = splat<T>(imm);
    else b = *LDSP(T, la);
    st.acc = bin_apply<T, K, TB>(st.acc, b);
    if constexpr (VAR & 1) hpoison<T>(st.poison, st.acc);
    return st;
}
// VAR bit0 = test the result, bit1 = operand is an LDS row (else acc)
template <typename T, int K, int VAR, bool TB = false> __device__ __forceinline__ BState<T> b_un(HARGS) {
    typedef typename VecOf<T>::type V;
    V x = st.acc;
    if constexpr (VAR & 2) x = *LDSP(T, la);
    st.acc = un_apply<T, K, TB>(x);
    if constexpr (VAR & 1) hpoison<T>(st.poison, st.acc);
    return st;
}
// ---- superinstructions (de_bind.h, fuse_tree): la = LDS address of row A | int8 (push row - row A) << 24
__device__ __forceinline__ uint32_t row_a(uint32_t la) { return la & 0xFFFFFFu; }
template <typename T> __device__ __forceinline__ uint32_t push_addr(uint32_t la) { return (la & 0xFFFFFFu) + (uint32_t)(((int32_t)la >> 24) * (int32_t)RowOf<T>::BYTES); }
template <typename T, bool PUSH, bool CHK> __device__ __forceinline__ BState<T> b_loadrow_f(HARGS) {
    if constexpr (PUSH) *LDSP(T, push_addr<T>(la)) = st.acc;
    const typename VecOf<T>::type v = *LDSP(T, PUSH ? row_a(la) : la);
    if constexpr (CHK) hpoison<T>(st.poison, v);
    st.acc = v;
    return st;
}
template <typename T> __device__ __forceinline__ BState<T> b_loadconst_push(HARGS) {
    *LDSP(T, la) = st.acc;
    st.acc = splat<T>(imm);
    return st;
}
template <typename T, int K, bool OUT, bool PUSH, bool CHK, bool TB = false> __device__ __forceinline__ BState<T> b_unrow_f(HARGS) {
    if constexpr (PUSH) *LDSP(T, push_addr<T>(la)) = st.acc;
    const typename VecOf<T>::type x = *LDSP(T, PUSH ? row_a(la) : la);
    if constexpr (CHK) hpoison<T>(st.poison, x);
    st.acc = un_apply<T, K, TB>(x);
    if constexpr (OUT) hpoison<T>(st.poison, st.acc);
    return st;
}
template <typename T, int K, bool OUT, bool TB = false> __device__ __forceinline__ BState<T> b_binrowc(HARGS) { // operand row tested, then acc = acc op row
    const typename VecOf<T>::type b = *LDSP(T, la);
    hpoison<T>(st.poison, b);
    st.acc = bin_apply<T, K, TB>(st.acc, b);
    if constexpr (OUT) hpoison<T>(st.poison, st.acc);
    return st;
}
// acc = row A op (row B | constant); row B's byte distance from row A travels in the immediate
template <typename T, int K, bool CST, bool OUT, bool PUSH, bool TB = false> __device__ __forceinline__ BState<T> b_bin2(HARGS) {
    typedef typename VecOf<T>::type V;
    if constexpr (PUSH) *LDSP(T, push_addr<T>(la)) = st.acc;
    const uint32_t a = PUSH ? row_a(la) : la;
    const V x = *LDSP(T, a);
    V b;
    if constexpr (CST) b = splat<T>(imm);
    else b = *LDSP(T, a + (uint32_t)imm);
    st.acc = bin_apply<T, K, TB>(x, b);
    if constexpr (OUT) hpoison<T>(st.poison, st.acc);
    return st;
}
// ---- FAST-PATH-ONLY handlers (Float32 cos / exp / sin, exact division) ------------------------------------------------------
// The wave-uniform range test comes first and a wavefront that fails it tail-calls the FULL handler of the same instruction (same
// arguments, nothing modified yet; a spill it may already have written is written again with the same value), which repeats the
// test and resolves it per element.  With the slow paths (OCML's Payne-Hanek reduction, the ldexp form of exp, the generic IEEE
// division) out of the function the argument is dead after its last use and the result is computed in place: no v_mov of the
// accumulator around the body (4-5 of them before: 6-10 % of the handler).  Two phases, so that every handler shape (plain,
// fused with a spill / a row test, fused with the end of the tree) can place its tail call between them:
//   un_pretest: the part of the arithmetic the range test reads (x log2 e | the multiple of pi), and the test — PER LANE: the caller
//               ORs the planes' results and takes ONE ballot;
//   un_finish : the rest.  Same operations in the same order as un_apply's fast paths: the same bits.
struct UnPre { DeF2 ta, tb, ka, kb; }; // exp: ta, tb = x log2 e;  trig: ta, tb = the multiple n, ka, kb = the magic sums (parity)
template <int K, bool TB> __device__ __forceinline__ bool un_pretest(VecOf<float>::type x, UnPre &p) {
    const DeF2 xa = {x[0], x[1]}, xb = {x[2], x[3]};
    if constexpr (K == 1) {
        p.ta = xa * DE_F2(0x1.715476p+0f);
        p.tb = xb * DE_F2(0x1.715476p+0f);
        if constexpr (TB) return false; // turbo exp has no slow path
        else return any_abs_exceeds_f32x4(p.ta[0], p.ta[1], p.tb[0], p.tb[1], DE_EXP_DIRECT_BOUND_T);
    } else {
        constexpr bool SIN = K == 2;
        const DeF2 ta = SIN ? xa * DE_F2(DE_TRIG_INV_PI) : __builtin_elementwise_fma(xa, DE_F2(DE_TRIG_INV_PI), DE_F2(0.5f));
        const DeF2 tb = SIN ? xb * DE_F2(DE_TRIG_INV_PI) : __builtin_elementwise_fma(xb, DE_F2(DE_TRIG_INV_PI), DE_F2(0.5f));
        p.ka = ta + DE_F2(DE_TRIG_MAGIC);
        p.kb = tb + DE_F2(DE_TRIG_MAGIC);
        p.ta = p.ka - DE_F2(DE_TRIG_MAGIC);
        p.tb = p.kb - DE_F2(DE_TRIG_MAGIC);
        return any_abs_exceeds_f32x4(p.ta[0], p.ta[1], p.tb[0], p.tb[1], DE_TRIG_FAST_BOUND_M);
    }
}
template <int K, bool TB> __device__ __forceinline__ VecOf<float>::type un_finish(VecOf<float>::type x, const UnPre &p) {
    typedef VecOf<float>::type V;
    const DeF2 xa = {x[0], x[1]}, xb = {x[2], x[3]};
    if constexpr (K == 1) {
        const DeF2 a = turbo_exp_f32x2(xa, p.ta), b = turbo_exp_f32x2(xb, p.tb);
        return V{a[0], a[1], b[0], b[1]};
    } else {
        constexpr bool SIN = K == 2;
        DeF2 sa = trig_poly_f32x2<SIN, TB>(xa, p.ta), sb = trig_poly_f32x2<SIN, TB>(xb, p.tb);
        if constexpr (!TB) { // exact mode: results within 2^-12 of +-pi/2 are exactly +-1 (de_device_ops.h)
#ifndef DE_TRIG_NO_EXTREMUM_FIX
            const DeF2 ra = trig_reduced_f32x2<SIN, TB>(xa, p.ta), rb = trig_reduced_f32x2<SIN, TB>(xb, p.tb); // (common subexpressions of trig_poly)
            const DeF2 za = ra * ra, zb = rb * rb;
            const bool near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(za[0], za[1]), zb[0]), zb[1]) > (0x1.3bd3ccp+1f - 8.0e-4f);
            if (__builtin_expect(__ballot(near) != 0ull, 0)) {
                asm volatile("" ::: "memory"); // a REAL branch: if-converted (the fix computed for every wave, then selected), cos / sin cost 59 instead of 37 VALU instructions
                sa[0] = trig_extremum_fix(ra[0], sa[0]); sa[1] = trig_extremum_fix(ra[1], sa[1]);
                sb[0] = trig_extremum_fix(rb[0], sb[0]); sb[1] = trig_extremum_fix(rb[1], sb[1]);
            }
#endif
        }
        const DeF2 ya = fast_trig_sign_f32x2(sa, p.ka), yb = fast_trig_sign_f32x2(sb, p.kb);
        return V{ya[0], ya[1], yb[0], yb[1]};
    }
}
// ... for ALL planes of a dispatch (two planes, exact mode): both planes' polynomials first, ONE wave-uniform extremum decision for the
// dispatch (per plane it would be a branch between the planes' instruction streams: the scheduler could not interleave them), then the
// signs.  The same operations per element as un_finish: the same bits.
template <int K, bool TB, int G> __device__ __forceinline__ void un_finish_planes(VecOf<float>::type (&out)[G], const VecOf<float>::type (&x)[G], const UnPre (&p)[G]) {
    typedef VecOf<float>::type V;
    if constexpr (K == 1 || TB || G == 1) {
        DE_UNROLL for (int g = 0; g < G; g++) out[g] = un_finish<K, TB>(x[g], p[g]);
    } else {
        constexpr bool SIN = K == 2;
        DeF2 sa[G], sb[G];
        DE_UNROLL for (int g = 0; g < G; g++) {
            sa[g] = trig_poly_f32x2<SIN, TB>(DeF2{x[g][0], x[g][1]}, p[g].ta);
            sb[g] = trig_poly_f32x2<SIN, TB>(DeF2{x[g][2], x[g][3]}, p[g].tb);
        }
#ifndef DE_TRIG_NO_EXTREMUM_FIX
        DeF2 ra[G], rb[G];
        bool near = false;
        DE_UNROLL for (int g = 0; g < G; g++) {
            ra[g] = trig_reduced_f32x2<SIN, TB>(DeF2{x[g][0], x[g][1]}, p[g].ta);
            rb[g] = trig_reduced_f32x2<SIN, TB>(DeF2{x[g][2], x[g][3]}, p[g].tb);
            const DeF2 za = ra[g] * ra[g], zb = rb[g] * rb[g];
            near |= __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(za[0], za[1]), zb[0]), zb[1]) > (0x1.3bd3ccp+1f - 8.0e-4f);
        }
        if (__builtin_expect(__ballot(near) != 0ull, 0)) {
            asm volatile("" ::: "memory"); // a REAL branch (see un_finish)
            DE_UNROLL for (int g = 0; g < G; g++) {
                sa[g][0] = trig_extremum_fix(ra[g][0], sa[g][0]); sa[g][1] = trig_extremum_fix(ra[g][1], sa[g][1]);
                sb[g][0] = trig_extremum_fix(rb[g][0], sb[g][0]); sb[g][1] = trig_extremum_fix(rb[g][1], sb[g][1]);
            }
        }
#endif
        DE_UNROLL for (int g = 0; g < G; g++) {
            const DeF2 ya = fast_trig_sign_f32x2(sa[g], p[g].ka), yb = fast_trig_sign_f32x2(sb[g], p[g].kb);
            out[g] = V{ya[0], ya[1], yb[0], yb[1]};
        }
    }
}
#if DE_NO_LOSS
#define HFAST_LOSS_C
#elif DE_TG == 1
#define HFAST_LOSS_C VecOf<float>::type ly0, VecOf<float>::type lw0,
#else
#define HFAST_LOSS_C VecOf<float>::type ly0, VecOf<float>::type ly1, VecOf<float>::type lw0, VecOf<float>::type lw1,
#endif
#define HFAST_ARGS HState<float> st, HFAST_LOSS_C uint32_t lds0, ConstU4Ptr code, uint64_t outp, uint32_t la, uint32_t w1, uint64_t w23, uint64_t okp, uint64_t ldo, \
                   uint64_t skip, uint32_t left, uint32_t flags
#define HFAST_PASS st, HL_PASS_C lds0, code, outp, la, w1, w23, okp, ldo, skip, left, flags
#define HFAST_NEXT(W) [[clang::musttail]] return arg_next<float>(w1, w23)(st, HL_PASS_C lds0, code_at(code, 1), outp, (W).x, (W).y, ((uint64_t)(W).w << 32) | (W).z, okp, ldo, skip, left, flags)
// the end of a tree behind a fast-path body: what h_chain_end does (T = float)
#define HFAST_END_TAIL()                                                                                                    \
    {                                                                                                                       \
        const U32x4 wn = code[1];                                                                                           \
        const uint32_t tree = *reinterpret_cast<const DE_CONSTANT uint32_t *>(code); /* the end record's operand word */      \
        const GPtr row = reinterpret_cast<GPtr>(outp + (uint64_t)tree * (uint64_t)(uint32_t)ldo);                                               \
        FOR_PLANES DE_OUT_STORE(VecOf<float>::type, row + lds0 + (uint32_t)g * DE_PLANE_BYTES, st.acc[g]); \
        HTREE_END_TAIL(wn, code_at(code, 2), code);                                                                                     \
    }
#define PLANE_ADDR(A, g) ((A) + (uint32_t)(g) * DE_PLANE_BYTES)
// (the planes of a fast handler: every plane's range test first — one wave-uniform decision for the whole dispatch: a wavefront
// that fails it on ANY plane tail-calls the full handler, which redoes all planes —, then every plane's arithmetic)
// cos / exp / sin on the accumulator or a row (VAR as in b_un)
template <int K, int VAR, bool TB> __device__ __noinline__ HState<float> h_un_fast(HFAST_ARGS) {
    typedef float T;
    constexpr int G = TG<T>::G;
    const U32x4 w = *code;
    VecOf<float>::type x[G];
    FOR_PLANES x[g] = (VAR & 2) ? *LDSP(T, PLANE_ADDR(lds0 + la, g)) : st.acc[g];
    UnPre p[G];
    bool slow = false;
    FOR_PLANES slow |= un_pretest<K, TB>(x[g], p[g]);
    if (__builtin_expect(__ballot(slow) != 0ull, 0)) [[clang::musttail]] return h_chain<T, &b_un<T, K, VAR, TB>>(HFAST_PASS);
#if DE_TG == 1
    FOR_PLANES {
        st.acc[g] = un_finish<K, TB>(x[g], p[g]);
        if constexpr (VAR & 1) hpoison<T>(st.poison, st.acc[g]);
    }
#else
    un_finish_planes<K, TB, G>(st.acc, x, p);
    if constexpr (VAR & 1) FOR_PLANES hpoison<T>(st.poison, st.acc[g]);
#endif
    HFAST_NEXT(w);
}
// ... as the last instruction of a tree (the end-fused form of b_un<K, 1>: accumulator operand, tested result)
template <int K, bool TB> __device__ __noinline__ HState<float> h_un_end_fast(HFAST_ARGS) {
    typedef float T;
    constexpr int G = TG<T>::G;
    UnPre p[G];
    bool slow = false;
    FOR_PLANES slow |= un_pretest<K, TB>(st.acc[g], p[g]);
    if (__builtin_expect((int32_t)flags < 0, 0)) [[clang::musttail]] return h_chain_end<T, &b_un<T, K, 1, TB>>(HFAST_PASS);
    if (__builtin_expect(wave_any(slow), 0)) [[clang::musttail]] return h_chain_end<T, &b_un<T, K, 1, TB>>(HFAST_PASS);
#if DE_TG == 1
    FOR_PLANES {
        st.acc[g] = un_finish<K, TB>(st.acc[g], p[g]);
        hpoison<T>(st.poison, st.acc[g]);
    }
#else
    un_finish_planes<K, TB, G>(st.acc, st.acc, p);
    FOR_PLANES hpoison<T>(st.poison, st.acc[g]);
#endif
    HFAST_END_TAIL()
}
// ... fused with a spill of the accumulator and / or the validity test of its row operand (b_unrow_f)
template <int K, bool OUT, bool PUSH, bool CHK, bool TB> __device__ __noinline__ HState<float> h_unrow_fast(HFAST_ARGS) {
    typedef float T;
    constexpr int G = TG<T>::G;
    const U32x4 w = *code;
    const uint32_t a = lds0 + la;
    VecOf<float>::type x[G];
    FOR_PLANES {
        if constexpr (PUSH) *LDSP(T, push_addr<T>(PLANE_ADDR(a, g))) = st.acc[g]; // first, as in b_unrow_f (the full handler would write it again: same value)
        x[g] = *LDSP(T, PUSH ? row_a(PLANE_ADDR(a, g)) : PLANE_ADDR(a, g));
    }
    UnPre p[G];
    bool slow = false;
    FOR_PLANES slow |= un_pretest<K, TB>(x[g], p[g]);
    if (__builtin_expect(__ballot(slow) != 0ull, 0)) [[clang::musttail]] return h_chain<T, &b_unrow_f<T, K, OUT, PUSH, CHK, TB>>(HFAST_PASS);
#if DE_TG == 1
    FOR_PLANES {
        if constexpr (CHK) hpoison<T>(st.poison, x[g]);
        st.acc[g] = un_finish<K, TB>(x[g], p[g]);
        if constexpr (OUT) hpoison<T>(st.poison, st.acc[g]);
    }
#else
    if constexpr (CHK) FOR_PLANES hpoison<T>(st.poison, x[g]);
    un_finish_planes<K, TB, G>(st.acc, x, p);
    if constexpr (OUT) FOR_PLANES hpoison<T>(st.poison, st.acc[g]);
#endif
    HFAST_NEXT(w);
}
// The exact Float32 divisions (K = 4: acc / operand, 5: operand / acc; VAR as in b_bin) likewise: range test, then either the
// tail call into the full handler (generic IEEE expansion) or the packed fast path computed in place.
template <int K, int VAR> __device__ __noinline__ HState<float> h_div_fast(HFAST_ARGS) {
    typedef float T;
    typedef VecOf<float>::type V;
    constexpr int G = TG<T>::G;
    const U32x4 w = *code;
    if constexpr (VAR & 2) { // constant operand
        bool unsafe = false;
        FOR_PLANES unsafe |= !div_samples_safe(st.acc[g]);
        if (__builtin_expect(!div_const_in_range(w1), 0)) [[clang::musttail]] return h_chain<T, &b_bin<T, K, VAR, false>>(HFAST_PASS);
        if (__builtin_expect(wave_any(unsafe), 0)) [[clang::musttail]] return h_chain<T, &b_bin<T, K, VAR, false>>(HFAST_PASS);
        FOR_PLANES st.acc[g] = div_const<K>(st.acc[g], w1);
    } else {
        V num[G], den[G];
        bool unsafe = false;
        FOR_PLANES {
            const V b = *LDSP(T, PLANE_ADDR(lds0 + la, g));
            num[g] = K == 4 ? st.acc[g] : b;
            den[g] = K == 4 ? b : st.acc[g];
            unsafe |= !div_operands_safe(num[g], den[g]);
        }
        if (__builtin_expect(__ballot(unsafe) != 0ull, 0)) [[clang::musttail]] return h_chain<T, &b_bin<T, K, VAR, false>>(HFAST_PASS);
        FOR_PLANES st.acc[g] = div_safe(num[g], den[g]);
    }
    if constexpr (VAR & 1) FOR_PLANES hpoison<T>(st.poison, st.acc[g]);
    HFAST_NEXT(w);
}
// ... as the last instruction of a tree (the end-fused forms of b_bin<K, 1> / <K, 3>: tested result)
template <int K, bool CST> __device__ __noinline__ HState<float> h_div_end_fast(HFAST_ARGS) {
    typedef float T;
    typedef VecOf<float>::type V;
    constexpr int G = TG<T>::G;
    if constexpr (CST) {
        bool unsafe = false;
        FOR_PLANES unsafe |= !div_samples_safe(st.acc[g]);
        if (__builtin_expect(((int32_t)flags < 0) | !div_const_in_range(w1), 0)) [[clang::musttail]] return h_chain_end<T, &b_bin<T, K, 3, false>>(HFAST_PASS);
        if (__builtin_expect(wave_any(unsafe), 0)) [[clang::musttail]] return h_chain_end<T, &b_bin<T, K, 3, false>>(HFAST_PASS);
        FOR_PLANES st.acc[g] = div_const<K>(st.acc[g], w1);
    } else {
        V num[G], den[G];
        bool unsafe = false;
        FOR_PLANES {
            const V b = *LDSP(T, PLANE_ADDR(lds0 + la, g));
            num[g] = K == 4 ? st.acc[g] : b;
            den[g] = K == 4 ? b : st.acc[g];
            unsafe |= !div_operands_safe(num[g], den[g]);
        }
        if (__builtin_expect((int32_t)flags < 0, 0)) [[clang::musttail]] return h_chain_end<T, &b_bin<T, K, 1, false>>(HFAST_PASS);
        if (__builtin_expect(wave_any(unsafe), 0)) [[clang::musttail]] return h_chain_end<T, &b_bin<T, K, 1, false>>(HFAST_PASS);
        FOR_PLANES st.acc[g] = div_safe(num[g], den[g]);
    }
    FOR_PLANES hpoison<T>(st.poison, st.acc[g]);
    HFAST_END_TAIL()
}
// ... fused with the validity test of the row operand (b_binrowc) / with the load of row A, a constant or row B as the other
// operand and possibly a spill (b_bin2): the exact divisions among the superinstructions
template <int K, bool OUT> __device__ __noinline__ HState<float> h_divrowc_fast(HFAST_ARGS) {
    typedef float T;
    typedef VecOf<float>::type V;
    constexpr int G = TG<T>::G;
    const U32x4 w = *code;
    V b[G], num[G], den[G];
    bool unsafe = false;
    FOR_PLANES {
        b[g] = *LDSP(T, PLANE_ADDR(lds0 + la, g));
        num[g] = K == 4 ? st.acc[g] : b[g];
        den[g] = K == 4 ? b[g] : st.acc[g];
        unsafe |= !div_operands_safe(num[g], den[g]);
    }
    if (__builtin_expect(__ballot(unsafe) != 0ull, 0)) [[clang::musttail]] return h_chain<T, &b_binrowc<T, K, OUT, false>>(HFAST_PASS);
    FOR_PLANES {
        hpoison<T>(st.poison, b[g]);
        st.acc[g] = div_safe(num[g], den[g]);
        if constexpr (OUT) hpoison<T>(st.poison, st.acc[g]);
    }
    HFAST_NEXT(w);
}
template <int K, bool CST, bool OUT, bool PUSH> __device__ __noinline__ HState<float> h_div2_fast(HFAST_ARGS) {
    typedef float T;
    typedef VecOf<float>::type V;
    constexpr int G = TG<T>::G;
    const U32x4 w = *code;
    const uint32_t a0 = lds0 + la;
    V x[G];
    FOR_PLANES {
        const uint32_t ag = PLANE_ADDR(a0, g);
        if constexpr (PUSH) *LDSP(T, push_addr<T>(ag)) = st.acc[g]; // first, as in b_bin2: row B may be this very slot (the full handler would write it again: same value)
        x[g] = *LDSP(T, PUSH ? row_a(ag) : ag);
    }
    if constexpr (CST) {
        bool unsafe = false;
        FOR_PLANES unsafe |= !div_samples_safe(x[g]);
        if (__builtin_expect(!div_const_in_range(w1), 0)) [[clang::musttail]] return h_chain<T, &b_bin2<T, K, CST, OUT, PUSH, false>>(HFAST_PASS);
        if (__builtin_expect(wave_any(unsafe), 0)) [[clang::musttail]] return h_chain<T, &b_bin2<T, K, CST, OUT, PUSH, false>>(HFAST_PASS);
        FOR_PLANES st.acc[g] = div_const<K>(x[g], w1);
    } else {
        V num[G], den[G];
        bool unsafe = false;
        FOR_PLANES {
            const uint32_t ag = PLANE_ADDR(a0, g);
            const V b = *LDSP(T, (PUSH ? row_a(ag) : ag) + w1);
            num[g] = K == 4 ? x[g] : b;
            den[g] = K == 4 ? b : x[g];
            unsafe |= !div_operands_safe(num[g], den[g]);
        }
        if (__builtin_expect(__ballot(unsafe) != 0ull, 0)) [[clang::musttail]] return h_chain<T, &b_bin2<T, K, CST, OUT, PUSH, false>>(HFAST_PASS);
        FOR_PLANES st.acc[g] = div_safe(num[g], den[g]);
    }
    if constexpr (OUT) FOR_PLANES hpoison<T>(st.poison, st.acc[g]);
    HFAST_NEXT(w);
}
// generic handlers: de_opcode in la[31:24].  SRC: 0 row, 1 const, 2 acc.  INJ: Inf-injection of the fused deg1 kernels.
template <typename T, int SRC, bool INJ> __device__ __forceinline__ BState<T> b_gen(HARGS) {
    typedef typename VecOf<T>::type V;
    const uint32_t aux = la >> 24;
    VG<T, 1> a, b;
    a.v[0] = st.acc;
    if constexpr (SRC == 0) b.v[0] = *LDSP(T, la & 0xFFFFFFu);
    else if constexpr (SRC == 1) { const T c = imm_from<T>(imm); DE_UNROLL for (int i = 0; i < VecOf<T>::W; i++) b.v[0][i] = c; }
    else b.v[0] = st.acc;
    const V in = b.v[0];
    a = cold_op<T, 1>(aux, a, b);
    st.acc = a.v[0];
    if constexpr (INJ) { DE_UNROLL for (int i = 0; i < VecOf<T>::W; i++) st.acc[i] = M<T>::isfinite(in[i]) ? st.acc[i] : M<T>::inf(); }
    return st;
}
// Unary operators outside the binder's hot set, without the detour through the generic switch (cold_op): K = gun_index
// (de_bind.h) 3 neg 4 square 5 cube 6 abs 7 log 8 safe_log 9 sqrt 10 safe_sqrt 11 tanh 12 relu — cold_op's expressions.
template <typename T, int K, int SRC> __device__ __forceinline__ BState<T> b_un2(HARGS) { // SRC 0: row, 1: accumulator
    using m = M<T>;
    typename VecOf<T>::type x = st.acc;
    if constexpr (SRC == 0) x = *LDSP(T, la & 0xFFFFFFu);
    DE_UNROLL for (int i = 0; i < VecOf<T>::W; i++) {
        const T v = x[i];
        T r;
        if constexpr (K == 3) r = -v;
        else if constexpr (K == 4) r = v * v;
        else if constexpr (K == 5) r = (v * v) * v;
        else if constexpr (K == 6) r = m::abs(v);
        else if constexpr (K == 7) r = m::log(v);
        else if constexpr (K == 8) r = v <= T(0) ? m::nan() : m::log(v);
        else if constexpr (K == 9) r = m::sqrt(v);
        else if constexpr (K == 10) r = v < T(0) ? m::nan() : m::sqrt(v);
        else if constexpr (K == 11) r = m::tanh(v);
        else r = v < T(0) ? T(0) : v;
        st.acc[i] = r;
    }
    return st;
}
// acc = max(acc, b) / min(acc, b) (K 6 / 7; cold_op's expressions) with b = a row (SRC 0) or a constant (SRC 1)
template <typename T, int K, int SRC> __device__ __forceinline__ BState<T> b_maxmin(HARGS) {
    typename VecOf<T>::type b;
    if constexpr (SRC == 0) b = *LDSP(T, la & 0xFFFFFFu);
    else { const T c = imm_from<T>(imm); DE_UNROLL for (int i = 0; i < VecOf<T>::W; i++) b[i] = c; }
    DE_UNROLL for (int i = 0; i < VecOf<T>::W; i++) st.acc[i] = K == 6 ? jl_max(st.acc[i], b[i]) : jl_min(st.acc[i], b[i]);
    return st;
}
template <typename T> __device__ __forceinline__ BState<T> b_tern(HARGS) { // acc = op3(row B, row C, acc)
    const uint32_t aux = la >> 24, lb = la & 0xFFFFFFu;
    VG<T, 1> a, b, c;
    a.v[0] = st.acc;
    b.v[0] = *LDSP(T, lb);
    c.v[0] = *LDSP(T, lb + (uint32_t)imm); // imm = byte distance from row B to row C
    a = cold_op3<T, 1>(aux, a, b, c);
    st.acc = a.v[0];
    return st;
}
template <typename T> __device__ __forceinline__ BState<T> b_nop(HARGS) { return st; }

// Operand = params[row, class of the sample] (src/ParametricExpression.jl:381-389).  The kernel leaves, per thread, the
// byte offsets of its samples' parameter columns in an LDS row of their own (the "class row", whose byte offset is the
// record's immediate) and the table's address behind them (Float64: in the same 16-byte vector; Float32: in the row
// behind), so the handler needs no kernel argument: 4 (2) gathers through the vector cache.  la[23] = validity-test the
// operand, la[31:24] = DOP_LOAD or the operator applied to (acc, operand) — the hot ones inline, the rest through cold_op.
template <typename T, bool TB> __device__ __noinline__ HState<T> h_param(HCHAIN_ARGS) {
    constexpr int VW = VecOf<T>::W;
    const U32x4 w = *code;
    const uint32_t op = la >> 24;
    FOR_PLANES { // (the class row has the planes of every row; the table's address is read once per plane: same value)
        const uint32_t crow = lds0 + (uint32_t)arg_imm<T>(w1, w23) + (uint32_t)g * DE_PLANE_BYTES;
        const U32x4 cv = *reinterpret_cast<__attribute__((address_space(3))) U32x4 *>((uintptr_t)crow);
        uint64_t tab;
        if constexpr (sizeof(T) == 4) {
            typedef uint32_t U2 __attribute__((ext_vector_type(2)));
            const U2 pv = *reinterpret_cast<__attribute__((address_space(3))) U2 *>((uintptr_t)(crow + RowOf<T>::BYTES));
            tab = ((uint64_t)pv.y << 32) | pv.x;
        } else tab = ((uint64_t)cv.w << 32) | cv.z;
        const char *__restrict__ pb = reinterpret_cast<const char *>(tab) + (size_t)(la & 0xFFFFu) * sizeof(T);
        VG<T, 1> av, bv;
        DE_UNROLL for (int i = 0; i < VW; i++) bv.v[0][i] = *reinterpret_cast<const T *>(pb + cv[i]);
        if (la & (1u << 23)) hpoison<T>(st.poison, bv.v[0]);
        switch (op) {
        case DOP_LOAD: st.acc[g] = bv.v[0]; break;
        case DE_B_ADD: st.acc[g] = bin_apply<T, 0>(st.acc[g], bv.v[0]); break;
        case DE_B_SUB: st.acc[g] = bin_apply<T, 1>(st.acc[g], bv.v[0]); break;
        case DOP_RSUB: st.acc[g] = bin_apply<T, 2>(st.acc[g], bv.v[0]); break;
        case DE_B_MUL: st.acc[g] = bin_apply<T, 3>(st.acc[g], bv.v[0]); break;
        case DE_B_DIV: st.acc[g] = bin_apply<T, 4, TB>(st.acc[g], bv.v[0]); break;
        case DOP_RDIV: st.acc[g] = bin_apply<T, 5, TB>(st.acc[g], bv.v[0]); break;
        case DE_U_COS: st.acc[g] = un_apply<T, 0, TB>(bv.v[0]); break; // unary operator on a parameter leaf
        case DE_U_EXP: st.acc[g] = un_apply<T, 1, TB>(bv.v[0]); break;
        case DE_U_SIN: st.acc[g] = un_apply<T, 2, TB>(bv.v[0]); break;
        default: av.v[0] = st.acc[g]; av = cold_op<T, 1>(op, av, bv); st.acc[g] = av.v[0]; break;
        }
    }
    HCHAIN_NEXT(w);
}

// TB = handlers of a DE_OPT_TURBO program: same ids, the division / cos / exp / sin handlers are the relaxed-accuracy
// instantiations (operators without a turbo version share the exact instantiation: TBK)
template <typename T, bool TB> __global__ void de_fill_handlers(uint64_t *t) {
#define TBK(K) (TB && (K) >= 4)
#define HB(K) t[BOP_BIN_BASE + 4 * K + 0] = (uint64_t)&h_chain<T, &b_bin<T, K, 0, TBK(K)>>; t[BOP_BIN_BASE + 4 * K + 1] = (uint64_t)&h_chain<T, &b_bin<T, K, 1, TBK(K)>>; \
              t[BOP_BIN_BASE + 4 * K + 2] = (uint64_t)&h_chain<T, &b_bin<T, K, 2, TBK(K)>>; t[BOP_BIN_BASE + 4 * K + 3] = (uint64_t)&h_chain<T, &b_bin<T, K, 3, TBK(K)>>;
#define HU(K) t[BOP_UN_BASE + 4 * K + 0] = (uint64_t)&h_chain<T, &b_un<T, K, 0, TB>>; t[BOP_UN_BASE + 4 * K + 1] = (uint64_t)&h_chain<T, &b_un<T, K, 1, TB>>; \
              t[BOP_UN_BASE + 4 * K + 2] = (uint64_t)&h_chain<T, &b_un<T, K, 2, TB>>; t[BOP_UN_BASE + 4 * K + 3] = (uint64_t)&h_chain<T, &b_un<T, K, 3, TB>>;
    t[BOP_LOAD_ROW] = (uint64_t)&h_chain<T, &b_load_row<T>>;
    t[BOP_LOAD_CONST] = (uint64_t)&h_chain<T, &b_load_const<T>>;
    t[BOP_PUSH] = (uint64_t)&h_chain<T, &b_push<T>>;
    t[BOP_CHECK_ROW] = (uint64_t)&h_chain<T, &b_check_row<T>>;
    t[BOP_CHECK_ACC] = (uint64_t)&h_chain<T, &b_check_acc<T>>;
    HB(0) HB(1) HB(2) HB(3) HB(4) HB(5)
    HU(0) HU(1) HU(2)
    t[BOP_GEN_ROW] = (uint64_t)&h_chain<T, &b_gen<T, 0, false>>;
    t[BOP_GEN_CONST] = (uint64_t)&h_chain<T, &b_gen<T, 1, false>>;
    t[BOP_GEN_ACC] = (uint64_t)&h_chain<T, &b_gen<T, 2, false>>;
    t[BOP_GEN_PARAM] = (uint64_t)&h_param<T, TB>;
    t[TOPX_END] = (uint64_t)&h_tree_end<T>;
    t[TOPX_AUX_BASE + 0] = (uint64_t)&h_tree_end_slow<T>;
    t[TOPX_AUX_BASE + 1] = (uint64_t)&h_tree_skip<T>;
#define HEB(K) t[TOPX_ENDV_BASE + K * 2] = (uint64_t)&h_chain_end<T, &b_bin<T, K, 1, TBK(K)>>; t[TOPX_ENDV_BASE + K * 2 + 1] = (uint64_t)&h_chain_end<T, &b_bin<T, K, 3, TBK(K)>>;
    HEB(0) HEB(1) HEB(2) HEB(3) HEB(4) HEB(5)
#undef HEB
    t[TOPX_ENDV_BASE + 12] = (uint64_t)&h_chain_end<T, &b_un<T, 0, 1, TB>>;
    t[TOPX_ENDV_BASE + 13] = (uint64_t)&h_chain_end<T, &b_un<T, 1, 1, TB>>;
    t[TOPX_ENDV_BASE + 14] = (uint64_t)&h_chain_end<T, &b_un<T, 2, 1, TB>>;
    t[BOP_TERN] = (uint64_t)&h_chain<T, &b_tern<T>>;
    t[BOP_INJ_ACC] = (uint64_t)&h_chain<T, &b_gen<T, 2, true>>;
    t[BOP_INJ_ROW] = (uint64_t)&h_chain<T, &b_gen<T, 0, true>>;
#define HX(K) t[TOPX_UN_BASE + (K - 3) * 2] = (uint64_t)&h_chain<T, &b_un2<T, K, 0>>; t[TOPX_UN_BASE + (K - 3) * 2 + 1] = (uint64_t)&h_chain<T, &b_un2<T, K, 1>>;
    HX(3) HX(4) HX(5) HX(6) HX(7) HX(8) HX(9) HX(10) HX(11) HX(12)
#undef HX
    t[TOPX_BIN_BASE + 0] = (uint64_t)&h_chain<T, &b_maxmin<T, 6, 0>>; t[TOPX_BIN_BASE + 1] = (uint64_t)&h_chain<T, &b_maxmin<T, 6, 1>>;
    t[TOPX_BIN_BASE + 2] = (uint64_t)&h_chain<T, &b_maxmin<T, 7, 0>>; t[TOPX_BIN_BASE + 3] = (uint64_t)&h_chain<T, &b_maxmin<T, 7, 1>>;
#undef HB
#undef HU
    // superinstructions
    t[top_loadrow(false, false)] = (uint64_t)&h_chain<T, &b_load_row<T>>;
    t[top_loadrow(false, true)] = (uint64_t)&h_chain<T, &b_loadrow_f<T, false, true>>;
    t[top_loadrow(true, false)] = (uint64_t)&h_chain<T, &b_loadrow_f<T, true, false>>;
    t[top_loadrow(true, true)] = (uint64_t)&h_chain<T, &b_loadrow_f<T, true, true>>;
    t[TOP_LOADCONST_PUSH] = (uint64_t)&h_chain<T, &b_loadconst_push<T>>;
#define TU1(K, O, P) t[top_unrow(K, O, P, false)] = (uint64_t)&h_chain<T, &b_unrow_f<T, K, O, P, false, TB>>; t[top_unrow(K, O, P, true)] = (uint64_t)&h_chain<T, &b_unrow_f<T, K, O, P, true, TB>>;
#define TU(K) TU1(K, false, false) TU1(K, false, true) TU1(K, true, false) TU1(K, true, true)
    TU(0) TU(1) TU(2)
#define TBC(K) t[top_binrowc(K, false)] = (uint64_t)&h_chain<T, &b_binrowc<T, K, false, TBK(K)>>; t[top_binrowc(K, true)] = (uint64_t)&h_chain<T, &b_binrowc<T, K, true, TBK(K)>>;
    TBC(0) TBC(1) TBC(2) TBC(3) TBC(4) TBC(5)
#define TB1(K, C, O) t[top_bin2(K, C, O, false)] = (uint64_t)&h_chain<T, &b_bin2<T, K, C, O, false, TBK(K)>>; t[top_bin2(K, C, O, true)] = (uint64_t)&h_chain<T, &b_bin2<T, K, C, O, true, TBK(K)>>;
#define TBF(K) TB1(K, false, false) TB1(K, false, true) TB1(K, true, false) TB1(K, true, true)
    TBF(0) TBF(1) TBF(2) TBF(3) TBF(4) TBF(5)
#undef TU1
#undef TU
#undef TBC
#undef TB1
#undef TBF
    if constexpr (sizeof(T) == 4) { // Float32: the fast-path-only forms (h_*_fast above) replace the full handlers in the table
#define HUF(K) t[BOP_UN_BASE + 4 * K + 0] = (uint64_t)&h_un_fast<K, 0, TB>; t[BOP_UN_BASE + 4 * K + 1] = (uint64_t)&h_un_fast<K, 1, TB>; \
               t[BOP_UN_BASE + 4 * K + 2] = (uint64_t)&h_un_fast<K, 2, TB>; t[BOP_UN_BASE + 4 * K + 3] = (uint64_t)&h_un_fast<K, 3, TB>; \
               t[TOPX_ENDV_BASE + 12 + K] = (uint64_t)&h_un_end_fast<K, TB>;
#define HURF1(K, O, P) t[top_unrow(K, O, P, false)] = (uint64_t)&h_unrow_fast<K, O, P, false, TB>; t[top_unrow(K, O, P, true)] = (uint64_t)&h_unrow_fast<K, O, P, true, TB>;
#define HURF(K) HURF1(K, false, false) HURF1(K, false, true) HURF1(K, true, false) HURF1(K, true, true)
        HUF(0) HUF(2) HURF(0) HURF(2)
        if constexpr (!TB) { // (turbo exp and the turbo divisions have no slow path)
            HUF(1) HURF(1)
#define HDF(K) t[BOP_BIN_BASE + 4 * K + 0] = (uint64_t)&h_div_fast<K, 0>; t[BOP_BIN_BASE + 4 * K + 1] = (uint64_t)&h_div_fast<K, 1>; \
               t[BOP_BIN_BASE + 4 * K + 2] = (uint64_t)&h_div_fast<K, 2>; t[BOP_BIN_BASE + 4 * K + 3] = (uint64_t)&h_div_fast<K, 3>; \
               t[TOPX_ENDV_BASE + K * 2] = (uint64_t)&h_div_end_fast<K, false>; t[TOPX_ENDV_BASE + K * 2 + 1] = (uint64_t)&h_div_end_fast<K, true>; \
               t[top_binrowc(K, false)] = (uint64_t)&h_divrowc_fast<K, false>; t[top_binrowc(K, true)] = (uint64_t)&h_divrowc_fast<K, true>;
#define HD2(K, C, O) t[top_bin2(K, C, O, false)] = (uint64_t)&h_div2_fast<K, C, O, false>; t[top_bin2(K, C, O, true)] = (uint64_t)&h_div2_fast<K, C, O, true>;
#define HD2K(K) HD2(K, false, false) HD2(K, false, true) HD2(K, true, false) HD2(K, true, true)
            HDF(4) HDF(5) HD2K(4) HD2K(5)
#undef HD2K
#undef HD2
#undef HDF
        }
#undef HURF
#undef HURF1
#undef HUF
    }
#undef TBK
}

// PRIORITY TILES (round 3).  A tree that is incomplete only on a few samples is evaluated on every tile until the workgroup holding
// such a sample has run — 1.1 ms of the 7.6 ms headline.  Those samples are not anywhere: a value overflows where a
// feature is largest or smallest, a quotient where one is closest to zero.  On the benchmark's data the 3 F tiles that hold, per
// feature, its largest value, its smallest value and its value closest to zero flag 534 of the 557 incomplete trees (first 16 tiles
// of the launch: 258; 16 random ones: 263; tools/exp_extremes.py).  So the launch first finds those tiles (this kernel: one pass over X,
// a record-breaking atomicMax per statistic and workgroup; F <= DE_PRIO_MAX_F) and runs them as its first workgroups — a second time, nothing is reordered: the pairs
// stay where map_block puts them and find their trees flagged.  Order only: flags and the rows of complete trees cannot change.
// key = orderable(value) << 32 | tile; statistic 3 f + 0: x, + 1: -x, + 2: -|x| (NaN / Inf rank first everywhere).
__device__ __forceinline__ uint32_t orderable_f32(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// Grid-stride over the samples (consecutive threads read consecutive samples: X is F contiguous values per sample), every thread keeps
// its best (value, tile) per statistic in registers (F <= FMAX: compile-time indices), one wave + block reduction of the 64-bit keys
// per workgroup, then at most 3 F atomics per workgroup: ~0.1 ms for 10^7 x 5 Float32 (the first version — a wave per tile, a shuffle
// reduction per tile and feature — took 0.49 ms).
template <typename T, int FMAX>
__global__ void __launch_bounds__(256) de_tile_extremes_kernel(const T *__restrict__ X, int64_t N, int64_t ldX, int F, int tile_shift,
                                                               unsigned long long *__restrict__ keys, const uint8_t *__restrict__ ok_init,
                                                               uint8_t *__restrict__ ok, int32_t n_ok) {
    // (round 6) the launch's first kernel also sets the flags to their initial values: one stream operation less in front of a small launch
    for (int32_t i = (int32_t)(blockIdx.x * 256 + threadIdx.x); i < n_ok; i += (int32_t)(gridDim.x * 256)) ok[i] = ok_init[i];
    const float inf = __builtin_inff();
    float v[3][FMAX];
    uint32_t at[3][FMAX];
    DE_UNROLL for (int f = 0; f < FMAX; f++) DE_UNROLL for (int k = 0; k < 3; k++) { v[k][f] = -inf; at[k][f] = 0u; }
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256) {
        const uint32_t tile = (uint32_t)(j >> tile_shift);
        DE_UNROLL for (int f = 0; f < FMAX; f++) {
            if (f < F) {
                const float x = (float)X[f + ldX * j];
                const bool fin = __builtin_fabsf(x) < inf; // false for NaN and Inf: they rank first in every statistic
                const float c[3] = {fin ? x : inf, fin ? -x : inf, fin ? -__builtin_fabsf(x) : inf};
                DE_UNROLL for (int k = 0; k < 3; k++)
                    if (c[k] > v[k][f]) { v[k][f] = c[k]; at[k][f] = tile; }
            }
        }
    }
    __shared__ unsigned long long best[4][3 * FMAX];
    DE_UNROLL for (int f = 0; f < FMAX; f++) {
        if (f < F) {
            DE_UNROLL for (int k = 0; k < 3; k++) {
                unsigned long long key = ((unsigned long long)orderable_f32(v[k][f]) << 32) | (unsigned long long)at[k][f];
                DE_UNROLL for (int m = 32; m >= 1; m >>= 1) {
                    const unsigned long long o = __shfl_xor(key, m, 64);
                    key = o > key ? o : key;
                }
                if ((threadIdx.x & 63) == 0) best[threadIdx.x >> 6][3 * f + k] = key;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 3 * F) {
        unsigned long long key = best[0][threadIdx.x];
        DE_UNROLL for (int w = 1; w < 4; w++) key = best[w][threadIdx.x] > key ? best[w][threadIdx.x] : key;
        unsigned long long *slot = keys + threadIdx.x;
        if (key > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, key);
    }
}

// COMPACTION OF THE LIVE TREES (round 4).  After the probe launch of the priority tiles most incomplete trees are flagged (534 of 557 on
// the benchmark's data), but they stay where they are in the population: a 64-tree chunk of the launch proper holds ~28 live trees, so the
// launch runs 2.3 x the workgroups for the same live work, each staging X for fewer trees and walking ~15 runs of skipped trees (0.45 ms of
// the 6.7 ms headline).  This kernel — one workgroup between the probe and the launch proper — RE-LINKS the chained stream:
// the records of every tree whose flag is still 1 are copied, in population order, into a second stream in which the live trees follow
// each other (a tree's end record — and its last instruction's record when that one is end-fused, the header's DE_HDR_FUSED_END bit —
// names the first handler of the next LIVE tree, which is what the header record of that tree names in the original stream).  Records are
// position-independent otherwise; the tree's index travels in its end record.  The launch proper then runs dense chunks over the compact
// stream (trees flagged later still drop out through the skip mask).  Order only: which trees run where — flags and complete rows cannot
// change (tests/test_gpu_early_exit.py).  ctrl = {n_live, n_chunks, trees per chunk, 0}.
// FUSED-LOSS launches also re-name the ENDS of the trees while they re-link them (LossEnds, by value): in the shared stream a tree ends in
// h_tree_end — or, 85 % of the bench population, in an end-fused last instruction — and both reach the loss epilogue only through their
// out-of-line detours (flags & HF_SLOW: the full form of the last instruction, then a tail call into h_tree_end_slow).  In the compact stream
// of a loss launch the last instruction is named in its PLAIN form (plain[k] for endv[k]) and names h_tree_end_slow (`slow_end`) directly.
struct LossEnds {
    uint64_t end, slow_end; // h_tree_end, h_tree_end_slow; slow_end == 0: not a loss launch, nothing is re-named
    uint64_t endv[TOPX_ENDV_COUNT], plain[TOPX_ENDV_COUNT];
};
template <bool F32> __device__ __forceinline__ uint64_t rec_next(const U32x4 &r, uint64_t hi) { return F32 ? (((uint64_t)r.w << 32) | r.z) : (hi | r.y); }
template <bool F32> __device__ __forceinline__ void rec_set_next(U32x4 &r, uint64_t h) {
    if (F32) { r.z = (uint32_t)h; r.w = (uint32_t)(h >> 32); } else r.y = (uint32_t)h;
}
template <bool F32>
__global__ void __launch_bounds__(1024) de_compact_live_kernel(const U32x4 *__restrict__ code, const int32_t *__restrict__ code_off, const uint8_t *__restrict__ ok,
                                                              int32_t n_trees, U32x4 *__restrict__ ccode, int32_t *__restrict__ coff, int32_t *__restrict__ live_idx,
                                                              int32_t *__restrict__ ctrl, int64_t n_tiles, int64_t want_blocks, int32_t tpc_max, const LossEnds le,
                                                              int64_t var_stride) {
    // (a wave group's stream variants, KArgs::var_stride: workgroup v re-links variant v; the offsets, the tree list and the plan it writes
    // are the same values in every workgroup, and each reads back only what it wrote itself)
    code += (int64_t)blockIdx.x * var_stride;
    ccode += (int64_t)blockIdx.x * var_stride;
    __shared__ int32_t wsum[2][16];
    __shared__ int32_t run[2]; // live trees / records placed so far
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { run[0] = 0; run[1] = 1; } // record 0 of the compact stream = the head record
    __syncthreads();
    for (int32_t b = 0; b < n_trees; b += 1024) {
        const int32_t t = b + tid;
        const bool live = t < n_trees && ok[t] != 0;
        const int32_t len = live ? code_off[t + 1] - code_off[t] : 0; // instruction records + the end record
        int32_t s0 = live ? 1 : 0, s1 = len;
        DE_UNROLL for (int d = 1; d < 64; d <<= 1) { // inclusive scan inside the wave
            const int32_t u0 = __shfl_up(s0, d, 64), u1 = __shfl_up(s1, d, 64);
            if (lane >= d) { s0 += u0; s1 += u1; }
        }
        if (lane == 63) { wsum[0][wv] = s0; wsum[1][wv] = s1; }
        __syncthreads();
        int32_t o0 = run[0], o1 = run[1];
        for (int w = 0; w < wv; w++) { o0 += wsum[0][w]; o1 += wsum[1][w]; }
        if (live) {
            live_idx[o0 + s0 - 1] = t;
            coff[o0 + s0 - 1] = o1 + s1 - len;
        }
        __syncthreads();
        if (tid == 1023) { run[0] = o0 + s0; run[1] = o1 + s1; }
        __syncthreads();
    }
    const int32_t n_live = run[0];
    if (tid == 0) {
        coff[n_live] = run[1];
        int32_t nc = 0, tpc = 1;
        chunk_plan(n_live, n_tiles, tpc_max, want_blocks, &nc, &tpc, nullptr);
        ctrl[0] = n_live;
        ctrl[1] = nc;
        ctrl[2] = tpc;
        ctrl[3] = 0;
    }
    // (live_idx / coff were written by this workgroup: visible to it behind the barriers above)
    for (int32_t k = tid; k < n_live; k += 1024) {
        const int32_t t = live_idx[k], tn = k + 1 < n_live ? live_idx[k + 1] : -1;
        const int32_t src = code_off[t], len = code_off[t + 1] - src, dst = coff[k];
        const U32x4 hdr = code[src - 1]; // this tree's header: length word (F32: .y, F64: .z) | DE_HDR_FUSED_END
        if (k == 0) ccode[0] = hdr;      // the head record names the first live tree's first handler
        { // four records in flight per thread: the copy is latency-bound (one workgroup, ~12 records per tree: 17 -> 10 us for 1000 trees)
            int32_t i = 0;
            for (; i + 4 <= len; i += 4) {
                const U32x4 r0 = code[src + i], r1 = code[src + i + 1], r2 = code[src + i + 2], r3 = code[src + i + 3];
                ccode[dst + i] = r0; ccode[dst + i + 1] = r1; ccode[dst + i + 2] = r2; ccode[dst + i + 3] = r3;
            }
            for (; i < len; i++) ccode[dst + i] = code[src + i];
        }
        if (tn >= 0) {
            const U32x4 hn = code[code_off[tn] - 1]; // names the next live tree's first handler (and carries its length word)
            U32x4 e = code[src + len - 1];
            e.y = hn.y; e.z = hn.z; e.w = hn.w;       // the operand word stays: this tree's index
            ccode[dst + len - 1] = e;
            if (((F32 ? hdr.y : hdr.z) & DE_HDR_FUSED_END) && len >= 2 && !le.slow_end) { // the end-fused last instruction names it too
                U32x4 q = code[src + len - 2];
                if (F32) { q.z = hn.z; q.w = hn.w; } else q.y = hn.y;
                ccode[dst + len - 2] = q;
            }
        }
        if (le.slow_end && len >= 2) { // fused loss: plain last instruction -> h_tree_end_slow (see LossEnds)
            const uint64_t hi = le.end & 0xFFFFFFFF00000000ull; // (Float64 records carry the low half; all handlers share the high one)
            U32x4 q = ccode[dst + len - 2];
            rec_set_next<F32>(q, le.slow_end);
            ccode[dst + len - 2] = q;
            if ((F32 ? hdr.y : hdr.z) & DE_HDR_FUSED_END) { // the record in front of the last instruction (>= 2 instructions: never the header) names endv[k]
                U32x4 f = ccode[dst + len - 3];
                const uint64_t named = rec_next<F32>(f, hi);
                DE_UNROLL for (int q2 = 0; q2 < (int)TOPX_ENDV_COUNT; q2++)
                    if (named == (F32 ? le.endv[q2] : (hi | (uint32_t)le.endv[q2]))) rec_set_next<F32>(f, le.plain[q2]);
                ccode[dst + len - 3] = f;
            }
        }
    }
}

// WW > 1: a WAVE GROUP (KArgs::var_stride) — WW waves of 64 lanes on one 64-lane tile; `tid` below is the lane's position in the TILE
// (what addresses rows, samples and output), `stid` its position in the workgroup (what the cooperative staging loops stride by).
template <typename T, bool PARAMS, bool LOSS = false, int WW = 1>
__global__ void __launch_bounds__(DE_TBLK * WW) de_eval_threaded_kernel(const KArgs<T> a) {
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    constexpr int BLK = DE_TBLK, G = TG<T>::G, PLANE = BLK * VW, TILE = PLANE * G, ROWV = BLK * G + 1; // (PLANE samples per plane, G planes per row)
    static_assert(WW == 1 || DE_TBLK == 64, "a wave group shares one 64-lane tile");
    constexpr int SBLK = BLK * WW; // threads that stage the tile
    extern __shared__ __align__(16) unsigned char smem_base[];
    unsigned char *const smem_raw = smem_base + DE_SKIPLIST_BYTES; // [0, DE_SKIPLIST_BYTES): the live-tree list of h_tree_skip; rows behind it
    T *__restrict__ rows = reinterpret_cast<T *>(smem_raw);

    TileMap tm;
    int flag_protocol = a.skip_flagged;
    // a compacted launch (de_compact_live_kernel): the trees, chunks and trees per chunk the device planned for the live trees
    int32_t n_trees = a.n_trees, n_chunks = a.n_chunks, tpc = a.trees_per_chunk;
    if (a.ctrl) {
        const ConstI32Ptr ct = (ConstI32Ptr)(uintptr_t)a.ctrl;
        n_trees = ct[0];
        n_chunks = ct[1];
        tpc = ct[2];
        if (n_trees <= 0) return;
    }
    // (a wave group maps GROUPS of WW chunks; the host launches it without a tail split)
    const int32_t host_groups = WW > 1 ? (a.n_chunks + WW - 1) / WW : a.n_chunks, groups = WW > 1 ? (n_chunks + WW - 1) / WW : n_chunks + a.tail_split - 1;
    if (blockIdx.x < a.n_prio_blocks) { // a priority tile (see de_tile_extremes_kernel): its flags go straight to memory and come from there
        const uint32_t k = blockIdx.x / (uint32_t)host_groups;
        if (k >= a.n_prio) return;
        tm.tile = (int64_t)((uint32_t)a.prio[k] >> a.prio_shift);
        tm.chunk = (int32_t)(blockIdx.x % (uint32_t)host_groups);
        tm.valid = tm.tile < a.n_tiles;
        flag_protocol = 1;
    } else tm = a.map_group > 0 ? map_block_grouped(blockIdx.x - a.n_prio_blocks, groups, a.n_tiles, (uint32_t)a.map_group)
                                : map_block(blockIdx.x - a.n_prio_blocks, groups, a.n_tiles);
    if (!tm.valid) return;
    const int stid = threadIdx.x;
    const int tid = WW > 1 ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
    const int wave = WW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int64_t base = tm.tile * TILE;
    const int64_t last = a.N - 1;
    if constexpr (WW > 1) tm.chunk = tm.chunk * WW + wave; // (a wave past the last chunk stages with the others and leaves behind the barrier)
    int tA = tm.chunk * tpc; // this workgroup's (wave's) trees: [tA, tB) (of the compact stream in a compacted launch)
    int tB = (tA + tpc < n_trees) ? tA + tpc : n_trees;
    if (WW > 1 && tm.chunk >= n_chunks) tA = tB = 0;
    if (WW == 1 && a.tail_split > 1 && tm.chunk >= n_chunks - 1) { // the plan's last chunk in `tail_split` pieces (chunk-major order: the last workgroups of the launch)
        const int b0 = (n_chunks - 1) * tpc, rem = n_trees - b0;
        const int sub = (rem + a.tail_split - 1) / a.tail_split;
        tA = b0 + (tm.chunk - (n_chunks - 1)) * sub;
        tB = tA + sub < n_trees ? tA + sub : n_trees;
        if (tA >= tB) return; // (before anything is staged: a workgroup is one wave here, no barrier is left waiting)
    }
    // early exit: the flags of the first <= 64 trees, requested before the X tile so that the two latencies overlap
    // (wave 0 reads them for the whole workgroup: two waves reading at different moments could see different flags, and the
    // workgroup shares ONE live-tree list)
    uint8_t f_first = 1;
    if (a.skip_flagged && (WW > 1 || tid < 64) && tA + tid < tB) f_first = skip_flag_load(a.ok + (a.live_idx ? a.live_idx[tA + tid] : tA + tid), flag_protocol, tm.tile);
    // the 64-bit skip mask travels from wave 0 to the others through the padding vector of LDS row 0 (16 unused bytes behind the
    // DE_TBLK vectors of every row)
    uint64_t *const mask_slot = reinterpret_cast<uint64_t *>(smem_raw + (size_t)BLK * G * 16);
    // ... and every wave's copy of the trees' record offsets (the live-tree list below), requested before the X tile as well: with
    // one tree per workgroup (the reference's own call shape, 1 tree x 5e7 samples) a load issued behind the barrier was +50 %
    int32_t co_first = 0;
    if (a.skip_flagged && tA + (tid & 63) < tB) co_first = a.code_off[tA + (tid & 63)];
    {
        const uint32_t F = (uint32_t)a.F;
        const uint32_t total = (uint32_t)TILE * F;
        if (a.x_vec && base + TILE <= a.N) {
            // The tile is TILE*F contiguous elements = exactly G*F 16-byte vectors per thread.  Up to 8 vector
            // loads are issued before the first LDS write (few trees per chunk = HBM-bound: memory-level
            // parallelism is what counts there); (sample, feature) of an element by a host-computed
            // reciprocal instead of a run-time division.
            const V *__restrict__ src = reinterpret_cast<const V *>(a.X + base * (int64_t)F);
            const uint32_t GF = (uint32_t)G * F;
            if constexpr (WW > 1) { // the same loop over the workgroup's SBLK threads (the tile has BLK * GF vectors)
                const uint32_t nv = (uint32_t)BLK * GF;
                for (uint32_t i0 = 0; i0 * SBLK < nv; i0 += 8) {
                    V buf[8];
                    DE_UNROLL for (uint32_t u = 0; u < 8; u++)
                        if (stid + (i0 + u) * SBLK < nv) buf[u] = src[stid + (i0 + u) * SBLK];
                    DE_UNROLL for (uint32_t u = 0; u < 8; u++) {
                        if (stid + (i0 + u) * SBLK < nv) {
                            const uint32_t e = (stid + (i0 + u) * SBLK) * VW;
                            uint32_t j = a.f_magic ? __umulhi(e, a.f_magic) : e;
                            uint32_t f = e - j * F;
                            DE_UNROLL for (int c = 0; c < VW; c++) {
                                rows[f * (ROWV * VW) + j] = buf[u][c];
                                ++f;
                                if (f == F) { f = 0; ++j; }
                            }
                        }
                    }
                }
            } else
            for (uint32_t i0 = 0; i0 < GF; i0 += 8) {
                V buf[8];
                DE_UNROLL for (uint32_t u = 0; u < 8; u++)
                    if (i0 + u < GF) buf[u] = src[tid + (i0 + u) * BLK];
                DE_UNROLL for (uint32_t u = 0; u < 8; u++) {
                    if (i0 + u < GF) {
                        const uint32_t e = (tid + (i0 + u) * BLK) * VW;
                        uint32_t j = a.f_magic ? __umulhi(e, a.f_magic) : e;
                        uint32_t f = e - j * F;
                        DE_UNROLL for (int c = 0; c < VW; c++) {
                            rows[f * (ROWV * VW) + j] = buf[u][c];
                            ++f;
                            if (f == F) { f = 0; ++j; }
                        }
                    }
                }
            }
        } else if (a.ldX == (int64_t)F && base + TILE <= a.N) {
            const T *__restrict__ src = a.X + base * (int64_t)F;
            for (uint32_t e = stid; e < total; e += SBLK) {
                const uint32_t j = e / F, f = e - j * F;
                rows[f * (ROWV * VW) + j] = src[e];
            }
        } else {
            for (uint32_t e = stid; e < total; e += SBLK) {
                const uint32_t j = e / F, f = e - j * F;
                int64_t jj = base + j;
                jj = jj < last ? jj : last;
                rows[f * (ROWV * VW) + j] = a.X[f + a.ldX * jj];
            }
        }
    }
    if (PARAMS && a.n_prows > 0) {
        stage_param_rows<T>(a, rows, ROWV * VW, base, TILE, stid, SBLK);
    } else if (PARAMS && WW == 1) { // (a wave group is only launched with staged parameter rows)
        // the class row: byte offsets of this thread's samples' parameter columns (the table has < 2^32 bytes: checked
        // on the host), and the table's address for h_param (see there)
        DE_UNROLL for (int g = 0; g < G; g++) {
            uint32_t cv[4] = {0u, 0u, 0u, 0u};
            DE_UNROLL for (int i = 0; i < VW; i++) {
                int64_t jj = base + g * PLANE + tid * VW + i;
                jj = jj < last ? jj : last;
                cv[i] = (uint32_t)((uint64_t)a.ld_params * sizeof(T) *
                                   (uint64_t)clamp_class((a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj]
                                                                           : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj]) - a.class_base,
                                                         a.n_classes));
            }
            const uint64_t tab = (uint64_t)(uintptr_t)a.params;
            unsigned char *crow = smem_raw + a.cls_row_off + g * DE_PLANE_BYTES + tid * 16;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<U32x4 *>(crow) = U32x4{cv[0], cv[1], cv[2], cv[3]};
                *reinterpret_cast<U32x4 *>(crow + RowOf<T>::BYTES) = U32x4{(uint32_t)tab, (uint32_t)(tab >> 32), 0u, 0u};
            } else {
                *reinterpret_cast<U32x4 *>(crow) = U32x4{cv[0], cv[1], (uint32_t)tab, (uint32_t)(tab >> 32)};
            }
        }
    }
    uint64_t m_first = 0ull; // (wave group: every wave has trees and flags of its own — its mask stays in registers)
    if constexpr (WW > 1) {
        if (a.skip_flagged) m_first = __ballot(f_first == 0);
    } else if (a.skip_flagged && tid < 64) { // (all 64 lanes of wave 0 take part in the ballot)
        const uint64_t m0 = __ballot(f_first == 0);
        if (tid == 0) *mask_slot = m0;
    }
    __syncthreads();

    const ConstU4Ptr code = (ConstU4Ptr)(uintptr_t)(a.code + (WW > 1 ? (int64_t)wave * a.var_stride : (int64_t)0));
    // this wave's live-tree list (h_tree_skip): wave 0 / a one-wave workgroup at LDS address 0, wave w of a group behind the rows
    const uint32_t list_at = (WW > 1 && wave > 0) ? a.list_off + (uint32_t)(wave - 1) * DE_SKIPLIST_BYTES : 0u;
    const ConstI32Ptr code_off = (ConstI32Ptr)(uintptr_t)a.code_off;
    const bool full = base + TILE <= a.N;
    // LDS byte address of this thread's vector in row 0 (the dynamic LDS segment starts at 0)
    // (the low 32 bits of a flat pointer into LDS are the LDS byte offset)
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw + tid * 16;
    // fused loss: this thread's residual targets and weights stay in registers for every tree of
    // the chunk; samples past N get weight 0 (their X columns are clamped copies of the last one)
    HLoss<T> yv, wv;
    DE_UNROLL for (int g = 0; g < 2; g++) { yv.v[g] = V{}; wv.v[g] = V{}; }
    if constexpr (LOSS) {
        DE_UNROLL for (int g = 0; g < G; g++) DE_UNROLL for (int i = 0; i < VW; i++) {
            const int64_t j = base + g * PLANE + tid * VW + i;
            const int64_t jj = j < last ? j : last;
            yv.v[g][i] = a.y[jj];
            wv.v[g][i] = j <= last ? (a.w ? a.w[jj] : T(1)) : T(0);
        }
    }

    if (tA >= tB) return;
    const uint64_t ldo = (uint64_t)a.ld_out * sizeof(T);
    // A chunk runs in sub-chunks of <= 64 trees: the flags of a sub-chunk are read when it starts (one 64-bit ballot; the later
    // ones see what other workgroups found in the meantime), then its trees run as ONE chain.  Without the early exit the whole
    // chunk is one sub-chunk.
    const int sub = 63; // (a chain runs <= 63 trees: bit (trees of the chain) of `skip` is the sentinel HTREE_END_TAIL tests; without the early exit the sub-chunks cost nothing)
    for (int t0 = tA; t0 < tB; t0 += sub) {
    const int t1 = t0 + sub < tB ? t0 + sub : tB;
    // the trees of this sub-chunk whose flag is already 0 (bit i: tree t0 + i): found incomplete by a workgroup that ran earlier
    // (or by the host: a non-finite constant) — the reference stops evaluating such a tree at its first non-finite array
    // (src/Evaluate.jl:26-32), this kernel stops at the next workgroup.  Agent scope: past this CU's vector cache.
    uint64_t skip = 0ull;
    if (a.skip_flagged) {
        if constexpr (WW > 1) { // (one sub-chunk: a wave group's chunks hold <= 63 trees — launch_threaded_t)
            if (t0 != tA) break;
            skip = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m_first >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m_first);
            skip &= (1ull << (t1 - t0)) - 1ull;
        } else {
        if (t0 != tA) { // a later sub-chunk: wave 0 reads its flags once every wave is done with the previous list
            __syncthreads();
            if (tid < 64) {
                const int i = t0 + tid;
                const uint8_t f = i >= t1 ? (uint8_t)1 : skip_flag_load(a.ok + (a.live_idx ? a.live_idx[i] : i), flag_protocol, tm.tile);
                const uint64_t m0 = __ballot(f == 0);
                if (tid == 0) *mask_slot = m0;
            }
            __syncthreads();
        }
        { // (the first sub-chunk's mask was written before the barrier behind the X staging)
            const uint64_t m = *mask_slot;
            skip = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
            skip &= (1ull << (t1 - t0)) - 1ull; // (the first mask holds the flags of 64 trees: the 64th belongs to the next sub-chunk)
        }
        } // one-wave workgroup
        {
            // the live trees' header addresses, in reverse order (see h_tree_skip); every wave writes the whole list (same values):
            // a wave reads only what it wrote itself, no barrier
            const int i = t0 + (tid & 63);
            if (i < t1 && !((skip >> (tid & 63)) & 1ull)) {
                const uint32_t r = (uint32_t)__builtin_popcountll((~skip & ((1ull << (t1 - t0)) - 1ull)) >> 1 >> (tid & 63));
                const uint64_t hdr = (uint64_t)(uintptr_t)(code + (t0 == tA ? co_first : a.code_off[i]) - 1);
                reinterpret_cast<uint32_t *>(smem_base + list_at)[r] = (uint32_t)hdr;
            }
        }
    }
    {
        // ONE call per sub-chunk: the trees t0..t1 are consecutive in the stream and every tree's end record (h_tree_end)
        // stores its results (fused loss: forms its loss partial) and runs on into the next tree; the call returns after the last one.
        int first = t0;
        if (skip & 1ull) { // leading skipped trees
            if (~skip == 0ull) continue;
            const int n = __builtin_ctzll(~skip);
            first += n;
            skip >>= n;
            if (first >= t1) continue;
        }
        skip |= 1ull << (t1 - first); // the sentinel behind the last tree of this chain (t1 - first <= 63)
        const ConstU4Ptr rec = code + code_off[first];
        HState<T> st;
        DE_UNROLL for (int g = 0; g < G; g++) DE_UNROLL for (int i = 0; i < VW; i++) st.acc[g][i] = T(0);
        st.poison = typename PoisonOf<T>::type{};
        const int64_t in_tile = a.N - base < (int64_t)TILE ? a.N - base : (int64_t)TILE;
        uint32_t flags = (flag_protocol == 1 ? 0u : HF_PLAIN_FLAG) | ((list_at >> 4) << HF_LIST_SHIFT);
        uint64_t outp, ldo_arg = ldo;
        if constexpr (LOSS) {
            flags |= HF_SLOW | HF_LOSS | (a.loss_kind == DE_LOSS_L1 ? (uint32_t)HF_LOSS_L1 : 0u) | ((full && !a.w) ? (uint32_t)HF_LOSS_PLAIN : 0u);
            outp = (uint64_t)(uintptr_t)(a.partial + ((int64_t)tm.tile * a.n_trees) * TWAVES + __builtin_amdgcn_readfirstlane(tid >> 6)); // (wave-uniform: an SGPR argument; a wave group: tid < 64)
            ldo_arg = (uint64_t)TWAVES * sizeof(T);
        } else {
            flags |= a.vec_store == 2 ? (HF_SLOW | HF_NO_STORE) : ((full && a.vec_store) ? 0u : (HF_SLOW | HF_SLOW_STORE | (uint32_t)in_tile));
            outp = (uint64_t)(uintptr_t)(a.out + base) - (uint64_t)(uint32_t)(uintptr_t)smem_raw;
        }
        const U32x4 hp = rec[-1], hd = *rec; // the first handler's address is in the record in front (the previous tree's end record / the head record)
        st = arg_next<T>(hp.y, ((uint64_t)hp.w << 32) | hp.z)(st,
#if DE_NO_LOSS
#elif DE_TG == 1
                                                              yv.v[0], wv.v[0],
#else
                                                              yv.v[0], yv.v[1], wv.v[0], wv.v[1],
#endif
                                                              lds0, rec + 1, outp, hd.x, hd.y, ((uint64_t)hd.w << 32) | hd.z, (uint64_t)(uintptr_t)a.ok,
                                                              ldo_arg, skip, (uint32_t)(t1 - first), flags);
        (void)st;
    }
    } // sub-chunks
}

// ---- fused loss, passes 2 and 3: deterministic fixed-order reduction of the per-wave partials.
// Pass 2: thread = one (tree, wave) column, block row = one segment of tiles; coalesced reads.
template <typename T>
__global__ void __launch_bounds__(256) de_loss_reduce_tiles_kernel(const T *__restrict__ partial, int64_t n_cols, int64_t n_tiles,
                                                                  int64_t tiles_per_seg, double *__restrict__ seg_sum) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= n_cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * tiles_per_seg;
    const int64_t r1 = r0 + tiles_per_seg < n_tiles ? r0 + tiles_per_seg : n_tiles;
    double s = 0.0;
    for (int64_t r = r0; r < r1; ++r) s += (double)partial[r * n_cols + col];
    seg_sum[(int64_t)blockIdx.y * n_cols + col] = s;
}
// Pass 3: thread = one tree; NaN where the evaluation was incomplete (what `tree(X)` NaN-fills to,
// src/EvaluationHelpers.jl:29-33, gives any sum-type loss).
template <typename T>
__global__ void __launch_bounds__(256) de_loss_finish_kernel(const double *__restrict__ seg_sum, int64_t n_trees, int32_t n_segs,
                                                            const uint8_t *__restrict__ ok, T *__restrict__ loss) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_trees) return;
    double s = 0.0;
    for (int32_t g = 0; g < n_segs; ++g)
        for (int w = 0; w < TWAVES; ++w) s += seg_sum[((int64_t)g * n_trees + t) * TWAVES + w];
    loss[t] = ok[t] ? (T)s : M<T>::nan();
}

// ---------------------------------------------------------------------------
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// Kernel geometry: G groups of 16-byte vectors per thread, BLK threads per workgroup.
// Defaults chosen on MI355X; DE_EVAL_G / DE_EVAL_BLOCK override
// them for experiments.
static void eval_geometry(int dtype, int *G, int *BLK) {
    *G = env_int("DE_EVAL_G", 1);
    *BLK = env_int("DE_EVAL_BLOCK", 256);
    if (*G != 1 && *G != 2) *G = 1;
    if (*BLK != 128 && *BLK != 256) *BLK = 256;
    if (dtype != DE_F32) { *G = 1; *BLK = 256; }
}

size_t eval_lds_bytes(int dtype, int F, int n_slots, int *K_out) {
    int G, BLK;
    eval_geometry(dtype, &G, &BLK);
    const size_t rowv = (size_t)BLK * G + 1;
    const size_t bytes = (size_t)(F + n_slots) * rowv * 16;
    if (K_out) *K_out = (dtype == DE_F32 ? 4 : 2) * G;
    return bytes <= 160 * 1024 ? bytes : 0;
}

static void eval_geometry(int dtype, int *G, int *BLK);
bool eval_uses_threaded();
static int g_cu_count = 0;

static int cu_count() {
    if (g_cu_count == 0) {
        const int forced = env_int("DE_CU_COUNT", 0); // experiments: < 0 disables the small-grid re-split
        if (forced != 0) { g_cu_count = forced; return forced < 0 ? 0 : forced; }
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cu_count = prop.multiProcessorCount;
        if (g_cu_count <= 0) g_cu_count = 256; // MI355X
    }
    return g_cu_count < 0 ? 0 : g_cu_count;
}

// Tree chunking: chunks of ~64 trees keep workgroups short (fine-grained tail) while the
// X-tile staging (one L2 read of the tile per chunk) stays a few percent of the work; with few
// sample tiles, split further so the grid still covers the chip several times.
// (`waves` > 1: the chunks of a wave group — one per wave, 1 / waves of the trees each: a workgroup keeps the trees, and the record
// footprint, of a one-wave workgroup)
static int32_t plan_tpc_max(int waves) {
    const int64_t tpc_env = env_int("DE_EVAL_TPC", 63); // trees per chunk (experiments: X staging per tree against the tail of a short launch)
    const int64_t t = (tpc_env < 1 ? 63 : (tpc_env > 63 && waves > 1 ? 63 : tpc_env)) / (waves > 1 ? waves : 1);
    return (int32_t)(t < 1 ? 1 : t);
}
static void plan_chunks(int64_t n_trees, int64_t n_tiles, int32_t *n_chunks_out, int32_t *tpc_out, int32_t *nc0_out = nullptr, int waves = 1) {
    chunk_plan(n_trees, n_tiles, plan_tpc_max(waves), (int64_t)cu_count() * 4 * 8, n_chunks_out, tpc_out, nc0_out);
    if (*n_chunks_out < 1) *n_chunks_out = 1;
}

void eval_plan(int dtype, int64_t n_trees, int64_t N, int32_t *tile, int32_t *n_chunks, int32_t *trees_per_chunk, int waves) {
    int G, BLK;
    eval_geometry(dtype, &G, &BLK);
    if (eval_uses_threaded()) { G = tg_planes(dtype); BLK = TBLK; }
    *tile = BLK * G * (dtype == DE_F32 ? 4 : 2);
    plan_chunks(n_trees, (N + *tile - 1) / *tile, n_chunks, trees_per_chunk, nullptr, eval_uses_threaded() ? waves : 1);
    if (eval_uses_threaded() && waves > 1) { // a wave group: the workgroups per sample tile, and the trees of one (they share its staged X tile)
        *n_chunks = (*n_chunks + waves - 1) / waves;
        *trees_per_chunk *= waves;
    }
}

template <typename T, int G, int BLK>
static hipError_t launch_eval_t(const EvalArgs &e, hipStream_t stream, const char **kname) {
    constexpr int VW = VecOf<T>::W;
    constexpr int TILE = BLK * VW * G;
    KArgs<T> a;
    a.code = e.code;
    a.code_off = e.code_off;
    a.X = static_cast<const T *>(e.X);
    a.out = static_cast<T *>(e.out);
    a.ok = e.ok;
    a.params = static_cast<const T *>(e.params);
    a.classes = e.classes;
    a.N = e.N;
    a.ldX = e.ldX;
    a.ld_out = e.ld_out;
    a.ld_params = e.ld_params;
    a.prow_base = e.prow_base;
    a.n_prows = e.n_prows;
    a.n_tiles = (e.N + TILE - 1) / TILE;
    a.F = e.F;
    a.n_trees = e.n_trees;
    a.n_slots = e.n_slots;
    a.xstride = 0;
    a.classes_is_i64 = e.classes_is_i64;
    a.class_base = e.class_base;
    a.n_classes = e.n_classes > 0 ? e.n_classes : 1;
    a.vec_store = (reinterpret_cast<uintptr_t>(e.out) % 16 == 0 && (e.ld_out * sizeof(T)) % 16 == 0) ? 1 : 0;

    int32_t tpc, nch;
    plan_chunks(e.n_trees, a.n_tiles, &nch, &tpc);
    a.trees_per_chunk = tpc;
    a.n_chunks = nch;
    a.skip_flagged = (e.early_exit && e.skip_flagged && tpc <= 64) ? 2 : 0; // (2: the flag protocol of launch_threaded_t)
    a.x_vec = 0;
    a.f_magic = 0;

    const int64_t tile_groups = (a.n_tiles + 7) / 8;
    const int64_t blocks = tile_groups * 8 * a.n_chunks;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    void (*kern)(const KArgs<T>);
    if (e.early_exit) kern = e.uses_params ? de_eval_tape_kernel<T, G, BLK, true, true> : de_eval_tape_kernel<T, G, BLK, true, false>;
    else kern = e.uses_params ? de_eval_tape_kernel<T, G, BLK, false, true> : de_eval_tape_kernel<T, G, BLK, false, false>;
    a.cert_max = e.cert_max;
    if constexpr (G == 1 && BLK == 256) {
        if (e.cert_max) { // the certificate pass (de_eval_sum_certificate): early-exit flag semantics, no output
            if (!e.early_exit || e.direct) return hipErrorInvalidValue;
            kern = e.uses_params ? de_eval_tape_kernel<T, 1, 256, true, true, false, true> : de_eval_tape_kernel<T, 1, 256, true, false, false, true>;
        }
    } else if (e.cert_max) return hipErrorInvalidValue;
    if (kname) *kname = e.cert_max ? "de_eval_tape_kernel<cert>" : "de_eval_tape_kernel";
    size_t lds = (size_t)(a.F + a.n_slots) * ((size_t)BLK * G + 1) * 16;
    if constexpr (G == 1 && BLK == 256) {
        if (e.direct) {
            kern = e.early_exit ? (e.uses_params ? de_eval_tape_kernel<T, 1, 256, true, true, true> : de_eval_tape_kernel<T, 1, 256, true, false, true>)
                                : (e.uses_params ? de_eval_tape_kernel<T, 1, 256, false, true, true> : de_eval_tape_kernel<T, 1, 256, false, false, true>);
            lds = (size_t)(a.n_slots > 0 ? a.n_slots : 1) * 257 * 16;
            if (kname) *kname = "de_eval_tape_kernel<direct>";
        }
    }
    if (lds > 64 * 1024) {
        hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLK), lds, stream, a);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_eval_geo(const EvalArgs &a, hipStream_t stream, const char **kn, int G, int BLK) {
    if (G == 2) return BLK == 128 ? launch_eval_t<T, 2, 128>(a, stream, kn) : launch_eval_t<T, 2, 256>(a, stream, kn);
    return BLK == 128 ? launch_eval_t<T, 1, 128>(a, stream, kn) : launch_eval_t<T, 1, 256>(a, stream, kn);
}

// ---- threaded variant: handler table + launch ---------------------------------------------
template <typename T, bool TB> static hipError_t fetch_handlers(uint64_t *host_table) {
    uint64_t *d = nullptr;
    hipError_t st = hipMalloc(reinterpret_cast<void **>(&d), TOPX_TABLE * sizeof(uint64_t));
    if (st != hipSuccess) return st;
    hipLaunchKernelGGL((de_fill_handlers<T, TB>), dim3(1), dim3(1), 0, 0, d);
    st = hipMemcpy(host_table, d, TOPX_TABLE * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return st;
}

// Handler addresses are baked into every record of the instruction streams and belong to ONE device's copy of the code object
// (every device loads its own): the caches of the three handler tables (eval here, gradient and reverse in de_grad_kernels.hip) are
// keyed by the CURRENT device — the device of the context that creates the program (de_api_program.cpp sets it before it asks).  A process may
// therefore hold contexts on several GPUs (round 3 refused every device but the first: ADVICE r3).
hipError_t handler_device_slot(int *slot) {
    int dev = 0;
    const hipError_t st = hipGetDevice(&dev);
    if (st != hipSuccess) return st;
    if (dev < 0 || dev >= DE_MAX_DEVICES) return hipErrorInvalidDevice;
    *slot = dev;
    return hipSuccess;
}

hipError_t eval_handler_table(int dtype, bool turbo, uint64_t *table) {
    struct Cache { uint64_t t[3][TOPX_TABLE]; bool have[3] = {false, false, false}; }; // Float32, Float64, Float32 turbo (Float64 has no relaxed operators)
    static std::unique_ptr<Cache> caches[DE_MAX_DEVICES];
    static std::mutex mu; // contexts on several host threads may ask at once
    int dev = 0;
    { const hipError_t dst = handler_device_slot(&dev); if (dst != hipSuccess) return dst; }
    const std::lock_guard<std::mutex> lock(mu);
    if (!caches[dev]) caches[dev].reset(new Cache());
    Cache &c = *caches[dev];
    const int k = dtype == DE_F32 ? (turbo ? 2 : 0) : 1;
    if (!c.have[k]) {
        hipError_t st = k == 0 ? fetch_handlers<float, false>(c.t[k]) : (k == 2 ? fetch_handlers<float, true>(c.t[k]) : fetch_handlers<double, false>(c.t[k]));
        if (st != hipSuccess) return st;
        c.have[k] = true;
    }
    for (int i = 0; i < (int)TOPX_TABLE; i++) table[i] = c.t[k][i];
    return hipSuccess;
}

bool eval_uses_threaded() { return env_int("DE_EVAL_THREADED", 1) != 0 && env_int("DE_EVAL_G", 1) == 1 && env_int("DE_EVAL_BLOCK", 256) == 256; }

// The same statistics for PACKED, 16-byte aligned Float32 X (ldX == F): a thread takes groups of four consecutive samples = F 16-byte
// vectors (4 F consecutive floats), so the pass issues a quarter of the load instructions of the scalar loop — which reads X at 1.8 TB/s
// (0.118 ms for the headline's 200 MB, 2 % of the step).  F is a template parameter: element s * F + f of the group is a fixed register.
// The <= 3 samples behind the last full group go through the scalar statistics of thread 0 of workgroup 0.
template <int F>
__global__ void __launch_bounds__(256) de_tile_extremes_vec_kernel(const float *__restrict__ X, int64_t N, int tile_shift, unsigned long long *__restrict__ keys,
                                                                   const uint8_t *__restrict__ ok_init, uint8_t *__restrict__ ok, int32_t n_ok) {
    for (int32_t i = (int32_t)(blockIdx.x * 256 + threadIdx.x); i < n_ok; i += (int32_t)(gridDim.x * 256)) ok[i] = ok_init[i]; // (see de_tile_extremes_kernel)
    typedef float V4 __attribute__((ext_vector_type(4)));
    const float inf = __builtin_inff();
    float v[3][F];
    uint32_t at[3][F];
    DE_UNROLL for (int f = 0; f < F; f++) DE_UNROLL for (int k = 0; k < 3; k++) { v[k][f] = -inf; at[k][f] = 0u; }
    const int64_t groups = N >> 2;
    const V4 *__restrict__ Xv = reinterpret_cast<const V4 *>(X);
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        float e[4 * F];
        DE_UNROLL for (int q = 0; q < F; q++) {
            const V4 w = Xv[g * F + q];
            e[4 * q] = w[0]; e[4 * q + 1] = w[1]; e[4 * q + 2] = w[2]; e[4 * q + 3] = w[3];
        }
        const uint32_t tile = (uint32_t)((g << 2) >> tile_shift); // (four samples of one group lie in one unit of >= 4 samples)
        DE_UNROLL for (int sm = 0; sm < 4; sm++)
            DE_UNROLL for (int f = 0; f < F; f++) {
                const float x = e[sm * F + f];
                const bool fin = __builtin_fabsf(x) < inf;
                const float c[3] = {fin ? x : inf, fin ? -x : inf, fin ? -__builtin_fabsf(x) : inf};
                DE_UNROLL for (int k = 0; k < 3; k++)
                    if (c[k] > v[k][f]) { v[k][f] = c[k]; at[k][f] = tile; }
            }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t j = groups << 2; j < N; j++) {
            const uint32_t tile = (uint32_t)(j >> tile_shift);
            DE_UNROLL for (int f = 0; f < F; f++) {
                const float x = X[f + (int64_t)F * j];
                const bool fin = __builtin_fabsf(x) < inf;
                const float c[3] = {fin ? x : inf, fin ? -x : inf, fin ? -__builtin_fabsf(x) : inf};
                DE_UNROLL for (int k = 0; k < 3; k++)
                    if (c[k] > v[k][f]) { v[k][f] = c[k]; at[k][f] = tile; }
            }
        }
    __shared__ unsigned long long best[4][3 * F];
    DE_UNROLL for (int f = 0; f < F; f++)
        DE_UNROLL for (int k = 0; k < 3; k++) {
            unsigned long long key = ((unsigned long long)orderable_f32(v[k][f]) << 32) | (unsigned long long)at[k][f];
            DE_UNROLL for (int m = 32; m >= 1; m >>= 1) {
                const unsigned long long o = __shfl_xor(key, m, 64);
                key = o > key ? o : key;
            }
            if ((threadIdx.x & 63) == 0) best[threadIdx.x >> 6][3 * f + k] = key;
        }
    __syncthreads();
    if ((int)threadIdx.x < 3 * F) {
        unsigned long long key = best[0][threadIdx.x];
        DE_UNROLL for (int w = 1; w < 4; w++) key = best[w][threadIdx.x] > key ? best[w][threadIdx.x] : key;
        unsigned long long *slot = keys + threadIdx.x;
        if (key > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, key);
    }
}

// The pre-pass of the priority tiles (de_tile_extremes_kernel) for X[F, N]: keys[3 F] = orderable(value) << 32 | (sample / DE_PRIO_UNIT).
hipError_t launch_tile_extremes(int dtype, const void *X, int64_t N, int64_t ldX, int F, void *keys, hipStream_t stream, const uint8_t *ok_init, uint8_t *ok,
                                int32_t n_ok) {
    if (!keys || F < 1 || F > DE_PRIO_MAX_F || N < 1) return hipErrorInvalidValue;
    if (!ok_init || !ok) n_ok = 0;
    // (the whole key array, 192 bytes: a multiple of 16 is ONE fill kernel, 3 F x 8 = 120 bytes were two)
    hipError_t st = hipMemsetAsync(keys, 0, (size_t)3 * DE_PRIO_MAX_F * sizeof(unsigned long long), stream);
    if (st != hipSuccess) return st;
    int shift = 0;
    while ((1 << shift) < DE_PRIO_UNIT) ++shift;
    const int64_t want = (N + 255) / 256;
    const dim3 grid((unsigned)(want < 2048 ? want : 2048));
    if (dtype == DE_F32 && ldX == F && N >= 4 && (reinterpret_cast<uintptr_t>(X) & 15u) == 0 && DE_PRIO_UNIT >= 4 && env_int("DE_PRIO_VEC", 1)) {
        const int64_t wantv = ((N >> 2) + 255) / 256;
        const dim3 gv((unsigned)(wantv < 2048 ? wantv : 2048));
        unsigned long long *k = static_cast<unsigned long long *>(keys);
        const float *x = static_cast<const float *>(X);
        switch (F) {
#define DE_TEV(FF) case FF: hipLaunchKernelGGL((de_tile_extremes_vec_kernel<FF>), gv, dim3(256), 0, stream, x, N, shift, k, ok_init, ok, n_ok); break;
            DE_TEV(1) DE_TEV(2) DE_TEV(3) DE_TEV(4) DE_TEV(5) DE_TEV(6) DE_TEV(7) DE_TEV(8)
#undef DE_TEV
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (dtype == DE_F32)
        hipLaunchKernelGGL((de_tile_extremes_kernel<float, DE_PRIO_MAX_F>), grid, dim3(256), 0, stream, static_cast<const float *>(X), N, ldX, F, shift,
                           static_cast<unsigned long long *>(keys), ok_init, ok, n_ok);
    else
        hipLaunchKernelGGL((de_tile_extremes_kernel<double, DE_PRIO_MAX_F>), grid, dim3(256), 0, stream, static_cast<const double *>(X), N, ldX, F, shift,
                           static_cast<unsigned long long *>(keys), ok_init, ok, n_ok);
    return hipGetLastError();
}
bool prio_tiles_wanted(int64_t N, int F, int64_t n_trees) {
    // the pre-pass costs a read of X; what it saves grows with the trees: at 10^7 samples 32 trees lose 13 %, 64 break even, 125 gain 6 %,
    // 250+ gain 13 % (tools/exp_prio_trees.py); with the probe launch it pays from 512 sample tiles on (1000 trees: 256 tiles +8 %, 512 -11 %,
    // 1024 -10 %, 4096 -13 %; tools/exp_prio_smallN.py)
    return F >= 1 && F <= DE_PRIO_MAX_F && n_trees >= env_int("DE_PRIO_MIN_TREES", 96) && (N + 255) / 256 >= env_int("DE_PRIO_MIN_TILES", 512) &&
           !env_int("DE_NO_PRIO_TILES", 0);
}

template <typename T>
static hipError_t launch_threaded_t(const EvalArgs &e, hipStream_t stream, const char **kname) {
    constexpr int VW = VecOf<T>::W;
    constexpr int TILE = TBLK * VW * TG<T>::G;
    KArgs<T> a;
    a.code = e.code;
    a.code_off = e.code_off;
    a.X = static_cast<const T *>(e.X);
    a.out = static_cast<T *>(e.out);
    a.ok = e.ok;
    a.params = static_cast<const T *>(e.params);
    a.classes = e.classes;
    a.N = e.N;
    a.ldX = e.ldX;
    a.ld_out = e.ld_out;
    a.ld_params = e.ld_params;
    a.prow_base = e.prow_base;
    a.n_prows = e.n_prows;
    a.n_tiles = (e.N + TILE - 1) / TILE;
    a.F = e.F;
    a.n_trees = e.n_trees;
    a.n_slots = e.n_slots;
    a.xstride = 0;
    a.classes_is_i64 = e.classes_is_i64;
    a.class_base = e.class_base;
    a.n_classes = e.n_classes > 0 ? e.n_classes : 1;
    a.vec_store = (reinterpret_cast<uintptr_t>(e.out) % 16 == 0 && (e.ld_out * sizeof(T)) % 16 == 0 && (uint64_t)e.ld_out * sizeof(T) < (1ull << 32)) ? 1 : 0; // (the fast ends of a tree take a 32-bit row stride)
    a.x_vec = (e.F >= 1 && e.ldX == e.F && reinterpret_cast<uintptr_t>(e.X) % 16 == 0 && (int64_t)TILE * e.F < 0x10000000LL) ? 1 : 0;
    a.f_magic = e.F > 1 ? (uint32_t)((0x100000000ull + (uint64_t)e.F - 1) / (uint64_t)e.F) : 0u;
    if (!env_int("DE_X_VEC", 1)) { a.x_vec = 0; a.f_magic = 0; } // scalar staging loop (A/B and the test of the vector path)
    if (env_int("DE_DEBUG_NO_STORE", 0)) a.vec_store = 2;
    // wave groups (KArgs::var_stride): parametric programs with staged parameter rows whose stream exists in e.waves variants
    const int WW = (TBLK == 64 && (e.waves == 2 || e.waves == 4 || e.waves == 8) && (!e.uses_params || e.n_prows > 0)) ? e.waves : 1;
    a.var_stride = WW > 1 ? e.var_stride : 0;
    int32_t tpc, nch, nc0;
    plan_chunks(e.n_trees, a.n_tiles, &nch, &tpc, &nc0, WW);
    a.trees_per_chunk = tpc;
    a.n_chunks = nch;
    a.live_idx = a.ctrl = nullptr;
    a.cert_max = nullptr;
    a.skip_flagged = (e.early_exit && e.skip_flagged && a.F + a.n_slots >= 1) ? 1 : 0;
    // Flag protocol (skip_flag_load, de_device_ops.h): 2 = through the caches + refresher tiles (default), 1 = agent scope for every access
    // The plain eval kernel writes 20+ GB per launch: flag lines leave the L1s / L2s all the time and protocol 2 is as good as 1 on the
    // headline (7.52 / 7.59 ms) and far better with few trees.  The fused-loss variant writes almost nothing: under protocol 2 a CU
    // keeps re-reading its stale L1 line (10.7 instead of 8.4 ms), so it uses protocol 1 unless its chunks are tiny.
    if (a.skip_flagged) { const int pr = env_int("DE_SKIP_PROTOCOL", (e.loss && tpc >= 8) ? 1 : 2); a.skip_flagged = pr >= 1 && pr <= 3 ? pr : 2; }
    // block order: chunk-fastest, or the chunk slower than a group of sample tiles (map_block_grouped) when there are several chunks
    const int64_t tiles8 = (a.n_tiles + 7) / 8;
    a.map_group = 0;
    if (a.n_tiles >= 64 && a.n_chunks > 1) {
        // Default: chunk-MAJOR (one group = all tiles) while the X tile is small beside the rows a workgroup writes (F <= trees per chunk / 8):
        // every workgroup in flight then walks the same ~10 KB of records — scalar-cache misses 12 % -> 1.3 % of the record loads
        // (tools/pmc_smem.sh), complete trees -2.5 %, fused loss -4 %, headline -1.2 % (tools/exp_map_group.sh; groups of 128 ... 2048 tiles
        // measured WORSE than either end) — and X is re-read from HBM once per chunk instead of once (+7 % traffic at F = 5, 16 chunks).
        // Wide X keeps the chunk-fastest order (its tile is re-served by the XCD's L2).
        const int64_t g = env_int("DE_MAP_GROUP", (int64_t)a.F * 8 <= (int64_t)a.trees_per_chunk * WW ? (int)(tiles8 > 0x3fffffff ? 0x3fffffff : tiles8) : 0);
        a.map_group = (int32_t)(g <= 0 ? 0 : (g > tiles8 ? tiles8 : g));
    }
    const int64_t tiles8g = a.map_group > 0 ? (tiles8 + a.map_group - 1) / a.map_group * a.map_group : tiles8;
    // The tail of the launch: under the chunk-major order the LAST workgroups all belong to the last chunk, and one workgroup of ~60 trees
    // runs ~100 us at full occupancy — a launch that fills the chip N times loses ~half a workgroup's time at its end (10^6 samples: ~9
    // fills, ~5 %).  The last chunk therefore runs as DE_TAIL_SPLIT (default 4) sub-chunks of >= 8 trees: the workgroups that finish the
    // launch are short.  Order of the trees' evaluation only; one wave per workgroup (the sub-chunks of an empty tail exit before staging).
    // Measured (same box, profiles/r5_ab_tail_split.txt): 10^6 samples (8 fills) -1.0 ... -1.5 %, 10^7 samples (80 fills) +0.8 % (the last
    // chunk's X tiles staged four times) — so only launches of <= DE_TAIL_SPLIT_FILLS (24) fills of the chip split their last chunk.
    a.tail_split = 1;
    const int64_t fills = (a.n_tiles * (int64_t)a.n_chunks) / ((int64_t)(cu_count() > 0 ? cu_count() : 256) * 21);
    if (TBLK == 64 && WW == 1 && a.map_group > 0 && a.n_chunks > 1 && fills <= env_int("DE_TAIL_SPLIT_FILLS", 24)) {
        const int ts = env_int("DE_TAIL_SPLIT", 4);
        const int most = tpc / 8; // (a sub-chunk keeps >= 8 trees when the chunk is full)
        a.tail_split = ts < 1 ? 1 : (ts > most ? (most < 1 ? 1 : most) : ts);
    }
    auto groups_of = [&](int64_t chunks) { return WW > 1 ? (chunks + WW - 1) / WW : chunks + a.tail_split - 1; }; // workgroups per sample tile
    int64_t blocks = tiles8g * 8 * groups_of(a.n_chunks);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    // priority tiles (de_tile_extremes_kernel): launches over >= 512 sample tiles and >= 96 trees with the early exit on; 3 F tiles, run
    // first and once more in place.  The pre-pass: a memset, one read of X, a dependent launch (0.11 ms at 10^7 samples)
    a.prio = nullptr;
    a.n_prio_blocks = a.n_prio = 0;
    bool ok_set = !e.ok_init; // e.ok_init: the flags' initial values, to be in e.ok before the first kernel reads them
    if (TBLK == 64 && a.skip_flagged && e.prio_keys && a.F >= 1 && prio_tiles_wanted(a.N, a.F, a.n_trees)) {
        const int np = 3 * a.F;
        hipError_t ps = e.prio_keys_ready ? hipSuccess
                                          : launch_tile_extremes(sizeof(T) == 4 ? DE_F32 : DE_F64, a.X, a.N, a.ldX, a.F, e.prio_keys, stream, e.ok_init, e.ok, e.n_trees);
        if (ps != hipSuccess) return ps;
        ok_set = !e.prio_keys_ready; // (the pre-pass, first kernel of the launch, has written the initial flags)
        const int tile_samples = TILE; // 512 / 128: a power of two
        a.prio_shift = 0;
        while ((DE_PRIO_UNIT << a.prio_shift) < tile_samples) ++a.prio_shift;
        a.prio = static_cast<const unsigned long long *>(e.prio_keys);
        a.n_prio = (uint32_t)np;
        a.n_prio_blocks = (uint32_t)(((int64_t)np * (WW > 1 ? groups_of(a.n_chunks) : (int64_t)a.n_chunks) + 7) / 8 * 8);
        if (!env_int("DE_PRIO_PROBE", 1)) { // (the priority tiles as the first workgroups of the ONE launch: they index the plain chunks)
            a.tail_split = 1;
            blocks = tiles8g * 8 * groups_of(a.n_chunks);
        }
        blocks += a.n_prio_blocks;
        if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    }
    if (!ok_set) {
        const hipError_t cs = hipMemcpyAsync(e.ok, e.ok_init, (size_t)e.n_trees, hipMemcpyDeviceToDevice, stream);
        if (cs != hipSuccess) return cs;
    }
    // rows: X, spill slots, then (parametric) the class row [+ the table-pointer row for Float32] of h_param
    a.cls_row_off = (uint32_t)((size_t)(a.F + a.n_slots) * RowOf<T>::BYTES);
    const int prm_rows = (e.uses_params && e.n_prows == 0) ? (sizeof(T) == 4 ? 2 : 1) : 0; // (the class row of h_param; staged parameter rows count as slots)
    if (e.uses_params && (uint64_t)e.ld_params * (uint64_t)e.n_classes * sizeof(T) > 0xFFFFFFFFull) return hipErrorInvalidValue; // 32-bit column offsets
    // (a wave group: the other waves' slot rows behind the parameter rows, then their live-tree lists)
    const size_t row_bytes_all = (size_t)(a.F + a.n_slots + prm_rows + (WW - 1) * e.wave_slots + env_int("DE_EXTRA_LDS_ROWS", 0)) * RowOf<T>::BYTES;
    a.list_off = (uint32_t)(DE_SKIPLIST_BYTES + row_bytes_all);
    const size_t lds = row_bytes_all + (size_t)WW * DE_SKIPLIST_BYTES;
    if (WW > 1 && (lds >> 4) > HF_LIST_MASK) return hipErrorInvalidValue;
    void (*kern)(const KArgs<T>) = e.uses_params ? de_eval_threaded_kernel<T, true> : de_eval_threaded_kernel<T, false>;
    if constexpr (TBLK == 64) { // (a build with wider tiles, -DDE_TBLK=128 for an A/B, has no wave groups: WW == 1 above)
        if (WW == 2) kern = e.uses_params ? de_eval_threaded_kernel<T, true, false, 2> : de_eval_threaded_kernel<T, false, false, 2>;
        if (WW == 4) kern = e.uses_params ? de_eval_threaded_kernel<T, true, false, 4> : de_eval_threaded_kernel<T, false, false, 4>;
        if (WW == 8) kern = e.uses_params ? de_eval_threaded_kernel<T, true, false, 8> : de_eval_threaded_kernel<T, false, false, 8>;
    }
    a.y = a.w = nullptr;
    a.partial = nullptr;
    a.loss_kind = 0;
    if (e.loss && DE_NO_LOSS) return hipErrorInvalidValue; // (a build without the loss arguments)
    if (e.loss) {
        kern = e.uses_params ? de_eval_threaded_kernel<T, true, true> : de_eval_threaded_kernel<T, false, true>;
        if constexpr (TBLK == 64) {
            if (WW == 2) kern = e.uses_params ? de_eval_threaded_kernel<T, true, true, 2> : de_eval_threaded_kernel<T, false, true, 2>;
            if (WW == 4) kern = e.uses_params ? de_eval_threaded_kernel<T, true, true, 4> : de_eval_threaded_kernel<T, false, true, 4>;
            if (WW == 8) kern = e.uses_params ? de_eval_threaded_kernel<T, true, true, 8> : de_eval_threaded_kernel<T, false, true, 8>;
        }
        a.y = static_cast<const T *>(e.loss->y);
        a.w = static_cast<const T *>(e.loss->w);
        a.partial = static_cast<T *>(e.loss->partial);
        a.loss_kind = e.loss->kind;
    }
    if (kname) *kname = "de_eval_threaded_kernel";
    if (lds > 64 * 1024) {
        hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    if (a.n_prio && env_int("DE_PRIO_PROBE", 1)) {
        // the priority tiles as a launch of their OWN in front, in chunks of DE_PRIO_PROBE_TPC trees (short workgroups, many of them): when the
        // launch proper starts, the flags are down for its very first workgroups too (no blind first wave: 9 % of the tiles at 10^6 samples)
        KArgs<T> pa = a;
        pa.tail_split = 1;
        pa.trees_per_chunk = env_int("DE_PRIO_PROBE_TPC", 8);
        pa.trees_per_chunk = pa.trees_per_chunk < 1 ? 1 : (pa.trees_per_chunk > 64 ? 64 : pa.trees_per_chunk); // (a divisor, and the skip mask has 64 bits)
        pa.n_chunks = (a.n_trees + pa.trees_per_chunk - 1) / pa.trees_per_chunk;
        pa.n_prio_blocks = (uint32_t)(((int64_t)pa.n_prio * (WW > 1 ? ((int64_t)pa.n_chunks + WW - 1) / WW : (int64_t)pa.n_chunks) + 7) / 8 * 8);
        hipLaunchKernelGGL(kern, dim3(pa.n_prio_blocks), dim3(TBLK * WW), lds, stream, pa);
        const hipError_t ps = hipGetLastError();
        if (ps != hipSuccess) return ps;
        blocks -= a.n_prio_blocks;
        a.n_prio_blocks = a.n_prio = 0;
        a.prio = nullptr;
        // ... and the launch proper runs DENSE chunks over the trees that are still live (de_compact_live_kernel re-links their records
        // into e.compact_code; chunk plan on the device: the grid below is the host's upper bound, workgroups beyond the device's plan exit)
        if (e.compact_code && e.compact_ints && tpc <= 64 && env_int("DE_COMPACT", 1)) {
            int32_t *coff = e.compact_ints, *live_idx = coff + (size_t)e.n_trees + 1, *ctrl = live_idx + e.n_trees;
            const int64_t want_blocks = (int64_t)cu_count() * 4 * 8;
            const int32_t tpc_max = plan_tpc_max(WW);
            LossEnds le;
            std::memset(&le, 0, sizeof le);
            if (e.loss && env_int("DE_LOSS_PLAIN_ENDS", 1)) {
                uint64_t table[TOPX_TABLE];
                const hipError_t ts = eval_handler_table(sizeof(T) == 4 ? DE_F32 : DE_F64, e.turbo, table);
                if (ts != hipSuccess) return ts;
                le.end = table[TOPX_END];
                le.slow_end = table[TOPX_AUX_BASE + 0];
                for (uint32_t k = 0; k < TOPX_ENDV_COUNT; k++) {
                    le.endv[k] = table[TOPX_ENDV_BASE + k];
                    le.plain[k] = k < 12 ? table[BOP_BIN_BASE + 4 * (k / 2) + ((k & 1) ? 3 : 1)] : table[BOP_UN_BASE + 4 * (k - 12) + 1]; // (de_bind.h topx_endv_of, inverted)
                }
            }
            if (sizeof(T) == 4)
                hipLaunchKernelGGL(de_compact_live_kernel<true>, dim3(a.var_stride ? WW : 1), dim3(1024), 0, stream, reinterpret_cast<const U32x4 *>(e.code), e.code_off, e.ok,
                                   e.n_trees, reinterpret_cast<U32x4 *>(e.compact_code), coff, live_idx, ctrl, a.n_tiles, want_blocks, tpc_max, le, a.var_stride);
            else
                hipLaunchKernelGGL(de_compact_live_kernel<false>, dim3(a.var_stride ? WW : 1), dim3(1024), 0, stream, reinterpret_cast<const U32x4 *>(e.code), e.code_off, e.ok,
                                   e.n_trees, reinterpret_cast<U32x4 *>(e.compact_code), coff, live_idx, ctrl, a.n_tiles, want_blocks, tpc_max, le, a.var_stride);
            const hipError_t cs = hipGetLastError();
            if (cs != hipSuccess) return cs;
            a.code = static_cast<const BoundInstr *>(e.compact_code);
            a.code_off = coff;
            a.live_idx = live_idx;
            a.ctrl = ctrl;
            if (e.compacted) *e.compacted = true;
            blocks = tiles8g * 8 * groups_of(nc0);
            if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(TBLK * WW), lds, stream, a);
    hipError_t st = hipGetLastError();
    if (st != hipSuccess || !e.loss) return st;
    int32_t n_segs = 1;
    st = launch_loss_reduce_tiles(sizeof(T) == 4 ? DE_F32 : DE_F64, a.partial, (int64_t)e.n_trees * TWAVES, a.n_tiles, e.loss->seg_sum, &n_segs, stream);
    if (st != hipSuccess) return st;
    hipLaunchKernelGGL(de_loss_finish_kernel<T>, dim3((unsigned)((e.n_trees + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const double *>(e.loss->seg_sum), (int64_t)e.n_trees, n_segs, e.ok, static_cast<T *>(e.loss->loss));
    return hipGetLastError();
}

hipError_t launch_loss_reduce_tiles(int dtype, const void *partial, int64_t n_cols, int64_t n_tiles, void *seg_sum,
                                    int32_t *n_segs_out, hipStream_t stream) {
    const int32_t n_segs = loss_segments(n_tiles);
    const int64_t tps = (n_tiles + n_segs - 1) / n_segs;
    const dim3 grid((unsigned)((n_cols + 255) / 256), (unsigned)n_segs);
    if (dtype == DE_F32)
        hipLaunchKernelGGL(de_loss_reduce_tiles_kernel<float>, grid, dim3(256), 0, stream, static_cast<const float *>(partial), n_cols,
                           n_tiles, tps, static_cast<double *>(seg_sum));
    else
        hipLaunchKernelGGL(de_loss_reduce_tiles_kernel<double>, grid, dim3(256), 0, stream, static_cast<const double *>(partial), n_cols,
                           n_tiles, tps, static_cast<double *>(seg_sum));
    *n_segs_out = n_segs;
    return hipGetLastError();
}

int32_t loss_segments(int64_t n_tiles) { return (int32_t)(n_tiles < 64 ? (n_tiles < 1 ? 1 : n_tiles) : 64); }
void loss_scratch_bytes(int dtype, int64_t n_trees, int64_t N, size_t *partial_bytes, size_t *seg_bytes) {
    const int64_t tile = ttile_samples(dtype);
    const int64_t n_tiles = (N + tile - 1) / tile;
    *partial_bytes = (size_t)n_tiles * (size_t)n_trees * TWAVES * (dtype == DE_F32 ? 4 : 8);
    *seg_bytes = (size_t)loss_segments(n_tiles) * (size_t)n_trees * TWAVES * sizeof(double);
}

// ---- multi-GPU flag exchange (de_dist.cpp): the two re-orderings around the one ncclAllGather of a step.  Rank r owns trees r, r + world,
// ...; it sends ceil(n / world) bytes (its flags, padded with 1 = complete) and receives world such blocks: entry [r][i] of the gathered
// buffer is tree r + i * world.  One tiny launch each instead of a memset + a copy in front and `world` strided 1-byte-row copies behind.
__global__ void __launch_bounds__(256) de_dist_pack_kernel(uint8_t *__restrict__ send, const uint8_t *__restrict__ ok_local, int64_t mine, int64_t per) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < per) send[i] = i < mine ? ok_local[i] : (uint8_t)1;
}
__global__ void __launch_bounds__(256) de_dist_unpack_kernel(uint8_t *__restrict__ ok_global, const uint8_t *__restrict__ recv, int64_t per, int32_t world, int64_t n_trees) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n_trees) ok_global[t] = recv[(t % world) * per + t / world];
}
// ---- constant subtrees, evaluated once per set of constants (round 6) ------------------------------------------------------------
// One thread per constant subtree ("fold"): a stack machine over the subtree's post-order tape slice, with the operator code of the
// threaded kernel's handlers (un_apply for cos / exp / sin, IEEE + - * /, cold_op / cold_op3 for the rest: bit for bit what the eval kernel
// computes for the same operator and argument), every node's output validity-tested as dispatch_constant_tree does
// (src/Evaluate.jl:1002-1067).
// Replaces the auxiliary PROGRAM of rounds 1-5 (a second population lowered, bound, threaded, uploaded and evaluated with N = 1: a
// quarter of de_program_create) for every subtree whose evaluation stack fits DE_FOLD_STACK values.
template <typename T>
__global__ void __launch_bounds__(64) de_fold_kernel(const de_tape_node_t *__restrict__ nodes, const int64_t *__restrict__ noff,
                                                     const int64_t *__restrict__ coff, const T *__restrict__ cvals, int64_t n_folds,
                                                     T *__restrict__ out, uint8_t *__restrict__ ok) {
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    constexpr int G = 1;
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= n_folds) return;
    V st[DE_FOLD_STACK];
    uint32_t leaf_bits = 0; // bit k: stack entry k is a constant LEAF (not an operator's result)
    int sp = 0;
    bool good = true;
    const int64_t c0 = coff[j];
    for (int64_t i = noff[j]; i < noff[j + 1]; i++) {
        const de_tape_node_t nd = nodes[i];
        V acc[G];
        if (nd.degree == 0) {
            const T c = cvals[c0 + nd.arg];
            FOR_I acc[0][i] = c;
            leaf_bits |= 1u << sp;
        } else if (nd.degree == 1) {
            V x_[G];
            x_[0] = st[--sp];
            // cos / exp / sin OF A CONSTANT LEAF: the binder has no hot form for that operand kind (BOP_GEN_CONST -> cold_op -> the math
            // library's function), so the eval kernel computes cos(c) with OCML's cosf and cos(x) with the fast polynomial — both within
            // 2 ulp, not the same bits (cos(-0.456): ...d1 against ...d2).  A folded subtree carries what the eval kernel would compute.
            const bool of_leaf = ((leaf_bits >> sp) & 1u) != 0;
            leaf_bits &= ~(1u << sp);
            const uint32_t op = nd.op;
            // cos / exp / sin: the functions of the THREADED kernel's hot handlers (un_apply: the vectorised trig, the direct exp with its
            // per-element select) — the flat-switch kernel's scalar forms differ from them in the last bit now and then
            // (tests/test_gpu_round6.py found cos(-0.456...): 2 of 300 trees), and a folded subtree must carry the bits the eval kernel
            // would have computed.  Everything else: cold_op, which the threaded kernel's generic handlers call too.
            if (!of_leaf && op == DE_U_COS) acc[0] = un_apply<T, 0>(x_[0]);
            else if (!of_leaf && op == DE_U_EXP) acc[0] = un_apply<T, 1>(x_[0]);
            else if (!of_leaf && op == DE_U_SIN) acc[0] = un_apply<T, 2>(x_[0]);
            else { acc[0] = x_[0]; COLD_CALL(x_) }
        } else if (nd.degree == 2) {
            V b[G];
            b[0] = st[--sp];
            acc[0] = st[--sp];
            leaf_bits &= ~(3u << sp);
            const uint32_t op = nd.op;
            if (op == DE_B_ADD) { FOR_I acc[0][i] = acc[0][i] + b[0][i]; }
            else if (op == DE_B_SUB) { FOR_I acc[0][i] = acc[0][i] - b[0][i]; }
            else if (op == DE_B_MUL) { FOR_I acc[0][i] = acc[0][i] * b[0][i]; }
            else if (op == DE_B_DIV) { FOR_I acc[0][i] = acc[0][i] / b[0][i]; }
            else COLD_CALL(b)
        } else {
            VG<T, G> av, bv, cv;
            av.v[0] = st[--sp]; // third argument
            cv.v[0] = st[--sp];
            bv.v[0] = st[--sp];
            leaf_bits &= ~(7u << sp);
            av = cold_op3<T, G>(nd.op, av, bv, cv);
            acc[0] = av.v[0];
        }
        good = good && M<T>::isfinite(acc[0][0]);
        st[sp++] = acc[0];
    }
    out[j] = st[0][0];
    ok[j] = good ? 1 : 0;
}
hipError_t launch_fold(int dtype, const void *nodes, const int64_t *noff, const int64_t *coff, const void *cvals, int64_t n_folds, void *out,
                       uint8_t *ok, hipStream_t stream) {
    if (n_folds <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_folds + 63) / 64)), block(64);
    if (dtype == DE_F32)
        hipLaunchKernelGGL(de_fold_kernel<float>, grid, block, 0, stream, static_cast<const de_tape_node_t *>(nodes), noff, coff,
                           static_cast<const float *>(cvals), n_folds, static_cast<float *>(out), ok);
    else
        hipLaunchKernelGGL(de_fold_kernel<double>, grid, block, 0, stream, static_cast<const de_tape_node_t *>(nodes), noff, coff,
                           static_cast<const double *>(cvals), n_folds, static_cast<double *>(out), ok);
    return hipGetLastError();
}

hipError_t launch_dist_pack(uint8_t *send, const uint8_t *ok_local_dev, int64_t mine, int64_t per, hipStream_t stream) {
    if (per <= 0) return hipSuccess;
    hipLaunchKernelGGL(de_dist_pack_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, stream, send, ok_local_dev, mine, per);
    return hipGetLastError();
}
hipError_t launch_dist_unpack(uint8_t *ok_global_dev, const uint8_t *recv, int64_t per, int32_t world, int64_t n_trees, hipStream_t stream) {
    if (n_trees <= 0) return hipSuccess;
    hipLaunchKernelGGL(de_dist_unpack_kernel, dim3((unsigned)((n_trees + 255) / 256)), dim3(256), 0, stream, ok_global_dev, recv, per, world, n_trees);
    return hipGetLastError();
}

hipError_t launch_eval(int dtype, const EvalArgs &a, hipStream_t stream, const char **kernel_name) {
    if (a.ok_init && !(a.threaded && !a.cert_max && !a.direct)) { // (the threaded launch sets the flags itself: fused into its pre-pass when that runs)
        const hipError_t cs = hipMemcpyAsync(a.ok, a.ok_init, (size_t)a.n_trees, hipMemcpyDeviceToDevice, stream);
        if (cs != hipSuccess) return cs;
    }
    if (a.cert_max && !a.direct) return dtype == DE_F32 ? launch_eval_t<float, 1, 256>(a, stream, kernel_name) : launch_eval_t<double, 1, 256>(a, stream, kernel_name);
    if (a.direct) return dtype == DE_F32 ? launch_eval_t<float, 1, 256>(a, stream, kernel_name) : launch_eval_t<double, 1, 256>(a, stream, kernel_name);
    if (a.threaded) return dtype == DE_F32 ? launch_threaded_t<float>(a, stream, kernel_name) : launch_threaded_t<double>(a, stream, kernel_name);
    int G, BLK;
    eval_geometry(dtype, &G, &BLK);
    if (dtype == DE_F32) return launch_eval_geo<float>(a, stream, kernel_name, G, BLK);
    return launch_eval_t<double, 1, 256>(a, stream, kernel_name);
}

} // namespace de
