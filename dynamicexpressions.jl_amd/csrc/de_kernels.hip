// de_kernels.hip — gfx950 (CDNA4 / MI355X) kernels for batched expression-tree
// evaluation.  Replaces the inner loops of the reference's src/Evaluate.jl
// (deg*_eval and the fused deg1_l*/deg2_* kernels, :366-993) and
// src/EvaluateDerivative.jl (grad_degn_eval :340-365, diff_degn_eval :99-119).
//
// Design:
//  * grid  = sample tiles x tree chunks; a workgroup (256 threads = 4 wave64)
//    owns TILE = 256*K consecutive samples and loops over a chunk of trees.
//  * The X tile ([F, TILE], feature-fastest in HBM) is read ONCE per workgroup with
//    fully coalesced loads and transposed into LDS as xs[f][sample], so a leaf
//    read is one conflict-free ds_read_b128 per thread (K=4 f32 samples).
//  * Each tree is a wave-uniform accumulator program (de_program.h): instruction
//    words are fetched through the SCALAR cache (constant address space ->
//    s_load_dwordx4, next instruction prefetched while the current one executes),
//    decode and dispatch run on the scalar unit, the VALU only sees operator
//    arithmetic on K independent samples per lane.  Intermediates live in
//    registers; only the rare both-children-are-subtrees case spills one value
//    per sample to LDS.
//  * NaN/Inf flag: per-lane predicate, one wavefront ballot per tree, one byte
//    store per failing wave (no atomics).
//  * blockIdx -> (tile, chunk) is XCD-aware: all chunks of one X tile run on the
//    same XCD, so the tile is fetched from HBM once and re-served by that XCD's L2.
//  * No MFMA: this is an elementwise map, not a contraction.
#include <hip/hip_runtime.h>

#include <cstdlib>

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
    const Instr *code;       // padded with one trailing instruction (prefetch reads pc+1)
    const int32_t *code_off; // n_trees + 1
    const T *X;
    T *out;
    uint8_t *ok;
    const T *params;
    const void *classes;
    int64_t N, ldX, ld_out, ld_params, n_tiles;
    int32_t F, n_trees, trees_per_chunk, n_chunks, n_slots, xstride;
    int32_t classes_is_i64, class_base, vec_store;
};

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

// Everything that is not on the fast path of the interpreter loop.
template <typename T, int G, typename V>
__device__ __forceinline__ void apply_cold_op(uint32_t op, V (&acc)[G], const V (&b)[G]) {
    using m = M<T>;
    constexpr int VW = VecOf<T>::W;
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
}

// acc = op3(b, c, acc): b, c from spill slots, acc = third argument
template <typename T, int G, typename V>
__device__ __forceinline__ void apply_op3(uint32_t op, V (&acc)[G], const V (&b)[G], const V (&c)[G]) {
    using m = M<T>;
    constexpr int VW = VecOf<T>::W;
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
}

// Validity accumulation without touching the scalar unit: poison = fma(v, 0, poison) stays
// +0 while every tested value is finite and turns (and stays) NaN at the first Inf/NaN.
template <typename T, int G, typename V>
__device__ __forceinline__ void poison_with(T &poison, const V (&v)[G]) {
    constexpr int VW = VecOf<T>::W;
    FOR_G FOR_I poison = M<T>::fma(v[g][i], T(0), poison);
}

// cos/sin/exp over the G*VW samples of a thread.  Float32 uses the fast versions of
// de_device_ops.h with ONE divergent fix-up region for out-of-range arguments.
template <typename T, int G, typename V, bool SIN>
__device__ __forceinline__ void vec_trig(V (&out)[G], const V (&x)[G]) {
    constexpr int VW = VecOf<T>::W;
    if constexpr (sizeof(T) == 4) {
        bool big = false;
        V r[G];
        FOR_G FOR_I {
            r[g][i] = fast_trig_f32<SIN>(x[g][i]);
            big |= M<T>::abs(x[g][i]) > DE_TRIG_FAST_BOUND;
        }
        if (big) {
            FOR_G FOR_I if (M<T>::abs(x[g][i]) > DE_TRIG_FAST_BOUND) r[g][i] = SIN ? M<T>::sin(x[g][i]) : M<T>::cos(x[g][i]);
        }
        FOR_G out[g] = r[g];
    } else {
        V r[G];
        FOR_G FOR_I r[g][i] = SIN ? M<T>::sin(x[g][i]) : M<T>::cos(x[g][i]);
        FOR_G out[g] = r[g];
    }
}
template <typename T, int G, typename V>
__device__ __forceinline__ void vec_exp(V (&out)[G], const V (&x)[G]) {
    constexpr int VW = VecOf<T>::W;
    V r[G];
    if constexpr (sizeof(T) == 4) { FOR_G FOR_I r[g][i] = fast_exp_f32(x[g][i]); }
    else { FOR_G FOR_I r[g][i] = M<T>::exp(x[g][i]); }
    FOR_G out[g] = r[g];
}

// The operators of the headline workload, expanded once per operand source so that a
// constant operand stays in an SGPR and a unary operator works on acc in place (no
// v_mov traffic).  IN(g,i) yields operand B of sample (g,i).
#define DE_FAST_BINARY(IN)                                                               \
    if (op == DE_B_ADD) { FOR_G FOR_I acc[g][i] = acc[g][i] + IN(g, i); }                \
    else if (op == DE_B_MUL) { FOR_G FOR_I acc[g][i] = acc[g][i] * IN(g, i); }           \
    else if (op == DE_B_SUB) { FOR_G FOR_I acc[g][i] = acc[g][i] - IN(g, i); }           \
    else if (op == DOP_RSUB) { FOR_G FOR_I acc[g][i] = IN(g, i) - acc[g][i]; }           \
    else if (op == DE_B_DIV) { FOR_G FOR_I acc[g][i] = acc[g][i] / IN(g, i); }           \
    else if (op == DOP_RDIV) { FOR_G FOR_I acc[g][i] = IN(g, i) / acc[g][i]; }
#define DE_FAST_UNARY(IN)                                                                \
    if (op == DE_U_COS) { V x_[G]; FOR_G FOR_I x_[g][i] = IN(g, i); vec_trig<T, G, V, false>(acc, x_); } \
    else if (op == DE_U_EXP) { V x_[G]; FOR_G FOR_I x_[g][i] = IN(g, i); vec_exp<T, G, V>(acc, x_); }    \
    else if (op == DE_U_SIN) { V x_[G]; FOR_G FOR_I x_[g][i] = IN(g, i); vec_trig<T, G, V, true>(acc, x_); }

// XCD-aware block mapping: hardware dispatches block b to XCD b % 8 (observed, used
// for L2 affinity only — correctness never depends on it).  All chunks of a sample
// tile get block ids with the same residue, i.e. run on one XCD back to back.
struct TileMap {
    int64_t tile;
    int32_t chunk;
    bool valid;
};
__device__ __forceinline__ TileMap map_block(uint32_t bid, int32_t n_chunks, int64_t n_tiles) {
    const uint32_t xcd = bid & 7u, idx = bid >> 3;
    TileMap m;
    m.chunk = (int32_t)(idx % (uint32_t)n_chunks);
    m.tile = (int64_t)(idx / (uint32_t)n_chunks) * 8 + xcd;
    m.valid = m.tile < n_tiles;
    return m;
}

template <typename T, int G, int BLK, bool EE, bool PARAMS>
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
    {
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
    int64_t cls[G][VW];
    if (PARAMS) {
        FOR_G FOR_I {
            int64_t jj = base + g * GT + tid * VW + i;
            jj = jj < last ? jj : last;
            cls[g][i] = (a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj]
                                          : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj]) -
                        a.class_base;
        }
    }
    __syncthreads();

    const ConstU4Ptr code = (ConstU4Ptr)(uintptr_t)a.code;
    const ConstI32Ptr code_off = (ConstI32Ptr)(uintptr_t)a.code_off;
    const int t0 = tm.chunk * a.trees_per_chunk;
    const int t1 = (t0 + a.trees_per_chunk < a.n_trees) ? t0 + a.trees_per_chunk : a.n_trees;
    const bool full = base + TILE <= a.N;
    const int F = a.F;

    int pe = code_off[t0];
    for (int tree = t0; tree < t1; ++tree) {
        int pc = pe;
        pe = code_off[tree + 1];
        V acc[G];
        FOR_G FOR_I acc[g][i] = T(0);
        T poison = T(0);
        U32x4 nxt = code[pc]; // scalar load; a tree has at least one instruction
        for (; pc < pe; ++pc) {
            const U32x4 w = nxt;
            nxt = code[pc + 1]; // prefetch (the code buffer carries one trailing pad instruction)
            const uint32_t hdr = w.x;
            const uint32_t op = hdr & H_OP_MASK;
            const uint32_t src = (hdr >> H_SRC_SHIFT) & H_SRC_MASK;
            if (hdr & H_PUSH) {
                V *__restrict__ s = rowsv + (F + ((hdr >> H_PUSH_SHIFT) & H_SLOT_MASK)) * ROWV + tid;
                FOR_G s[g * BLK] = acc[g];
            }
            V inj[G]; // input of a fused deg1 operator (early_exit=false only)
            if (!EE && (hdr & H_INJECT)) { FOR_G inj[g] = acc[g]; }
            if (src == SRC_ROW) {
                V b[G];
                const V *__restrict__ s = rowsv + (w.y & 0xFFFFu) * ROWV + tid;
                FOR_G b[g] = s[g * BLK];
                if (EE && (hdr & H_CHECK_B)) poison_with<T, G, V>(poison, b);
                if (!EE && (hdr & H_INJECT)) { FOR_G inj[g] = b[g]; }
#define IN_ROW(g, i) b[g][i]
                if (op == DOP_LOAD) { FOR_G acc[g] = b[g]; }
                else DE_FAST_BINARY(IN_ROW)
                else DE_FAST_UNARY(IN_ROW)
                else if (op >= DE_T_FMA && op < DOP_LOAD) {
                    V c[G];
                    const V *__restrict__ s2 = rowsv + (F + ((hdr >> H_POPC_SHIFT) & H_SLOT_MASK)) * ROWV + tid;
                    FOR_G c[g] = s2[g * BLK];
                    apply_op3<T, G, V>(op, acc, b, c);
                } else apply_cold_op<T, G, V>(op, acc, b);
            } else if (src == SRC_CONST) {
                const T c = imm_of<T>(w.z, w.w);
                if (!EE && (hdr & H_INJECT)) { FOR_G FOR_I inj[g][i] = c; }
#define IN_CONST(g, i) c
                if (op == DOP_LOAD) { FOR_G FOR_I acc[g][i] = c; }
                else DE_FAST_BINARY(IN_CONST)
                else {
                    V b[G];
                    FOR_G FOR_I b[g][i] = c;
                    apply_cold_op<T, G, V>(op, acc, b);
                }
            } else if (PARAMS && src == SRC_PARAM) {
                V b[G];
                const T *__restrict__ s = a.params + (w.y & 0xFFFFu);
                FOR_G FOR_I b[g][i] = s[a.ld_params * cls[g][i]];
                if (EE && (hdr & H_CHECK_B)) poison_with<T, G, V>(poison, b);
                if (!EE && (hdr & H_INJECT)) { FOR_G inj[g] = b[g]; }
                if (op == DOP_LOAD) { FOR_G acc[g] = b[g]; }
                else DE_FAST_BINARY(IN_ROW)
                else apply_cold_op<T, G, V>(op, acc, b);
            } else { // SRC_ACC: a unary operator applied to the accumulator in place
#define IN_ACC(g, i) acc[g][i]
                DE_FAST_UNARY(IN_ACC)
                else {
                    V b[G];
                    FOR_G b[g] = acc[g];
                    apply_cold_op<T, G, V>(op, acc, b);
                }
            }
            if (op != DOP_LOAD) {
                if (!EE && (hdr & H_INJECT)) { // is_valid(x_l) ? op(x_l) : Inf  (src/Evaluate.jl:722)
                    FOR_G FOR_I if (!M<T>::isfinite(inj[g][i])) acc[g][i] = M<T>::inf();
                }
                if (hdr & (EE ? H_CHECK_OUT : H_CHECK_ALWAYS)) poison_with<T, G, V>(poison, acc);
            }
        }
        // ---- store out[tree][...]: one 16-byte store per group, coalesced over the wave
        T *__restrict__ o = a.out + (int64_t)tree * a.ld_out + base + tid * VW;
        if (full && a.vec_store) {
            FOR_G *reinterpret_cast<V *>(o + g * GT) = acc[g];
        } else {
            FOR_G FOR_I if (base + g * GT + tid * VW + i < a.N) o[g * GT + i] = acc[g][i];
        }
        // ---- completion flag: one ballot per wave, one byte store per failing wave
        if (__ballot(poison != poison) != 0ull && (tid & 63) == 0) a.ok[tree] = 0;
    }
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
    (void)dtype;
}

size_t eval_lds_bytes(int dtype, int F, int n_slots, int *K_out) {
    int G, BLK;
    eval_geometry(dtype, &G, &BLK);
    const size_t rowv = (size_t)BLK * G + 1;
    const size_t bytes = (size_t)(F + n_slots) * rowv * 16;
    if (K_out) *K_out = (dtype == DE_F32 ? 4 : 2) * G;
    return bytes <= 160 * 1024 ? bytes : 0;
}

static int g_cu_count = 0;

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
    a.n_tiles = (e.N + TILE - 1) / TILE;
    a.F = e.F;
    a.n_trees = e.n_trees;
    a.n_slots = e.n_slots;
    a.xstride = 0;
    a.classes_is_i64 = e.classes_is_i64;
    a.class_base = e.class_base;
    a.vec_store = (reinterpret_cast<uintptr_t>(e.out) % 16 == 0 && (e.ld_out * sizeof(T)) % 16 == 0) ? 1 : 0;

    if (g_cu_count == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_cu_count = prop.multiProcessorCount;
        if (g_cu_count <= 0) g_cu_count = 256;
    }
    // Tree chunking: chunks of ~64 trees keep workgroups short (fine-grained tail) while
    // the X-tile staging (one L2 read of the tile per chunk) stays a few percent of the work;
    // with few sample tiles, split further so the grid still covers the chip several times.
    int64_t n_chunks = (e.n_trees + 63) / 64;
    const int64_t want_blocks = (int64_t)g_cu_count * 4 * 8;
    if (a.n_tiles * n_chunks < want_blocks) n_chunks = (want_blocks + a.n_tiles - 1) / a.n_tiles;
    const int64_t max_chunks = (e.n_trees + 7) / 8; // >= 8 trees per chunk
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    if (n_chunks < 1) n_chunks = 1;
    a.trees_per_chunk = (int32_t)((e.n_trees + n_chunks - 1) / n_chunks);
    a.n_chunks = (int32_t)((e.n_trees + a.trees_per_chunk - 1) / a.trees_per_chunk);

    const int64_t tile_groups = (a.n_tiles + 7) / 8;
    const int64_t blocks = tile_groups * 8 * a.n_chunks;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds = (size_t)(a.F + a.n_slots) * ((size_t)BLK * G + 1) * 16;

    void (*kern)(const KArgs<T>);
    if (e.early_exit) kern = e.uses_params ? de_eval_tape_kernel<T, G, BLK, true, true> : de_eval_tape_kernel<T, G, BLK, true, false>;
    else kern = e.uses_params ? de_eval_tape_kernel<T, G, BLK, false, true> : de_eval_tape_kernel<T, G, BLK, false, false>;
    if (kname) *kname = "de_eval_tape_kernel";
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

hipError_t launch_eval(int dtype, const EvalArgs &a, hipStream_t stream, const char **kernel_name) {
    int G, BLK;
    eval_geometry(dtype, &G, &BLK);
    if (dtype == DE_F32) return launch_eval_geo<float>(a, stream, kernel_name, G, BLK);
    return launch_eval_geo<double>(a, stream, kernel_name, G, BLK);
}

} // namespace de
