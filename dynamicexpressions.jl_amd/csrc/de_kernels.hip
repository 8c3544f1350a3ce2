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

template <typename T, int K> struct VecOf;
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };

template <typename T> __device__ __forceinline__ T imm_of(uint32_t w2, uint32_t w3);
template <> __device__ __forceinline__ float imm_of<float>(uint32_t w2, uint32_t) { return __uint_as_float(w2); }
template <> __device__ __forceinline__ double imm_of<double>(uint32_t w2, uint32_t w3) {
    return __longlong_as_double((long long)(((unsigned long long)w3 << 32) | w2));
}

#define DE_UNROLL _Pragma("unroll")

// acc = f(b) for every sample
#define U_CASE(OPC, EXPR)                                    \
    case OPC:                                                \
        DE_UNROLL for (int i = 0; i < K; i++) {              \
            const T x = b[i];                                \
            acc[i] = (EXPR);                                 \
        }                                                    \
        break;
// acc = f(acc, b)
#define B_CASE(OPC, EXPR)                                    \
    case OPC:                                                \
        DE_UNROLL for (int i = 0; i < K; i++) {              \
            const T x = acc[i], y = b[i];                    \
            acc[i] = (EXPR);                                 \
        }                                                    \
        break;

// Everything that is not on the fast path of the interpreter loop.
template <typename T, int K, typename V>
__device__ __forceinline__ void apply_cold_op(uint32_t op, V &acc, const V &b) {
    using m = M<T>;
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
        DE_UNROLL for (int i = 0; i < K; i++) {
            const T c = m::cos(b[i]);
            acc[i] = c * c;
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
template <typename T, int K, typename V>
__device__ __forceinline__ void apply_op3(uint32_t op, V &acc, const V &b, const V &c) {
    using m = M<T>;
    DE_UNROLL for (int i = 0; i < K; i++) {
        const T x = b[i], y = c[i], z = acc[i];
        T r;
        switch (op) {
        case DE_T_FMA: r = m::fma(x, y, z); break;
        case DE_T_CLAMP: r = x > z ? z : (x < y ? y : x); break;
        case DE_T_ADD3: r = (x + y) + z; break;
        default: r = jl_max(jl_max(x, y), z); break;
        }
        acc[i] = r;
    }
}

template <typename T, int K, typename V> __device__ __forceinline__ bool any_nonfinite(const V &v) {
    bool bad = false;
    DE_UNROLL for (int i = 0; i < K; i++) bad |= !M<T>::isfinite(v[i]);
    return bad;
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
    const uint32_t xcd = bid & 7u, idx = bid >> 3;
    TileMap m;
    m.chunk = (int32_t)(idx % (uint32_t)n_chunks);
    m.tile = (int64_t)(idx / (uint32_t)n_chunks) * 8 + xcd;
    m.valid = m.tile < n_tiles;
    return m;
}

template <typename T, int K, bool EE, bool PARAMS>
__global__ void __launch_bounds__(BLOCK) de_eval_tape_kernel(const KArgs<T> a) {
    typedef typename VecOf<T, K>::type V;
    constexpr int TILE = BLOCK * K;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *__restrict__ xs = reinterpret_cast<T *>(smem_raw);
    const int xstride = a.xstride; // multiple of K: every row starts 16-byte aligned
    V *__restrict__ xsv = reinterpret_cast<V *>(smem_raw);
    V *__restrict__ stkv = reinterpret_cast<V *>(xs + (size_t)a.F * xstride);
    const int xstride_v = xstride / K;

    const TileMap tm = map_block(blockIdx.x, a.n_chunks, a.n_tiles);
    if (!tm.valid) return;
    const int tid = threadIdx.x;
    const int64_t base = tm.tile * TILE;
    const int64_t last = a.N - 1;

    // ---- stage the X tile: coalesced HBM/L2 read, transposed LDS write ----------
    {
        const uint32_t F = (uint32_t)a.F;
        const uint32_t total = (uint32_t)TILE * F;
        if (a.ldX == (int64_t)F && base + TILE <= a.N) {
            const T *__restrict__ src = a.X + base * (int64_t)F; // contiguous TILE*F elements
            for (uint32_t e = tid; e < total; e += BLOCK) {
                const uint32_t j = e / F, f = e - j * F;
                xs[f * xstride + j] = src[e];
            }
        } else { // ragged tail / strided X: clamp to the last real sample
            for (uint32_t e = tid; e < total; e += BLOCK) {
                const uint32_t j = e / F, f = e - j * F;
                int64_t jj = base + j;
                jj = jj < last ? jj : last;
                xs[f * xstride + j] = a.X[f + a.ldX * jj];
            }
        }
    }
    const int my = tid * K; // first sample of this thread inside the tile
    int64_t cls[K];
    if (PARAMS) {
        DE_UNROLL for (int i = 0; i < K; i++) {
            int64_t jj = base + my + i;
            jj = jj < last ? jj : last;
            cls[i] = (a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj]
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

    int pe = code_off[t0];
    for (int tree = t0; tree < t1; ++tree) {
        int pc = pe;
        pe = code_off[tree + 1];
        V acc;
        DE_UNROLL for (int i = 0; i < K; i++) acc[i] = T(0);
        bool bad = false;
        U32x4 nxt = code[pc]; // scalar load; a tree has at least one instruction
        for (; pc < pe; ++pc) {
            const U32x4 w = nxt;
            nxt = code[pc + 1]; // prefetch (the code buffer carries one trailing pad instruction)
            const uint32_t hdr = w.x;
            const uint32_t op = hdr & H_OP_MASK;
            const uint32_t src = (hdr >> H_SRC_SHIFT) & H_SRC_MASK;
            if (hdr & H_PUSH) stkv[((hdr >> H_PUSH_SHIFT) & H_SLOT_MASK) * BLOCK + tid] = acc;
            V b;
            if (src == SRC_FEAT) {
                b = xsv[(w.y & 0xFFFFu) * xstride_v + tid];
            } else if (src == SRC_CONST) {
                const T c = imm_of<T>(w.z, w.w);
                DE_UNROLL for (int i = 0; i < K; i++) b[i] = c;
            } else if (src == SRC_ACC) {
                b = acc;
            } else if (src == SRC_POP) {
                b = stkv[((hdr >> H_POP_SHIFT) & H_SLOT_MASK) * BLOCK + tid];
            } else {
                if (PARAMS) {
                    const T *__restrict__ s = a.params + (w.y & 0xFFFFu);
                    DE_UNROLL for (int i = 0; i < K; i++) b[i] = s[a.ld_params * cls[i]];
                } else {
                    b = acc;
                }
            }
            if (EE && (hdr & H_CHECK_B)) bad |= any_nonfinite<T, K, V>(b);
            // ---- fast path: the operators of the headline workload ----------------
            if (op == DOP_LOAD) {
                acc = b;
            } else {
                if (op == DE_B_ADD) acc = acc + b;
                else if (op == DE_B_MUL) acc = acc * b;
                else if (op == DE_B_SUB) acc = acc - b;
                else if (op == DOP_RSUB) acc = b - acc;
                else if (op == DE_B_DIV) acc = acc / b;
                else if (op == DOP_RDIV) acc = b / acc;
                else if (op == DE_U_COS) { DE_UNROLL for (int i = 0; i < K; i++) acc[i] = M<T>::cos(b[i]); }
                else if (op == DE_U_EXP) { DE_UNROLL for (int i = 0; i < K; i++) acc[i] = M<T>::exp(b[i]); }
                else if (op >= DE_T_FMA && op < DOP_LOAD) {
                    const V c = stkv[((hdr >> H_POPC_SHIFT) & H_SLOT_MASK) * BLOCK + tid];
                    apply_op3<T, K, V>(op, acc, b, c);
                } else {
                    apply_cold_op<T, K, V>(op, acc, b);
                }
                if (!EE && (hdr & H_INJECT)) { // is_valid(x_l) ? op(x_l) : Inf  (src/Evaluate.jl:722)
                    DE_UNROLL for (int i = 0; i < K; i++)
                        if (!M<T>::isfinite(b[i])) acc[i] = M<T>::inf();
                }
                if (EE || (hdr & H_CHECK_ALWAYS)) bad |= any_nonfinite<T, K, V>(acc);
            }
        }
        // ---- store out[tree][base + my .. +K) ---------------------------------
        T *__restrict__ o = a.out + (int64_t)tree * a.ld_out + base + my;
        if (full && a.vec_store) {
            *reinterpret_cast<V *>(o) = acc;
        } else {
            DE_UNROLL for (int i = 0; i < K; i++)
                if (base + my + i < a.N) o[i] = acc[i];
        }
        // ---- completion flag: one ballot per wave, one byte store per failing wave
        if (__ballot(bad) != 0ull && (tid & 63) == 0) a.ok[tree] = 0;
    }
}

// ---------------------------------------------------------------------------
size_t eval_lds_bytes(int dtype, int F, int n_slots, int *K_out) {
    const size_t es = dtype == DE_F32 ? 4 : 8;
    const int K = dtype == DE_F32 ? 4 : 2;
    const size_t tile = (size_t)BLOCK * K;
    const size_t xstride = tile + 16 / es;
    const size_t bytes = ((size_t)F * xstride + (size_t)n_slots * tile) * es;
    if (K_out) *K_out = K;
    return bytes <= 160 * 1024 ? bytes : 0;
}

static int g_cu_count = 0;

template <typename T> struct KName;
template <> struct KName<float> { static const char *get(bool ee, bool p) {
    return ee ? (p ? "de_eval_tape_kernel<float, 4, true, true>" : "de_eval_tape_kernel<float, 4, true, false>")
              : (p ? "de_eval_tape_kernel<float, 4, false, true>" : "de_eval_tape_kernel<float, 4, false, false>"); } };
template <> struct KName<double> { static const char *get(bool ee, bool p) {
    return ee ? (p ? "de_eval_tape_kernel<double, 2, true, true>" : "de_eval_tape_kernel<double, 2, true, false>")
              : (p ? "de_eval_tape_kernel<double, 2, false, true>" : "de_eval_tape_kernel<double, 2, false, false>"); } };

template <typename T, int K>
static hipError_t launch_eval_t(const EvalArgs &e, hipStream_t stream, const char **kname) {
    constexpr int TILE = BLOCK * K;
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
    a.xstride = TILE + (int)(16 / sizeof(T));
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
    const size_t lds = ((size_t)a.F * a.xstride + (size_t)a.n_slots * TILE) * sizeof(T);

    void (*kern)(const KArgs<T>);
    if (e.early_exit) kern = e.uses_params ? de_eval_tape_kernel<T, K, true, true> : de_eval_tape_kernel<T, K, true, false>;
    else kern = e.uses_params ? de_eval_tape_kernel<T, K, false, true> : de_eval_tape_kernel<T, K, false, false>;
    if (kname) *kname = KName<T>::get(e.early_exit, e.uses_params);
    if (lds > 64 * 1024) {
        hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_eval(int dtype, const EvalArgs &a, hipStream_t stream, const char **kernel_name) {
    if (dtype == DE_F32) return launch_eval_t<float, 4>(a, stream, kernel_name);
    return launch_eval_t<double, 2>(a, stream, kernel_name);
}

} // namespace de
