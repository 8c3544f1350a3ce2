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
//    words come through the scalar cache (s_load_dwordx4), decode and dispatch run
//    on the scalar unit, the VALU only sees operator arithmetic on K independent
//    samples per lane.  Intermediates live in registers; only the rare
//    both-children-are-subtrees case spills one value per sample to LDS.
//  * NaN/Inf flag: per-lane predicate, one wavefront ballot per tree, one byte
//    store per failing wave (no atomics).
//  * blockIdx -> (tile, chunk) is XCD-aware: all chunks of one X tile run on the
//    same XCD, so the tile is fetched from HBM once and re-served by that XCD's L2.
//  * No MFMA: this is an elementwise map, not a contraction.
#include <hip/hip_runtime.h>

#include "de_device_ops.h"
#include "de_kernels.h"

namespace de {

template <typename T> struct KArgs {
    const Instr *__restrict__ code;
    const int32_t *__restrict__ code_off;
    const T *__restrict__ X;
    T *__restrict__ out;
    uint8_t *__restrict__ ok;
    const T *__restrict__ params;
    const void *__restrict__ classes;
    int64_t N, ldX, ld_out, ld_params, n_tiles;
    int32_t F, n_trees, trees_per_chunk, n_chunks, n_slots, xstride;
    int32_t classes_is_i64, class_base, uses_params, vec_store;
};

template <typename T> __device__ __forceinline__ T imm_of(const Instr &ins);
template <> __device__ __forceinline__ float imm_of<float>(const Instr &ins) { return ins.imm.f32; }
template <> __device__ __forceinline__ double imm_of<double>(const Instr &ins) { return ins.imm.f64; }

#define DE_UNROLL _Pragma("unroll")

// acc = op(b) for every sample
#define U_CASE(OPC, EXPR)                                    \
    case OPC:                                                \
        DE_UNROLL for (int i = 0; i < K; i++) {              \
            const T x = b[i];                                \
            acc[i] = (EXPR);                                 \
        }                                                    \
        break;
// acc = op(acc, b)
#define B_CASE(OPC, EXPR)                                    \
    case OPC:                                                \
        DE_UNROLL for (int i = 0; i < K; i++) {              \
            const T x = acc[i], y = b[i];                    \
            acc[i] = (EXPR);                                 \
        }                                                    \
        break;

template <typename T, int K>
__device__ __forceinline__ void apply_op(uint32_t op, T (&acc)[K], const T (&b)[K]) {
    using m = M<T>;
    switch (op) {
        U_CASE(DOP_LOAD, x)
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
        U_CASE(DE_U_EXP2, m::exp2(x))
        U_CASE(DE_U_LOG, m::log(x))
        U_CASE(DE_U_LOG2, m::log2(x))
        U_CASE(DE_U_LOG10, m::log10(x))
        U_CASE(DE_U_LOG1P, m::log1p(x))
        U_CASE(DE_U_SIN, m::sin(x))
        U_CASE(DE_U_COS, m::cos(x))
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
        B_CASE(DE_B_ADD, x + y)
        B_CASE(DE_B_SUB, x - y)
        B_CASE(DOP_RSUB, y - x)
        B_CASE(DE_B_MUL, x * y)
        B_CASE(DE_B_DIV, x / y)
        B_CASE(DOP_RDIV, y / x)
        B_CASE(DE_B_POW, m::pow(x, y))
        B_CASE(DE_B_MAX, jl_max(x, y))
        B_CASE(DE_B_MIN, jl_min(x, y))
        B_CASE(DE_B_MOD, jl_mod(x, y))
        B_CASE(DE_B_REM, m::fmod(x, y))
        B_CASE(DE_B_GREATER, x > y ? T(1) : T(0))
        B_CASE(DE_B_POW_ABS2, jl_pow_abs2(x, y))
    default: break;
    }
}

// acc = op3(b, c, acc): b, c from spill slots, acc = third argument
template <typename T, int K>
__device__ __forceinline__ void apply_op3(uint32_t op, T (&acc)[K], const T (&b)[K], const T (&c)[K]) {
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

template <typename T, int K> __device__ __forceinline__ bool any_nonfinite(const T (&v)[K]) {
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

template <typename T, int K, bool EE>
__global__ void __launch_bounds__(BLOCK) de_eval_tape_kernel(const KArgs<T> a) {
    constexpr int TILE = BLOCK * K;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *__restrict__ xs = reinterpret_cast<T *>(smem_raw);
    const int xstride = a.xstride;
    T *__restrict__ stk = xs + (size_t)a.F * xstride;

    const TileMap tm = map_block(blockIdx.x, a.n_chunks, a.n_tiles);
    if (!tm.valid) return;
    const int tid = threadIdx.x;
    const int64_t base = tm.tile * TILE;
    const int64_t last = a.N - 1;

    // ---- stage the X tile: coalesced HBM read, transposed LDS write ----------
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
    if (a.uses_params) {
        DE_UNROLL for (int i = 0; i < K; i++) {
            int64_t jj = base + my + i;
            jj = jj < last ? jj : last;
            cls[i] = (a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj]
                                       : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj]) -
                     a.class_base;
        }
    } else {
        DE_UNROLL for (int i = 0; i < K; i++) cls[i] = 0;
    }
    __syncthreads();

    const int t0 = tm.chunk * a.trees_per_chunk;
    const int t1 = (t0 + a.trees_per_chunk < a.n_trees) ? t0 + a.trees_per_chunk : a.n_trees;
    const bool full = base + TILE <= a.N;

    for (int tree = t0; tree < t1; ++tree) {
        int pc = a.code_off[tree];
        const int pe = a.code_off[tree + 1];
        T acc[K];
        DE_UNROLL for (int i = 0; i < K; i++) acc[i] = T(0);
        bool bad = false;
        for (; pc < pe; ++pc) {
            const Instr ins = a.code[pc]; // wave-uniform: scalar load
            const uint32_t hdr = ins.hdr;
            const uint32_t op = hdr & H_OP_MASK;
            const uint32_t src = (hdr >> H_SRC_SHIFT) & H_SRC_MASK;
            if (hdr & H_PUSH) {
                T *__restrict__ s = stk + ((hdr >> H_PUSH_SHIFT) & H_SLOT_MASK) * TILE + my;
                DE_UNROLL for (int i = 0; i < K; i++) s[i] = acc[i];
            }
            T b[K];
            switch (src) {
            case SRC_FEAT: {
                const T *__restrict__ s = xs + (ins.feat & 0xFFFFu) * xstride + my;
                DE_UNROLL for (int i = 0; i < K; i++) b[i] = s[i];
                break;
            }
            case SRC_CONST: {
                const T c = imm_of<T>(ins);
                DE_UNROLL for (int i = 0; i < K; i++) b[i] = c;
                break;
            }
            case SRC_POP: {
                const T *__restrict__ s = stk + ((hdr >> H_POP_SHIFT) & H_SLOT_MASK) * TILE + my;
                DE_UNROLL for (int i = 0; i < K; i++) b[i] = s[i];
                break;
            }
            case SRC_PARAM: {
                const T *__restrict__ s = a.params + (ins.feat & 0xFFFFu);
                DE_UNROLL for (int i = 0; i < K; i++) b[i] = s[a.ld_params * cls[i]];
                break;
            }
            default: // SRC_ACC
                DE_UNROLL for (int i = 0; i < K; i++) b[i] = acc[i];
                break;
            }
            if (EE && (hdr & H_CHECK_B)) bad |= any_nonfinite<T, K>(b);
            if (op >= DE_T_FMA && op < DOP_LOAD) {
                T c[K];
                const T *__restrict__ s = stk + ((hdr >> H_POPC_SHIFT) & H_SLOT_MASK) * TILE + my;
                DE_UNROLL for (int i = 0; i < K; i++) c[i] = s[i];
                apply_op3<T, K>(op, acc, b, c);
            } else {
                if (hdr & H_SWAP) {
                    DE_UNROLL for (int i = 0; i < K; i++) {
                        const T t = acc[i];
                        acc[i] = b[i];
                        b[i] = t;
                    }
                }
                apply_op<T, K>(op, acc, b);
                if (!EE && (hdr & H_INJECT)) { // is_valid(x_l) ? op(x_l) : Inf  (src/Evaluate.jl:722)
                    DE_UNROLL for (int i = 0; i < K; i++)
                        if (!M<T>::isfinite(b[i])) acc[i] = M<T>::inf();
                }
            }
            if (EE ? (op != DOP_LOAD) : ((hdr & H_CHECK_ALWAYS) != 0)) bad |= any_nonfinite<T, K>(acc);
        }
        // ---- store out[tree][base + my .. +K) ---------------------------------
        T *__restrict__ o = a.out + (int64_t)tree * a.ld_out + base + my;
        if (full && a.vec_store) {
            if constexpr (K * sizeof(T) == 16) {
                float4 v;
                __builtin_memcpy(&v, acc, 16);
                *reinterpret_cast<float4 *>(o) = v;
            } else {
                DE_UNROLL for (int i = 0; i < K; i++) o[i] = acc[i];
            }
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
    a.uses_params = e.uses_params ? 1 : 0;
    a.vec_store = (reinterpret_cast<uintptr_t>(e.out) % 16 == 0 && (e.ld_out * sizeof(T)) % 16 == 0) ? 1 : 0;

    if (g_cu_count == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_cu_count = prop.multiProcessorCount;
        if (g_cu_count <= 0) g_cu_count = 256;
    }
    // Tree chunking: enough workgroups to fill the chip many times over (tail
    // quantisation), but chunks long enough to amortise the X-tile staging.
    const int64_t want_blocks = (int64_t)g_cu_count * 4 * 12;
    int64_t n_chunks = (want_blocks + a.n_tiles - 1) / a.n_tiles;
    const int64_t max_chunks = (e.n_trees + 15) / 16; // >= 16 trees per chunk
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    if (n_chunks < 1) n_chunks = 1;
    a.trees_per_chunk = (int32_t)((e.n_trees + n_chunks - 1) / n_chunks);
    a.n_chunks = (int32_t)((e.n_trees + a.trees_per_chunk - 1) / a.trees_per_chunk);

    const int64_t tile_groups = (a.n_tiles + 7) / 8;
    const int64_t blocks = tile_groups * 8 * a.n_chunks;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds = ((size_t)a.F * a.xstride + (size_t)a.n_slots * TILE) * sizeof(T);

    auto kern = e.early_exit ? de_eval_tape_kernel<T, K, true> : de_eval_tape_kernel<T, K, false>;
    if (kname)
        *kname = sizeof(T) == 4 ? (e.early_exit ? "de_eval_tape_kernel<float, 4, true>" : "de_eval_tape_kernel<float, 4, false>")
                                : (e.early_exit ? "de_eval_tape_kernel<double, 2, true>" : "de_eval_tape_kernel<double, 2, false>");
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
