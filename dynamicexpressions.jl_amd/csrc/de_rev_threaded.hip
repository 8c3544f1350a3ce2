// de_rev_threaded.hip — fused loss + gradient by REVERSE accumulation (de_eval_loss_grad).
//
// The consumers of the fused reduction (the optimiser callback `G[i] = sum_j l'(e_j) dyhat/dc_i`,
// test/test_optim.jl:42-51; EvalPullback, src/ChainRules.jl:56-77) need   sum_j lp_j * d tree(x_j)/d theta_k
// for EVERY gradient row k, never the [n_grad, N] Jacobian itself.  Forward duals (de_grad_threaded.hip, the
// arithmetic of eval_grad_tree_array, src/EvaluateDerivative.jl:340-365) cost (1 + n_grad) values per node;
// one forward sweep that keeps every operator's PARTIALS plus one backward sweep over the same instruction
// stream costs ~2 values per node whatever n_grad is.  Same partial functions (ChainRules scalar rules,
// de_grad_common.h), same validity rule — a tree is incomplete iff a tested value, a partial or a gradient
// entry of some sample is non-finite (a non-finite partial always reaches the reference's gradient matrix:
// g * 0 = NaN in its dense update, :355-361) — but the products of a gradient entry are associated leaf-wards
// instead of root-wards, so entries agree with the forward Jacobian to rounding, not bit for bit.
//
// Machine: the eval accumulator machine (acc + spill slots in LDS).  Forward, every operator stores its
// partial(s) in LDS rows of its own (+, - store nothing).  Backward, the instruction stream runs in reverse
// with the ADJOINT in the accumulator and adjoint spill slots in the same LDS rows: a unary operator multiplies
// the adjoint by its partial; a binary operator sends (adjoint x partial) to its operand — a spill slot, or a
// leaf whose gradient row is reduced over the wavefront right there — and continues with the accumulator side.
// Dispatch is direct-threaded (as in de_kernels.hip / de_grad_threaded.hip): a handler loads the next record first, runs its
// body and tail-calls the next handler with that record's operand words in SGPRs; each sweep ends in an end record (r_end)
// that returns to the kernel.
// Instruction words (BoundInstr): x = handler offset, y = LDS byte offset added to the lane's base, z/w = immediate.
//   forward, row operand   : y = operand row, z = partial row (| generic opcode << 24)
//   forward, const operand : y = partial row (| generic opcode << 24), z/w = the constant
//   backward               : y = partial row (or slot row for POP), z = slot byte offset | gradient column (bit 31: accumulate)
// One module per element type: build.sh compiles this file with -DDE_RT_T=float|double -DDE_RT_TAG=f|d.
#include "de_grad_common.h"
#include "de_bind.h"

#include <cstring>

#if !defined(DE_RT_T) || !defined(DE_RT_TAG)
#error "compile with -DDE_RT_T=<float|double> -DDE_RT_TAG=<f|d>"
#endif

namespace de {
#define DE_RT_CAT2(a, b) a##b
#define DE_RT_CAT(a, b) DE_RT_CAT2(a, b)
#define DE_RT_NAME(prefix) DE_RT_CAT(prefix, DE_RT_TAG)
namespace DE_RT_NAME(rtm_) { // per-module names: see de_grad_threaded.hip

template <typename T> struct RImm;
template <> struct RImm<float> { typedef uint64_t type; }; // (z | w << 32: the fused backward handlers carry a second column word in w)
template <> struct RImm<double> { typedef uint64_t type; };
template <typename T> __device__ __forceinline__ T rimm_from(typename RImm<T>::type b);
template <> __device__ __forceinline__ float rimm_from<float>(uint64_t b) { return __uint_as_float((uint32_t)b); }
template <> __device__ __forceinline__ double rimm_from<double>(uint64_t b) { return __longlong_as_double((long long)b); }

// Named GState so that irpatch.py recognises the handlers (return type %"struct.de::<module>::GState").
template <typename T> struct GState {
    T x;       // forward: accumulator; backward: adjoint d tree / d (current accumulator)
    T lp;      // backward: w_j * l'(e_j) of the lane's sample (0 = excluded)
    T vpoison; // NaN once a tested VALUE was non-finite
    T gpoison; // NaN once a partial or a gradient entry was non-finite (counts only for trees with gradient rows)
    uint32_t lds0;  // the lane's LDS base
    uint32_t stage; // LDS address of this tree's column sums (the wave's staging area)
};
template <typename T> using RBodyFn = GState<T> (*)(GState<T>, uint32_t, typename RImm<T>::type);
#define RHARGS GState<T> st, uint32_t la, typename RImm<T>::type imm
// what the stream points at: rh_chain<T, &body>.  `code` = the NEXT record; (la, imm) = this instruction's operand words (la
// still without the lane's base: st.lds0 is added here); nx = the handler of the NEXT record (address - hbase): x of record k names the
// handler of record k + 1, the end record of a sweep names the sweep's first handler (round 4, as in de_grad_threaded.hip) — a handler
// knows its successor at entry, loads the next record into the successor's argument registers and jumps without waiting for it;
// hbase = the module's handler base.  irpatch.py: code, hbase, nx, la, imm in SGPRs.
#define RCHAIN_ARGS GState<T> st, ConstU4Ptr code, uint64_t hbase, uint32_t nx, uint32_t la, typename RImm<T>::type imm
template <typename T> using RHandlerFn = GState<T> (*)(GState<T>, ConstU4Ptr, uint64_t, uint32_t, uint32_t, typename RImm<T>::type);
template <typename T> __device__ __forceinline__ typename RImm<T>::type rrec_imm(const U32x4 &w);
template <> __device__ __forceinline__ uint64_t rrec_imm<float>(const U32x4 &w) { return ((uint64_t)w.w << 32) | w.z; }
template <> __device__ __forceinline__ uint64_t rrec_imm<double>(const U32x4 &w) { return ((uint64_t)w.w << 32) | w.z; }
// record address + 1 WITHOUT a carry into the high half (the stream lies inside one 4 GiB window: prog_malloc in de_api.cpp guarantees it, the callers check the allocation)
__device__ __forceinline__ ConstU4Ptr rcode_next(ConstU4Ptr c) {
    const uint64_t a = (uint64_t)(uintptr_t)c;
    return (ConstU4Ptr)(uintptr_t)((a & 0xFFFFFFFF00000000ull) | (uint64_t)((uint32_t)a + 16u));
}
#define RCHAIN_NEXT(W) [[clang::musttail]] return reinterpret_cast<RHandlerFn<T>>(hbase + nx)(st, rcode_next(code), hbase, (W).x, (W).y, rrec_imm<T>(W))
#define RH(...) (uint64_t)&rh_chain<T, &__VA_ARGS__>
#define RLDS(T, addr) (reinterpret_cast<__attribute__((address_space(3))) T *>((uintptr_t)(addr)))
template <typename T> constexpr uint32_t rrow_bytes() { return (uint32_t)(64 * sizeof(T)); }
template <typename T> __device__ __forceinline__ void rpoison(T &p, T v) { p = M<T>::fma(v, T(0), p); }

enum { RS_LEAF = 0, RS_SLOT = 1, RS_CONST = 2, RS_ACC = 3 };

// ---- forward handlers -----------------------------------------------------------------------------------------
template <typename T, int SRC> __device__ __forceinline__ T roperand(GState<T> &st, uint32_t la, typename RImm<T>::type imm) {
    if constexpr (SRC == RS_CONST) return rimm_from<T>(imm);
    else if constexpr (SRC == RS_ACC) return st.x;
    else {
        const T v = *RLDS(T, la & 0xFFFFFFu);
        if constexpr (SRC == RS_LEAF) rpoison<T>(st.vpoison, v); // every leaf operand is tested where it is read
        return v;
    }
}
// address of the instruction's first partial row
template <typename T, int SRC> __device__ __forceinline__ uint32_t rprow(const GState<T> &st, uint32_t la, typename RImm<T>::type imm) {
    if constexpr (SRC == RS_CONST || SRC == RS_ACC) return la & 0xFFFFFFu;
    else return st.lds0 + ((uint32_t)imm & 0xFFFFFFu);
}
template <typename T, int SRC> __device__ __forceinline__ GState<T> f_load(RHARGS) {
    st.x = roperand<T, SRC>(st, la, imm);
    return st;
}
template <typename T> __device__ __forceinline__ GState<T> f_push(RHARGS) {
    *RLDS(T, la) = st.x;
    return st;
}
template <typename T> __device__ __forceinline__ GState<T> f_check(RHARGS) {
    rpoison<T>(st.vpoison, st.x);
    return st;
}
template <typename T> __device__ __forceinline__ GState<T> r_nop(RHARGS) { return st; }

// binary hot ops, K = 0 ADD, 1 SUB, 2 RSUB, 3 MUL, 4 DIV, 5 RDIV (K 2/5: left = operand).  Partials are stored
// as (d/d acc, d/d operand); + and - store nothing (the backward handlers know them).
template <typename T, int K, int SRC, bool CHK> __device__ __forceinline__ GState<T> f_bin(RHARGS) {
    const T b = roperand<T, SRC>(st, la, imm);
    constexpr bool REV = (K == 2 || K == 5);
    const T lx = REV ? b : st.x, ly = REV ? st.x : b;
    if constexpr (K == 0) st.x = lx + ly;
    else if constexpr (K == 1 || K == 2) st.x = lx - ly;
    else {
        T v, gl, gr;
        if constexpr (K == 3) { v = lx * ly; gl = ly; gr = lx; }
        else if constexpr (K == 6 || K == 7) { // max / min (binary_vg's expressions)
            const bool gt = lx > ly;
            v = K == 6 ? jl_max(lx, ly) : jl_min(lx, ly);
            gl = (K == 6) == gt ? T(1) : T(0);
            gr = (K == 6) == gt ? T(0) : T(1);
        }
        else { v = lx / ly; gl = T(1) / ly; gr = -(v * gl); }
        const uint32_t pr = rprow<T, SRC>(st, la, imm);
        const T ga = REV ? gr : gl, gb = REV ? gl : gr;
        *RLDS(T, pr) = ga;
        *RLDS(T, pr + rrow_bytes<T>()) = gb;
        rpoison<T>(st.gpoison, ga);
        rpoison<T>(st.gpoison, gb);
        st.x = v;
    }
    if constexpr (CHK) rpoison<T>(st.vpoison, st.x);
    return st;
}
// unary hot ops (K: 0 cos, 1 exp, 2 sin, 3.. gun_inline); SRC = RS_ACC or RS_LEAF (fused leaf load)
template <typename T, int K, int SRC, bool CHK> __device__ __forceinline__ GState<T> f_un(RHARGS) {
    const T b = roperand<T, SRC>(st, la, imm);
    T y, g;
    if constexpr (K >= 3) { // the cheap unary operators, same expressions as the generic table (gun_inline, de_grad_common.h)
        const UG<T> r = gun_inline<T, K>(b);
        y = r.y;
        g = r.g;
    } else if constexpr (sizeof(T) == 4) {
        if constexpr (K == 1) { y = (T)fast_exp_f32((float)b); g = y; }
        else {
            float sn, cs;
            fast_sincos_f32((float)b, &sn, &cs);
            if (M<T>::abs(b) > T(DE_TRIG_FAST_BOUND)) { sn = sinf((float)b); cs = cosf((float)b); } // per element, as everywhere
            if constexpr (K == 0) { y = (T)cs; g = (T)-sn; } else { y = (T)sn; g = (T)cs; }
        }
    } else {
        if constexpr (K == 0) { y = M<T>::cos(b); g = -M<T>::sin(b); }
        else if constexpr (K == 1) { y = M<T>::exp(b); g = y; }
        else { y = M<T>::sin(b); g = M<T>::cos(b); }
    }
    const uint32_t pr = rprow<T, SRC>(st, la, imm);
    *RLDS(T, pr) = g;
    rpoison<T>(st.gpoison, g);
    st.x = y;
    if constexpr (CHK) rpoison<T>(st.vpoison, st.x);
    return st;
}
// generic operators through the noinline value+partials functions of de_grad_common.h: acc = op(b) (unary) or
// op(acc, b) / op(b, acc); partial rows as above (unary: one row)
template <typename T> __device__ __noinline__ GState<T> r_gen_apply(GState<T> st, uint32_t gop, T b, uint32_t pr) {
    if (gop < DE_B_ADD) {
        const UG<T> r = unary_vg<T>(gop, b);
        st.x = r.y;
        *RLDS(T, pr) = r.g;
        rpoison<T>(st.gpoison, r.g);
        return st;
    }
    uint32_t fop = gop;
    bool rev = false;
    switch (gop) {
    case DOP_RSUB: fop = DE_B_SUB; rev = true; break;
    case DOP_RDIV: fop = DE_B_DIV; rev = true; break;
    case DOP_RPOW: fop = DE_B_POW; rev = true; break;
    case DOP_RMOD: fop = DE_B_MOD; rev = true; break;
    case DOP_RREM: fop = DE_B_REM; rev = true; break;
    case DOP_RGREATER: fop = DE_B_GREATER; rev = true; break;
    case DOP_RPOW_ABS2: fop = DE_B_POW_ABS2; rev = true; break;
    default: break;
    }
    const BG<T> r = rev ? binary_vg<T>(fop, b, st.x) : binary_vg<T>(fop, st.x, b);
    const T ga = rev ? r.gy : r.gx, gb = rev ? r.gx : r.gy;
    st.x = r.v;
    *RLDS(T, pr) = ga;
    *RLDS(T, pr + rrow_bytes<T>()) = gb;
    rpoison<T>(st.gpoison, ga);
    rpoison<T>(st.gpoison, gb);
    return st;
}
template <typename T, int SRC> __device__ __forceinline__ GState<T> f_gen(RHARGS) {
    const uint32_t gop = (SRC == RS_CONST || SRC == RS_ACC) ? (la >> 24) : ((uint32_t)imm >> 24);
    const T b = roperand<T, SRC>(st, la, imm);
    return r_gen_apply<T>(st, gop, b, rprow<T, SRC>(st, la, imm));
}
// acc = op3(slot B, slot C, acc): la = partial rows (3) | op << 24, imm = row index B | row index C << 16
template <typename T> __device__ __forceinline__ GState<T> f_tern(RHARGS) {
    const uint32_t lb = st.lds0 + ((uint32_t)imm & 0xFFFFu) * rrow_bytes<T>(), lc = st.lds0 + (((uint32_t)imm >> 16) & 0xFFFFu) * rrow_bytes<T>();
    const uint32_t pr = la & 0xFFFFFFu;
    const TG<T> r = ternary_vg<T>(la >> 24, *RLDS(T, lb), *RLDS(T, lc), st.x);
    st.x = r.v;
    *RLDS(T, pr) = r.g0;
    *RLDS(T, pr + rrow_bytes<T>()) = r.g1;
    *RLDS(T, pr + 2 * rrow_bytes<T>()) = r.g2;
    rpoison<T>(st.gpoison, r.g0);
    rpoison<T>(st.gpoison, r.g1);
    rpoison<T>(st.gpoison, r.g2);
    return st;
}

// ---- backward handlers ----------------------------------------------------------------------------------------
// The gradient row of this tree receives  sum over the wave of  lp * v   (v = d tree / d leaf of the lane's sample).
// colw: [15:0] column, [29:16] LDS accumulation row, [31:30] 0 = reduce now; rows several leaves share are summed per
// sample first: 1 = first leaf (store), 2 = middle (add), 3 = last (add, then reduce).
template <typename T> __device__ __forceinline__ void r_reduce(GState<T> &st, T v, uint32_t colw) {
    const uint32_t md = colw >> 30;
    if (md != 0u) { // wave-uniform
        const uint32_t ar = st.lds0 + ((colw >> 16) & 0x3FFFu) * rrow_bytes<T>();
        if (md == 1u) { *RLDS(T, ar) = v; return; }
        v += *RLDS(T, ar);
        if (md == 2u) { *RLDS(T, ar) = v; return; }
    }
    rpoison<T>(st.gpoison, v);
    const T c = st.lp == T(0) ? T(0) : st.lp * v; // weight 0 (and samples past N) really excludes the sample
    const T s = wave_sum_to_lane63(c);
    // a global store here would stall the next handler (every function entry waits for vmcnt = 0): the sums are
    // staged in LDS and written once per batch of trees by the interpreter loop
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 63u) *RLDS(T, st.stage + (colw & 0xFFFFu) * (uint32_t)sizeof(T)) = s;
}
template <typename T> __device__ __forceinline__ GState<T> r_un(RHARGS) { // also: binary with an untracked leaf operand
    st.x = st.x * *RLDS(T, la);
    return st;
}
template <typename T> __device__ __forceinline__ GState<T> r_neg(RHARGS) {
    st.x = -st.x;
    return st;
}
template <typename T> __device__ __forceinline__ GState<T> r_pop(RHARGS) { // reverse of PUSH: continue with the slot's adjoint
    st.x = *RLDS(T, la);
    return st;
}
template <typename T, bool ADD> __device__ __forceinline__ GState<T> r_slotacc(RHARGS) { // reverse of "acc = shared row": the row's adjoint receives the accumulator's
    if constexpr (ADD) *RLDS(T, la) += st.x;
    else *RLDS(T, la) = st.x;
    return st;
}
template <typename T> __device__ __forceinline__ GState<T> r_popadd(RHARGS) { // reverse of a PUSH whose value ALSO stays in the accumulator: both adjoints
    st.x = st.x + *RLDS(T, la);
    return st;
}
template <typename T> __device__ __forceinline__ GState<T> r_leaf(RHARGS) { // reverse of a LOAD of a tracked leaf
    r_reduce<T>(st, st.x, (uint32_t)imm);
    return st;
}
// PK: 0 partial rows at la, 1 ADD, 2 SUB (acc - b), 3 RSUB (b - acc).  OK: 0 slot (imm = byte offset), 1 column (imm), 2 slot that
// ACCUMULATES (a shared row with several consumers: the one that runs first in the backward sweep stores, the others add)
template <typename T, int PK, int OK> __device__ __forceinline__ GState<T> r_bin(RHARGS) {
    T ab, aa;
    if constexpr (PK == 0) { aa = st.x * *RLDS(T, la); ab = st.x * *RLDS(T, la + rrow_bytes<T>()); }
    else if constexpr (PK == 1) { aa = st.x; ab = st.x; }
    else if constexpr (PK == 2) { aa = st.x; ab = -st.x; }
    else { aa = -st.x; ab = st.x; }
    if constexpr (OK == 0) *RLDS(T, st.lds0 + (uint32_t)imm) = ab;
    else if constexpr (OK == 2) *RLDS(T, st.lds0 + (uint32_t)imm) += ab;
    else r_reduce<T>(st, ab, (uint32_t)imm);
    st.x = aa;
    return st;
}
template <typename T> __device__ __forceinline__ GState<T> r_tern(RHARGS) {
    const uint32_t lb = st.lds0 + ((uint32_t)imm & 0xFFFFu) * rrow_bytes<T>(), lc = st.lds0 + (((uint32_t)imm >> 16) & 0xFFFFu) * rrow_bytes<T>();
    const uint32_t pr = la & 0xFFFFFFu;
    *RLDS(T, lb) = st.x * *RLDS(T, pr);
    *RLDS(T, lc) = st.x * *RLDS(T, pr + rrow_bytes<T>());
    st.x = st.x * *RLDS(T, pr + 2 * rrow_bytes<T>());
    return st;
}

template <typename T, RBodyFn<T> BODY> __device__ __noinline__ GState<T> rh_chain(RCHAIN_ARGS) {
    const U32x4 w = *code;
    st = BODY(st, st.lds0 + la, imm);
    RCHAIN_NEXT(w);
}
template <typename T> __device__ __noinline__ GState<T> r_end(GState<T> st, ConstU4Ptr, uint64_t, uint32_t, uint32_t, typename RImm<T>::type) { return st; }

// ---- fused pairs / triples (round 4) --------------------------------------------------------------------------------------------------
// One sample per lane makes this kernel dispatch-bound (VALU 40 % busy, as many scalar as vector instructions): every dispatch saved is
// time saved.  The C5 pullback's streams (37.8 dispatches per tree) are full of fixed sequences — every PUSH is followed by the load (or the
// unary function of a leaf) that starts the next subtree; backwards, every r_pop follows the r_leaf of that load, 1.9 r_leaf per tree follow
// the r_un of a unary function of a leaf and 1.9 follow an r_bin whose operand is a tracked leaf.  The encoder (de_api_grad.cpp
// ensure_rev_threaded) emits them as ONE record: la = two 16-bit LDS byte offsets relative to the lane's base (a wave's rows span < 64 KB:
// checked there), imm = the column word(s).  Same arithmetic in the same order: the same bits as the unfused stream (DE_REV_NO_FUSE=1).
template <typename T, int SRC> __device__ __noinline__ GState<T> rh_pushload(RCHAIN_ARGS) { // la = push slot | leaf row << 16 (CONST: imm = the constant)
    const U32x4 w = *code;
    *RLDS(T, st.lds0 + (la & 0xFFFFu)) = st.x;
    if constexpr (SRC == RS_CONST) st.x = rimm_from<T>(imm);
    else {
        const T v = *RLDS(T, st.lds0 + (la >> 16));
        rpoison<T>(st.vpoison, v);
        st.x = v;
    }
    RCHAIN_NEXT(w);
}
template <typename T, int K, bool CHK> __device__ __noinline__ GState<T> rh_pushun(RCHAIN_ARGS) { // la = push slot | leaf row << 16, imm = partial row
    const U32x4 w = *code;
    *RLDS(T, st.lds0 + (la & 0xFFFFu)) = st.x;
    st = f_un<T, K, RS_LEAF, CHK>(st, st.lds0 + (la >> 16), imm);
    RCHAIN_NEXT(w);
}
template <typename T, bool PRE, bool POP> __device__ __noinline__ GState<T> rh_leafx(RCHAIN_ARGS) { // la = partial row (PRE) | slot row << 16 (POP), imm = column word
    const U32x4 w = *code;
    if constexpr (PRE) st.x = st.x * *RLDS(T, st.lds0 + (la & 0xFFFFu));
    r_reduce<T>(st, st.x, (uint32_t)imm);
    if constexpr (POP) st.x = *RLDS(T, st.lds0 + (la >> 16));
    RCHAIN_NEXT(w);
}
template <typename T, int PK, bool POP> __device__ __noinline__ GState<T> rh_bincolx(RCHAIN_ARGS) { // r_bin<PK, column>, r_leaf, [r_pop]: imm = column | leaf's column << 32
    const U32x4 w = *code;
    st = r_bin<T, PK, 1>(st, st.lds0 + (la & 0xFFFFu), (typename RImm<T>::type)(uint32_t)imm);
    r_reduce<T>(st, st.x, (uint32_t)(imm >> 32));
    if constexpr (POP) st.x = *RLDS(T, st.lds0 + (la >> 16));
    RCHAIN_NEXT(w);
}

template <typename T> __global__ void de_rev_fill_handlers(uint64_t *t) {
    for (int i = 0; i < (int)ROP_COUNT; i++) t[i] = RH(r_nop<T>);
    t[rop_load(RS_LEAF)] = RH(f_load<T, RS_LEAF>);
    t[rop_load(RS_SLOT)] = RH(f_load<T, RS_SLOT>);
    t[rop_load(RS_CONST)] = RH(f_load<T, RS_CONST>);
    t[ROP_PUSH] = RH(f_push<T>);
    t[ROP_CHECK] = RH(f_check<T>);
#define RB2(K, S) t[rop_bin(K, S, false)] = RH(f_bin<T, K, S, false>); t[rop_bin(K, S, true)] = RH(f_bin<T, K, S, true>);
#define RB1(K) RB2(K, RS_LEAF) RB2(K, RS_SLOT) RB2(K, RS_CONST)
    RB1(0) RB1(1) RB1(2) RB1(3) RB1(4) RB1(5) RB1(6) RB1(7)
#define RU2(K, S) t[rop_un(K, S, false)] = RH(f_un<T, K, S, false>); t[rop_un(K, S, true)] = RH(f_un<T, K, S, true>);
#define RU1(K) RU2(K, RS_ACC) RU2(K, RS_LEAF)
    RU1(0) RU1(1) RU1(2) RU1(3) RU1(4) RU1(5) RU1(6) RU1(7) RU1(8) RU1(9) RU1(10) RU1(11) RU1(12)
    t[rop_gen(RS_LEAF)] = RH(f_gen<T, RS_LEAF>);
    t[rop_gen(RS_SLOT)] = RH(f_gen<T, RS_SLOT>);
    t[rop_gen(RS_CONST)] = RH(f_gen<T, RS_CONST>);
    t[rop_gen(RS_ACC)] = RH(f_gen<T, RS_ACC>);
    t[ROP_TERN] = RH(f_tern<T>);
    t[ROP_PARAM] = (uint64_t)&r_end<T>; // the end record of either sweep (the id is a leftover of round 1's parameter handler)
    t[ROP_R_UN] = RH(r_un<T>);
    t[ROP_R_NEG] = RH(r_neg<T>);
    t[ROP_R_POP] = RH(r_pop<T>);
    t[ROP_R_LEAF] = RH(r_leaf<T>);
#define RR(PK) t[rop_rbin(PK, 0)] = RH(r_bin<T, PK, 0>); t[rop_rbin(PK, 1)] = RH(r_bin<T, PK, 1>);
    RR(0) RR(1) RR(2) RR(3)
    t[ROP_R_TERN] = RH(r_tern<T>);
#define RUS(K) t[rop_un_slot(K, false)] = RH(f_un<T, K, RS_SLOT, false>); t[rop_un_slot(K, true)] = RH(f_un<T, K, RS_SLOT, true>);
    RUS(0) RUS(1) RUS(2) RUS(3) RUS(4) RUS(5) RUS(6) RUS(7) RUS(8) RUS(9) RUS(10) RUS(11) RUS(12)
#undef RUS
    t[ROP_R_POPADD] = RH(r_popadd<T>);
    t[ROP_R_SLOTACC_BASE + 0] = RH(r_slotacc<T, false>);
    t[ROP_R_SLOTACC_BASE + 1] = RH(r_slotacc<T, true>);
    t[ROP_R_BINACC_BASE + 0] = RH(r_bin<T, 0, 2>);
    t[ROP_R_BINACC_BASE + 1] = RH(r_bin<T, 1, 2>);
    t[ROP_R_BINACC_BASE + 2] = RH(r_bin<T, 2, 2>);
    t[ROP_R_BINACC_BASE + 3] = RH(r_bin<T, 3, 2>);
    t[ROP_F_PUSHLOAD_BASE + 0] = (uint64_t)&rh_pushload<T, RS_LEAF>;
    t[ROP_F_PUSHLOAD_BASE + 1] = (uint64_t)&rh_pushload<T, RS_CONST>;
#define RPU(K) t[rop_pushun(K, false)] = (uint64_t)&rh_pushun<T, K, false>; t[rop_pushun(K, true)] = (uint64_t)&rh_pushun<T, K, true>;
    RPU(0) RPU(1) RPU(2) RPU(3) RPU(4) RPU(5) RPU(6) RPU(7) RPU(8) RPU(9) RPU(10) RPU(11) RPU(12)
#undef RPU
    t[rop_leafx(false, true)] = (uint64_t)&rh_leafx<T, false, true>;
    t[rop_leafx(true, false)] = (uint64_t)&rh_leafx<T, true, false>;
    t[rop_leafx(true, true)] = (uint64_t)&rh_leafx<T, true, true>;
#define RBX(PK) t[rop_bincolx(PK, false)] = (uint64_t)&rh_bincolx<T, PK, false>; t[rop_bincolx(PK, true)] = (uint64_t)&rh_bincolx<T, PK, true>;
    RBX(0) RBX(1) RBX(2) RBX(3)
#undef RBX
}

// One sample per lane; wave-major LDS: per wave rows [0,F) = its slice of the X tile, [F, F+n_slots) spill slots
// (values forward, adjoints backward), then the partial rows of the tree being evaluated.
template <typename T, bool PARAMS>
__global__ void __launch_bounds__(GBLK) de_rev_threaded_kernel(const GArgs<T> a, const uint64_t hbase, const uint32_t param_off) {
    extern __shared__ __align__(16) unsigned char rtsmem[];
    T *__restrict__ rows = reinterpret_cast<T *>(rtsmem);
    const GTileMap tm = gmap_block_prio(a, blockIdx.x);
    if (!tm.valid) return;
    const int flag_protocol = tm.prio ? 1 : a.skip_flagged;
    // few per-lane values may live across the handler calls: they sit in callee-saved VGPRs, which the ABI hands out
    // in blocks of 8 at v40, v56, v72, v88 — a fourth block costs a fifth of the occupancy
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // regular tiling, or class-aligned tiles of a by-class reduction: a tile never straddles two classes, samples past
    // the end of its class are clamped copies with weight 0 like the samples past N
    const ConstI64Ptr tile_range = (ConstI64Ptr)(uintptr_t)a.tile_range;
    const int64_t base = a.tile_range ? tile_range[2 * tm.tile] : tm.tile * GBLK;
    const int64_t last = a.tile_range ? tile_range[2 * tm.tile + 1] : a.N - 1;
    const int F = a.F, R = a.rev_rows; // rows per wave
    {
        const uint32_t Fu = (uint32_t)a.FX, total = (uint32_t)GBLK * Fu;
        for (uint32_t e = tid; e < total; e += GBLK) {
            const uint32_t j = e / Fu, f = e - j * Fu;
            int64_t jj = base + j;
            jj = jj < last ? jj : last;
            rows[((j >> 6) * (uint32_t)R + f) * 64 + (j & 63)] = a.X[f + a.ldX * jj];
        }
    }
    if (PARAMS) { // rows FX .. F: params[:, class of the sample] (src/ParametricExpression.jl:381-389), read through the caches
        const uint32_t Pu = (uint32_t)(F - a.FX), total = (uint32_t)GBLK * Pu;
        for (uint32_t e = tid; e < total; e += GBLK) {
            const uint32_t j = e / Pu, q = e - j * Pu;
            int64_t jj = base + j;
            jj = jj < last ? jj : last;
            const int64_t cl = clamp_class((a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj] : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj]) - a.class_base, a.n_classes);
            rows[((j >> 6) * (uint32_t)R + (uint32_t)a.FX + q) * 64 + (j & 63)] = a.params[q + a.ld_params * cl];
        }
    }
    const int64_t j = base + tid, jj = j < last ? j : last;
    const T yv = a.y[jj];
    const T wv = j <= last ? (a.w ? a.w[jj] : T(1)) : T(0);
    __syncthreads();

    const ConstU4Ptr code = (ConstU4Ptr)(uintptr_t)a.code;
    const ConstI32Ptr code_off = (ConstI32Ptr)(uintptr_t)a.code_off;
    const ConstI32Ptr code_mid = (ConstI32Ptr)(uintptr_t)a.rev_mid;
    const ConstI64Ptr col_off = (ConstI64Ptr)(uintptr_t)a.col_off;
    const ConstI32Ptr n_grad = (ConstI32Ptr)(uintptr_t)a.n_grad;
    const ConstI32Ptr tree_ids = (ConstI32Ptr)(uintptr_t)a.tree_ids;
    const int t0 = tm.chunk * a.trees_per_chunk;
    const int t1 = (t0 + a.trees_per_chunk < a.n_trees) ? t0 + a.trees_per_chunk : a.n_trees;
    GState<T> st;
    st.lds0 = (uint32_t)(uintptr_t)rtsmem + (uint32_t)(wave * R) * rrow_bytes<T>() + (uint32_t)(tid & 63) * (uint32_t)sizeof(T);
    const int64_t n_cols = col_off[a.n_all_trees];

    // per-wave staging of the column sums: [stage_cols] elements after the wave's rows
    const int SC = a.rev_stage_cols;
    const uint32_t stage0 = (uint32_t)(uintptr_t)rtsmem + (uint32_t)(wave * R + (R - a.rev_stage_rows)) * rrow_bytes<T>();
#define RT_LANE() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
    int64_t stage_col0 = 0; // global column of stage[0]
    int staged = 0;         // columns staged so far
    auto flush = [&]() {
        T *dst = a.partial + ((int64_t)tm.tile * n_cols + stage_col0) * 4 + wave;
        for (int i = RT_LANE(); i < staged; i += 64) dst[(int64_t)i * 4] = *RLDS(T, stage0 + (uint32_t)i * (uint32_t)sizeof(T));
        staged = 0;
    };
    const uint64_t skip = gskip_mask(a.ok, tree_ids, t0, t1, flag_protocol, (int64_t)tm.tile);
    for (int ti = t0; ti < t1; ++ti) {
        if ((skip >> (ti - t0)) & 1ull) continue; // already incomplete (early exit): its reductions are NaN whatever the partials hold
        const int tree = tree_ids[ti];
        const int64_t c0 = col_off[tree];
        const int nc = 1 + n_grad[tree];
        if (staged > 0 && (c0 != stage_col0 + staged || staged + nc > SC)) flush();
        if (staged == 0) stage_col0 = c0;
        for (int i = RT_LANE(); i < nc; i += 64) *RLDS(T, stage0 + (uint32_t)(staged + i) * (uint32_t)sizeof(T)) = T(0); // rows no leaf touches are 0
        st.x = T(0);
        st.lp = T(0);
        st.vpoison = T(0);
        st.gpoison = T(0);
        st.stage = stage0 + (uint32_t)staged * (uint32_t)sizeof(T);
        staged += nc;
        int pc = code_off[tree];
        const int pm = code_mid[tree], pe = code_off[tree + 1];
        {   // forward sweep: one chain, ending in the end record in front of code[pm]
            const ConstU4Ptr rec = code + pc;
            const U32x4 hd = *rec;
            const uint32_t first = code[pm - 1].x; // the sweep's end record names its first handler
            st = reinterpret_cast<RHandlerFn<T>>(hbase + first)(st, rec + 1, hbase, hd.x, hd.y, rrec_imm<T>(hd));
        }
        rpoison<T>(st.vpoison, st.x);
        { // loss term and the seed of the backward sweep
            const T e = st.x - yv;
            T l, lp;
            if (a.loss_mode == 1 + DE_LOSS_L2) { l = wv * (e * e); lp = wv * (T(2) * e); }
            else if (a.loss_mode == 1 + DE_LOSS_L1) { l = wv * M<T>::abs(e); lp = wv * jl_sign(e); }
            else { l = wv * (st.x * yv); lp = wv * yv; } // DE_LOSS_PULLBACK: y holds the cotangent dY
            if (wv == T(0)) { l = T(0); lp = T(0); }
            const T s = wave_sum_to_lane63(l);
            if (RT_LANE() == 63) *RLDS(T, st.stage) = s;
            st.lp = lp;
            st.x = T(1);
        }
        {   // backward sweep (instructions stored in execution order), ending in the tree's last record
            const ConstU4Ptr rec = code + pm;
            const U32x4 hd = *rec;
            const uint32_t first = code[pe - 1].x;
            st = reinterpret_cast<RHandlerFn<T>>(hbase + first)(st, rec + 1, hbase, hd.x, hd.y, rrec_imm<T>(hd));
        }
        const bool bad = (st.vpoison != st.vpoison) || (nc > 1 && st.gpoison != st.gpoison);
        if (__ballot(bad) != 0ull) gflag_incomplete(a.ok + tree, flag_protocol == 1);
    }
    if (staged > 0) flush();
}

} // module namespace

hipError_t DE_RT_NAME(rev_thr_fetch_)(uint64_t *host_table) {
    using namespace DE_RT_NAME(rtm_);
    uint64_t *d = nullptr;
    hipError_t st = hipMalloc(reinterpret_cast<void **>(&d), ROP_COUNT * sizeof(uint64_t));
    if (st != hipSuccess) return st;
    hipLaunchKernelGGL((de_rev_fill_handlers<DE_RT_T>), dim3(1), dim3(1), 0, 0, d);
    st = hipMemcpy(host_table, d, ROP_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return st;
}

hipError_t DE_RT_NAME(rev_thr_launch_)(const GradArgs &ga, int group, hipStream_t stream) {
    using namespace DE_RT_NAME(rtm_);
    typedef DE_RT_T T;
    static int rt_gcu = 0;
    const EvalArgs &e = ga.e;
    const GradArgs::RevGroup &grp = ga.rev_groups[group];
    if (grp.n <= 0) return hipSuccess;
    GArgs<T> a;
    std::memset(&a, 0, sizeof a);
    a.code = ga.rev_code;
    a.code_off = ga.rev_code_off;
    a.rev_mid = ga.rev_code_mid;
    a.rev_rows = grp.rows;
    a.rev_stage_cols = ga.rev_stage_cols;
    a.rev_stage_rows = (int32_t)(((size_t)ga.rev_stage_cols * sizeof(T) + 64 * sizeof(T) - 1) / (64 * sizeof(T)));
    a.X = static_cast<const T *>(e.X);
    a.n_grad = ga.n_grad;
    a.ok = e.ok;
    a.params = static_cast<const T *>(e.params);
    a.classes = e.classes;
    a.N = e.N;
    a.ldX = e.ldX;
    a.ld_params = e.ld_params;
    a.n_tiles = ga.rev_tile_range ? ga.rev_n_tiles : (e.N + GBLK - 1) / GBLK;
    a.tile_range = ga.rev_tile_range;
    a.FX = e.F;
    a.F = e.F + (e.uses_params ? ga.P : 0); // leaf rows: X, then the parameters gathered by class
    a.P = ga.P;
    a.n_trees = grp.n;
    a.n_all_trees = e.n_trees;
    a.tree_ids = ga.rev_ids + grp.first;
    a.n_slots = e.n_slots;
    a.mode = ga.mode;
    a.classes_is_i64 = e.classes_is_i64;
    a.class_base = e.class_base;
    a.n_classes = e.n_classes > 0 ? e.n_classes : 1;
    a.uses_params = e.uses_params ? 1 : 0;
    a.check = 1;
    a.skip_flagged = e.skip_flagged ? 1 : 0;
    a.diff_g0 = -1;
    a.loss_mode = 1 + ga.loss->kind;
    a.y = static_cast<const T *>(ga.loss->y);
    a.w = static_cast<const T *>(ga.loss->w);
    a.partial = static_cast<T *>(ga.loss->partial);
    a.col_off = ga.col_off;
    if (rt_gcu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) rt_gcu = prop.multiProcessorCount;
        if (rt_gcu <= 0) rt_gcu = 256;
    }
    int64_t n_chunks = (grp.n + 31) / 32;
    const int64_t want_blocks = (int64_t)rt_gcu * 4 * 8;
    if (a.n_tiles * n_chunks < want_blocks) n_chunks = (want_blocks + a.n_tiles - 1) / a.n_tiles;
    const int64_t max_chunks = (grp.n + 3) / 4;
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    if (n_chunks < 1) n_chunks = 1;
    a.trees_per_chunk = (int32_t)((grp.n + n_chunks - 1) / n_chunks);
    if (a.skip_flagged) a.skip_flagged = a.trees_per_chunk >= 8 ? 1 : 2;
    if (a.skip_flagged) { const char *pv = getenv("DE_SKIP_PROTOCOL"); if (pv && *pv >= '1' && *pv <= '3') a.skip_flagged = *pv - '0'; } // (experiments) // flag protocol (skip_flag_load, de_device_ops.h): these kernels write little, their L1 lines go stale under 2 (reverse kernel 17.0 / 16.0 ms); 2 only for tiny chunks (many tiles on one flag line)
    a.n_chunks = (int32_t)((grp.n + a.trees_per_chunk - 1) / a.trees_per_chunk);
    int64_t blocks = ((a.n_tiles + 7) / 8) * 8 * a.n_chunks;
    a.prio = nullptr;
    a.n_prio = a.n_prio_blocks = a.prio_shift = 0;
    if (a.skip_flagged && ga.prio_ready && !ga.rev_tile_range) blocks += gprio_setup(a, e.prio_keys, e.F, GBLK); // (class-aligned tiles: no)
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds = 4 * (size_t)a.rev_rows * 64 * sizeof(T);
    void (*kern)(const GArgs<T>, uint64_t, uint32_t) = e.uses_params ? de_rev_threaded_kernel<T, true> : de_rev_threaded_kernel<T, false>;
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    if (a.n_prio) { // the priority tiles as a launch of their own in front, in short chunks (de_kernels.hip launch_threaded_t: no blind first wave)
        GArgs<T> pa = a;
        pa.trees_per_chunk = 4;
        pa.n_chunks = (int32_t)((grp.n + 3) / 4);
        pa.n_prio_blocks = (uint32_t)(((int64_t)pa.n_prio * pa.n_chunks + 7) / 8 * 8);
        hipLaunchKernelGGL(kern, dim3(pa.n_prio_blocks), dim3(GBLK), lds, stream, pa, ga.rev_handler_base, ga.rev_param_off);
        const hipError_t ps = hipGetLastError();
        if (ps != hipSuccess) return ps;
        blocks -= a.n_prio_blocks;
        a.n_prio = a.n_prio_blocks = 0;
        a.prio = nullptr;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(GBLK), lds, stream, a, ga.rev_handler_base, ga.rev_param_off);
    return hipGetLastError();
}

} // namespace de
