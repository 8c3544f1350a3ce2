// de_grad_threaded.hip — forward-mode gradient kernel, threaded-code variant (gfx950).
// Same semantics as de_grad_tape_kernel (de_grad_kernels.hip; reference src/EvaluateDerivative.jl:
// eval_grad_tree_array :193-243, grad_degn_eval :340-365, grad_deg0_eval :367-404): every sample
// carries a dual number (x, d[0..GC)) through the bound UNFOLDED program.  The flat switch of that
// kernel costs ~37 scalar instructions per interpreted instruction and a gradient wave holds ONE
// sample per lane, so scalar issue — not the dual arithmetic — was its limit.  Here every
// (operator, operand kind, check) combination is its own leaf function; operand kinds are resolved on the
// host (leaf row / spill slot / constant), so handlers are straight-line code.  Dispatch is DIRECT-THREADED like the
// eval kernel's (de_kernels.hip): a handler loads the next record first (s_load_dwordx4, overlapping its own LDS reads
// and arithmetic), runs its body and tail-calls the next handler (s_setpc_b64) with that record's operand words in
// SGPRs — knowing its successor since entry, without waiting for that load (round 4) —; the tree's end record (g_end) returns to
// the kernel for the epilogue.  6 scalar + 1 vector instruction per dispatch against 14 + 2 for the call/return loop of round 1.
//
// Instruction word (16 B, built by de_api_grad.cpp ensure_grad_threaded from the bound program):
//   x = handler address - handler base OF THE NEXT RECORD (end record: of the tree's first record); the base travels with the chain
//       in SGPRs: one code object module per window width
//   y = LDS byte offset of the operand (leaf row or spill slot base) | aux << 24
//         aux = gradient row seeded by a leaf/constant operand (0xFF: none in this mode);
//         for GOP_GEN_CONST y[23:16] = de_opcode (no LDS operand), for GOP_TERN aux = de_opcode
//   z,w = constant bits | de_opcode (generic row/acc handlers) | byte distance slot B -> slot C (TERN)
// The window offset g0 travels in the state (a uniform VGPR that never moves): seed = aux - g0.
//
// ONE translation unit (= one code object module) per (element type, window width): for an indirect call
// LLVM sizes the kernel's register file for the most expensive address-taken function of the MODULE,
// so with every instantiation in one module the Float32 GC=5 kernel was allocated the 104 VGPRs of the
// Float64 GC=8 handlers (4 waves/SIMD instead of 8).  build.sh compiles this file with
// -DDE_GT_T=float|double -DDE_GT_TAG=f|d -DDE_GT_GC=1..6,8 -DDE_GT_VS=1|2; de_grad_kernels.hip dispatches.
//
// Samples per lane (DE_GT_VS): Float32 windows of <= 6 rows also exist in a variant that carries TWO
// consecutive samples per lane as float2 — the dense dual update  g1*d1[k] + g2*d2[k]  is all multiplies and
// adds, which gfx950 issues two per lane per instruction (v_pk_mul_f32 / v_pk_add_f32), and a dispatch is
// amortised over twice the samples.  It doubles the LDS rows, so the host uses it for the trees that need at
// most one spill slot (11 rows x 512 B x 4 waves = 22.5 KB: 7 workgroups per CU); with two slots (34.8 KB: 4
// workgroups) the lost occupancy costs more than the packed arithmetic saves (measured).  The state must stay
// within the 16 dwords the calling convention passes in registers (2*(1 + GC) + poison + g0).
#include <algorithm>

#include "de_grad_common.h"

#ifndef DE_GT_T
#error "compile with -DDE_GT_T=<float|double> -DDE_GT_TAG=<f|d> -DDE_GT_GC=<n> -DDE_GT_VS=<1|2>"
#endif
#ifndef DE_GT_VS
#define DE_GT_VS 1
#endif

namespace de {

#define DE_GT_CAT4(a, b, c, d, e) a##b##c##d##e
#define DE_GT_CAT5(a, b, c, d, e) DE_GT_CAT4(a, b, c, d, e)
#define DE_GT_NAME(prefix) DE_GT_CAT5(prefix, DE_GT_TAG, DE_GT_GC, v, DE_GT_VS)
// Every module instantiates the SAME templates with different DE_GT_VS: their host-side kernel stubs are
// linkonce symbols the linker would merge across modules (launching some other module's kernel), so each
// module's templates live in a namespace of their own.
namespace DE_GT_NAME(gtm_) {

template <typename T> struct GImm;
template <> struct GImm<float> { typedef uint32_t type; };
template <> struct GImm<double> { typedef uint64_t type; };
template <typename T> __device__ __forceinline__ T gimm_from(typename GImm<T>::type b);
template <> __device__ __forceinline__ float gimm_from<float>(uint32_t b) { return __uint_as_float(b); }
template <> __device__ __forceinline__ double gimm_from<double>(uint64_t b) { return __longlong_as_double((long long)b); }

constexpr int VS = DE_GT_VS; // samples per lane
template <typename T> struct LaneVec { typedef T type __attribute__((ext_vector_type(DE_GT_VS))); };
#define LV(T) typename LaneVec<T>::type
template <typename T> __device__ __forceinline__ LV(T) lv_splat(T c) {
    LV(T) v;
    DE_UNROLL for (int i = 0; i < VS; i++) v[i] = c;
    return v;
}
template <typename T, int GC> struct GState {
    LV(T) x;
    LV(T) d[GC];
    T poison; // one per lane: fma-accumulated over its samples
    uint32_t g0;
};
template <typename T, int GC> struct GDual {
    LV(T) x;
    LV(T) d[GC];
};
#define GHARGS GState<T, GC> st, uint32_t la, typename GImm<T>::type imm
template <typename T, int GC> using GBodyFn = GState<T, GC> (*)(GState<T, GC>, uint32_t, typename GImm<T>::type);
// what the stream points at: gh_chain<T, GC, &body>.  `code` = the NEXT record; (la, imm) = this instruction's operand words; nx = the
// handler of the NEXT record (address - hbase) — it stands in THIS instruction's record (round 4: x of record k names the handler of
// record k + 1, the end record names the tree's first handler), so a handler knows its successor at entry, loads the next record
// straight into the successor's argument registers (the aligned SGPR quad nx, la, imm) and jumps WITHOUT waiting for the load:
// the callee's entry wait completes it, the jump's instruction fetch overlaps with it.  (Before, the target came out of the loaded
// record: every handler sat out the scalar-cache latency before it could jump — the eval kernel dropped that in round 2.)
// lds0 = the lane's LDS base; hbase = handler base of this module.  csrc/irpatch.py moves code, hbase, nx, la, imm to SGPRs.
#define GCHAIN_ARGS GState<T, GC> st, uint32_t lds0, ConstU4Ptr code, uint64_t hbase, uint32_t nx, uint32_t la, typename GImm<T>::type imm
template <typename T, int GC> using GHandlerFn = GState<T, GC> (*)(GState<T, GC>, uint32_t, ConstU4Ptr, uint64_t, uint32_t, uint32_t, typename GImm<T>::type);
template <typename T> __device__ __forceinline__ typename GImm<T>::type grec_imm(const U32x4 &w);
template <> __device__ __forceinline__ uint32_t grec_imm<float>(const U32x4 &w) { return w.z; }
template <> __device__ __forceinline__ uint64_t grec_imm<double>(const U32x4 &w) { return ((uint64_t)w.w << 32) | w.z; }
// record address + 1 WITHOUT a carry into the high half (the stream lies inside one 4 GiB window: prog_malloc in de_api.cpp guarantees it, the callers check the allocation)
__device__ __forceinline__ ConstU4Ptr gcode_next(ConstU4Ptr c) {
    const uint64_t a = (uint64_t)(uintptr_t)c;
    return (ConstU4Ptr)(uintptr_t)((a & 0xFFFFFFFF00000000ull) | (uint64_t)((uint32_t)a + 16u));
}
#define GCHAIN_NEXT(W) [[clang::musttail]] return reinterpret_cast<GHandlerFn<T, GC>>(hbase + nx)(st, lds0, gcode_next(code), hbase, (W).x, (W).y, grec_imm<T>(W))
#define GH(...) (uint64_t)&gh_chain<T, GC, &__VA_ARGS__>
#define GLDS(T, addr) (reinterpret_cast<__attribute__((address_space(3))) LV(T) *>((uintptr_t)(addr)))
// LDS is laid out wave-major: wave w owns rows [w*R, (w+1)*R), a row = the 64*VS samples of that wave
// (512 B for Float32 x 2) — everything a wave touches is private to it and contiguous, which the epilogue
// uses to turn the [samples, G] gradient block into coalesced 16-byte stores.
template <typename T> constexpr uint32_t grow_bytes() { return (uint32_t)(64 * VS * sizeof(T)); }
template <typename T> __device__ __forceinline__ void gpoison(T &poison, LV(T) v) {
    DE_UNROLL for (int i = 0; i < VS; i++) poison = M<T>::fma(v[i], T(0), poison);
}

// operand kinds (de_bind.h GSRC_*)
enum { GS_LEAF = GSRC_LEAF, GS_SLOT = GSRC_SLOT, GS_CONST = GSRC_CONST, GS_ACC = GSRC_ACC };

// An operand: value + how its gradient is represented.  SV (seed variant, de_bind.h): 0 = the one-hot
// row index is read at run time (aux - g0), 1 = no gradient component in this window, 2 + k = component k.
// For LEAF/CONST operands with SV >= 1 the one-hot vector is never materialised: the reference's dense
//   g * db[k]   is   g * 1 = g  for k = seed  and  g * 0  elsewhere  (g * 0 is still computed once: it is NaN
// for an infinite partial and carries the sign of zero, exactly as in grad_degn_eval :340-365).
template <typename T, int GC, int SRC, int SV> struct GOperand {
    LV(T) x;
    LV(T) d[GC]; // only meaningful for SLOT/ACC operands and run-time seeds
};
template <typename T, int GC, int SRC, int SV> __device__ __forceinline__ GOperand<T, GC, SRC, SV> goperand(GState<T, GC> &st, uint32_t la, typename GImm<T>::type imm) {
    GOperand<T, GC, SRC, SV> b;
    if constexpr (SRC == GS_SLOT) {
        b.x = *GLDS(T, la);
        DE_UNROLL for (int k = 0; k < GC; k++) b.d[k] = *GLDS(T, la + (1 + k) * grow_bytes<T>());
    } else if constexpr (SRC == GS_ACC) {
        b.x = st.x;
        DE_UNROLL for (int k = 0; k < GC; k++) b.d[k] = st.d[k];
    } else {
        if constexpr (SRC == GS_LEAF) {
            b.x = *GLDS(T, SV == 0 ? (la & 0xFFFFFFu) : la); // host leaves aux = 0 for known seeds
            gpoison<T>(st.poison, b.x); // every leaf operand is tested where it is read (:239-242)
        } else b.x = lv_splat<T>(gimm_from<T>(imm));
        if constexpr (SV == 0) {
            const int seed = (int)(la >> 24) - (int)(st.g0 & 0xFFFFu);
            DE_UNROLL for (int k = 0; k < GC; k++) b.d[k] = lv_splat<T>((k == seed) ? T(1) : T(0));
        }
    }
    return b;
}
// g * db[k]
template <typename T, int GC, int SRC, int SV> __device__ __forceinline__ void gscale(LV(T) g, const GOperand<T, GC, SRC, SV> &b, LV(T) (&out)[GC]) {
    if constexpr ((SRC == GS_LEAF || SRC == GS_CONST) && SV >= 1) {
        const LV(T) z = g * lv_splat<T>(T(0));
        DE_UNROLL for (int k = 0; k < GC; k++) out[k] = (k == SV - 2) ? g : z; // g * 1 == g bit for bit
    } else {
        DE_UNROLL for (int k = 0; k < GC; k++) out[k] = g * b.d[k];
    }
}

template <typename T, int GC, int SRC, int SV> __device__ __forceinline__ GState<T, GC> g_load(GHARGS) {
    const GOperand<T, GC, SRC, SV> b = goperand<T, GC, SRC, SV>(st, la, imm);
    st.x = b.x;
    if constexpr ((SRC == GS_LEAF || SRC == GS_CONST) && SV >= 1) { DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = lv_splat<T>((k == SV - 2) ? T(1) : T(0)); }
    else { DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = b.d[k]; }
    return st;
}
template <typename T, int GC> __device__ __forceinline__ GState<T, GC> g_push(GHARGS) {
    *GLDS(T, la) = st.x;
    DE_UNROLL for (int k = 0; k < GC; k++) *GLDS(T, la + (1 + k) * grow_bytes<T>()) = st.d[k];
    return st;
}
// PUSH + LOAD in one dispatch.  LEAF: la = the row (| run-time seed << 24), imm = byte distance from the row to the slot;
// CONST: la = the slot (| run-time seed << 24), imm = the constant.
template <typename T, int GC, int SRC, int SV> __device__ __forceinline__ GState<T, GC> g_pushload(GHARGS) {
    uint32_t slot = la & 0xFFFFFFu;
    if constexpr (SRC == GS_LEAF) slot += (uint32_t)imm;
    *GLDS(T, slot) = st.x;
    DE_UNROLL for (int k = 0; k < GC; k++) *GLDS(T, slot + (1 + k) * grow_bytes<T>()) = st.d[k];
    const GOperand<T, GC, SRC, SV> b = goperand<T, GC, SRC, SV>(st, la, imm);
    st.x = b.x;
    if constexpr (SV >= 1) { DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = lv_splat<T>((k == SV - 2) ? T(1) : T(0)); }
    else { DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = b.d[k]; }
    return st;
}
template <typename T, int GC> __device__ __forceinline__ GState<T, GC> g_check_acc(GHARGS) {
    gpoison<T>(st.poison, st.x);
    return st;
}
template <typename T, int GC> __device__ __forceinline__ GState<T, GC> g_nop(GHARGS) { return st; }

// binary hot ops: value v and partials (gl, gr) w.r.t. (left, right); K 2/5 (RSUB/RDIV): left = operand.
// The formulas (and their operation order) are the oracle's / the switch kernel's: d = gl*dl + gr*dr dense.
template <typename T, int GC, int K, int SRC, int SV, bool CHK> __device__ __forceinline__ GState<T, GC> g_bin(GHARGS) {
    const GOperand<T, GC, SRC, SV> b = goperand<T, GC, SRC, SV>(st, la, imm);
    constexpr bool REV = (K == 2 || K == 5);
    const LV(T) lx = REV ? b.x : st.x, ly = REV ? st.x : b.x;
    LV(T) v, gl, gr;
    if constexpr (K == 0) { v = lx + ly; gl = lv_splat<T>(T(1)); gr = lv_splat<T>(T(1)); }
    else if constexpr (K == 1 || K == 2) { v = lx - ly; gl = lv_splat<T>(T(1)); gr = lv_splat<T>(T(-1)); }
    else if constexpr (K == 3) { v = lx * ly; gl = ly; gr = lx; }
    else if constexpr (K == 6 || K == 7) { // max / min: the expressions of binary_vg (ties: the second argument takes the gradient of max)
        DE_UNROLL for (int i = 0; i < VS; i++) {
            const bool gt = lx[i] > ly[i];
            v[i] = K == 6 ? jl_max(lx[i], ly[i]) : jl_min(lx[i], ly[i]);
            gl[i] = (K == 6) == gt ? T(1) : T(0);
            gr[i] = (K == 6) == gt ? T(0) : T(1);
        }
    }
    else { v = lx / ly; gl = lv_splat<T>(T(1)) / ly; gr = -(v / ly); }
    st.x = v;
    LV(T) sb[GC];
    gscale<T, GC, SRC, SV>(REV ? gl : gr, b, sb); // the operand's term
    if constexpr (REV) { DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = sb[k] + gr * st.d[k]; }
    else { DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = gl * st.d[k] + sb[k]; }
    if constexpr (CHK) gpoison<T>(st.poison, st.x);
    return st;
}
// unary hot ops (K: 0 cos, 1 exp, 2 sin, 3.. see gun_inline)
template <typename T, int GC, int K, int SRC, int SV, bool CHK> __device__ __forceinline__ GState<T, GC> g_un(GHARGS) {
    const GOperand<T, GC, SRC, SV> b = goperand<T, GC, SRC, SV>(st, la, imm);
    LV(T) y, g;
    if constexpr (K >= 3) {
        DE_UNROLL for (int i = 0; i < VS; i++) {
            const UG<T> r = gun_inline<T, K>(b.x[i]);
            y[i] = r.y;
            g[i] = r.g;
        }
    } else if constexpr (sizeof(T) == 4) {
        if constexpr (K == 1) { DE_UNROLL for (int i = 0; i < VS; i++) { y[i] = (T)fast_exp_f32((float)b.x[i]); g[i] = y[i]; } }
        else {
            bool big = false;
            DE_UNROLL for (int i = 0; i < VS; i++) {
                float sn, cs;
                fast_sincos_f32((float)b.x[i], &sn, &cs);
                if constexpr (K == 0) { y[i] = (T)cs; g[i] = (T)-sn; } else { y[i] = (T)sn; g[i] = (T)cs; }
                big |= M<T>::abs(b.x[i]) > T(DE_TRIG_FAST_BOUND);
            }
            // |x| > 1e5: OCML's full-range functions, per ELEMENT — a sample's value must not depend on its wave
            // neighbours (inline: a call would make this handler a non-leaf function)
            if (__ballot(big) != 0ull) {
                DE_UNROLL for (int i = 0; i < VS; i++) {
                    if (M<T>::abs(b.x[i]) > T(DE_TRIG_FAST_BOUND)) {
                        const float sn = sinf((float)b.x[i]), cs = cosf((float)b.x[i]);
                        if constexpr (K == 0) { y[i] = (T)cs; g[i] = (T)-sn; } else { y[i] = (T)sn; g[i] = (T)cs; }
                    }
                }
            }
        }
    } else {
        DE_UNROLL for (int i = 0; i < VS; i++) {
            if constexpr (K == 0) { y[i] = M<T>::cos(b.x[i]); g[i] = -M<T>::sin(b.x[i]); }
            else if constexpr (K == 1) { y[i] = M<T>::exp(b.x[i]); g[i] = y[i]; }
            else { y[i] = M<T>::sin(b.x[i]); g[i] = M<T>::cos(b.x[i]); }
        }
    }
    st.x = y;
    gscale<T, GC, SRC, SV>(g, b, st.d);
    if constexpr (CHK) gpoison<T>(st.poison, st.x);
    return st;
}
// generic (cold) operators through the noinline value+partials functions of de_grad_common.h
template <typename T, int GC> __device__ __noinline__ GState<T, GC> g_gen_apply(GState<T, GC> st, uint32_t gop, GDual<T, GC> b) {
    if (gop < DE_B_ADD) {
        LV(T) g;
        DE_UNROLL for (int i = 0; i < VS; i++) {
            const UG<T> r = unary_vg<T>(gop, b.x[i]);
            st.x[i] = r.y;
            g[i] = r.g;
        }
        DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = g * b.d[k];
    } else {
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
        LV(T) gx, gy;
        DE_UNROLL for (int i = 0; i < VS; i++) {
            const BG<T> r = rev ? binary_vg<T>(fop, b.x[i], st.x[i]) : binary_vg<T>(fop, st.x[i], b.x[i]);
            st.x[i] = r.v;
            gx[i] = r.gx;
            gy[i] = r.gy;
        }
        if (rev) { DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = gx * b.d[k] + gy * st.d[k]; }
        else { DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = gx * st.d[k] + gy * b.d[k]; }
    }
    return st;
}
template <typename T, int GC, int SRC> __device__ __forceinline__ GState<T, GC> g_gen(GHARGS) {
    uint32_t gop;
    if constexpr (SRC == GS_CONST) gop = ((la - (st.g0 & 0xFFFF0000u)) >> 16) & 0xFFu; // no LDS operand: the opcode rides in la[23:16] above the lane's LDS base, whose upper half st.g0 carries
    else gop = (uint32_t)imm;
    const GOperand<T, GC, SRC, 0> o = goperand<T, GC, SRC, 0>(st, la, imm);
    GDual<T, GC> b;
    b.x = o.x;
    DE_UNROLL for (int k = 0; k < GC; k++) b.d[k] = o.d[k];
    return g_gen_apply<T, GC>(st, gop, b);
}
template <typename T, int GC> __device__ __forceinline__ GState<T, GC> g_tern(GHARGS) { // acc = op3(slot B, slot C, acc)
    const uint32_t lb = la & 0xFFFFFFu, lc = lb + (uint32_t)imm;
    const LV(T) xb = *GLDS(T, lb), xc = *GLDS(T, lc);
    LV(T) g0, g1, g2;
    DE_UNROLL for (int i = 0; i < VS; i++) {
        const TG<T> r = ternary_vg<T>(la >> 24, xb[i], xc[i], st.x[i]);
        st.x[i] = r.v;
        g0[i] = r.g0; g1[i] = r.g1; g2[i] = r.g2;
    }
    DE_UNROLL for (int k = 0; k < GC; k++)
        st.d[k] = (g0 * *GLDS(T, lb + (1 + k) * grow_bytes<T>()) + g1 * *GLDS(T, lc + (1 + k) * grow_bytes<T>())) + g2 * st.d[k];
    return st;
}

// ---- direct-threaded dispatch (see the head of the file) ----------------------------------------------------------
template <typename T, int GC, GBodyFn<T, GC> BODY> __device__ __noinline__ GState<T, GC> gh_chain(GCHAIN_ARGS) {
    const U32x4 w = *code;
    st = BODY(st, lds0 + la, imm);
    GCHAIN_NEXT(w);
}
template <typename T, int GC> __device__ __noinline__ GState<T, GC> g_end(GState<T, GC> st, uint32_t, ConstU4Ptr, uint64_t, uint32_t, uint32_t, typename GImm<T>::type) { return st; }

// handler table: seed variants enumerated at compile time
template <typename T, int GC, int SV> __device__ __forceinline__ void fill_seeded(uint64_t *t) {
    if constexpr (SV < GC + 2) {
        t[gop_load(GC, GS_LEAF, SV)] = GH(g_load<T, GC, GS_LEAF, SV>);
        t[gop_load(GC, GS_CONST, SV)] = GH(g_load<T, GC, GS_CONST, SV>);
        t[gop_pushload(GC, GS_LEAF, SV)] = GH(g_pushload<T, GC, GS_LEAF, SV>);
        t[gop_pushload(GC, GS_CONST, SV)] = GH(g_pushload<T, GC, GS_CONST, SV>);
#define GB1(K) t[gop_bin(GC, K, GS_LEAF, SV, false)] = GH(g_bin<T, GC, K, GS_LEAF, SV, false>); \
               t[gop_bin(GC, K, GS_LEAF, SV, true)] = GH(g_bin<T, GC, K, GS_LEAF, SV, true>);    \
               t[gop_bin(GC, K, GS_CONST, SV, false)] = GH(g_bin<T, GC, K, GS_CONST, SV, false>); \
               t[gop_bin(GC, K, GS_CONST, SV, true)] = GH(g_bin<T, GC, K, GS_CONST, SV, true>);
        GB1(0) GB1(1) GB1(2) GB1(3) GB1(4) GB1(5) GB1(6) GB1(7)
#undef GB1
#define GU1(K) t[gop_un(GC, K, GS_LEAF, SV, false)] = GH(g_un<T, GC, K, GS_LEAF, SV, false>); \
               t[gop_un(GC, K, GS_LEAF, SV, true)] = GH(g_un<T, GC, K, GS_LEAF, SV, true>);
        GU1(0) GU1(1) GU1(2) GU1(3) GU1(4) GU1(5) GU1(6) GU1(7) GU1(8) GU1(9) GU1(10) GU1(11) GU1(12)
#undef GU1
        fill_seeded<T, GC, SV + 1>(t);
    }
}
template <typename T, int GC> __global__ void de_grad_fill_handlers(uint64_t *t) {
    fill_seeded<T, GC, 0>(t);
    t[gop_load(GC, GS_SLOT, 0)] = GH(g_load<T, GC, GS_SLOT, 0>);
    t[gop_push(GC)] = GH(g_push<T, GC>);
    t[gop_check_acc(GC)] = GH(g_check_acc<T, GC>);
#define GB2(K) t[gop_bin(GC, K, GS_SLOT, 0, false)] = GH(g_bin<T, GC, K, GS_SLOT, 0, false>); \
               t[gop_bin(GC, K, GS_SLOT, 0, true)] = GH(g_bin<T, GC, K, GS_SLOT, 0, true>);
    GB2(0) GB2(1) GB2(2) GB2(3) GB2(4) GB2(5) GB2(6) GB2(7)
#undef GB2
#define GU2(K, S) t[gop_un(GC, K, S, 0, false)] = GH(g_un<T, GC, K, S, 0, false>); t[gop_un(GC, K, S, 0, true)] = GH(g_un<T, GC, K, S, 0, true>);
#define GU3(K) GU2(K, GS_SLOT) GU2(K, GS_ACC)
    GU3(0) GU3(1) GU3(2) GU3(3) GU3(4) GU3(5) GU3(6) GU3(7) GU3(8) GU3(9) GU3(10) GU3(11) GU3(12)
#undef GU3
#undef GU2
    t[gop_gen(GC, GS_LEAF)] = GH(g_gen<T, GC, GS_LEAF>);
    t[gop_gen(GC, GS_SLOT)] = GH(g_gen<T, GC, GS_SLOT>);
    t[gop_gen(GC, GS_CONST)] = GH(g_gen<T, GC, GS_CONST>);
    t[gop_gen(GC, GS_ACC)] = GH(g_gen<T, GC, GS_ACC>);
    t[gop_param(GC)] = (uint64_t)&g_end<T, GC>; // the end record of every tree (the id is a leftover of round 1's parameter handler)
    t[gop_tern(GC)] = GH(g_tern<T, GC>);
}

// End of a tree: store (or reduce) the wave's results.  Out of line on purpose, and fed with plain values
// (taking the address of the kernel-argument struct would make every load from it look divergent): nothing
// of the epilogue stays live in VGPRs across the handler calls of the interpreter loop.
template <typename T, int GC, bool SHARE>
__device__ __noinline__ void g_epilogue_loss(GState<T, GC> st, const T *y, const T *w, T *pp, int64_t N, int loss_mode, int G, int g0, int64_t tile) {
    // pp = the partial sums of the (tile of 256 x VS samples, tree) pair this wave's samples belong to: [1 + n_grad][4 waves]
    // (shared leaf rows: `tile` counts 64 x VS samples — a quarter of such a tile, entry tile & 3 of its four)
    constexpr int TILE = GBLK * VS, WSAMP = 64 * VS;
    const int tid = threadIdx.x, wave = SHARE ? (int)(tile & 3) : tid >> 6, lane = tid & 63;
    const int64_t jl = SHARE ? tile * WSAMP + (int64_t)lane * VS : tile * TILE + (int64_t)tid * VS, last = N - 1; // this lane's first sample
    T l = T(0);
    LV(T) lp, wv;
    DE_UNROLL for (int i = 0; i < VS; i++) {
        const int64_t j = jl + i;
        const int64_t jj = j < last ? j : last;
        const T yv = y[jj];
        wv[i] = j <= last ? (w ? w[jj] : T(1)) : T(0);
        const T e = st.x[i] - yv;
        T li;
        if (loss_mode == 1 + DE_LOSS_L2) { li = wv[i] * (e * e); lp[i] = wv[i] * (T(2) * e); }
        else if (loss_mode == 1 + DE_LOSS_L1) { li = wv[i] * M<T>::abs(e); lp[i] = wv[i] * jl_sign(e); }
        else { li = wv[i] * (st.x[i] * yv); lp[i] = wv[i] * yv; } // DE_LOSS_PULLBACK: y holds the cotangent dY
        if (wv[i] == T(0)) { li = T(0); lp[i] = T(0); } // weight 0 (and samples past N) really excludes the sample
        l += li;
    }
    pp += wave;
    if (g0 == 0) {
        const T s = wave_sum_to_lane63(l);
        if (lane == 63) pp[0] = s;
    }
    DE_UNROLL for (int k = 0; k < GC; k++) {
        if (g0 + k < G) { // wave-uniform
            T c = T(0);
            DE_UNROLL for (int i = 0; i < VS; i++) c += wv[i] == T(0) ? T(0) : lp[i] * st.d[k][i];
            const T s = wave_sum_to_lane63(c);
            if (lane == 63) pp[(int64_t)(1 + g0 + k) * 4] = s;
        }
    }
}
// The Jacobian block and the values are written once and never read by the kernel: non-temporal stores (round 5: the eval kernel's output
// stream gained 2 % from them; -DDE_GRAD_NT_STORE=0 for the A/B)
#ifndef DE_GRAD_NT_STORE
#define DE_GRAD_NT_STORE 1
#endif
#if DE_GRAD_NT_STORE
#define DE_G_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define DE_G_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif
template <typename T, int GC, bool SHARE>
__device__ __noinline__ void g_epilogue_store(GState<T, GC> st, T *out_row, T *grad_tree, int64_t N, int G, int g0, int64_t tile, uint32_t stage0) {
    // stage0 = LDS address of the wave's slot area; `tile` counts 64 x VS samples under shared leaf rows, 256 x VS otherwise
    constexpr int TILE = GBLK * VS, WSAMP = 64 * VS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t jl = SHARE ? tile * WSAMP + (int64_t)lane * VS : tile * TILE + (int64_t)tid * VS, last = N - 1; // this lane's first sample
    if (G <= GC) {
        // One window: the wave's gradient block — WSAMP samples x G rows, gradient index fastest
        // (src/EvaluateDerivative.jl:355-361) — is WSAMP*G contiguous elements in memory.  Transpose it
        // through the wave's (now idle) slot rows and write it with 16-byte stores; per-lane stores
        // would each touch 4 of every 4*G*VS bytes.
        DE_UNROLL for (int i = 0; i < VS; i++) {
            const int64_t j = jl + i;
            if (j <= last && out_row) DE_G_STORE(out_row + j, (T)st.x[i]);
        }
        const int64_t j0 = SHARE ? tile * WSAMP : tile * TILE + (int64_t)wave * WSAMP; // first sample of this wave
        const int64_t n_valid = N - j0 < WSAMP ? N - j0 : WSAMP; // samples of this wave inside N (may be <= 0)
        if (G > 0 && n_valid > 0) {
            DE_UNROLL for (int i = 0; i < VS; i++)
                DE_UNROLL for (int k = 0; k < GC; k++)
                    if (k < G) *reinterpret_cast<__attribute__((address_space(3))) T *>((uintptr_t)(stage0 + (uint32_t)(((lane * VS + i) * G + k) * sizeof(T)))) = st.d[k][i];
            __builtin_amdgcn_wave_barrier(); // LDS is in order within a wave; this only pins the compiler
            constexpr int PER16 = 16 / (int)sizeof(T);
            T *__restrict__ gdst = grad_tree + (int64_t)G * j0;
            const int total = (int)n_valid * G; // elements to write
            const bool aligned = (reinterpret_cast<uintptr_t>(gdst) & 15) == 0;
            for (int e = lane * PER16; e < total; e += 64 * PER16) {
                typedef T V16 __attribute__((ext_vector_type(PER16)));
                const V16 v = *reinterpret_cast<__attribute__((address_space(3))) V16 *>((uintptr_t)(stage0 + (uint32_t)(e * sizeof(T))));
                if (aligned && e + PER16 <= total) DE_G_STORE(reinterpret_cast<V16 *>(gdst + e), v);
                else { DE_UNROLL for (int q = 0; q < PER16; q++) if (e + q < total) gdst[e + q] = v[q]; }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else { // several windows: every window owns a few rows of the block — direct stores
        DE_UNROLL for (int i = 0; i < VS; i++) {
            const int64_t j = jl + i;
            if (j <= last) {
                if (out_row && g0 == 0) out_row[j] = st.x[i];
                T *__restrict__ gp = grad_tree + (int64_t)G * j + g0;
                DE_UNROLL for (int k = 0; k < GC; k++)
                    if (g0 + k < G) gp[k] = st.d[k][i];
            }
        }
    }
}

// VS consecutive samples per thread; wave-major LDS: per wave, rows [0,F) = its slice of the X tile, then each
// spill slot s owns 1+GC rows (x, d[0..GC)); at least GC rows follow the X rows (output staging).
template <typename T, int GC, bool PARAMS, bool SHARE = false>
__global__ void __launch_bounds__(GBLK) de_grad_threaded_kernel(const GArgs<T> a, const uint64_t hbase, const uint32_t param_off) {
    constexpr int TILE = GBLK * VS, WSAMP = 64 * VS; // samples per workgroup / per wave
    extern __shared__ __align__(16) unsigned char gtsmem[];
    T *__restrict__ rows = reinterpret_cast<T *>(gtsmem);

    const GTileMap tm = gmap_block_prio(a, blockIdx.x);
    if (!tm.valid) return;
    const int flag_protocol = tm.prio ? 1 : a.skip_flagged;
    const int tid = threadIdx.x;
    // SHARED LEAF ROWS (GArgs::share): the tile is one wave's 64 x VS samples, staged once for the four waves — rows [0, F) —, behind them
    // every wave's slot rows; wave w runs trees t0 + w, t0 + w + 4, ... of the chunk through stream variant w
    constexpr bool share = SHARE; // (a template parameter: the per-wave-copy kernel keeps the code it had)
    const int wave = share ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int64_t base = tm.tile * (share ? WSAMP : TILE);
    const int64_t last = a.N - 1;
    const int g0 = (int)blockIdx.y * GC; // first gradient component of this window
    const int F = a.F;
    const int slot_rows = a.n_slots * (1 + GC) > GC ? a.n_slots * (1 + GC) : GC;
    const int R = F + slot_rows; // rows per wave
    if (share) {
        const uint32_t Fu = (uint32_t)a.FX, total = (uint32_t)WSAMP * Fu;
        for (uint32_t e = tid; e < total; e += GBLK) {
            const uint32_t j = e / Fu, f = e - j * Fu;
            int64_t jj = base + j;
            jj = jj < last ? jj : last;
            rows[f * WSAMP + j] = a.X[f + a.ldX * jj];
        }
        if (PARAMS) {
            const uint32_t Pu = (uint32_t)(F - a.FX), totalp = (uint32_t)WSAMP * Pu;
            for (uint32_t e = tid; e < totalp; e += GBLK) {
                const uint32_t j = e / Pu, q = e - j * Pu;
                int64_t jj = base + j;
                jj = jj < last ? jj : last;
                const int64_t cl = clamp_class((a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj] : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj]) - a.class_base, a.n_classes);
                rows[((uint32_t)a.FX + q) * WSAMP + j] = a.params[q + a.ld_params * cl];
            }
        }
    } else {
    {
        const uint32_t Fu = (uint32_t)a.FX;
        const uint32_t total = (uint32_t)TILE * Fu;
        for (uint32_t e = tid; e < total; e += GBLK) {
            const uint32_t j = e / Fu, f = e - j * Fu;
            int64_t jj = base + j;
            jj = jj < last ? jj : last;
            rows[((j / WSAMP) * (uint32_t)R + f) * WSAMP + (j % WSAMP)] = a.X[f + a.ldX * jj];
        }
    }
    if (PARAMS) { // rows FX .. F: params[:, class of the sample] (src/ParametricExpression.jl:381-389), read through the caches
        const uint32_t Pu = (uint32_t)(F - a.FX), total = (uint32_t)TILE * Pu;
        for (uint32_t e = tid; e < total; e += GBLK) {
            const uint32_t j = e / Pu, q = e - j * Pu;
            int64_t jj = base + j;
            jj = jj < last ? jj : last;
            const int64_t cl = clamp_class((a.classes_is_i64 ? reinterpret_cast<const int64_t *>(a.classes)[jj] : (int64_t) reinterpret_cast<const int32_t *>(a.classes)[jj]) - a.class_base, a.n_classes);
            rows[((j / WSAMP) * (uint32_t)R + (uint32_t)a.FX + q) * WSAMP + (j % WSAMP)] = a.params[q + a.ld_params * cl];
        }
    }
    } // one copy of the leaf rows per wave
    __syncthreads();

    const ConstU4Ptr code = (ConstU4Ptr)(uintptr_t)(a.code + (share ? (int64_t)wave * a.var_stride : (int64_t)0));
    const ConstI32Ptr code_off = (ConstI32Ptr)(uintptr_t)a.code_off;
    const ConstI64Ptr col_off = (ConstI64Ptr)(uintptr_t)a.col_off;
    const ConstI64Ptr grad_off = (ConstI64Ptr)(uintptr_t)a.grad_off;
    const ConstI32Ptr n_grad = (ConstI32Ptr)(uintptr_t)a.n_grad;
    const int t0 = tm.chunk * a.trees_per_chunk;
    const int t1 = (t0 + a.trees_per_chunk < a.n_trees) ? t0 + a.trees_per_chunk : a.n_trees;
    const uint32_t wave_base = (uint32_t)(uintptr_t)gtsmem + (share ? 0u : (uint32_t)(wave * R) * grow_bytes<T>());
    const uint32_t lds0 = wave_base + (uint32_t)(tid & 63) * (uint32_t)(VS * sizeof(T));
    const uint32_t stage0 = wave_base + (uint32_t)(F + (share ? wave * slot_rows : 0)) * grow_bytes<T>(); // the wave's slot area, free between trees
    const int64_t ptile = share ? (int64_t)tm.tile >> 2 : (int64_t)tm.tile; // (the tile of 256 x VS samples the partial sums are kept by)

    const ConstI32Ptr tree_ids = (ConstI32Ptr)(uintptr_t)a.tree_ids;
    const uint64_t skip = gskip_mask(a.ok, tree_ids, t0, t1, flag_protocol, (int64_t)tm.tile);
    for (int ti = share ? t0 + wave : t0; ti < t1; ti += share ? 4 : 1) {
        if ((skip >> (ti - t0)) & 1ull) continue; // already incomplete (early exit)
        const int tree = tree_ids[ti];
        const int G = n_grad[tree];
        if (g0 >= G && g0 > 0) continue; // window without a component of this tree (window 0 always runs: x, flag)
        int pc = code_off[tree];
        const int pe = code_off[tree + 1];
        GState<T, GC> st;
        st.x = lv_splat<T>(T(0));
        DE_UNROLL for (int k = 0; k < GC; k++) st.d[k] = lv_splat<T>(T(0));
        st.poison = T(0);
        st.g0 = (uint32_t)g0 | (lds0 & 0xFFFF0000u); // window offset (< 2^16) | upper half of the lane's LDS base (g_gen)
        {   // one call per tree: the chain ends in the tree's end record (g_end)
            const ConstU4Ptr rec = code + pc;
            const U32x4 hd = *rec;
            const uint32_t first = code[pe - 1].x; // the end record names the tree's first handler (every other record: its successor's)
            st = reinterpret_cast<GHandlerFn<T, GC>>(hbase + first)(st, lds0, rec + 1, hbase, hd.x, hd.y, grec_imm<T>(hd));
        }
        // a non-finite d[k] always survives to the root (every update is linear in it), so the gradient
        // is validity-tested once, here; x was tested where the lowering kept a test (H_CHECK_OUT)
        T poison = st.poison;
        gpoison<T>(poison, st.x);
        // only the components this tree really has: a window is as wide as its bucket, and a column beyond
        // n_grad (the lone column of a tree without constants in constant mode) holds g*0 terms that are NaN
        // for an infinite partial although the reference's gradient matrix has no such row
        DE_UNROLL for (int k = 0; k < GC; k++) gpoison<T>(poison, g0 + k < G ? st.d[k] : lv_splat<T>(T(0)));
        if (a.loss_mode) {
            const int64_t n_cols = col_off[a.n_all_trees];
            g_epilogue_loss<T, GC, SHARE>(st, a.y, a.w, a.partial + (ptile * n_cols + col_off[tree]) * 4, a.N, a.loss_mode, G, g0, (int64_t)tm.tile);
        } else {
            g_epilogue_store<T, GC, SHARE>(st, a.out ? a.out + (int64_t)tree * a.ld_out : nullptr, a.grad + grad_off[tree], a.N, G, g0, (int64_t)tm.tile, stage0);
        }
        if (__ballot(poison != poison) != 0ull) gflag_incomplete(a.ok + tree, flag_protocol == 1);
    }
}

// ---- host side: entry points of this (type, window) module -----------------------------------------
} // module namespace

hipError_t DE_GT_NAME(grad_thr_fetch_)(uint64_t *host_table) {
    using namespace DE_GT_NAME(gtm_);
    uint64_t *d = nullptr;
    hipError_t st = hipMalloc(reinterpret_cast<void **>(&d), GOP_MAX * sizeof(uint64_t));
    if (st != hipSuccess) return st;
    hipLaunchKernelGGL((de_grad_fill_handlers<DE_GT_T, DE_GT_GC>), dim3(1), dim3(1), 0, 0, d);
    st = hipMemcpy(host_table, d, gop_count(DE_GT_GC) * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return st;
}

hipError_t DE_GT_NAME(grad_thr_launch_)(const GradArgs &ga, int bucket, hipStream_t stream) {
    using namespace DE_GT_NAME(gtm_);
    const GradArgs::Bucket &bk = ga.buckets[bucket];
    const int windows = bk.windows;
    typedef DE_GT_T T;
    constexpr int GC = DE_GT_GC;
    static int gt_gcu = 0;
    const EvalArgs &e = ga.e;
    GArgs<T> a;
    a.code = ga.threaded_code;
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
    const bool share = ga.gt_share;
    a.share = share ? 1 : 0;
    a.var_stride = share ? ga.gt_var_stride : 0;
    const int tile_samples = share ? 64 * VS : GBLK * VS;
    // (shared rows: whole groups of four tiles — a fused-loss launch must write every (tile of 256 x VS samples, wave) entry of the partial
    // sums, also the all-padding quarter tiles behind N, as the four-wave workgroup did)
    a.n_tiles = share ? 4 * ((e.N + GBLK * VS - 1) / (GBLK * VS)) : (e.N + tile_samples - 1) / tile_samples;
    a.FX = e.F;
    a.F = e.F + (e.uses_params ? ga.P : 0); // leaf rows: X, then the parameters gathered by class
    a.P = ga.P;
    a.n_trees = bk.n;
    a.n_all_trees = e.n_trees;
    a.tree_ids = bk.ids;
    a.n_slots = bk.n_slots; // spill slots the trees of this bucket need
    a.mode = ga.mode;
    a.classes_is_i64 = e.classes_is_i64;
    a.class_base = e.class_base;
    a.n_classes = e.n_classes > 0 ? e.n_classes : 1;
    a.uses_params = e.uses_params ? 1 : 0;
    a.check = 1;
    a.skip_flagged = e.skip_flagged ? 1 : 0;
    a.diff_g0 = -1;
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
    if (gt_gcu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) gt_gcu = prop.multiProcessorCount;
        if (gt_gcu <= 0) gt_gcu = 256;
    }
    int64_t n_chunks = (bk.n + 31) / 32;
    const int64_t want_blocks = (int64_t)gt_gcu * 4 * 8;
    if (a.n_tiles * n_chunks * windows < want_blocks) n_chunks = (want_blocks + a.n_tiles * windows - 1) / (a.n_tiles * windows);
    const int64_t max_chunks = (bk.n + 3) / 4;
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    if (n_chunks < 1) n_chunks = 1;
    a.trees_per_chunk = (int32_t)((bk.n + n_chunks - 1) / n_chunks);
    // ... unless the launch has priority tiles (below): the flags are then down before most workgroups first look, a stale line is rare and
    // the cached protocol wins (fused loss gradient 6.65 -> 5.98 ms; the reverse kernel keeps protocol 1: 15.9 against 16.9 ms)
    if (a.skip_flagged) a.skip_flagged = (a.trees_per_chunk >= 8 && !ga.prio_ready) ? 1 : 2;
    if (a.skip_flagged) { const char *pv = getenv("DE_SKIP_PROTOCOL"); if (pv && *pv >= '1' && *pv <= '3') a.skip_flagged = *pv - '0'; } // (experiments) // flag protocol (skip_flag_load, de_device_ops.h): these kernels write little, their L1 lines go stale under 2 (reverse kernel 17.0 / 16.0 ms); 2 only for tiny chunks (many tiles on one flag line)
    a.n_chunks = (int32_t)((bk.n + a.trees_per_chunk - 1) / a.trees_per_chunk);
    int64_t blocks = ((a.n_tiles + 7) / 8) * 8 * a.n_chunks;
    a.prio = nullptr;
    a.n_prio = a.n_prio_blocks = a.prio_shift = 0;
    if (a.skip_flagged && ga.prio_ready) blocks += gprio_setup(a, e.prio_keys, e.F, tile_samples);
    if (blocks <= 0 || blocks > 0x7fffffffLL || windows > 65535) return hipErrorInvalidValue;
    const size_t slot_rows = std::max<size_t>((size_t)a.n_slots * (1 + GC), (size_t)GC);
    const size_t lds = (share ? (size_t)a.F + 4 * slot_rows : 4 * ((size_t)a.F + slot_rows)) * 64 * VS * sizeof(T); // 4 waves x rows x one wave's samples (shared leaf rows: once)
    void (*kern)(const GArgs<T>, uint64_t, uint32_t) = e.uses_params ? de_grad_threaded_kernel<T, GC, true> : de_grad_threaded_kernel<T, GC, false>;
    if (share) kern = e.uses_params ? de_grad_threaded_kernel<T, GC, true, true> : de_grad_threaded_kernel<T, GC, false, true>;
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        hipError_t st = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (st != hipSuccess) return st;
    }
    if (a.n_prio) { // the priority tiles as a launch of their own in front, in short chunks (de_kernels.hip launch_threaded_t: no blind first wave)
        GArgs<T> pa = a;
        pa.trees_per_chunk = 4;
        pa.n_chunks = (int32_t)((bk.n + 3) / 4);
        pa.n_prio_blocks = (uint32_t)(((int64_t)pa.n_prio * pa.n_chunks + 7) / 8 * 8);
        hipLaunchKernelGGL(kern, dim3(pa.n_prio_blocks, (unsigned)windows), dim3(GBLK), lds, stream, pa, bk.handler_base, bk.param_handler_off);
        const hipError_t ps = hipGetLastError();
        if (ps != hipSuccess) return ps;
        blocks -= a.n_prio_blocks;
        a.n_prio = a.n_prio_blocks = 0;
        a.prio = nullptr;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)windows), dim3(GBLK), lds, stream, a, bk.handler_base, bk.param_handler_off);
    return hipGetLastError(); // the loss reduction passes run once, after the last bucket (de_grad_kernels.hip)
}


} // namespace de
