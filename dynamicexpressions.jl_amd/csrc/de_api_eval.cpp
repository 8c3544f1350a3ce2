// de_api_eval.cpp — C ABI (include/de_hip.h): de_eval, de_eval_loss, de_eval_sum_certificate, de_eval_tree_array: staging of host buffers,
// launch planning (priority tiles, probe launch, compaction) and the certificate program.
#include "de_api_internal.h"

extern "C" {
int stage_in(de_ctx *c, DevBuf &buf, const void *user, size_t bytes, Staged *s) {
    s->dev = const_cast<void *>(user);
    s->staged = false;
    if (!user || bytes == 0 || is_device_ptr(user)) return DE_OK;
    HIP_TRY(c, buf.reserve(bytes));
    HIP_TRY(c, hipMemcpyAsync(buf.p, user, bytes, hipMemcpyHostToDevice, c->stream));
    s->dev = buf.p;
    s->staged = true;
    return DE_OK;
}
int stage_out(de_ctx *c, DevBuf &buf, void *user, size_t bytes, Staged *s) {
    s->dev = user;
    s->staged = false;
    if (!user || bytes == 0 || is_device_ptr(user)) return DE_OK;
    HIP_TRY(c, buf.reserve(bytes));
    s->dev = buf.p;
    s->staged = true;
    return DE_OK;
}

int check_param_args(de_ctx *c, const de_program *p, const de_param_args_t *pa, int64_t N) {
    if (!p->uses_params) return DE_OK;
    if (!pa || !pa->params || !pa->classes)
        return fail(c, DE_ERR_INVALID_ARG, "program has parameter leaves: params/classes required "
                                           "(reference: \"You must pass the `classes::Vector` argument\")");
    if (pa->ld_params < p->n_params || pa->n_classes <= 0) return fail(c, DE_ERR_INVALID_ARG, "bad parameter matrix shape");
    // `@assert maximum(classes) <= size(parameters, 2)` (src/ParametricExpression.jl:378-379): checked here when the ids are
    // host memory; ids already on the device are the caller's to check (the kernels clamp them, so a bad id cannot fault)
    if (N > 0 && !is_device_ptr(pa->classes)) {
        int64_t lo = pa->class_base, hi = pa->class_base;
        if (pa->classes_is_i64) {
            const int64_t *q = static_cast<const int64_t *>(pa->classes);
            for (int64_t j = 0; j < N; j++) { lo = std::min(lo, q[j]); hi = std::max(hi, q[j]); }
        } else {
            const int32_t *q = static_cast<const int32_t *>(pa->classes);
            for (int64_t j = 0; j < N; j++) { lo = std::min<int64_t>(lo, q[j]); hi = std::max<int64_t>(hi, q[j]); }
        }
        if (lo < pa->class_base || hi - pa->class_base >= pa->n_classes)
            return fail(c, DE_ERR_OUT_OF_RANGE, "class id outside [%d, %lld): maximum(classes) <= size(parameters, 2) violated",
                        (int)pa->class_base, (long long)(pa->class_base + pa->n_classes));
    }
    return DE_OK;
}

struct LossReq {
    const void *y, *w;
    int32_t kind;
    void *loss;
};

struct CertReq {
    uint8_t *certified; // host, n_trees
    double *max_abs;    // host, n_trees, or null
};
static int eval_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                     void *out, int64_t ld_out, uint8_t *ok, const LossReq *lr, const CertReq *cr = nullptr);
static int ensure_cert_program(de_ctx *c, de_program *p);

// The certificate program (de_eval_sum_certificate): the eval program's generic form with the result of EVERY operator validity-tested —
// the exact elision of de_lower.cpp (a test is dropped when the value's consumer maps a non-finite input onto a tested non-finite output)
// keeps the FLAG exact but drops values the reference still sums — bound for the flat-switch kernel; plus, per tree, the largest
// |constant operand| (deg0_eval of a constant is an array of N copies: the reference sums that too).  A superset of the arrays the
// reference sums (the inner values of its fused 2/3-node kernels are never materialised there): sound, slightly conservative.
// Cached per constants generation (it used to be rebuilt serially, with a stream synchronisation and an upload, at EVERY call);
// built on the pool of host threads like every other per-tree pass.
static int ensure_cert_program(de_ctx *c, de_program *p) {
    if (p->cert_gen == p->consts_gen && p->d_cert_code && p->d_cert_off) return DE_OK;
    const std::vector<Instr> &src = p->folded ? p->fcode : p->code;
    const std::vector<int32_t> &off = p->folded ? p->fcode_off : p->code_off;
    const int prb = p->prows ? p->n_features + p->n_slots : -1;
    std::vector<BoundInstr> bc;
    std::vector<int32_t> boff;
    p->cert_cmax.assign((size_t)p->n_trees, 0.0);
    build_stream_by_trees(p->n_trees, &bc, &boff, [&](int64_t t, std::vector<BoundInstr> *out) {
        std::vector<Instr> tmp(src.begin() + off[(size_t)t], src.begin() + off[(size_t)t + 1]);
        double cm = 0.0;
        for (Instr &ins : tmp) {
            if ((ins.hdr & H_OP_MASK) != DOP_LOAD) ins.hdr |= H_CHECK_OUT;
            if (((ins.hdr >> H_SRC_SHIFT) & H_SRC_MASK) == SRC_CONST) {
                const double v = p->dtype == DE_F32 ? (double)ins.imm.f32 : ins.imm.f64;
                if (v == v) cm = std::max(cm, std::fabs(v));
            }
        }
        p->cert_cmax[(size_t)t] = cm;
        bind_tree(tmp.data(), tmp.size(), true, p->n_features, out, prb);
    });
    bc.push_back(BoundInstr{0u, 0u, 0u, 0u}); // (the kernel prefetches pc + 1)
    HIP_TRY(c, hipSetDevice(c->device));
    if (p->cert_cap < bc.size()) {
        if (p->d_cert_code) (void)hipFree(p->d_cert_code);
        p->d_cert_code = nullptr;
        p->cert_cap = 0;
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_cert_code), bc.size() * sizeof(BoundInstr)));
        p->cert_cap = bc.size();
    }
    if (!p->d_cert_off) HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&p->d_cert_off), boff.size() * sizeof(int32_t)));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // (an earlier certificate launch may still read the buffers)
    HIP_TRY(c, hipMemcpy(p->d_cert_code, bc.data(), bc.size() * sizeof(BoundInstr), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(p->d_cert_off, boff.data(), boff.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    p->cert_gen = p->consts_gen;
    return DE_OK;
}

int de_eval_sum_certificate(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                            uint8_t *ok, uint8_t *certified, double *max_abs) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (N < 0 || !ok || !certified || (p->n_trees > 0 && N > 0 && !X)) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (is_device_ptr(certified) || (max_abs && is_device_ptr(max_abs))) return fail(c, DE_ERR_INVALID_ARG, "certified / max_abs are host arrays");
    if (p->direct) return fail(c, DE_ERR_UNSUPPORTED, "de_eval_sum_certificate needs the LDS-tiled kernel (feature matrix too wide)");
    if (!(p->options & DE_OPT_EARLY_EXIT) || N == 0) {
        // early_exit = false: the reference sums nothing (src/Evaluate.jl:16-32 are no-ops), the flag is the constant part alone; N = 0: sum(empty) = 0
        for (int64_t t = 0; t < p->n_trees; t++) { certified[t] = 1; if (max_abs) max_abs[t] = 0.0; }
        if (is_device_ptr(ok)) { HIP_TRY(c, hipSetDevice(c->device)); HIP_TRY(c, hipMemcpy(ok, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice)); }
        else std::memcpy(ok, p->host_ok_eval.data(), (size_t)p->n_trees);
        return DE_OK;
    }
    const CertReq cr{certified, max_abs};
    DE_NOTHROW(c, eval_impl(c, p, X, N, ldX, pa, nullptr, N, ok, nullptr, &cr)); // (builds host vectors: no exception may leave the C ABI)
}

int de_eval(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
            void *out, int64_t ld_out, uint8_t *ok) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (N < 0 || !ok || (p->n_trees > 0 && N > 0 && (!X || !out))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (ld_out < N) return fail(c, DE_ERR_INVALID_ARG, "ld_out < N");
    DE_NOTHROW(c, eval_impl(c, p, X, N, ldX, pa, out, ld_out, ok, nullptr));
}

int de_eval_loss(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                 const void *y, const void *w, int32_t loss_kind, void *loss, uint8_t *ok) {
    if (!c || !p) return DE_ERR_INVALID_ARG;
    if (N < 0 || !ok || (p->n_trees > 0 && (!loss || (N > 0 && (!X || !y))))) return fail(c, DE_ERR_INVALID_ARG, "null buffer");
    if (loss_kind != DE_LOSS_L2 && loss_kind != DE_LOSS_L1) return fail(c, DE_ERR_INVALID_ARG, "unknown loss_kind %d", loss_kind);
    if (!p->threaded)
        return fail(c, DE_ERR_UNSUPPORTED, "de_eval_loss needs the LDS-tiled kernel (feature matrix too wide for this build)");
    const LossReq lr{y, w, loss_kind, loss};
    DE_NOTHROW(c, eval_impl(c, p, X, N, ldX, pa, nullptr, N, ok, &lr));
}

static int eval_impl(de_ctx_t *c, de_program_t *p, const void *X, int64_t N, int64_t ldX, const de_param_args_t *pa,
                     void *out, int64_t ld_out, uint8_t *ok, const LossReq *lr, const CertReq *cr) {
    if (p->ctx != c) return fail(c, DE_ERR_INVALID_ARG, "program belongs to another context");
    if (ldX < p->n_features) return fail(c, DE_ERR_INVALID_ARG, "ldX < n_features");
    int rc = check_param_args(c, p, pa, N);
    if (rc != DE_OK) return rc;
    if (p->n_trees == 0) return DE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t es = p->dtype == DE_F32 ? 4 : 8;
    const bool ok_dev = is_device_ptr(ok);
    if (N == 0) { // nothing to evaluate: only the constant part of the flag (sum(empty) is finite)
        if (ok_dev) HIP_TRY(c, hipMemcpyAsync(ok, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
        else std::memcpy(ok, p->host_ok_eval.data(), (size_t)p->n_trees);
        if (lr) { // empty sum = 0; NaN where a constant already failed the flag
            std::vector<unsigned char> z((size_t)p->n_trees * es);
            for (int64_t t = 0; t < p->n_trees; t++) {
                const double v = p->host_ok_eval[(size_t)t] ? 0.0 : std::nan("");
                if (p->dtype == DE_F32) reinterpret_cast<float *>(z.data())[t] = (float)v;
                else reinterpret_cast<double *>(z.data())[t] = v;
            }
            if (is_device_ptr(lr->loss)) {
                HIP_TRY(c, hipMemcpyAsync(lr->loss, z.data(), z.size(), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            } else std::memcpy(lr->loss, z.data(), z.size());
        }
        return DE_OK;
    }
    const bool direct = p->direct && !(p->threaded && !cr); // (the threaded kernel stages a wider X than the flat-switch kernel can: de_api_program.cpp make_threaded)

    Staged sX, sOut, sOk, sPar, sCls, sY, sW, sLoss;
    LossArgs la;
    std::memset(&la, 0, sizeof la);
    rc = stage_in(c, c->sX, X, (size_t)ldX * (size_t)N * es, &sX);
    if (rc) return rc;
    if (lr) {
        rc = stage_in(c, c->sY, lr->y, (size_t)N * es, &sY);
        if (rc) return rc;
        if (lr->w) {
            rc = stage_in(c, c->sW, lr->w, (size_t)N * es, &sW);
            if (rc) return rc;
        }
        rc = stage_out(c, c->sLoss, lr->loss, (size_t)p->n_trees * es, &sLoss);
        if (rc) return rc;
        size_t pb = 0, sb = 0;
        loss_scratch_bytes(p->dtype, p->n_trees, N, &pb, &sb);
        HIP_TRY(c, c->sPartial.reserve(pb));
        HIP_TRY(c, c->sSeg.reserve(sb));
        la.y = sY.dev;
        la.w = lr->w ? sW.dev : nullptr;
        la.kind = lr->kind;
        la.partial = c->sPartial.p;
        la.seg_sum = c->sSeg.p;
        la.loss = sLoss.dev;
    } else if (!cr) {
        rc = stage_out(c, c->sOut, out, ((size_t)(p->n_trees - 1) * (size_t)ld_out + (size_t)N) * es, &sOut);
        if (rc) return rc;
    }
    if (cr) {
        rc = ensure_cert_program(c, p);
        if (rc) return rc;
        HIP_TRY(c, c->sCert.reserve((size_t)p->n_trees * es));
        HIP_TRY(c, hipMemsetAsync(c->sCert.p, 0, (size_t)p->n_trees * es, c->stream));
    }
    // ok[] starts as the host-side (constant) part of the flag; the kernel only clears bytes
    if (ok_dev) {
        sOk.dev = ok;
    } else {
        HIP_TRY(c, c->sOk.reserve((size_t)p->n_trees));
        sOk.dev = c->sOk.p;
        sOk.staged = true;
    }
    // (the device copy of the initial flags goes in through the launch: EvalArgs.ok_init)
    if (!p->d_ok_eval) HIP_TRY(c, hipMemcpyAsync(sOk.dev, p->host_ok_eval.data(), (size_t)p->n_trees, hipMemcpyHostToDevice, c->stream));
    if (p->uses_params) {
        rc = stage_in(c, c->sParams, pa->params, (size_t)pa->ld_params * (size_t)pa->n_classes * es, &sPar);
        if (rc) return rc;
        rc = stage_in(c, c->sClasses, pa->classes, (size_t)N * (pa->classes_is_i64 ? 8 : 4), &sCls);
        if (rc) return rc;
    }
    EvalArgs a;
    std::memset(&a, 0, sizeof a);
    a.code = p->d_code;
    a.code_off = p->d_code_off;
    a.n_trees = (int32_t)p->n_trees;
    a.n_slots = p->n_slots + (p->prows ? p->n_params : 0); // (LDS rows behind X: spill slots, then the staged parameter rows)
    a.prow_base = p->prows ? p->n_features + p->n_slots : 0;
    a.n_prows = p->prows ? p->n_params : 0;
    a.uses_params = p->uses_params;
    a.X = sX.dev;
    a.N = N;
    a.ldX = ldX;
    a.F = p->n_features;
    a.out = sOut.dev;
    a.ld_out = ld_out;
    a.ok = static_cast<uint8_t *>(sOk.dev);
    a.ok_init = p->d_ok_eval;
    if (p->uses_params) {
        a.params = sPar.dev;
        a.ld_params = pa->ld_params;
        a.n_classes = pa->n_classes;
        a.classes = sCls.dev;
        a.classes_is_i64 = pa->classes_is_i64;
        a.class_base = pa->class_base;
    }
    a.early_exit = (p->options & DE_OPT_EARLY_EXIT) != 0;
    a.skip_flagged = a.early_exit && !(p->options & DE_OPT_FULL_EVAL) && tree_skip_enabled();
    a.turbo = (p->options & DE_OPT_TURBO) != 0;
    a.threaded = p->threaded && !direct;
    a.direct = direct;
    a.loss = lr ? &la : nullptr;
    HIP_TRY(c, c->sPrio.reserve((size_t)3 * DE_PRIO_MAX_F * sizeof(unsigned long long)));
    a.prio_keys = c->sPrio.p;
    a.prio_keys_ready = !sX.staged && dataset_keys(c, p->dtype, X, N, ldX, p->n_features, &a.prio_keys);
    a.compact_code = p->d_compact_code;
    a.compact_ints = p->d_compact_ints;
    a.waves = p->waves; // (wave groups: de_api_internal.h)
    a.wave_slots = p->n_slots;
    a.var_stride = p->var_stride;
    if (cr) { // the certificate pass: the un-elided program through the flat-switch kernel's CERT variant, nothing stored
        a.code = p->d_cert_code;
        a.code_off = p->d_cert_off;
        a.threaded = false;
        a.cert_max = c->sCert.p;
        a.out = nullptr;
        a.prio_keys = nullptr;
        a.compact_code = nullptr;
        a.compact_ints = nullptr;
    }
    HIP_TRY(c, time_begin(c));
    a.compacted = &p->last_compacted;
    p->last_compacted = false;
    HIP_TRY(c, launch_eval(p->dtype, a, c->stream, &c->last_kernel));
    HIP_TRY(c, time_end(c));
    if (sLoss.staged) HIP_TRY(c, hipMemcpyAsync(lr->loss, sLoss.dev, (size_t)p->n_trees * es, hipMemcpyDeviceToHost, c->stream));
    if (sOut.staged) {
        if (ld_out == N) // one block (the constant-folding population is 10^3..10^5 one-sample rows)
            HIP_TRY(c, hipMemcpyAsync(out, sOut.dev, (size_t)p->n_trees * (size_t)N * es, hipMemcpyDeviceToHost, c->stream));
        else
            for (int64_t t = 0; t < p->n_trees; t++) // rows are strided in the caller's buffer
                HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(out) + (size_t)t * (size_t)ld_out * es,
                                          static_cast<char *>(sOut.dev) + (size_t)t * (size_t)ld_out * es,
                                          (size_t)N * es, hipMemcpyDeviceToHost, c->stream));
    }
    if (sOk.staged) HIP_TRY(c, hipMemcpyAsync(ok, sOk.dev, (size_t)p->n_trees, hipMemcpyDeviceToHost, c->stream));
    if (cr) {
        // certified[t]: the reference's `complete` provably equals ok[t].  It tests isfinite(sum(x)) over N values (src/ValueInterface.jl:9)
        // where the device tests every element: the two differ only when all elements are finite and a (partial) sum overflows — impossible
        // while N * max|x| stays below the largest finite value.  ok[t] == 0 means some element is non-finite: its sum is too.
        std::vector<unsigned char> mx((size_t)p->n_trees * es);
        std::vector<uint8_t> okh((size_t)p->n_trees);
        HIP_TRY(c, hipMemcpyAsync(mx.data(), c->sCert.p, mx.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(okh.data(), sOk.dev, okh.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const double top = p->dtype == DE_F32 ? (double)std::numeric_limits<float>::max() : std::numeric_limits<double>::max();
        for (int64_t t = 0; t < p->n_trees; t++) {
            double m = p->dtype == DE_F32 ? (double)reinterpret_cast<float *>(mx.data())[t] : reinterpret_cast<double *>(mx.data())[t];
            m = std::max(m, p->cert_cmax[(size_t)t]);
            if (cr->max_abs) cr->max_abs[t] = m;
            cr->certified[t] = (!okh[(size_t)t] || m * (double)N * 1.001 < top || !(m == m)) ? 1 : 0; // (the margin: the summation's own roundings)
            if (!std::isfinite(m) && okh[(size_t)t]) cr->certified[t] = 0;
        }
    }
    if (sX.staged || sOut.staged || sOk.staged || sPar.staged || sCls.staged || sY.staged || sW.staged || sLoss.staged)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (sOut.staged && a.skip_flagged) {
        // host output buffer: the rows of incomplete trees were only partly written on the device, and the staging buffer is shared by
        // every program of the context — they would carry an earlier call's data.  NaN-fill them (what the callable sugar does anyway,
        // src/EvaluationHelpers.jl:29-33); the flags come from the caller's host array or, for a device `ok`, from a copy.
        std::vector<uint8_t> okh;
        const uint8_t *okp = ok;
        if (ok_dev) {
            okh.resize((size_t)p->n_trees);
            HIP_TRY(c, hipMemcpy(okh.data(), ok, (size_t)p->n_trees, hipMemcpyDeviceToHost));
            okp = okh.data();
        }
        for (int64_t t = 0; t < p->n_trees; t++) {
            if (okp[t]) continue;
            if (p->dtype == DE_F32) std::fill_n(static_cast<float *>(out) + (size_t)t * (size_t)ld_out, (size_t)N, std::nanf(""));
            else std::fill_n(static_cast<double *>(out) + (size_t)t * (size_t)ld_out, (size_t)N, std::nan(""));
        }
    }
    return DE_OK;
}

int de_eval_tree_array(de_ctx_t *c, int dtype, const de_tape_node_t *nodes, int64_t n_nodes, const void *consts,
                       int64_t n_consts, const void *X, int32_t n_features, int64_t N, uint32_t options, void *out,
                       uint8_t *ok) {
    if (!c) return DE_ERR_INVALID_ARG;
    const int64_t noff[2] = {0, n_nodes}, coff[2] = {0, n_consts};
    de_program_t *p = nullptr;
    int rc = de_program_create(c, dtype, nodes, noff, 1, consts, coff, n_features, 0, options, &p);
    if (rc != DE_OK) return rc;
    rc = de_eval(c, p, X, N, n_features, nullptr, out, N, ok);
    if (rc == DE_OK) rc = de_ctx_synchronize(c);
    de_program_destroy(p);
    return rc;
}

} // extern "C"
