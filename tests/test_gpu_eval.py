"""GPU parity tests of eval_tree_array through the C ABI (libde_hip.so) vs the CPU oracle and
the reference's golden vectors.  Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from helpers import case_X, case_tree, load_golden, parity_tolerance
from oracle import oracle

pytestmark = pytest.mark.gpu

CASES = [c for c in load_golden() if c["kind"] in ("eval", "flag", "param")]


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()  # fail loudly if libde_hip.so is missing
    return _api


def ctx_of(case, api):
    o = case.get("options", {})
    return api.EvalContext(early_exit=o.get("early_exit", True), use_fused=o.get("use_fused", True),
                           bumper=o.get("bumper", False))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_known_answers_on_gpu(case, api):
    tree, ops = case_tree(case)
    X = case_X(case)
    exp = case["expect"]
    if case["kind"] == "param":
        ex = api.ParametricExpression(tree, ops, np.asarray(exp["params"], dtype=X.dtype))
        out, ok = ex.eval_tree_array(X, exp["classes"], eval_context=ctx_of(case, api))
    else:
        out, ok = api.eval_tree_array(tree, X, ops, eval_context=ctx_of(case, api))
    assert ok == exp["ok"], f"{case['name']} ({case['cite']})"
    if ok and "y" in exp:
        want = np.asarray(exp["y"], dtype=np.float64)
        for i in exp.get("y_nonfinite_idx", []):
            assert not np.isfinite(out[i])
        m = np.isfinite(want)
        tol = max(exp.get("atol", 0), 1e-30) + max(exp.get("rtol", 0), 1e-5 if X.dtype == np.float32 else 1e-13) * np.abs(want[m])
        err = np.abs(out[m].astype(np.float64) - want[m])
        assert np.all(err <= tol), f"{case['name']}: max err {err.max()}"


ILL_FRACTION = {}  # label -> share of compared samples the tolerance model classed as ill-conditioned (printed, capped)


def compare_population(api, trees, ops, X, dtype, eval_context=None, use_torch=False, min_ok=1, max_ill=0.05, label=None, tally=None):
    pop = api.Population(trees, ops, dtype, n_features=X.shape[0], eval_context=eval_context)
    if use_torch:
        import torch
        Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda().t()
        out, ok = pop.eval(Xd)
        torch.cuda.synchronize()
        out, ok = out.cpu().numpy(), ok.cpu().numpy()
    else:
        out, ok = pop.eval(X)
    opts = (eval_context or api.EvalContext()).option_bits(ops)
    n_ok = n_quirk = n_cmp = n_ill = 0
    worst = 0.0
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, dtype)
        y, ok_el = oracle.eval_tree_array(tape, consts, X, opts, elementwise=True)
        _, ok_ref = oracle.eval_tree_array(tape, consts, X, opts)
        n_quirk += ok_el != ok_ref  # isfinite(sum(x)) overflow quirk: documented divergence
        assert bool(ok[t]) == ok_el, f"flag mismatch tree {t}: {de.string_tree(tree, ops)}"
        if ok_el:
            n_ok += 1
            m = np.isfinite(y)
            tol_all = parity_tolerance(tree, ops, X, dtype, opts)
            # same finite/non-finite pattern wherever the sample is well-conditioned (next to an overflow or
            # behind a chaotic intermediate, Inf-vs-finite is as implementation-dependent as the value)
            wc = np.isfinite(tol_all)
            assert np.array_equal(np.isfinite(out[t])[wc], m[wc])
            m = m & np.isfinite(out[t])
            tol = tol_all[m]
            n_cmp += int(m.sum())
            n_ill += int(np.isinf(tol).sum())
            err = np.abs(out[t][m].astype(np.float64) - y[m])
            assert np.all(err <= tol), f"value mismatch tree {t}: {de.string_tree(tree, ops)} max err {err.max()}"
            with np.errstate(divide="ignore", invalid="ignore"):
                rel = np.nanmax(np.where(np.abs(y[m]) > 0, err / np.abs(y[m]), 0)) if m.any() else 0
            worst = max(worst, float(rel))
    assert n_ok >= min_ok
    frac = n_ill / max(n_cmp, 1)
    ILL_FRACTION[label or f"{len(trees)} trees x {X.shape[1]} {np.dtype(dtype).name}"] = frac
    print(f"[eval parity {label or ''} {len(trees)} trees x {X.shape[1]} {np.dtype(dtype).name}] {n_cmp} samples bounded, "
          f"{n_ill} ({100.0 * frac:.2f} %) ill-conditioned (flags/finiteness only), worst rel err {worst:.3g}")
    if tally is not None:  # the caller caps the share over ALL its calls (tiny launches: a handful of samples each)
        tally["compared"] = tally.get("compared", 0) + n_cmp
        tally["ill"] = tally.get("ill", 0) + n_ill
    else:
        assert frac <= max_ill, f"{n_ill} of {n_cmp} samples ill-conditioned (cap {max_ill})"
    pop.close()
    return n_ok, n_quirk, worst


@pytest.mark.parametrize("N", [1, 63, 1024, 4099])
def test_random_population_f32_vs_oracle(api, N):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(200, seed=0xDE02)
    X = de.synth.random_X(5, N, seed=1)
    n_ok, n_quirk, worst = compare_population(api, trees, ops, X, np.float32, min_ok=20)
    assert n_quirk == 0


def test_random_population_f64_vs_oracle(api):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(100, seed=0xDE03, dtype=np.float64)
    X = de.synth.random_X(5, 2051, seed=2, dtype=np.float64)
    compare_population(api, trees, ops, X, np.float64, min_ok=20)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_chunk_and_tile_boundaries_of_the_chained_stream(api, dtype):
    """The eval kernel runs a chunk of consecutive trees as ONE chain (every tree's end record runs on into the next tree,
    csrc/de_kernels.hip h_tree_end): populations of 1 .. 130 trees (one tree, one short chunk, exactly 64, one more, three
    chunks; few samples re-split the launch into 8-tree chunks) x sample counts around the 256-sample tile (one sample, one
    short of a tile, exactly two tiles, ragged third tile), leaf-only trees included; values and flags against the oracle."""
    ops = de.synth.BENCH_OPERATORS
    rng = de.synth.Xoshiro256ss(77)
    pool = [de.synth.gen_random_tree_fixed_size(1 + (i * 5) % 23, ops, 5, rng, dtype) for i in range(130)]
    g = np.random.Generator(np.random.PCG64(3))
    tally = {}
    for n_trees in (1, 7, 8, 9, 64, 65, 130):
        for N in (1, 255, 257, 511, 1024, 1031):
            X = np.asfortranarray(g.standard_normal((5, N)).astype(dtype))
            compare_population(api, pool[:n_trees], ops, X, dtype, min_ok=0, tally=tally)
            if n_trees in (9, 65):  # the same on device tensors (16-byte aligned rows or not, by N)
                compare_population(api, pool[:n_trees], ops, X, dtype, use_torch=True, min_ok=0, tally=tally)
    share = tally["ill"] / max(tally["compared"], 1)
    print(f"[chunk / tile boundaries {np.dtype(dtype).name}] {tally['compared']} samples bounded over all launches, {100.0 * share:.2f} % ill-conditioned")
    assert tally["compared"] > 50000 and share <= 0.05  # (one-sample launches cannot be capped one by one: capped together)


def test_torch_device_tensors_zero_copy_path(api):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(64, seed=11)
    X = de.synth.random_X(5, 5000, seed=5)
    compare_population(api, trees, ops, X, np.float32, use_torch=True, min_ok=10)


def test_wide_operator_set_and_all_option_modes(api):
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^", "mod", "rem", "greater"),
                          unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin",
                                           "safe_sqrt", "relu", "sign", "round", "atan"))
    rng = de.synth.Xoshiro256ss(99)
    trees = [de.synth.gen_random_tree_fixed_size(5 + i % 20, ops, 3, rng, np.float32) for i in range(150)]
    g = np.random.Generator(np.random.PCG64(3))
    X = np.asfortranarray(g.standard_normal((3, 777)).astype(np.float32))
    X[1, 5] = np.inf
    X[0, 700] = np.nan
    for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False),
               api.EvalContext(bumper=True)):
        # ^ / pow_abs2 / mod / round next to their discontinuities: a larger ill-conditioned share than the bench set
        compare_population(api, trees, ops, X, np.float32, eval_context=ec, min_ok=0, max_ill=0.2, label="wide operator set")


def test_ieee_exact_operators_are_bit_identical(api):
    """+ - * / sqrt abs neg max min square are IEEE-exact on both sides: bit-for-bit equality."""
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min"),
                          unary_operators=("neg", "abs", "square", "safe_sqrt"))
    rng = de.synth.Xoshiro256ss(5)
    for dtype in (np.float32, np.float64):
        trees = [de.synth.gen_random_tree_fixed_size(4 + i % 28, ops, 5, rng, dtype) for i in range(120)]
        X = de.synth.random_X(5, 3000, seed=9, dtype=dtype)
        pop = api.Population(trees, ops, dtype, n_features=5, eval_context=api.EvalContext(early_exit=False))
        out, ok = pop.eval(X)
        for t, tree in enumerate(trees):
            tape, consts = de.flatten(tree, ops, dtype)
            y, ok_ref = oracle.eval_tree_array(tape, consts, X, 6)
            assert bool(ok[t]) == ok_ref  # early_exit=false: only constant folding can fail
            if not ok_ref:
                continue  # reference returns an unfilled buffer (src/Evaluate.jl:350-351)
            # NaNs compare equal whatever their sign/payload (x86 0/0 = -NaN, gfx950 = +NaN);
            # everything else, signed zeros and infinities included, must match bit for bit
            nan = np.isnan(y)
            assert np.array_equal(np.isnan(out[t]), nan), de.string_tree(tree, ops)
            ui = np.uint32 if dtype == np.float32 else np.uint64
            np.testing.assert_array_equal(out[t][~nan].view(ui), y[~nan].view(ui), err_msg=de.string_tree(tree, ops))


def test_empty_and_degenerate_inputs(api):
    ops = de.synth.BENCH_OPERATORS
    trees = [de.Node(feature=2), de.Node(val=1.5), de.Node(val=float("inf")),
             de.Node(2, de.Node(val=float("nan")))]
    pop = api.Population(trees, ops, np.float32, n_features=5)
    out, ok = pop.eval(np.zeros((5, 0), dtype=np.float32, order="F"))  # N = 0
    assert out.shape == (4, 0)
    assert list(ok) == [True, True, False, False]
    X = de.synth.random_X(5, 10, seed=3)
    out, ok = pop.eval(X)
    np.testing.assert_array_equal(out[0], X[1])
    np.testing.assert_array_equal(out[1], np.full(10, 1.5, np.float32))
    assert list(ok) == [True, True, False, False]
    with pytest.raises(ValueError):  # fewer features than the trees use (Expression validation)
        pop.eval(np.zeros((1, 4), dtype=np.float32, order="F"))
    empty = api.Population([], ops, np.float32, n_features=5)
    out, ok = empty.eval(X)
    assert out.shape == (0, 10) and ok.shape == (0,)


def test_set_constants_without_reflattening(api):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(20, seed=21)
    X = de.synth.random_X(5, 500, seed=4)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    allc = []
    for t in trees:
        cs, refs = de.get_scalar_constants(t)
        cs = (cs * 0.5 + 0.25).astype(np.float32)
        de.set_scalar_constants(t, cs, refs)
        allc.append(cs)
    pop.set_constants(np.concatenate(allc))
    out, ok = pop.eval(X)
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, np.float32)
        y, ok_el = oracle.eval_tree_array(tape, consts, X, elementwise=True)
        assert bool(ok[t]) == ok_el
        if ok_el:
            assert np.all(np.abs(out[t].astype(np.float64) - y) <= parity_tolerance(tree, ops, X, np.float32))


def test_full_size_properties_config_C2(api):
    """BASELINE config 2 at full size (1000 trees x 10^6 samples, f32): size-independent
    properties instead of an oracle run — (i) every tree's first 2048 and last 1000 samples equal
    a separate small launch on those columns (tiling/ragged-tail independence), (ii) the flag of
    the full run is the AND of the flags of a 4-way sample split, (iii) a checksum of the
    output is reproducible across two launches."""
    import torch
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(1000, seed=0xDE02)
    N = 10**6
    g = torch.Generator(device="cuda").manual_seed(1)
    Xd = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()
    pop = api.Population(trees, ops, np.float32, n_features=5)
    out, ok = pop.eval(Xd)
    head, ok_h = pop.eval(Xd[:, :2048])
    tail, ok_t = pop.eval(Xd[:, N - 1000:].t().contiguous().t())
    torch.cuda.synchronize()
    okc = ok.cpu().numpy()
    complete = torch.from_numpy(okc).cuda()
    assert torch.equal(out[complete][:, :2048], head[complete])
    assert torch.equal(out[complete][:, N - 1000:], tail[complete])
    parts = []
    for i in range(4):
        sl = Xd[:, i * (N // 4):(i + 1) * (N // 4)].t().contiguous().t()
        parts.append(pop.eval(sl)[1])
    assert torch.equal(ok, parts[0] & parts[1] & parts[2] & parts[3])
    out2, ok2 = pop.eval(Xd)
    assert torch.equal(ok, ok2)
    assert torch.equal(torch.nan_to_num(out[complete]).double().sum(1), torch.nan_to_num(out2[complete]).double().sum(1))
    assert 100 < int(okc.sum()) < 1000


def test_wide_feature_matrix_uses_direct_path(api):
    """F = 200 features does not fit the LDS tile: the direct variant gathers features from
    global memory; results must still match the oracle (and be bit-exact for exact operators)."""
    F = 200
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*"), unary_operators=("cos", "exp", "neg"))
    rng = de.synth.Xoshiro256ss(41)
    trees = [de.synth.gen_random_tree_fixed_size(3 + i % 25, ops, F, rng, np.float32) for i in range(60)]
    X = de.synth.random_X(F, 1500, seed=13)
    compare_population(api, trees, ops, X, np.float32, min_ok=5)
    ops2 = de.OperatorEnum(binary_operators=("+", "-", "*"), unary_operators=("neg", "abs"))
    trees2 = [de.synth.gen_random_tree_fixed_size(3 + i % 25, ops2, F, rng, np.float64) for i in range(40)]
    X64 = de.synth.random_X(F, 700, seed=14, dtype=np.float64)
    pop = api.Population(trees2, ops2, np.float64, n_features=F)
    out, ok = pop.eval(X64)
    assert pop.ctx.last_kernel_name().endswith("<direct>")
    for t, tree in enumerate(trees2):
        tape, consts = de.flatten(tree, ops2, np.float64)
        y, okr = oracle.eval_tree_array(tape, consts, X64, elementwise=True)
        assert bool(ok[t]) == okr
        if okr:
            np.testing.assert_array_equal(out[t], y)


def test_abi_error_paths_on_device(api):
    """Misuse returns status codes + messages, never crashes."""
    import ctypes as C
    lib = api.library()
    ctx = api.default_context()
    T = de.node.TAPE_DTYPE
    bad = np.array([(2, 64, 0)], dtype=T)  # operator without operands
    off = np.array([0, 1], dtype=np.int64)
    coff = np.zeros(2, dtype=np.int64)
    h = C.c_void_p()
    rc = lib.de_program_create(ctx._h, 0, bad.ctypes.data, off.ctypes.data, 1, None, coff.ctypes.data, 2, 0, 7, C.byref(h))
    assert rc == 2 and b"tree 0" in lib.de_last_error(ctx._h)
    unk = np.array([(0, 1, 0), (1, 250, 0)], dtype=T)  # opcode outside the table
    off2 = np.array([0, 2], dtype=np.int64)
    rc = lib.de_program_create(ctx._h, 0, unk.ctypes.data, off2.ctypes.data, 1, None, coff.ctypes.data, 2, 0, 7, C.byref(h))
    assert rc == 3
    rc = lib.de_program_create(ctx._h, 5, unk.ctypes.data, off2.ctypes.data, 1, None, coff.ctypes.data, 2, 0, 7, C.byref(h))
    assert rc == 1  # bad dtype
    pop = api.Population([de.Node(feature=2)], de.synth.BENCH_OPERATORS, np.float32, n_features=2)
    ok = np.zeros(1, np.uint8)
    out = np.zeros(4, np.float32)
    X = np.zeros((2, 4), np.float32, order="F")
    assert lib.de_eval(ctx._h, pop._h, X.ctypes.data, 4, 1, None, out.ctypes.data, 4, ok.ctypes.data) == 1  # ldX < F
    assert lib.de_eval(ctx._h, pop._h, X.ctypes.data, 4, 2, None, out.ctypes.data, 3, ok.ctypes.data) == 1  # ld_out < N
    assert lib.de_eval(ctx._h, pop._h, None, 4, 2, None, out.ctypes.data, 4, ok.ctypes.data) == 1  # null X
    assert lib.de_eval(ctx._h, pop._h, X.ctypes.data, 4, 2, None, out.ctypes.data, 4, ok.ctypes.data) == 0


def test_strided_buffers_through_the_c_abi(api):
    """ldX > n_features (a view into a wider matrix) and ld_out > N, host pointers, raw C ABI."""
    import ctypes as C
    lib = api.library()
    ctx = api.default_context()
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(8, seed=5)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    N, ldX, ld_out = 1300, 9, 1400
    g = np.random.Generator(np.random.PCG64(1))
    big = g.standard_normal((N, ldX)).astype(np.float32)  # row j = sample j, first 5 entries are the features
    out = np.full((8, ld_out), -7.0, np.float32)
    ok = np.zeros(8, np.uint8)
    assert lib.de_eval(ctx._h, pop._h, big.ctypes.data, N, ldX, None, out.ctypes.data, ld_out, ok.ctypes.data) == 0
    X = np.asfortranarray(big[:, :5].T)
    ref, okr = pop.eval(X)
    np.testing.assert_array_equal(out[okr, :N], ref[okr])  # (rows of incomplete trees are unspecified: early exit)
    assert np.all(out[:, N:] == -7.0)  # padding untouched
    np.testing.assert_array_equal(ok.astype(bool), okr)


def test_two_contexts_from_two_threads(api):
    """One de_ctx_t per calling thread: concurrent evaluation gives the same
    results as sequential evaluation."""
    import threading
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(50, seed=9)
    X = de.synth.random_X(5, 20000, seed=4)
    ref, okr = api.Population(trees, ops, np.float32, n_features=5).eval(X)
    results = {}

    def work(i):
        ctx = api.Context(0, stream=None)
        pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
        for _ in range(3):
            results[i] = pop.eval(X)
        pop.close()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        np.testing.assert_array_equal(results[i][0][okr], ref[okr])
        np.testing.assert_array_equal(results[i][1], okr)


def test_constant_inner_branch_of_a_fused_kernel_is_not_constant_folded_for_the_flag(api):
    """x + (c1 ^ c2) with a negative base: the reference evaluates the constant pair INSIDE the fused
    3-node kernel (deg2_branch0_eval, src/Evaluate.jl:795-871), not with dispatch_constant_tree, so its
    NaN clears `complete` only under early_exit.  (The device folds the pair into one constant; the flag
    must still follow the reference.)  With use_fused=false the pair IS a constant subtree of its own
    (_eval_tree_array -> dispatch_constant_tree) and the test is unconditional."""
    ops = de.OperatorEnum(binary_operators=("+", "*", "^"), unary_operators=("cos",))
    pw = de.Node(3, de.Node(val=-1.32), de.Node(val=-0.356))
    trees = [de.Node(1, de.Node(feature=2), pw.copy()), de.Node(2, pw.copy(), de.Node(feature=1)),
             de.Node(1, de.Node(1, de.Node(feature=1)), pw.copy())]  # the last one: not a fused shape
    X = de.synth.random_X(2, 300, seed=4)
    for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False),
               api.EvalContext(early_exit=False, use_fused=False)):
        opts = ec.option_bits(ops)
        pop = api.Population(trees, ops, np.float32, n_features=2, eval_context=ec)
        out, ok = pop.eval(X)
        for t, tree in enumerate(trees):
            tape, consts = de.flatten(tree, ops, np.float32)
            _, ok_ref = oracle.eval_tree_array(tape, consts, X, opts)
            assert bool(ok[t]) == ok_ref, (t, opts)
            if ok_ref:
                assert np.all(np.isnan(out[t]))
        pop.close()
    # the expectation itself, spelled out for the default fused dispatch
    pop = api.Population(trees, ops, np.float32, n_features=2, eval_context=api.EvalContext(early_exit=False))
    _, ok = pop.eval(X)
    assert list(ok) == [True, True, False]
    pop.close()


def test_set_constants_patches_every_derived_program_in_place(api, monkeypatch):
    """de_program_set_consts rewrites immediates inside the bound / fused / gradient instruction streams
    (no re-lowering).  After several updates every entry point must be bit-identical to a population
    created from scratch with the same constants — including folded constant subtrees, fused
    two-operand forms, and the gradient programs built before the update."""
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max"), unary_operators=("cos", "exp", "square", "safe_log"))
    rng = de.synth.Xoshiro256ss(31)
    trees = [de.synth.gen_random_tree_fixed_size(2 + i % 28, ops, 3, rng, np.float32) for i in range(300)]
    X = de.synth.random_X(3, 700, seed=9)
    y = np.cos(np.arange(700)).astype(np.float32)
    pop = api.Population(trees, ops, np.float32, n_features=3)
    pop.eval(X); pop.eval_grad(X, False); pop.eval_grad(X, "both"); pop.eval_loss_grad(X, y)  # build every derived program first
    monkeypatch.setenv("DE_LOSS_GRAD_REVERSE", "1")  # ... the reverse-accumulation program too
    pop.eval_loss_grad(X, y)
    monkeypatch.delenv("DE_LOSS_GRAD_REVERSE")
    g = np.random.Generator(np.random.PCG64(1))
    n_const = int(pop.n_consts.sum())

    def bits(a):
        return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)

    def same(a, b):  # NaN payloads/signs are not contractual
        a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
        assert a.shape == b.shape
        m = ~(np.isnan(a) & np.isnan(b))
        np.testing.assert_array_equal(bits(a)[m], bits(b)[m])

    def assign(n, it):  # constants in depth-first leaf order = the order of set_constants
        if n.degree == 0:
            if n.constant:
                n.val = float(next(it))
            return
        for ch in n.children[:n.degree]:
            assign(ch, it)

    for step in range(3):
        new = g.standard_normal(n_const).astype(np.float32)
        if step == 2:
            new[::17] = np.float32(np.inf)  # flags through the constant path too
        pop.set_constants(new)
        it = iter(new.tolist())
        fresh = [t.copy() for t in trees]
        for c in fresh:
            assign(c, it)
        ref = api.Population(fresh, ops, np.float32, n_features=3)
        oa, ka = pop.eval(X); ob, kb = ref.eval(X)
        assert np.array_equal(ka, kb) and ka.any()
        same(oa[ka], ob[kb])
        for variable in (False, "both"):
            oa, ga, ka = pop.eval_grad(X, variable); ob, gb, kb = ref.eval_grad(X, variable)
            assert np.array_equal(ka, kb)
            same(oa[ka], ob[kb])
            for t in np.nonzero(ka)[0]:
                same(ga[t], gb[t])
        la, da, ka = pop.eval_loss_grad(X, y); lb, db, kb = ref.eval_loss_grad(X, y)
        assert np.array_equal(ka, kb)
        same(la[ka], lb[kb])
        for t in np.nonzero(ka)[0]:
            same(da[t], db[t])
        la, ka = pop.eval_loss(X, y); lb, kb = ref.eval_loss(X, y)
        assert np.array_equal(ka, kb)
        same(la[ka], lb[kb])
        monkeypatch.setenv("DE_LOSS_GRAD_REVERSE", "1")  # patched reverse program == reverse program built from scratch
        la, da, ka = pop.eval_loss_grad(X, y); lb, db, kb = ref.eval_loss_grad(X, y)
        monkeypatch.delenv("DE_LOSS_GRAD_REVERSE")
        assert np.array_equal(ka, kb)
        same(la[ka], lb[kb])
        for t in np.nonzero(ka)[0]:
            same(da[t], db[t])
        ref.close()
    pop.close()


def test_eval_handlers_of_more_unary_operators_and_max_min(api, monkeypatch):
    """`neg square cube abs log safe_log sqrt safe_sqrt tanh relu` and max/min have eval handlers of their own (DESIGN
    §4.1); they must give the bits of the generic handler (same expressions), and the oracle's bits for the exact ones."""
    from helpers import sexpr_to_node
    una = ("neg", "square", "cube", "abs", "log", "safe_log", "sqrt", "safe_sqrt", "tanh", "relu")
    ops = de.OperatorEnum(binary_operators=("+", "*", "max", "min"), unary_operators=una)
    for dtype in (np.float32, np.float64):
        X = np.asfortranarray(de.synth.random_X(3, 3000, seed=4, dtype=dtype))
        trees = []
        for u in una:
            trees += [sexpr_to_node(f, ops) for f in (
                [u, ["x", 1]], [u, ["+", ["x", 1], ["*", ["x", 2], 0.5]]],
                ["*", [u, ["+", ["x", 1], 1.5]], [u, ["*", ["x", 3], ["x", 2]]]],
                ["max", [u, ["x", 2]], ["min", [u, ["x", 1]], 0.75]], ["min", ["max", ["x", 3], -0.25], [u, ["x", 3]]])]
        for ec in (api.EvalContext(), api.EvalContext(early_exit=False)):
            pop = api.Population(trees, ops, dtype, n_features=3, eval_context=ec)
            a, ka = pop.eval(X)
            pop.close()
            monkeypatch.setenv("DE_NO_CONST_UNARY_HOT", "1")
            pop = api.Population(trees, ops, dtype, n_features=3, eval_context=ec)
            b, kb = pop.eval(X)
            pop.close()
            monkeypatch.delenv("DE_NO_CONST_UNARY_HOT")
            assert np.array_equal(ka, kb)
            ui = np.uint32 if dtype == np.float32 else np.uint64
            if ec.early_exit:  # rows of incomplete trees (log / sqrt of negative samples) are unspecified: early exit
                a, b = a[ka], b[kb]
            m = ~(np.isnan(a) & np.isnan(b))
            np.testing.assert_array_equal(a.view(ui)[m], b.view(ui)[m])
        for t, tree in enumerate(trees):  # IEEE-exact operators: the oracle's bits
            if (t // 5) in (0, 1, 2, 3, 9):  # neg square cube abs relu
                tape, consts = de.flatten(tree, ops, dtype)
                y, ok = oracle.eval_tree_array(tape, consts, X)
                o, k = api.eval_tree_array(tree, X, ops)
                assert k == ok
                if ok:
                    np.testing.assert_array_equal(o.view(ui), y.view(ui))


def test_shared_subtrees_evaluate_like_their_expansion(api):
    """GraphNode-style sharing (src/Node.jl:138-166): the reference evaluates a shared node once per parent, so a DAG
    and its deep copy give the same bits; the constant of the shared node owns one gradient row per occurrence (the
    reference's shared `NodeIndex` row is their sum)."""
    ops = de.OperatorEnum(binary_operators=("+", "*"), unary_operators=("cos",))
    s = de.Node(1, de.Node(2, de.Node(feature=1), de.Node(val=0.75)))
    dag = de.Node(1, s, de.Node(2, s, s))
    tree = dag.copy()
    X = np.asfortranarray(np.linspace(-2, 2, 257, dtype=np.float64)[None, :])
    ya, oka = api.eval_tree_array(dag, X, ops)
    yb, okb = api.eval_tree_array(tree, X, ops)
    assert oka and okb
    np.testing.assert_array_equal(ya, yb)
    c, sn = np.cos(X[0] * 0.75), np.sin(X[0] * 0.75)
    np.testing.assert_allclose(ya, c + c * c, rtol=1e-14)
    _, ga, oka = api.eval_grad_tree_array(dag, X, ops, variable=False)
    _, gb, okb = api.eval_grad_tree_array(tree, X, ops, variable=False)
    assert oka and okb and ga.shape == (3, X.shape[1])
    np.testing.assert_array_equal(ga, gb)
    np.testing.assert_allclose(ga.sum(axis=0), -sn * X[0] * (1 + 2 * c), rtol=1e-12, atol=1e-14)
