"""GPU parity tests of eval_grad_tree_array / eval_diff_tree_array (and the parametric path)
through the C ABI vs the CPU oracle.  Run with `pytest -m gpu` on an MI355X."""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from helpers import case_X, case_tree, grad_tolerance, load_golden, parity_tolerance
from oracle import oracle

pytestmark = pytest.mark.gpu
GCASES = [c for c in load_golden() if c["kind"] == "grad"]
MODES = {"variable": (True, oracle.GRAD_VARIABLE), "constant": (False, oracle.GRAD_CONSTANT),
         "both": ("both", oracle.GRAD_BOTH)}


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


@pytest.mark.parametrize("case", GCASES, ids=[c["name"] for c in GCASES])
def test_golden_gradients_on_gpu(case, api):
    tree, ops = case_tree(case)
    X = case_X(case)
    exp = case["expect"]
    variable, omode = MODES[exp["mode"]]
    y, g, ok = api.eval_grad_tree_array(tree, X, ops, variable=variable)
    assert ok == exp["ok"], f"{case['name']} ({case['cite']})"
    if not ok:
        return
    tol = lambda want: max(exp.get("atol", 0), 1e-30) + max(exp.get("rtol", 0), 2e-13) * np.abs(want)  # noqa: E731
    if "y" in exp:
        want = np.asarray(exp["y"])
        assert np.all(np.abs(y - want) <= tol(want))
    if "grad" in exp:
        want = np.asarray(exp["grad"])
        assert g.shape == want.shape
        assert np.all(np.abs(g - want) <= tol(want)), case["name"]
    for row, vals in exp.get("grad_rows", {}).items():
        want = np.asarray(vals)
        assert np.all(np.abs(g[int(row)] - want) <= tol(want))
    if exp["mode"] == "variable":  # eval_diff_tree_array per feature == gradient row (test_derivatives.jl:84-92)
        for f in range(X.shape[0]):
            y2, d, okd = api.eval_diff_tree_array(tree, X, ops, f + 1)
            assert okd
            np.testing.assert_array_equal(d, g[f])
            np.testing.assert_array_equal(y2, y)


@pytest.fixture(params=["1-per-lane", "2-per-lane"], autouse=True)
def samples_per_lane(request, monkeypatch):
    """The threaded gradient modules exist with one and with two samples per lane (Float32, windows <= 6, trees with
    <= 1 spill slot); the library uses the latter from 65536 samples on — the tests force each."""
    monkeypatch.setenv("DE_GRAD_VS2_MIN_N", "0" if request.param == "2-per-lane" else "1000000000000")
    return request.param


ILL_FRACTION = {}  # (mode, dtype, N) -> share of Jacobian entries the tolerance model classed as ill-conditioned


def grad_compare(api, trees, ops, X, dtype, mode_name, exact=False):
    variable, omode = MODES[mode_name]
    pop = api.Population(trees, ops, dtype, n_features=X.shape[0])
    out, grads, ok = pop.eval_grad(X, variable)
    n_ok = 0
    n_ent = n_ill = 0
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, dtype)
        y, g, ok_el = oracle.eval_grad_tree_array(tape, consts, X, omode, elementwise=True)
        assert bool(ok[t]) == ok_el, f"flag mismatch tree {t} [{mode_name}]: {de.string_tree(tree, ops)}"
        assert grads[t].shape == g.shape
        if not ok_el:
            continue
        n_ok += 1
        if exact:
            ui = np.uint32 if dtype == np.float32 else np.uint64
            np.testing.assert_array_equal(out[t].view(ui), y.view(ui), err_msg=de.string_tree(tree, ops))
            np.testing.assert_array_equal(np.ascontiguousarray(grads[t]).view(ui), np.ascontiguousarray(g).view(ui),
                                          err_msg=de.string_tree(tree, ops))
            continue
        # A BOUND on every entry, not a pass fraction: the conditioned model of helpers.grad_tolerance (the gradient twin
        # of parity_tolerance); entries it classes as ill-conditioned are counted and capped below.
        tol = grad_tolerance(tree, ops, X, dtype, mode_name)
        assert tol is not None, f"no partial table for an operator of {de.string_tree(tree, ops)}"
        err = np.abs(np.asarray(grads[t], dtype=np.float64) - g.astype(np.float64))
        bad = err > tol
        assert not bad.any(), (f"gradient entry beyond its bound [{mode_name}] tree {t}: {de.string_tree(tree, ops)} "
                               f"worst err/tol {np.nanmax(np.where(np.isfinite(tol), err / tol, 0)):.3g}")
        n_ent += tol.size
        n_ill += int(np.isinf(tol).sum())
        tolx = parity_tolerance(tree, ops, X, dtype)
        m = np.isfinite(tolx)
        assert np.all(np.abs(out[t].astype(np.float64) - y)[m] <= tolx[m]), de.string_tree(tree, ops)
    pop.close()
    if not exact and n_ent:
        ILL_FRACTION[(mode_name, np.dtype(dtype).name, X.shape[1])] = n_ill / n_ent
        print(f"[grad parity {mode_name} {np.dtype(dtype).name} N={X.shape[1]}] {n_ent} entries bounded, "
              f"{n_ill} ({100.0 * n_ill / n_ent:.2f} %) ill-conditioned (not compared)")
        assert n_ill <= 0.05 * n_ent, f"{n_ill}/{n_ent} gradient entries ill-conditioned [{mode_name}]"
    return n_ok


@pytest.mark.parametrize("mode", ["variable", "constant", "both"])
def test_exact_operator_gradients_bit_identical(api, mode):
    """+ - * / square neg abs: every partial is IEEE-exact, the accumulation order is the
    reference's (g1*d1 + g2*d2): the whole Jacobian must match the oracle bit for bit."""
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*"), unary_operators=("neg", "square", "abs"))
    rng = de.synth.Xoshiro256ss(17)
    for dtype in (np.float32, np.float64):
        trees = [de.synth.gen_random_tree_fixed_size(3 + i % 26, ops, 4, rng, dtype) for i in range(80)]
        X = de.synth.random_X(4, 777, seed=12, dtype=dtype)
        assert grad_compare(api, trees, ops, X, dtype, mode, exact=True) > 10


@pytest.mark.parametrize("mode", ["variable", "constant", "both"])
@pytest.mark.parametrize("N", [1, 300, 2049])
def test_random_population_gradients_vs_oracle(api, mode, N):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(120, seed=0xDE03)
    X = de.synth.random_X(5, N, seed=6)
    assert grad_compare(api, trees, ops, X, np.float32, mode) > 5


def test_wide_operator_gradients_f64(api):
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^"),
                          unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin",
                                           "safe_sqrt", "relu", "atan", "custom_cos"))
    rng = de.synth.Xoshiro256ss(23)
    trees = [de.synth.gen_random_tree_fixed_size(4 + i % 18, ops, 3, rng, np.float64) for i in range(100)]
    X = de.synth.random_X(3, 500, seed=8, dtype=np.float64)
    for mode in ("variable", "both"):
        grad_compare(api, trees, ops, X, np.float64, mode)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_every_hot_unary_and_max_min_on_every_operand_kind(api, dtype):
    """The gradient kernels serve `neg square cube abs log safe_log sqrt safe_sqrt tanh relu` (and max/min) with hot
    handlers of their own instead of the generic one: each operator applied to a feature row, a
    constant leaf, the accumulator, a spilled subtree and (parametric) a parameter row must give the oracle's
    Jacobian — bit for bit for the IEEE-exact ones — in all three modes, and the same bits as the generic handler."""
    from helpers import sexpr_to_node
    una = ("neg", "square", "cube", "abs", "log", "safe_log", "sqrt", "safe_sqrt", "tanh", "relu", "cos", "exp", "sin")
    ops = de.OperatorEnum(binary_operators=("+", "*", "max", "min"), unary_operators=una)
    exact_ops = {"neg", "square", "cube", "abs", "relu"}
    X = np.asfortranarray(np.abs(de.synth.random_X(3, 300, seed=4, dtype=dtype)) + dtype(0.25))  # positive: log / sqrt defined
    import os
    for u in una:
        forms = [
            [u, ["x", 1]],                                             # feature row
            [u, 1.75],                                                 # constant leaf
            [u, ["+", ["x", 1], ["*", ["x", 2], 0.5]]],                # accumulator
            ["*", [u, ["+", ["x", 1], 1.5]], [u, ["*", ["x", 3], ["x", 2]]]],  # a spilled subtree on one side
            ["max", [u, ["x", 2]], ["min", [u, 0.75], ["x", 3]]],      # max / min with rows, constants, slots
        ]
        trees = [sexpr_to_node(f, ops) for f in forms]
        for mode in ("variable", "constant", "both"):
            assert grad_compare(api, trees, ops, X, dtype, mode, exact=u in exact_ops) == len(trees), (u, mode)
        # same bits as the generic handler
        variable, _ = MODES["both"]
        pop = api.Population(trees, ops, dtype, n_features=3)
        a = pop.eval_grad(X, variable)
        pop.close()
        os.environ["DE_NO_CONST_UNARY_HOT"] = "1"
        try:
            pop = api.Population(trees, ops, dtype, n_features=3)
            b = pop.eval_grad(X, variable)
            pop.close()
        finally:
            del os.environ["DE_NO_CONST_UNARY_HOT"]
        if u in exact_ops or u in ("log", "safe_log", "sqrt", "safe_sqrt", "tanh"):  # same library functions on both paths
            for t in range(len(trees)):
                np.testing.assert_array_equal(np.asarray(a[1][t]), np.asarray(b[1][t]), err_msg=f"{u} tree {t}")
                np.testing.assert_array_equal(a[0][t], b[0][t])
    # parameter rows
    pops = de.OperatorEnum(binary_operators=("+", "*", "max", "min"), unary_operators=("square", "tanh", "safe_log", "cos"))
    ptrees = [sexpr_to_node(f, pops, de.ParametricNode) for f in (
        ["square", ["p", 1]], ["tanh", ["p", 2]], ["max", ["x", 1], ["p", 1]], ["*", ["safe_log", ["p", 2]], ["min", ["p", 1], ["x", 2]]],
        ["+", ["cos", ["p", 1]], ["square", ["*", ["p", 2], ["x", 3]]]])]
    g = np.random.Generator(np.random.PCG64(2))
    params = np.asfortranarray(np.abs(g.standard_normal((2, 3))).astype(dtype) + dtype(0.5))
    classes = g.integers(1, 4, X.shape[1])
    pop = api.Population(ptrees, pops, dtype, n_features=3, n_params=2)
    for variable, om in ((True, oracle.GRAD_VARIABLE), ("both", oracle.GRAD_BOTH)):
        out, grads, ok = pop.eval_grad(X, variable, params=params, classes=classes)
        for t, tree in enumerate(ptrees):
            tape, consts = de.flatten(tree, pops, dtype)
            t2, PX = oracle.parametric_to_plain(tape, X, params, classes)
            y, gg, ok_o = oracle.eval_grad_tree_array(t2, consts, PX, om, elementwise=True)
            assert bool(ok[t]) == ok_o
            np.testing.assert_allclose(out[t], y, rtol=2e-6 if dtype == np.float32 else 1e-13)
            np.testing.assert_allclose(np.asarray(grads[t]), gg, rtol=1e-5 if dtype == np.float32 else 1e-12, atol=1e-6 if dtype == np.float32 else 1e-14)
    pop.close()


def test_many_constants_use_several_windows(api):
    """A tree with 19 constants: constant-mode gradient is computed in three 8-wide windows."""
    ops = de.OperatorEnum(binary_operators=("+", "*"), unary_operators=("cos",))
    t = de.Node(feature=1)
    for i in range(19):
        t = de.Node(1 + i % 2, t, de.Node(val=0.5 + 0.1 * i))
    t2 = de.Node(1, de.Node(1, t.copy()), de.Node(feature=2))
    X = de.synth.random_X(2, 1000, seed=2, dtype=np.float64)
    grad_compare(api, [t, t2, de.Node(feature=2)], ops, X, np.float64, "constant", exact=False)
    grad_compare(api, [t, t2, de.Node(val=2.0)], ops, X, np.float64, "both", exact=False)
    # + and * only: exact
    ops2 = de.OperatorEnum(binary_operators=("+", "*"))
    t3 = de.Node(feature=1)
    for i in range(19):
        t3 = de.Node(1 + i % 2, t3, de.Node(val=0.5 + 0.1 * i))
    grad_compare(api, [t3], ops2, X, np.float64, "constant", exact=True)


def test_parametric_eval_and_constant_gradient_config_C5_shape(api):
    """BASELINE config 5 in miniature: ParametricNode trees, P=8 per-class parameters, eval and
    the constant-mode gradient, against the reference's own reduction (gather params above X,
    re-index leaves — src/ParametricExpression.jl:381-389) run through the oracle."""
    ops = de.synth.BENCH_OPERATORS
    P, F, C, N = 8, 5, 16, 1500
    trees = de.synth.random_population(60, seed=0xDE05, nfeatures=F, node_type=de.ParametricNode, nparams=P)
    g = np.random.Generator(np.random.PCG64(5))
    params = np.asfortranarray(g.standard_normal((P, C)).astype(np.float32))
    classes = g.integers(1, C + 1, N).astype(np.int64)
    X = de.synth.random_X(F, N, seed=7)
    pop = api.Population(trees, ops, np.float32, n_features=F, n_params=P)
    out, ok = pop.eval(X, params, classes)
    outg, grads, okg = pop.eval_grad(X, False, params, classes)
    outb, gradsb, okb = pop.eval_grad(X, "both", params, classes)
    n_ok = n_ent = n_ill = 0
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, np.float32)
        y, ok_el = oracle.eval_tree_array_parametric(tape, consts, X, params, classes.astype(np.int32), 1, elementwise=True)
        assert bool(ok[t]) == ok_el, de.string_tree(tree, ops)
        t2, PX = oracle.parametric_to_plain(tape, X, params, classes)
        yg, gg, okg_el = oracle.eval_grad_tree_array(t2, consts, PX, oracle.GRAD_CONSTANT, elementwise=True)
        yb, gb, okb_el = oracle.eval_grad_tree_array(t2, consts, PX, oracle.GRAD_BOTH, elementwise=True)
        assert bool(okg[t]) == okg_el and bool(okb[t]) == okb_el
        assert gradsb[t].shape == (P + F + len(consts), N)
        if ok_el:
            n_ok += 1
            tol = parity_tolerance(tree, ops, X, np.float32, 7, params, classes - 1)
            assert np.all(np.abs(out[t].astype(np.float64) - y) <= tol), de.string_tree(tree, ops)
            assert np.mean(np.isinf(tol)) < 0.5
        for got, want, okm, mname in ((grads[t], gg, okg_el, "constant"), (gradsb[t], gb, okb_el, "both")):
            if not (okm and want.size):
                continue
            tolg = grad_tolerance(tree, ops, X, np.float32, mname, params, classes, 1)
            err = np.abs(np.asarray(got, dtype=np.float64) - want.astype(np.float64))
            assert not (err > tolg).any(), f"{mname} gradient beyond its bound: {de.string_tree(tree, ops)}"
            n_ent += tolg.size
            n_ill += int(np.isinf(tolg).sum())
    assert n_ok > 5
    print(f"[C5 miniature] {n_ent} gradient entries bounded, {100.0 * n_ill / max(n_ent, 1):.2f} % ill-conditioned")
    assert n_ill <= 0.05 * n_ent
    with pytest.raises(ValueError):  # "You must pass the `classes::Vector` argument"
        pop.eval(X)


def test_diff_has_no_validity_test(api):
    ops = de.OperatorEnum(binary_operators=("+", "/"), unary_operators=("cos",))
    tree = de.Node(2, de.Node(feature=1), de.Node(val=0.0))  # x1 / 0
    X = de.synth.random_X(1, 50, seed=1, dtype=np.float64)
    y, d, ok = api.eval_diff_tree_array(tree, X, ops, 1)
    tape, consts = de.flatten(tree, ops, np.float64)
    yo, do, oko = oracle.eval_diff_tree_array(tape, consts, X, 0)
    assert ok and oko and np.all(np.isinf(y))
    np.testing.assert_array_equal(y, yo)
    np.testing.assert_array_equal(np.isnan(d), np.isnan(do))  # (1/0)*1 + (-(Inf/0))*0 = NaN
    _, _, okg = api.eval_grad_tree_array(tree, X, ops, variable=True)
    assert not okg


def test_random_population_diff_vs_oracle(api):
    """eval_diff_tree_array on a random population: every direction equals the matching row of
    the variable-mode Jacobian bit for bit (same arithmetic), and matches the oracle."""
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(40, seed=77)
    X = de.synth.random_X(5, 900, seed=3)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    _, grads, ok_grad = pop.eval_grad(X, True)
    assert ok_grad.sum() >= 15
    n_ent = n_ill = 0
    for direction in (1, 3, 5):
        out, dout, ok = pop.eval_diff(X, direction)
        assert ok.all()
        for t, tree in enumerate(trees):
            if ok_grad[t]:  # (ABI v2: the host rows of a tree the Jacobian call found incomplete — a non-finite entry in ANY row — are NaN)
                np.testing.assert_array_equal(dout[t], grads[t][direction - 1])
            tape, consts = de.flatten(tree, ops, np.float32)
            yo, do, _ = oracle.eval_diff_tree_array(tape, consts, X, direction - 1)
            tolg = grad_tolerance(tree, ops, X, np.float32, "variable")[direction - 1]
            m = np.isfinite(do) & np.isfinite(yo) & np.isfinite(dout[t])
            err = np.abs(dout[t].astype(np.float64) - do.astype(np.float64))
            assert not (err[m] > tolg[m]).any(), de.string_tree(tree, ops)
            n_ent += int(m.sum())
            n_ill += int(np.isinf(tolg[m]).sum())
    assert n_ill <= 0.05 * n_ent, f"{n_ill}/{n_ent} derivative entries ill-conditioned"


def test_full_size_gradient_properties_config_C3(api):
    """BASELINE config 3 at full size (1000 trees x 10^6, variable=true): the head of every
    Jacobian equals a separate small launch; flags are the AND over a 4-way sample split."""
    import torch
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(1000, seed=0xDE02)
    N = 10**6
    g = torch.Generator(device="cuda").manual_seed(1)
    Xd = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()
    pop = api.Population(trees, ops, np.float32, n_features=5)
    out, grads, ok = pop.eval_grad(Xd, True)
    out_h, grads_h, ok_h = pop.eval_grad(Xd[:, :1536], True)
    torch.cuda.synchronize()
    for t in range(0, 1000, 37):
        if bool(ok[t]):
            assert torch.equal(grads[t][:, :1536], grads_h[t])
            assert torch.equal(out[t, :1536], out_h[t])
    parts = []
    for i in range(4):
        sl = Xd[:, i * (N // 4):(i + 1) * (N // 4)].t().contiguous().t()
        parts.append(pop.eval_grad(sl, True)[2])
    assert torch.equal(ok, parts[0] & parts[1] & parts[2] & parts[3])
    assert 50 < int(ok.sum()) < 1000


def test_gradient_flag_ignores_columns_the_tree_does_not_have(api):
    """A tree without constants has a 0-row gradient in constant mode: an infinite partial (d safe_sqrt / dx at 0)
    must not clear `complete` through the window's unused column (found by tests/fuzz/fuzz_grad.py)."""
    ops = de.OperatorEnum(binary_operators=("max", "+"), unary_operators=("safe_sqrt", "relu", "square"))
    t0 = de.Node(1, de.Node(1, de.Node(2, de.Node(feature=2))), de.Node(3, de.Node(feature=1)))  # max(safe_sqrt(relu(x2)), square(x1))
    t1 = de.Node(2, t0.copy(), de.Node(val=0.5))  # same + one constant: the partial now meets a real column
    for dtype in (np.float32, np.float64):
        X = de.synth.random_X(2, 300, seed=3, dtype=dtype)  # x2 < 0 for about half of the samples
        pop = api.Population([t0, t1], ops, dtype, n_features=2)
        for mode_name in ("constant", "variable", "both"):
            variable, omode = MODES[mode_name]
            out, grads, ok = pop.eval_grad(X, variable)
            for t, tree in enumerate((t0, t1)):
                tape, consts = de.flatten(tree, ops, dtype)
                _, g, ok_ref = oracle.eval_grad_tree_array(tape, consts, X, omode, elementwise=True)
                assert bool(ok[t]) == ok_ref, (dtype.__name__, mode_name, t)
                assert grads[t].shape == g.shape
        assert pop.n_grad(0, 1) == 0
        out, grads, ok = pop.eval_grad(X, False)
        assert bool(ok[0])
        pop.close()
