"""GPU parity tests of the fused loss reduction (de_eval_loss): the residual sum a
consumer of eval_tree_array computes right after the call — `sum(abs2, tree(X, operators) .- y)`,
test/test_optim.jl:95,99 — evaluated without writing the [n_trees, N] output.
Checked against (a) the CPU oracle's outputs reduced in float64 and (b) the device's own de_eval
outputs reduced in float64 (same values, different summation order)."""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from helpers import parity_tolerance, path_abs_jacobian
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


def ref_loss(out64, y, w, kind):
    e = out64 - y.astype(np.float64)
    term = np.abs(e) if kind == "L1" else e * e
    if w is not None:
        term = np.where(w != 0, w.astype(np.float64) * term, 0.0)
    return term.sum(), e


def check_losses(api, trees, ops, X, y, w, kind, dtype, eval_context=None, min_ok=1):
    pop = api.Population(trees, ops, dtype, n_features=X.shape[0], eval_context=eval_context)
    loss, ok = pop.eval_loss(X, y, weights=w, loss=kind)
    out, ok_eval = pop.eval(X)
    assert np.array_equal(ok, ok_eval), "fused loss and plain eval disagree on the completion flags"
    opts = (eval_context or api.EvalContext()).option_bits(ops)
    eps = np.finfo(dtype).eps
    n_ok = 0
    for t, tree in enumerate(trees):
        if not ok[t]:
            assert np.isnan(loss[t]), f"tree {t}: incomplete evaluation must give a NaN loss"
            continue
        n_ok += 1
        # (b) same values, other summation order: only rounding of the sum itself
        self_ref, _ = ref_loss(out[t].astype(np.float64), y, w, kind)
        if np.isnan(self_ref):  # complete=true does not promise finite values (early_exit=false, untested leaves)
            assert np.isnan(loss[t])
            continue
        e_self = out[t].astype(np.float64) - y
        big = np.abs(e_self).max() if kind == "L1" else (e_self * e_self).max()
        if max(self_ref, big) > 0.25 * np.finfo(dtype).max:  # a term or the sum overflows T, as sum(abs2, ...) in T does
            assert np.isposinf(loss[t]) or abs(float(loss[t]) - self_ref) <= 64 * eps * self_ref
            continue
        assert abs(float(loss[t]) - self_ref) <= 64 * eps * abs(self_ref) + 1e-300, (t, loss[t], self_ref)
        # (a) the oracle's values: each sample may differ by its parity tolerance
        tape, consts = de.flatten(tree, ops, dtype)
        yo, ok_el = oracle.eval_tree_array(tape, consts, X, opts, elementwise=True)
        assert ok_el
        tol = parity_tolerance(tree, ops, X, dtype, opts)
        if not np.all(np.isfinite(tol)):
            continue  # a chaotic sample: no meaningful bound on the sum
        want, e = ref_loss(yo.astype(np.float64), y, w, kind)
        ww = np.ones_like(tol) if w is None else np.abs(w.astype(np.float64))
        slack = (ww * tol).sum() if kind == "L1" else (ww * (2 * np.abs(e) * tol + tol * tol)).sum()
        assert abs(float(loss[t]) - want) <= slack + 64 * eps * abs(want) + 1e-300, (t, loss[t], want, slack)
    assert n_ok >= min_ok
    pop.close()


@pytest.mark.parametrize("N", [1, 63, 1024, 4099])
@pytest.mark.parametrize("kind", ["L2", "L1"])
def test_fused_loss_f32_vs_oracle(api, N, kind):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(150, seed=0xDE02)
    X = de.synth.random_X(5, N, seed=1)
    g = np.random.Generator(np.random.PCG64(N))
    y = g.standard_normal(N).astype(np.float32)
    check_losses(api, trees, ops, X, y, None, kind, np.float32, min_ok=15)


def test_fused_loss_with_weights_and_excluded_samples(api):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(100, seed=77)
    N = 3001
    X = de.synth.random_X(5, N, seed=3)
    g = np.random.Generator(np.random.PCG64(9))
    y = g.standard_normal(N).astype(np.float32)
    w = g.uniform(0, 2, N).astype(np.float32)
    w[::7] = 0
    check_losses(api, trees, ops, X, y, w, "L2", np.float32, min_ok=10)


def test_fused_loss_f64(api):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(80, seed=0xDE03, dtype=np.float64)
    N = 2051
    X = de.synth.random_X(5, N, seed=2, dtype=np.float64)
    y = np.cos(np.arange(N, dtype=np.float64))
    check_losses(api, trees, ops, X, y, None, "L2", np.float64, min_ok=15)


def test_fused_loss_all_option_modes_and_nonfinite_inputs(api):
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min"),
                          unary_operators=("cos", "exp", "safe_log", "neg", "square", "abs", "tanh", "safe_sqrt"))
    rng = de.synth.Xoshiro256ss(99)
    trees = [de.synth.gen_random_tree_fixed_size(5 + i % 20, ops, 3, rng, np.float32) for i in range(100)]
    g = np.random.Generator(np.random.PCG64(3))
    X = np.asfortranarray(g.standard_normal((3, 777)).astype(np.float32))
    X[1, 5] = np.inf
    y = g.standard_normal(777).astype(np.float32)
    for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False)):
        check_losses(api, trees, ops, X, y, None, "L2", np.float32, eval_context=ec, min_ok=0)


def test_fused_loss_device_tensors_reproducible_and_exact_properties(api):
    """Size-independent properties at 10^6 samples: loss(tree_t, y = tree_t(X)) == 0 exactly;
    doubling every weight doubles every loss bit-for-bit; two runs are bit-identical."""
    import torch
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(200, seed=0xDE02)
    N = 10**6 + 37
    pop = api.Population(trees, ops, np.float32, n_features=5)
    g = torch.Generator(device="cuda").manual_seed(4)
    X = torch.randn((N, 5), generator=g, device="cuda").t()
    out, ok = pop.eval(X)
    t_ok = [int(t) for t in torch.nonzero(ok).flatten()[:3]]
    assert t_ok
    for t in t_ok:
        loss, ok2 = pop.eval_loss(X, out[t])
        assert torch.equal(ok, ok2)
        assert float(loss[t]) == 0.0
        assert bool((loss[ok] >= 0).all()) and bool(torch.isnan(loss[~ok]).all())
    y = torch.randn(N, generator=g, device="cuda")
    w = torch.rand(N, generator=g, device="cuda")
    l1, _ = pop.eval_loss(X, y, weights=w)
    l1b, _ = pop.eval_loss(X, y, weights=w)
    l2, _ = pop.eval_loss(X, y, weights=2 * w)
    torch.cuda.synchronize()
    assert torch.equal(l1[ok], l1b[ok])
    assert torch.equal(2 * l1[ok], l2[ok])
    # against the materialised outputs, reduced in float64 on the device
    want = (w.double() * (out.double() - y.double()) ** 2).sum(dim=1)
    rel = ((l1.double() - want).abs() / want.abs().clamp_min(1e-300))[ok]
    assert float(rel.max()) < 1e-5
    pop.close()


def test_fused_loss_parametric_expression(api):
    ops = de.OperatorEnum(binary_operators=("+", "*", "-"), unary_operators=("cos", "exp"))
    rng = de.synth.Xoshiro256ss(21)
    trees = [de.synth.gen_random_tree_fixed_size(9 + i % 8, ops, 2, rng, np.float32, de.ParametricNode, 2)
             for i in range(40)]
    N, P, Cn = 1500, 2, 4
    g = np.random.Generator(np.random.PCG64(5))
    X = np.asfortranarray(g.standard_normal((2, N)).astype(np.float32))
    params = np.asfortranarray(g.standard_normal((P, Cn)).astype(np.float32))
    classes = g.integers(1, Cn + 1, N)
    y = g.standard_normal(N).astype(np.float32)
    pop = api.Population(trees, ops, np.float32, n_features=2, n_params=P)
    loss, ok = pop.eval_loss(X, y, params=params, classes=classes)
    out, ok_e = pop.eval(X, params=params, classes=classes)
    assert np.array_equal(ok, ok_e) and ok.any()
    want = ((out.astype(np.float64) - y) ** 2).sum(axis=1)
    assert np.allclose(loss[ok], want[ok], rtol=1e-5)
    pop.close()


def test_fused_loss_empty_and_error_paths(api):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(5, seed=1)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    loss, ok = pop.eval_loss(np.zeros((5, 0), np.float32), np.zeros(0, np.float32))
    assert np.array_equal(loss, np.zeros(5, np.float32)) and ok.all()
    X = de.synth.random_X(5, 10, seed=1)
    with pytest.raises(ValueError):
        pop.eval_loss(X, np.zeros(9, np.float32))
    lib = api.library()
    okb = np.zeros(5, np.uint8)
    lossb = np.zeros(5, np.float32)
    rc = lib.de_eval_loss(pop.ctx._h, pop._h, X.ctypes.data, 10, 5, None, None, None, 0, lossb.ctypes.data, okb.ctypes.data)
    assert rc == 1  # DE_ERR_INVALID_ARG: y is null
    yb = np.zeros(10, np.float32)
    rc = lib.de_eval_loss(pop.ctx._h, pop._h, X.ctypes.data, 10, 5, None, yb.ctypes.data, None, 7, lossb.ctypes.data, okb.ctypes.data)
    assert rc == 1 and b"loss_kind" in lib.de_last_error(pop.ctx._h)
    pop.close()


# ---- fused loss + gradient (de_eval_loss_grad) -------------------------------------------------
@pytest.fixture(params=["forward", "forward-2-per-lane", "reverse"], autouse=True)
def accumulation(request, monkeypatch):
    """Every fused loss+gradient test runs with all kernels: forward duals (de_grad_threaded.hip) with one sample per
    lane and with the two-samples-per-lane modules (which the library only uses from 65536 samples on), and reverse
    accumulation (de_rev_threaded.hip; picked by gradient width unless DE_LOSS_GRAD_REVERSE says)."""
    monkeypatch.setenv("DE_LOSS_GRAD_REVERSE", "1" if request.param == "reverse" else "0")
    monkeypatch.setenv("DE_GRAD_VS2_MIN_N", "0" if request.param == "forward-2-per-lane" else "1000000000000")
    return request.param


MODES = {"variable": (True, oracle.GRAD_VARIABLE), "constant": (False, oracle.GRAD_CONSTANT),
         "both": ("both", oracle.GRAD_BOTH)}


def ref_loss_grad(out64, g64, y, w, kind, gabs=None):
    """(loss, dloss[k], magnitude of the gradient terms) from a materialised evaluation + Jacobian.  The
    magnitude is sum_j |lp_j| * gabs[k, j]: with gabs = the path-absolute Jacobian (helpers.path_abs_jacobian)
    it bounds the rounding of ANY association of the per-sample products and sums — forward duals and reverse
    accumulation differ exactly there, e.g. d/dx (x * (c / x)) is rounding noise in both, but different noise."""
    y = y.astype(np.float64)
    ww = np.ones_like(y) if w is None else w.astype(np.float64)
    if kind == "pullback":
        l, lp = out64 * y, y
    else:
        e = out64 - y
        l, lp = (e * e, 2 * e) if kind == "L2" else (np.abs(e), np.sign(e))
    keep = ww != 0
    terms = (ww * lp)[None, keep] * g64[:, keep]
    mag = np.abs(terms).sum(axis=1)
    if gabs is not None:
        with np.errstate(all="ignore"):
            mag = np.maximum(mag, np.nan_to_num((np.abs(ww * lp)[None, keep] * gabs[:, keep]).sum(axis=1), nan=np.inf, posinf=np.inf))
    return (ww * l)[keep].sum(), terms.sum(axis=1), mag


def check_loss_grads(api, trees, ops, X, y, w, kind, dtype, mode_name, oracle_exact=False, min_ok=1, **pkw):
    variable, omode = MODES[mode_name]
    pop = api.Population(trees, ops, dtype, n_features=X.shape[0], n_params=pkw.pop("n_params", 0))
    loss, dls, ok = pop.eval_loss_grad(X, y, weights=w, loss=kind, variable=variable, **pkw)
    out, grads, ok_g = pop.eval_grad(X, variable, **pkw)
    assert np.array_equal(ok, ok_g)
    eps = np.finfo(dtype).eps
    fmax = np.finfo(dtype).max
    tiny = float(np.finfo(dtype).tiny) * X.shape[1]  # terms below the normal range flush/round
    n_ok = 0
    for t, tree in enumerate(trees):
        assert dls[t].shape == (grads[t].shape[0],)
        if not ok[t]:
            assert np.isnan(loss[t]) and np.all(np.isnan(dls[t]))  # src/ChainRules.jl:62-64
            continue
        srcs = [(out[t].astype(np.float64), np.asarray(grads[t], dtype=np.float64))]
        if oracle_exact:  # IEEE-exact operators: the oracle's Jacobian is bit-identical, so it is a second anchor
            tape, consts = de.flatten(tree, ops, dtype)
            yo, go, ok_o = oracle.eval_grad_tree_array(tape, consts, X, omode, elementwise=True)
            assert ok_o
            srcs.append((yo.astype(np.float64), go.astype(np.float64)))
        gabs = path_abs_jacobian(tree, ops, X, mode_name, pkw.get("params"), pkw.get("classes"), pkw.get("class_base", 1))
        for o64, g64 in srcs:
            want_l, want_g, mag = ref_loss_grad(o64, g64, y, w, kind, gabs)
            if not np.all(np.isfinite(want_g)) or max(abs(want_l), mag.max(initial=0)) > 0.05 * fmax:
                continue  # a term overflows T
            assert abs(float(loss[t]) - want_l) <= 64 * eps * abs(want_l) + 64 * eps * np.abs(o64 * y).sum() * (kind == "pullback") + tiny
            err = np.abs(dls[t].astype(np.float64) - want_g)
            assert np.all(err <= 64 * eps * mag + tiny), (t, mode_name, de.string_tree(tree, ops), dls[t], want_g)
        n_ok += 1
    assert n_ok >= min_ok
    pop.close()
    return n_ok


@pytest.mark.parametrize("mode", ["constant", "variable", "both"])
@pytest.mark.parametrize("kind", ["L2", "L1", "pullback"])
def test_fused_loss_grad_exact_operators_vs_oracle(api, mode, kind):
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*"), unary_operators=("neg", "square", "abs"))
    rng = de.synth.Xoshiro256ss(17)
    for dtype in (np.float32, np.float64):
        trees = [de.synth.gen_random_tree_fixed_size(3 + i % 26, ops, 4, rng, dtype) for i in range(60)]
        N = 777
        X = de.synth.random_X(4, N, seed=12, dtype=dtype)
        g = np.random.Generator(np.random.PCG64(2))
        y = g.standard_normal(N).astype(dtype)
        w = g.uniform(0, 2, N).astype(dtype)
        w[::5] = 0
        check_loss_grads(api, trees, ops, X, y, w, kind, dtype, mode, oracle_exact=True, min_ok=10)


@pytest.mark.parametrize("N", [1, 300, 2049])
def test_fused_loss_grad_random_population(api, N):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(120, seed=0xDE03)
    X = de.synth.random_X(5, N, seed=6)
    y = np.sin(np.arange(N)).astype(np.float32)
    for mode in ("constant", "variable", "both"):
        check_loss_grads(api, trees, ops, X, y, None, "L2", np.float32, mode, min_ok=5)


def test_fused_loss_grad_many_constants_use_several_windows(api):
    ops = de.OperatorEnum(binary_operators=("+", "*"))
    t = de.Node(feature=1)
    for i in range(19):
        t = de.Node(1 + i % 2, t, de.Node(val=0.5 + 0.1 * i))
    X = de.synth.random_X(2, 1000, seed=2, dtype=np.float64)
    y = np.cos(np.arange(1000.0))
    for mode in ("constant", "both"):
        check_loss_grads(api, [t, de.Node(feature=2), t.copy()], ops, X, y, None, "L2", np.float64, mode,
                         oracle_exact=True, min_ok=3)


def test_fused_loss_grad_parametric_and_device_tensors(api):
    import torch
    ops = de.OperatorEnum(binary_operators=("+", "*", "-"), unary_operators=("cos", "exp"))
    rng = de.synth.Xoshiro256ss(21)
    trees = [de.synth.gen_random_tree_fixed_size(9 + i % 8, ops, 2, rng, np.float32, de.ParametricNode, 2)
             for i in range(40)]
    N, P, Cn = 1500, 2, 4
    g = np.random.Generator(np.random.PCG64(5))
    X = np.asfortranarray(g.standard_normal((2, N)).astype(np.float32))
    params = np.asfortranarray(g.standard_normal((P, Cn)).astype(np.float32))
    classes = g.integers(1, Cn + 1, N)
    y = g.standard_normal(N).astype(np.float32)
    for mode in ("constant", "variable"):
        check_loss_grads(api, trees, ops, X, y, None, "L2", np.float32, mode, min_ok=5, n_params=P,
                         params=params, classes=classes)
    # device tensors, repeatability, and the optimiser identity: loss from loss_grad == loss from eval_loss
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(100, seed=3)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    gen = torch.Generator(device="cuda").manual_seed(8)
    Nd = 200_003
    Xd = torch.randn((Nd, 5), generator=gen, device="cuda").t()
    yd = torch.randn(Nd, generator=gen, device="cuda")
    l1, d1, ok1 = pop.eval_loss_grad(Xd, yd)
    l2, d2, ok2 = pop.eval_loss_grad(Xd, yd)
    torch.cuda.synchronize()
    assert torch.equal(ok1, ok2) and torch.equal(l1[ok1], l2[ok1])
    for a, b, k in zip(d1, d2, ok1.tolist()):
        if k:
            assert torch.equal(a, b)
    out, grads, okg = pop.eval_grad(Xd, False)
    for t in torch.nonzero(ok1).flatten().tolist()[:20]:
        want = (2 * (out[t].double() - yd.double())[None, :] * grads[t].double()).sum(dim=1)
        mag = (2 * (out[t].double() - yd.double())[None, :] * grads[t].double()).abs().sum(dim=1)
        assert bool(((d1[t].double() - want).abs() <= 64 * 1.2e-7 * mag + 1e-30).all())
    pop.close()


# ---- parameter gradients reduced by class (de_eval_loss_grad_by_class) -----------------------------
def test_parameter_gradient_by_class_reference_known_answer(api):
    """test/test_parametric_expression.jl:321-372: ex = x*x - cos(2.5*y) + p1, params [0.1 0.2], loss
    sum(abs2, ex(X, classes) - y_true): the gradient w.r.t. the constant and w.r.t. the parameter MATRIX equal
    the closed form (there: Zygote through the hand-written prediction).  X / classes are re-drawn with our
    PRNG (MersenneTwister streams are not reproducible here); the assertion is the reference's."""
    from helpers import sexpr_to_node
    ops = de.OperatorEnum(binary_operators=("+", "*", "-"), unary_operators=("cos",))
    tree = sexpr_to_node(["+", ["-", ["*", ["x", 1], ["x", 1]], ["cos", ["*", 2.5, ["x", 2]]]], ["p", 1]], ops,
                         de.ParametricNode)
    g = np.random.Generator(np.random.PCG64(0))
    for N in (32, 1000):
        X = np.asfortranarray(g.random((2, N)))
        classes = g.integers(1, 3, N)
        true_params, init = np.array([[0.5, 2.0]]), np.array([[0.1, 0.2]])
        y = X[0] * X[0] - np.cos(2.6 * X[1]) + true_params[0, classes - 1]
        pred = X[0] * X[0] - np.cos(2.5 * X[1]) + init[0, classes - 1]
        e = pred - y
        true_val = (e * e).sum()
        true_dc = (2 * e * np.sin(2.5 * X[1]) * X[1]).sum()
        true_dp = np.array([[(2 * e)[classes == 1].sum(), (2 * e)[classes == 2].sum()]])
        pop = api.Population([tree], ops, np.float64, n_features=2, n_params=1)
        loss, dls, dp, ok = pop.eval_loss_grad_by_class(X, y, init, classes, variable="both")
        assert ok[0] and dp.shape == (1, 1, 2)
        np.testing.assert_allclose(loss[0], true_val, rtol=1e-12)
        # rows of :both on a parametric tree: parameters, features, constants
        np.testing.assert_allclose(dls[0][3], true_dc, rtol=1e-10)
        np.testing.assert_allclose(dp[0], true_dp, rtol=1e-10)
        np.testing.assert_allclose(dls[0][0], true_dp.sum(), rtol=1e-10)  # the dloss row is the sum over classes
        pop.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["L2", "pullback"])
def test_parameter_gradient_by_class_matches_scatter_of_jacobian(api, dtype, kind, monkeypatch):
    """dparams[t][p, c] == sum over the samples of class c of l'(e_j) * Jacobian row p — the scatter-add the
    reference's AD performs through `parameters[i, classes[j]]` (src/ParametricExpression.jl:381-384); classes
    arrive unordered (the wrapper groups them), some classes are empty, weights exclude samples."""
    ops = de.OperatorEnum(binary_operators=("+", "*", "-", "/"), unary_operators=("cos", "exp"))
    rng = de.synth.Xoshiro256ss(33)
    P, Cn, N = 3, 7, 2500
    trees = [de.synth.gen_random_tree_fixed_size(7 + i % 12, ops, 2, rng, dtype, de.ParametricNode, P) for i in range(60)]
    g = np.random.Generator(np.random.PCG64(9))
    X = np.asfortranarray(g.standard_normal((2, N)).astype(dtype))
    params = np.asfortranarray(g.standard_normal((P, Cn)).astype(dtype))
    classes = g.choice([1, 2, 4, 5, 7], N)  # classes 3 and 6 have no sample
    y = g.standard_normal(N).astype(dtype)
    w = (g.random(N) > 0.2).astype(dtype) * g.random(N).astype(dtype)
    pop = api.Population(trees, ops, dtype, n_features=2, n_params=P)
    for variable in (True, "both"):
        loss, dls, dp, ok = pop.eval_loss_grad_by_class(X, y, params, classes, weights=w, loss=kind, variable=variable)
        loss1, dls1, ok1 = pop.eval_loss_grad(X, y, weights=w, loss=kind, variable=variable, params=params, classes=classes)
        out, grads, okg = pop.eval_grad(X, variable, params=params, classes=classes)
        assert np.array_equal(ok, okg) and np.array_equal(ok, ok1) and ok.sum() > 10
        assert dp.shape == (len(trees), P, Cn)
        eps, fmax = np.finfo(dtype).eps, float(np.finfo(dtype).max)
        for t in range(len(trees)):
            if not ok[t]:
                assert np.isnan(loss[t]) and np.isnan(dls[t]).all() and np.isnan(dp[t]).all()
                continue
            o64, g64 = out[t].astype(np.float64), np.asarray(grads[t], dtype=np.float64)
            lp = y.astype(np.float64) if kind == "pullback" else 2 * (o64 - y)
            terms = (w.astype(np.float64) * lp)[None, :] * g64
            gabs = path_abs_jacobian(trees[t], ops, X, "variable" if variable is True else "both", params, classes)
            aterms = np.abs(terms) if gabs is None else np.maximum(np.abs(terms), np.nan_to_num(
                np.abs(w.astype(np.float64) * lp)[None, :] * gabs, nan=np.inf))  # conditioning under any association
            for c in range(Cn):
                sel = classes == c + 1
                want, mag = terms[:P, sel].sum(axis=1), aterms[:P, sel].sum(axis=1)
                big = mag > 0.25 * fmax  # a sum beyond the range of the type may be +-Inf
                assert np.all((np.abs(dp[t][:, c] - want) <= 64 * eps * mag + 1e-300) | big), (t, c)
                if not sel.any():
                    assert np.all(dp[t][:, c] == 0)
            want, mag = terms.sum(axis=1), aterms.sum(axis=1)
            assert np.all((np.abs(dls[t] - want) <= 64 * eps * mag + 1e-300) | (mag > 0.25 * fmax))
            assert loss[t] == loss1[t] or abs(loss[t] - loss1[t]) <= 64 * eps * abs(loss1[t]) + 1e-300 or kind == "pullback"
    # run-to-run reproducible (fixed-order reduction), and grouped=True input gives the same bits
    a = pop.eval_loss_grad_by_class(X, y, params, classes, weights=w, loss=kind)
    order = np.argsort(classes, kind="stable")
    b = pop.eval_loss_grad_by_class(np.asfortranarray(X[:, order]), y[order], params, classes[order], weights=w[order],
                                    loss=kind, grouped=True)
    assert np.array_equal(a[2], b[2], equal_nan=True) and np.array_equal(a[0], b[0], equal_nan=True)
    # one pass over class-aligned tiles (reverse kernel) == one call per class: same tiles, same reduction order
    monkeypatch.setenv("DE_BY_CLASS_ONE_PASS", "0")
    c2 = pop.eval_loss_grad_by_class(X, y, params, classes, weights=w, loss=kind)
    assert np.array_equal(a[2], c2[2], equal_nan=True) and np.array_equal(a[0], c2[0], equal_nan=True)
    assert all(np.array_equal(u, v, equal_nan=True) for u, v in zip(a[1], c2[1])) and np.array_equal(a[3], c2[3])
    pop.close()


def test_parameter_gradient_by_class_edge_cases(api):
    """One class (the by-class matrix is the shared row), no samples (zeros), int64 class ids, L1 loss, Float64."""
    ops = de.OperatorEnum(binary_operators=("+", "*", "-"), unary_operators=("cos",))
    rng = de.synth.Xoshiro256ss(8)
    P = 2
    trees = [de.synth.gen_random_tree_fixed_size(5 + i % 9, ops, 3, rng, np.float64, de.ParametricNode, P) for i in range(24)]
    g = np.random.Generator(np.random.PCG64(3))
    N = 700
    X = np.asfortranarray(g.standard_normal((3, N)))
    y = g.standard_normal(N)
    pop = api.Population(trees, ops, np.float64, n_features=3, n_params=P)
    # a single class: dparams[:, 0] is the parameter rows of the plain fused gradient
    params1 = np.asfortranarray(g.standard_normal((P, 1)))
    ones = np.ones(N, dtype=np.int64)
    loss, dls, dp, ok = pop.eval_loss_grad_by_class(X, y, params1, ones, loss="L1", variable=True)
    loss1, dls1, ok1 = pop.eval_loss_grad(X, y, loss="L1", variable=True, params=params1, classes=ones)
    assert np.array_equal(ok, ok1) and ok.sum() > 10 and dp.shape == (24, P, 1)
    for t in np.nonzero(ok)[0]:
        np.testing.assert_allclose(dp[t][:, 0], dls1[t][:P], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(dls[t], dls1[t], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(loss[t], loss1[t], rtol=1e-13)
    # int64 ids with a base of 0, three classes of which the middle one is empty
    params3 = np.asfortranarray(g.standard_normal((P, 3)))
    cls = np.where(g.random(N) < 0.5, 0, 2).astype(np.int64)
    a = pop.eval_loss_grad_by_class(X, y, params3, cls, class_base=0)
    b = pop.eval_loss_grad_by_class(X, y, params3, cls.astype(np.int32), class_base=0)
    assert np.array_equal(a[2], b[2], equal_nan=True) and np.all(a[2][a[3]][:, :, 1] == 0)
    # no samples at all
    l0, d0, p0, k0 = pop.eval_loss_grad_by_class(np.zeros((3, 0)), np.zeros(0), params3, np.zeros(0, dtype=np.int64), class_base=0)
    assert k0.all() and np.all(l0 == 0) and np.all(p0 == 0) and all(np.all(d == 0) for d in d0)
    pop.close()


def test_parameter_gradient_by_class_device_tensors_and_errors(api):
    import torch
    trees = de.synth.random_population(50, seed=0xDE05, node_type=de.ParametricNode, nparams=8)
    ops = de.synth.BENCH_OPERATORS
    pop = api.Population(trees, ops, np.float32, n_features=5, n_params=8)
    gen = torch.Generator(device="cuda").manual_seed(4)
    N, Cn = 50_001, 16
    Xd = torch.randn((N, 5), generator=gen, device="cuda").t()
    dY = torch.randn(N, generator=gen, device="cuda")
    params = torch.randn((Cn, 8), generator=gen, device="cuda").t()
    classes = torch.randint(1, Cn + 1, (N,), generator=gen, device="cuda", dtype=torch.int32)
    loss, dls, dp, ok = pop.eval_loss_grad_by_class(Xd, dY, params, classes, loss="pullback")
    torch.cuda.synchronize()
    assert dp.shape == (50, 8, Cn) and ok.sum().item() > 5
    out, grads, okg = pop.eval_grad(Xd, "both", params=params, classes=classes)
    assert torch.equal(ok, okg)
    for t in torch.nonzero(ok).flatten().tolist()[:10]:
        terms = dY.double()[None, :] * grads[t][:8].double()
        for c in (0, 7, 15):
            sel = classes == c + 1
            want, mag = terms[:, sel].sum(dim=1), terms[:, sel].abs().sum(dim=1)
            assert bool(((dp[t][:, c].double() - want).abs() <= 64 * 1.2e-7 * mag + 1e-30).all())
    # misuse: constant mode has no parameter rows; a plain population has no parameters
    with pytest.raises(ValueError):
        pop.eval_loss_grad_by_class(Xd, dY, params, classes, variable=False)
    pop.close()
    plain = api.Population(de.synth.random_population(3, seed=1), ops, np.float32, n_features=5)
    with pytest.raises(ValueError):
        plain.eval_loss_grad_by_class(Xd, dY, params, classes)
    plain.close()


def test_fused_loss_grad_empty_and_error_paths(api):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(5, seed=1)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    loss, dls, ok = pop.eval_loss_grad(np.zeros((5, 0), np.float32), np.zeros(0, np.float32))
    assert np.array_equal(loss, np.zeros(5, np.float32)) and ok.all()
    assert all(np.array_equal(d, np.zeros_like(d)) for d in dls)
    X = de.synth.random_X(5, 10, seed=1)
    with pytest.raises(ValueError):
        pop.eval_loss_grad(X, np.zeros(9, np.float32))
    with pytest.raises(KeyError):
        pop.eval_loss(X, np.zeros(10, np.float32), loss="pullback")  # value-only entry point has no cotangent mode
    pop.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reverse_kernel_fused_records_give_the_same_bits(api, dtype, monkeypatch):
    """The reverse-accumulation streams fuse fixed sequences into one record (csrc/de_api_grad.cpp ensure_rev_threaded: PUSH + load / unary of
    a leaf; [r_un] r_leaf [r_pop]; r_bin<column> r_leaf [r_pop]) — the same arithmetic in the same order: losses, gradients and flags
    are bit-identical to the unfused streams (DE_REV_NO_FUSE=1), for plain and parametric populations and the three modes."""
    import dynamicexpressions_jl_amd as de
    monkeypatch.setenv("DE_LOSS_GRAD_REVERSE", "1")
    ops = de.synth.BENCH_OPERATORS
    g = np.random.Generator(np.random.PCG64(77))
    N = 1500
    X = np.asfortranarray(g.standard_normal((4, N)).astype(dtype))
    y = g.standard_normal(N).astype(dtype)
    w = (g.random(N) > 0.1).astype(dtype)
    plain = de.synth.random_population(120, seed=0xF05E, dtype=dtype, nfeatures=4)
    par = de.synth.random_population(120, seed=0xF05F, dtype=dtype, nfeatures=4, node_type=de.ParametricNode, nparams=3)
    params = np.asfortranarray(g.standard_normal((3, 5)).astype(dtype))
    classes = g.integers(1, 6, N)
    it = np.uint32 if dtype == np.float32 else np.uint64
    for trees, P, kw in ((plain, 0, {}), (par, 3, dict(params=params, classes=classes))):
        res = {}
        for nofuse in ("0", "1"):
            if nofuse == "1":
                monkeypatch.setenv("DE_REV_NO_FUSE", "1")
            else:
                monkeypatch.delenv("DE_REV_NO_FUSE", raising=False)
            pop = api.Population(trees, ops, dtype, n_features=4, n_params=P)
            res[nofuse] = [pop.eval_loss_grad(X, y, weights=w, loss=kind, variable=v, **kw) for v in (False, True, "both") for kind in ("L2", "pullback")]
            pop.close()
        for (l0, d0, k0), (l1, d1, k1) in zip(res["0"], res["1"]):
            assert np.array_equal(k0, k1) and 0 < k0.sum() < len(trees)
            assert np.array_equal(np.asarray(l0).view(it), np.asarray(l1).view(it))
            for a, b in zip(d0, d1):
                assert np.array_equal(np.asarray(a).view(it), np.asarray(b).view(it))
