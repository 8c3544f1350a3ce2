"""GPU parity tests of the fused loss reduction (de_eval_loss): the residual sum a
consumer of eval_tree_array computes right after the call — `sum(abs2, tree(X, operators) .- y)`,
test/test_optim.jl:95,99 — evaluated without writing the [n_trees, N] output.
Checked against (a) the CPU oracle's outputs reduced in float64 and (b) the device's own de_eval
outputs reduced in float64 (same values, different summation order)."""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from helpers import parity_tolerance
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
MODES = {"variable": (True, oracle.GRAD_VARIABLE), "constant": (False, oracle.GRAD_CONSTANT),
         "both": ("both", oracle.GRAD_BOTH)}


def ref_loss_grad(out64, g64, y, w, kind):
    """(loss, dloss[k], abs-sum of the gradient terms) from a materialised evaluation + Jacobian."""
    y = y.astype(np.float64)
    ww = np.ones_like(y) if w is None else w.astype(np.float64)
    if kind == "pullback":
        l, lp = out64 * y, y
    else:
        e = out64 - y
        l, lp = (e * e, 2 * e) if kind == "L2" else (np.abs(e), np.sign(e))
    keep = ww != 0
    terms = (ww * lp)[None, keep] * g64[:, keep]
    return (ww * l)[keep].sum(), terms.sum(axis=1), np.abs(terms).sum(axis=1)


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
        for o64, g64 in srcs:
            want_l, want_g, mag = ref_loss_grad(o64, g64, y, w, kind)
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
