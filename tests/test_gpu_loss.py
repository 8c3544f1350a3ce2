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
