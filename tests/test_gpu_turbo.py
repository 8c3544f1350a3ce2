"""DE_OPT_TURBO — `EvalContext(turbo=true)`, the reference's LoopVectorization path
(ext/DynamicExpressionsLoopVectorizationExt.jl:24-278) — on the GPU: relaxed-accuracy Float32 `/`, exp, cos, sin.
The reference's own turbo results drift from the plain path's (SLEEF; test/test_supposition_consistency.jl:106-108
compares them with `isapprox`), so parity is what north_star states for Float32: <= 1e-5 relative against the oracle
(enforced through the same conditioned tolerance as the exact mode), flags equal except through the documented domain
edges (csrc/de_device_ops.h).  The per-operator accuracy is measured in ulps and written to gpurun_out/ulp_turbo_f32.json."""
import json
import os

import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from helpers import parity_tolerance
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


def rel_err(got, want64):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.abs(got.astype(np.float64) - want64) / np.abs(want64)


def ulps32(got, want64):
    u = np.spacing(np.abs(want64.astype(np.float32))).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / u


def run1(api, name, degree, X, turbo=True):
    if degree == 1:
        ops = de.OperatorEnum(binary_operators=("+",), unary_operators=(name,))
        tree = de.Node(1, de.Node(feature=1))
    else:
        ops = de.OperatorEnum(binary_operators=(name,))
        tree = de.Node(1, de.Node(feature=1), de.Node(feature=2))
    out, _ = api.eval_tree_array(tree, np.asfortranarray(X), ops, eval_context=api.EvalContext(turbo=turbo, early_exit=False))
    return out


@pytest.mark.parametrize("name,f64", [("cos", np.cos), ("sin", np.sin)])
def test_turbo_trig_accuracy(api, name, f64):
    g = np.random.Generator(np.random.PCG64(17))
    x = np.concatenate([g.uniform(-4, 4, 500_000), g.uniform(-100, 100, 500_000), g.standard_normal(100_000) * 1e-3,
                        [0.0, -0.0, 1e-30, 1e-45]]).astype(np.float32)
    out = run1(api, name, 1, x[None, :])
    want = f64(x.astype(np.float64))
    e = rel_err(out, want)
    e[want == 0] = np.abs(out[want == 0])
    REPORT[name] = dict(max_rel=float(np.nanmax(e)), max_ulp=float(np.nanmax(ulps32(out, want))), range="|x| <= 100", points=int(x.size))
    assert np.nanmax(e) <= 1e-6, f"{name}: {np.nanmax(e):.3g} at x = {x[np.nanargmax(e)]!r}"
    # |x| <= 1e4: the two-term pi keeps the ABSOLUTE error at the 1e-7 level (relative error only where |f| is not tiny)
    xl = g.uniform(-1e4, 1e4, 500_000).astype(np.float32)
    ol, wl = run1(api, name, 1, xl[None, :]), f64(xl.astype(np.float64))
    assert np.max(np.abs(ol.astype(np.float64) - wl)) <= 3e-7
    REPORT[name + " |x|<=1e4"] = dict(max_abs=float(np.max(np.abs(ol.astype(np.float64) - wl))), points=int(xl.size))
    # non-finite in, non-finite out; arguments beyond 1e5 take the full range reduction of the exact mode (cos(exp(exp(x)))
    # is common in random trees: a value outside [-1, 1] there would change flags downstream)
    xs = np.array([np.inf, -np.inf, np.nan, 3e38, -1e30, 1.5e7, -2.5e9], dtype=np.float32)
    o = run1(api, name, 1, xs[None, :])
    assert np.all(~np.isfinite(o[:3]))
    assert ulps32(o[3:], f64(xs[3:].astype(np.float64))).max() <= 2.0
    xm = g.uniform(-1e7, 1e7, 200_000).astype(np.float32)
    xm[::2] = g.uniform(-1e5, 1e5, xm[::2].size).astype(np.float32)  # waves that mix both paths
    assert np.max(np.abs(run1(api, name, 1, xm[None, :]).astype(np.float64) - f64(xm.astype(np.float64)))) <= 3e-7
    # and the exact mode is untouched by the option: bit-identical with and without an unrelated context field
    np.testing.assert_array_equal(run1(api, name, 1, x[None, :1000], turbo=False), run1(api, name, 1, x[None, :1000], turbo=False))
    assert np.any(run1(api, name, 1, x[None, :100000], turbo=False) != out[:100000])  # the option really selects other code


def test_turbo_exp_accuracy(api):
    g = np.random.Generator(np.random.PCG64(18))
    x = np.concatenate([g.uniform(-87, 88, 1_000_000), g.uniform(-2, 2, 500_000), [0.0, -0.0, 88.7, -87.3]]).astype(np.float32)
    out = run1(api, "exp", 1, x[None, :])
    want = np.exp(x.astype(np.float64))
    normal = (want < np.finfo(np.float32).max) & (want >= np.finfo(np.float32).tiny)
    e = rel_err(out[normal], want[normal])
    REPORT["exp"] = dict(max_rel=float(e.max()), max_ulp=float(ulps32(out[normal], want[normal]).max()), range="normal results", points=int(normal.sum()))
    assert e.max() <= 1e-6, f"exp: {e.max():.3g}"
    xs = np.array([200.0, np.inf, np.nan, -200.0, -1e30], dtype=np.float32)
    o = run1(api, "exp", 1, xs[None, :])
    assert not np.isfinite(o[0]) and not np.isfinite(o[1]) and np.isnan(o[2]) and o[3] == 0 and o[4] == 0


def test_turbo_division_accuracy(api):
    g = np.random.Generator(np.random.PCG64(19))
    N = 1 << 20
    a = (10.0 ** g.uniform(-18, 18, N) * g.choice([-1, 1], N)).astype(np.float32)
    b = (10.0 ** g.uniform(-18, 18, N) * g.choice([-1, 1], N)).astype(np.float32)
    out = run1(api, "/", 2, np.stack([a, b]))
    want = a.astype(np.float64) / b.astype(np.float64)
    m = (np.abs(want) < 1e38) & (np.abs(want) > 1e-37)
    e = rel_err(out[m], want[m])
    REPORT["/"] = dict(max_rel=float(e.max()), max_ulp=float(ulps32(out[m], want[m]).max()), range="1e-18 <= |x|, |y| <= 1e18", points=int(m.sum()))
    assert e.max() <= 3e-7
    sp = np.array([[1.0, 0.0, 0.0, np.inf, 1.0, np.nan], [0.0, 0.0, 3.0, np.inf, np.inf, 2.0]], dtype=np.float32)
    o = run1(api, "/", 2, sp)
    assert np.isposinf(o[0]) and np.isnan(o[1]) and o[2] == 0 and np.isnan(o[3]) and o[4] == 0 and np.isnan(o[5])
    # x / c and c / x (constant operand forms) follow the same contract
    ops = de.OperatorEnum(binary_operators=("/",))
    for tree, ref in ((de.Node(1, de.Node(feature=1), de.Node(val=3.7)), a.astype(np.float64) / np.float32(3.7)),
                      (de.Node(1, de.Node(val=-0.3), de.Node(feature=1)), np.float32(-0.3) / a.astype(np.float64))):
        o, _ = api.eval_tree_array(tree, np.asfortranarray(a[None, :]), ops, eval_context=api.EvalContext(turbo=True, early_exit=False))
        mm = (np.abs(ref) < 1e38) & (np.abs(ref) > 1e-37)
        assert rel_err(o[mm], ref[mm]).max() <= 3e-7


@pytest.mark.parametrize("N", [1000, 4099])
def test_turbo_bench_population_within_north_star_of_the_oracle(api, N):
    """north_star: 1e-5 relative for Float32.  The same conditioned bound as the exact mode (helpers.parity_tolerance:
    1e-5*|y| + the measured amplification of one-ulp differences) must hold with turbo operators, flags included."""
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(300, seed=0xDE02)
    X = de.synth.random_X(5, N, seed=1)
    pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(turbo=True))
    out, ok = pop.eval(X)
    exact, ok_x = api.Population(trees, ops, np.float32, n_features=5).eval(X)
    n_ok = n_flag = n_cmp = n_ill = 0
    worst = 0.0
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, np.float32)
        y, ok_el = oracle.eval_tree_array(tape, consts, X, elementwise=True)
        if bool(ok[t]) != ok_el:
            # only through a documented domain edge — a value that overflows / hits a pole in one mode only — i.e. on a tree the
            # tolerance model itself classes as ill-conditioned somewhere; anywhere else a flag difference is a bug
            n_flag += 1
            print("turbo flag differs:", de.string_tree(tree, ops))
            assert np.isinf(parity_tolerance(tree, ops, X, np.float32)).any(), f"turbo flag differs on a well-conditioned tree: {de.string_tree(tree, ops)}"
            continue
        if not ok_el:
            continue
        n_ok += 1
        tol = parity_tolerance(tree, ops, X, np.float32)
        m = np.isfinite(tol) & np.isfinite(out[t])
        err = np.abs(out[t].astype(np.float64) - y)
        assert np.all(err[m] <= tol[m]), f"turbo beyond the north-star bound: {de.string_tree(tree, ops)}"
        n_cmp += int(m.sum())
        n_ill += int(np.isinf(tol).sum())
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(np.abs(y[m]) > 0, err[m] / np.abs(y[m]), 0)
        wc = np.isfinite(tol[m]) & (tol[m] <= 1.0001e-5 * np.abs(y[m]) + 1e-37 + 8e-7 * np.abs(y[m]))
        if wc.any():
            worst = max(worst, float(r[wc].max()))
    print(f"[turbo parity N={N}] {n_ok} complete trees, {n_cmp} samples bounded, {100.0 * n_ill / max(n_cmp, 1):.2f} % ill-conditioned, "
          f"worst rel err on well-conditioned samples {worst:.3g}, {n_flag} flag differences")
    REPORT[f"bench population N={N}"] = dict(worst_rel_well_conditioned=worst, flag_differences=n_flag, trees=len(trees))
    assert n_ok > 50 and n_flag <= 1 and n_ill <= 0.05 * n_cmp
    assert int((np.asarray(ok_x) != np.asarray(ok)).sum()) <= n_flag  # the exact mode has the oracle's flags: turbo differs from it only where counted above


def test_turbo_is_ignored_where_it_has_no_meaning(api):
    """Float64 programs, gradients and wide-X programs run the exact operators whatever the bit says."""
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(30, seed=5, dtype=np.float64)
    X = de.synth.random_X(5, 500, seed=2, dtype=np.float64)
    a, ka = api.Population(trees, ops, np.float64, n_features=5, eval_context=api.EvalContext(turbo=True)).eval(X)
    b, kb = api.Population(trees, ops, np.float64, n_features=5).eval(X)
    assert np.array_equal(ka, kb)
    np.testing.assert_array_equal(a[ka], b[kb])
    t32 = de.synth.random_population(30, seed=6)
    X32 = de.synth.random_X(5, 500, seed=3)
    pa = api.Population(t32, ops, np.float32, n_features=5, eval_context=api.EvalContext(turbo=True))
    pb = api.Population(t32, ops, np.float32, n_features=5)
    _, ga, ka = pa.eval_grad(X32, True)
    _, gb, kb = pb.eval_grad(X32, True)
    assert np.array_equal(ka, kb)
    for t in np.nonzero(ka)[0]:
        np.testing.assert_array_equal(np.asarray(ga[t]), np.asarray(gb[t]))


def test_zz_write_turbo_report():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ulp_turbo_f32.json"), "w") as fh:
        json.dump(dict(what="DE_OPT_TURBO Float32 operators against float64 references (tests/test_gpu_turbo.py)", operators=REPORT),
                  fh, indent=1, sort_keys=True)
    assert len(REPORT) >= 5
