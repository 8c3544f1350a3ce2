"""Accuracy of the hand-written Float32 hot operators (csrc/de_device_ops.h: cos, sin, exp) measured
in ulps against float64 references through the public API (one-operator trees), plus the
Float64 operators against the oracle's libm.  north_star tolerance: 1e-5 relative for Float32;
these operators are held to <= 2 ulp (2.4e-7)."""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


def ulps(got32, want64):
    want32 = want64.astype(np.float32)
    u = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got32.astype(np.float64) - want64) / u


def inputs_trig():
    g = np.random.Generator(np.random.PCG64(7))
    parts = [g.uniform(-4, 4, 400_000), g.uniform(-100, 100, 400_000), g.uniform(-1e4, 1e4, 400_000),
             g.uniform(-1e5, 1e5, 800_000), g.standard_normal(200_000) * 1e-3,
             np.array([0.0, -0.0, 1e-30, -1e-30, 1e-45, 99999.99, -99999.99, 1e5, -1e5])]
    # floats next to the zeros of cos and sin: multiples of pi/2, +-2 ulp
    parts.append(g.uniform(-2.4e-4, 2.4e-4, 5000))
    k = np.arange(1, 60000, 7, dtype=np.float64)
    z = (k * (np.pi / 2)).astype(np.float32)
    parts += [z, np.nextafter(z, np.float32(np.inf)), np.nextafter(z, np.float32(-np.inf)), -z]
    return np.concatenate([np.asarray(p, dtype=np.float32) for p in parts])


@pytest.mark.parametrize("name,f64", [("cos", np.cos), ("sin", np.sin)])
def test_fast_trig_within_two_ulp_up_to_1e5(api, name, f64):
    ops = de.OperatorEnum(binary_operators=("+",), unary_operators=(name,))
    x = inputs_trig()
    X = np.asfortranarray(x[None, :])
    tree = de.Node(1, de.Node(feature=1))
    out, ok = api.eval_tree_array(tree, X, ops)
    assert ok
    e = ulps(out, f64(x.astype(np.float64)))
    assert e.max() <= 2.0, f"{name}: max {e.max():.3f} ulp at x = {x[np.argmax(e)]!r}"
    assert e.mean() < 0.5
    # where the correctly rounded result is exactly +-1 or the argument itself, so is ours
    # (up to the last 1 % of that band: the device measures the distance to pi/2 in float32)
    true64 = f64(x.astype(np.float64))
    ext = 1.0 - np.abs(true64) < 0.98 * 2.0 ** -25
    assert ext.sum() > 100
    np.testing.assert_array_equal(out[ext], np.sign(true64[ext]).astype(np.float32))
    # the other lowered forms of the same operator use the same code: acc-source after a load, and
    # the fused constant form  cos(x * 1)
    tree2 = de.Node(1, de.Node(2 if False else 1, de.Node(feature=1), de.Node(val=0.0)))  # cos(x + 0)
    out2, ok2 = api.eval_tree_array(tree2, X, ops)
    assert ok2
    np.testing.assert_array_equal(out2, out)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sin_keeps_the_sign_of_zero(api, dtype):
    """sin(-0) = -0 (IEEE; Julia): visible through 1 / sin(neg(relu(x))) = -Inf (found by tests/fuzz/fuzz_gpu.py 102: the
    Float32 fast path added a +0 correction term to the -0 argument).  Eval (packed and turbo handlers), the value output of
    the gradient kernels (their sincos), and a wave that mixes zeros with ordinary arguments."""
    ops = de.OperatorEnum(binary_operators=("/",), unary_operators=("sin", "cos"))
    x = np.array([-0.0, 0.0, 1.5, -1.5, -0.0, 1e-45, -1e-45, 3.0] * 40, dtype=dtype)
    X = np.asfortranarray(x[None, :])
    want = np.sin(x)
    tree = de.Node(1, de.Node(feature=1))
    for ec in (api.EvalContext(), api.EvalContext(turbo=True), api.EvalContext(early_exit=False)):
        out, ok = api.eval_tree_array(tree, X, ops, eval_context=ec)
        assert ok
        np.testing.assert_array_equal(np.signbit(out), np.signbit(want))
        np.testing.assert_array_equal(out[x == 0], x[x == 0])
    # 1 / sin(-0) = -Inf, 1 / sin(+0) = +Inf
    inv = de.Node(1, de.Node(val=1.0), tree)
    o, _ = api.eval_tree_array(inv, X, ops, eval_context=api.EvalContext(early_exit=False))
    assert np.all(np.isneginf(o[(x == 0) & np.signbit(x)])) and np.all(np.isposinf(o[(x == 0) & ~np.signbit(x)]))
    og, grad, okg = api.eval_grad_tree_array(tree, X, ops, variable=True)
    assert okg
    np.testing.assert_array_equal(np.signbit(og), np.signbit(want))
    np.testing.assert_array_equal(grad[0][x == 0], np.ones(int((x == 0).sum()), dtype=dtype))  # cos(+-0) = 1


@pytest.mark.parametrize("name,f64", [("cos", np.cos), ("sin", np.sin)])
def test_trig_beyond_fast_range_uses_full_range_reduction(api, name, f64):
    ops = de.OperatorEnum(binary_operators=("+",), unary_operators=(name,))
    g = np.random.Generator(np.random.PCG64(8))
    x = np.concatenate([g.uniform(1e5, 1e9, 50_000), -g.uniform(1e5, 1e9, 50_000),
                        10.0 ** g.uniform(9, 38, 50_000), [3.0e38, -3.0e38, 100000.01]]).astype(np.float32)
    # mixed waves: fast and slow elements side by side
    x[::3] = g.uniform(-50, 50, x[::3].size).astype(np.float32)
    X = np.asfortranarray(x[None, :])
    out, ok = api.eval_tree_array(de.Node(1, de.Node(feature=1)), X, ops)
    assert ok
    # float64 libm reduces exactly for |x| < 2^1024: a valid reference for float32 inputs
    e = ulps(out, f64(x.astype(np.float64)))
    assert e.max() <= 2.0, f"{name}: max {e.max():.3f} ulp at x = {x[np.argmax(e)]!r}"


def test_fast_exp_within_two_ulp_and_edges(api):
    ops = de.OperatorEnum(binary_operators=("+",), unary_operators=("exp",))
    g = np.random.Generator(np.random.PCG64(9))
    x = np.concatenate([g.uniform(-104, 89, 1_000_000), g.uniform(-2, 2, 500_000), g.standard_normal(100_000) * 1e-4,
                        [0.0, -0.0, 88.72283, 88.7229, -87.3365, -103.97, -103.98, 1e-45, -1e-45]]).astype(np.float32)
    X = np.asfortranarray(x[None, :])
    # full_eval: the tree is incomplete (overflowing samples) and the test reads its values — without it a tree whose flag is
    # already 0 is not evaluated by the workgroups that start later (early exit, DE_OPT_EARLY_EXIT in include/de_hip.h)
    out, ok = api.eval_tree_array(de.Node(1, de.Node(feature=1)), X, ops, eval_context=api.EvalContext(full_eval=True))
    assert not ok
    want = np.exp(x.astype(np.float64))
    fin = want < np.finfo(np.float32).max
    assert np.all(np.isposinf(out[~fin]))
    assert ok == bool(fin.all())
    normal = fin & (want >= np.finfo(np.float32).tiny)
    e = ulps(out[normal], want[normal])
    assert e.max() <= 2.0, f"exp: max {e.max():.3f} ulp"
    sub = fin & ~normal  # gradual underflow: within one subnormal step
    assert np.all(np.abs(out[sub].astype(np.float64) - want[sub]) <= 1.5 * 1.4012984643e-45)
    # far tails and specials (no early exit: values, not flags)
    xs = np.array([-200.0, -1e30, 200.0, 1e30, np.inf, -np.inf, np.nan], dtype=np.float32)
    o, _ = api.eval_tree_array(de.Node(1, de.Node(feature=1)), np.asfortranarray(xs[None, :]), ops,
                               eval_context=api.EvalContext(early_exit=False))
    assert o[0] == 0 and o[1] == 0 and np.isposinf(o[2]) and np.isposinf(o[3]) and np.isposinf(o[4]) and o[5] == 0
    assert np.isnan(o[6])


def test_division_fast_path_is_bit_identical_to_ieee(api):
    """The packed Newton division (csrc/de_kernels.hip div_apply) must equal the correctly rounded
    quotient bit for bit: in-range operands (fast path), and waves that mix in zeros, Inf, NaN,
    denormals and huge/tiny values (generic path), for all lowered forms x/y, c/x, x/c."""
    ops = de.OperatorEnum(binary_operators=("/",))
    g = np.random.Generator(np.random.PCG64(11))
    N = 1 << 20
    mags = [(-1, 1), (-12, 12), (-38, 38), (-45, 38)]
    for lo, hi in mags:
        a = (10.0 ** g.uniform(lo, hi, N) * g.choice([-1, 1], N)).astype(np.float32)
        b = (10.0 ** g.uniform(lo, hi, N) * g.choice([-1, 1], N)).astype(np.float32)
        if lo < -20:  # sprinkle specials into some waves
            idx = g.integers(0, N, 2000)
            a[idx[:500]] = 0.0; a[idx[500:700]] = -0.0; b[idx[700:900]] = 0.0
            a[idx[900:1100]] = np.inf; b[idx[1100:1300]] = -np.inf; a[idx[1300:1500]] = np.nan
            b[idx[1500:1700]] = np.float32(1e-42); a[idx[1700:2000]] = np.float32(-3e-41)
        X = np.asfortranarray(np.stack([a, b]))
        ec = api.EvalContext(early_exit=False)
        with np.errstate(all="ignore"):
            want = a / b
            forms = [(de.Node(1, de.Node(feature=1), de.Node(feature=2)), want),
                     (de.Node(1, de.Node(val=1.5), de.Node(feature=2)), np.float32(1.5) / b),
                     (de.Node(1, de.Node(feature=1), de.Node(val=-3.0)), a / np.float32(-3.0)),
                     (de.Node(1, de.Node(1, de.Node(feature=1), de.Node(feature=2)), de.Node(feature=2)), (a / b) / b)]
        for tree, w in forms:
            out, _ = api.eval_tree_array(tree, X, ops, eval_context=ec)
            nan = np.isnan(w)
            assert np.array_equal(np.isnan(out), nan)
            np.testing.assert_array_equal(out[~nan].view(np.uint32), w[~nan].view(np.uint32))


def test_division_with_a_constant_operand_is_bit_identical_to_ieee(api):
    """x / c and c / x take their own fast path (csrc/de_kernels.hip div_const: scalar range test of the constant, ONE refined
    reciprocal per wavefront for x / c): many constants — random significands, exponents inside, at the edges of and outside
    the fast range [2^-39, 2^40), powers of two, denormals, zero, Inf — in every lowered form (first instruction of a tree,
    mid-tree on the accumulator, end of tree), samples in range and with specials mixed in: the IEEE quotient bit for bit."""
    ops = de.OperatorEnum(binary_operators=("/", "*"))
    g = np.random.Generator(np.random.PCG64(12))
    N = 1 << 14
    consts = list((g.uniform(1, 2, 96) * 2.0 ** g.integers(-45, 45, 96) * g.choice([-1, 1], 96)).astype(np.float32))
    consts += [np.float32(v) for v in (2.0 ** -40, 2.0 ** -39, np.nextafter(np.float32(2.0 ** -39), np.float32(0)), 2.0 ** 40,
                                       np.nextafter(np.float32(2.0 ** 40), np.float32(0)), 1.0, -1.0, 3.0, 1e-42, 0.0, -0.0, np.inf,
                                       1.0000001, 1.9999999, 3.4e38, 1.2e-38)]
    x1, x2 = de.Node(feature=1), de.Node(feature=2)
    trees, want_fn = [], []
    for c in consts:
        cn = de.Node(val=float(c))
        trees += [de.Node(1, x1, cn), de.Node(1, cn, x1), de.Node(1, de.Node(2, x1, x2), cn), de.Node(1, cn, de.Node(2, x1, x2)),
                  de.Node(2, de.Node(1, x1, cn), x2), de.Node(2, de.Node(1, cn, de.Node(2, x1, x2)), x2)]
        want_fn += [lambda a, b, c=c: a / c, lambda a, b, c=c: c / a, lambda a, b, c=c: (a * b) / c, lambda a, b, c=c: c / (a * b),
                    lambda a, b, c=c: (a / c) * b, lambda a, b, c=c: (c / (a * b)) * b]
    for special in (False, True):
        a = (2.0 ** g.uniform(-18, 18, N) * g.choice([-1, 1], N)).astype(np.float32)
        b = (2.0 ** g.uniform(-18, 18, N) * g.choice([-1, 1], N)).astype(np.float32)
        if special:
            idx = g.integers(0, N, 600)
            a[idx[:100]] = 0.0; a[idx[100:200]] = np.inf; a[idx[200:300]] = np.nan; a[idx[300:400]] = np.float32(1e-42)
            a[idx[400:500]] = np.float32(3e38); b[idx[500:600]] = np.float32(2e-30)
        X = np.asfortranarray(np.stack([a, b]))
        pop = api.Population(trees, ops, np.float32, n_features=2, eval_context=api.EvalContext(early_exit=False))
        out, _ = pop.eval(X)
        pop.close()
        with np.errstate(all="ignore"):
            for t, fn in enumerate(want_fn):
                w = fn(a, b).astype(np.float32)
                nan = np.isnan(w)
                assert np.array_equal(np.isnan(out[t]), nan), (t, consts[t // 6])
                np.testing.assert_array_equal(out[t][~nan].view(np.uint32), w[~nan].view(np.uint32), err_msg=f"form {t % 6}, c = {consts[t // 6]!r}")


def test_exp_and_powers_round_into_the_last_subnormal_bits(api):
    """Results c * 2^-149: Julia (and glibc, the oracle) round them to the nearest subnormal, 2^-149 for c in (1/2, 1);
    OCML's expf/powf return 0 there, which `safe_log(x ^ y)` turns into NaN against a finite value (found by
    tests/fuzz/fuzz_gpu.py) — Float32 `exp`, `^` and `pow_abs2` go through the ldexp-based exp instead."""
    from oracle import oracle
    ops = de.OperatorEnum(binary_operators=("^", "pow_abs2", "*"), unary_operators=("exp", "safe_log"))
    C = np.array([0.4, 0.6, 0.9, 1.4, 1.6, 2.4, 2.6, 3.6, 100.3])
    a = (np.log(C) - 149 * np.log(2.0)).astype(np.float32)
    X = np.asfortranarray(np.stack([a, np.full_like(a, 0.5), np.full_like(a, -0.5)]))
    want = np.array([0, 1, 1, 1, 2, 2, 3, 4, 100], dtype=np.float64) * 2.0 ** -149
    x1, x2, x3 = (de.Node(feature=i) for i in (1, 2, 3))
    expo = de.Node(3, x1, de.Node(val=-1.4426950408889634))  # 0.5 ^ expo = exp(x1)
    trees = {
        "exp": de.Node(1, x1),
        "pow": de.Node(1, x2, expo),
        "pow_abs2": de.Node(2, x2, expo),
        "pow_abs2 of a negative base": de.Node(2, x3, expo),
    }
    for name, tree in trees.items():
        for ec in (api.EvalContext(), api.EvalContext(use_fused=False), api.EvalContext(early_exit=False)):
            y, ok = api.eval_tree_array(tree, X, ops, eval_context=ec)
            tape, consts = de.flatten(tree, ops, np.float32)
            yo, oko = oracle.eval_tree_array(tape, consts, X, ec.option_bits(ops), elementwise=True)
            assert ok and oko, name
            np.testing.assert_array_equal(y, yo, err_msg=name)
            np.testing.assert_array_equal(y.astype(np.float64), want, err_msg=name)
    # the consumer that made it visible: log of the smallest subnormal is finite
    y, ok = api.eval_tree_array(de.Node(2, trees["pow"]), X[:, 1:], ops)
    assert ok and np.all(np.isfinite(y))
