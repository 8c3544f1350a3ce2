"""The callers either side of the hot path and the reference's many-operator tests, on the GPU:

* callable sugar `tree(X, operators)` / `ex(X)`: NaN-fill when the evaluation is incomplete
  (src/EvaluationHelpers.jl:29-33), gradient sugar `ex'(X)` likewise (:56-62);
* "Test many operators" — test/test_evaluation.jl:293-350 and test/test_derivatives.jl:172-209: an OperatorEnum with
  more than 15 operators per degree switches the reference to its non-fused dispatch (src/Evaluate.jl:14,496,607); values
  and gradients must equal those of the small enum."""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from oracle import oracle

pytestmark = pytest.mark.gpu

BASIC = de.OperatorEnum(binary_operators=("+", "-", "*", "/"), unary_operators=("sin", "cos"))
# 100 copies of `(x, y) -> x + y` and `x -> x^2` behind the basic operators (test_evaluation.jl:300-330)
MANY = de.OperatorEnum(binary_operators=("+", "-", "*", "/") + ("+",) * 100, unary_operators=("sin", "cos") + ("square",) * 100)


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


def test_more_than_15_operators_known_answer(api):
    # tree = Node(op=1, children=(Node(op=num_ops/2, children=(3.0, x2)),))  = (3.0 + x2)^2   (test_evaluation.jl:313-320)
    ops = de.OperatorEnum(binary_operators=("+",) * 100, unary_operators=("square",) * 100)
    assert ops.fuse_flags() == (False, False)
    tree = de.Node(1, de.Node(50, de.Node(val=3.0), de.Node(feature=2)))
    X = np.asfortranarray(np.random.Generator(np.random.PCG64(3)).standard_normal((2, 10)))
    out = api.Expression(tree, ops)(X)
    np.testing.assert_array_equal(out, (3.0 + X[1]) ** 2)


def test_many_operator_enum_matches_basic_enum_values_and_gradients(api):
    rng = de.synth.Xoshiro256ss(2026)
    g = np.random.Generator(np.random.PCG64(4))
    n_nan = n_cmp = 0
    for _ in range(100):
        tree = de.synth.gen_random_tree_fixed_size(20, BASIC, 3, rng, np.float64)
        X = np.asfortranarray(g.standard_normal((3, 10)))
        basic, many = api.Expression(tree, BASIC)(X), api.Expression(tree, MANY)(X)
        # @test (all(isnan, basic_eval) && all(isnan, many_ops_eval)) || basic_eval ≈ many_ops_eval
        if np.all(np.isnan(basic)):
            assert np.all(np.isnan(many))
            n_nan += 1
        else:
            np.testing.assert_array_equal(basic, many)  # same device functions either way: equal bits, not just ≈
            n_cmp += 1
        gb, gm = api.Expression(tree, BASIC).grad(X), api.Expression(tree, MANY).grad(X)  # tree'(X, operators)
        assert gb.shape == gm.shape == (3, 10)
        if np.all(np.isnan(gb)):
            assert np.all(np.isnan(gm))
        else:
            np.testing.assert_array_equal(gb, gm)
        # and the oracle with the reference's option bits for each enum agrees on the flag
        tape, consts = de.flatten(tree, BASIC, np.float64)
        _, ok_b = oracle.eval_tree_array(tape, consts, X, 7)
        _, ok_m = oracle.eval_tree_array(tape, consts, X, 1)  # > 15 operators: no fused kernels
        assert ok_b == (not np.all(np.isnan(basic))) and ok_m == (not np.all(np.isnan(many)))
    assert n_cmp > 30


def test_callable_sugar_nan_fills_incomplete_evaluations(api):
    ops = de.OperatorEnum(binary_operators=("+", "*", "/", "-"), unary_operators=("cos", "exp", "safe_log"))
    x1, x2 = de.Node(feature=1), de.Node(feature=2)
    X = np.asfortranarray(np.array([[1.0, 2.0, 3.0], [0.5, 0.0, -1.0]], dtype=np.float32))
    # x1 / x2 hits 2/0 on the second sample: eval_tree_array reports complete = false, the sugar returns all NaN
    bad = de.Node(3, x1, x2)
    y, ok = api.eval_tree_array(bad, X, ops)
    assert not ok
    filled = api.Expression(bad, ops)(X)
    assert filled.shape == (3,) and np.all(np.isnan(filled))
    # a complete evaluation is returned as is
    good = de.Node(1, de.Node(2, x1, x2), de.Node(1, x1))  # x1*x2 + cos(x1)
    np.testing.assert_allclose(api.Expression(good, ops)(X), X[0] * X[1] + np.cos(X[0]), rtol=2e-6)
    # the gradient sugar NaN-fills the whole matrix (src/EvaluationHelpers.jl:56-62); safe_log(x2) is NaN for x2 <= 0
    gbad = api.Expression(de.Node(3, x2), ops).grad(X, variable=True)
    assert gbad.shape == (2, 3) and np.all(np.isnan(gbad))
    ggood = api.Expression(good, ops).grad(X, variable=True)
    np.testing.assert_allclose(ggood, np.stack([X[1] - np.sin(X[0]), X[0]]), rtol=3e-6, atol=1e-6)
    # early_exit = false: values come back with their non-finite samples in place (test_evaluation.jl:352-387)
    raw = api.Expression(bad, ops)(X, eval_context=api.EvalContext(early_exit=False))
    assert raw[0] == 2.0 and np.isinf(raw[1]) and raw[2] == -3.0
    # a feature beyond size(X, 1) is a usage error, not data (src/Expression.jl:401-409)
    with pytest.raises(ValueError):
        api.Expression(de.Node(feature=3), ops)(X)


def test_pullback_dX_times_dY(api):
    """EvalPullback (src/ChainRules.jl:56-77): dtree = sum_j dconst[:, j] * dY[j]  and  dX = dX .* dY'."""
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(40, seed=12)
    X = de.synth.random_X(5, 3000, seed=2)
    dY = np.random.Generator(np.random.PCG64(9)).standard_normal(3000).astype(np.float32)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    out, grads, ok = pop.eval_grad(X, True)
    dX, okp = pop.eval_pullback_dX(X, dY)
    assert np.array_equal(ok, okp)
    for t in range(len(trees)):
        if ok[t]:
            np.testing.assert_array_equal(dX[t], np.asarray(grads[t]) * dY[None, :])
        else:
            assert np.all(np.isnan(dX[t]))  # src/ChainRules.jl:62-64: NaN cotangent when incomplete
    pop.close()


def test_simplify_tree_folds_on_the_device_with_the_device_bits(api):
    """simplify_tree! (src/Simplify.jl:118-136) folds ANY operator whose children are constants.  The host twin evaluates
    the feature-free subtrees with the device library (one launch): the folded constants are the bits the device's own
    constant folding produces, relu(-2) is +0 (so 1/relu(-2) stays +Inf), and operators without a numpy table entry fold."""
    from dynamicexpressions_jl_amd import simplify
    ops = de.OperatorEnum(binary_operators=("+", "*", "/", "mod", "pow_abs2"), unary_operators=("cos", "relu", "gamma", "exp"))
    c = lambda v: de.Node(val=v)  # noqa: E731
    x1 = de.Node(feature=1)
    # x1 * (cos(2.5) + mod(7.5, 2.0))  +  pow_abs2(gamma(3.5), 0.3) / relu(-2.0)
    t = de.Node(1, de.Node(2, x1, de.Node(1, de.Node(1, c(2.5)), de.Node(4, c(7.5), c(2.0)))),
                de.Node(3, de.Node(5, de.Node(3, c(3.5)), c(0.3)), de.Node(2, c(-2.0))))
    X = np.asfortranarray(np.linspace(-1, 1, 64, dtype=np.float32)[None, :])
    before, _ = api.eval_tree_array(t.copy(), X, ops, eval_context=api.EvalContext(early_exit=False))
    s = simplify.simplify_tree(t.copy(), ops, np.float32, use_device=True)
    # cos(2.5) + mod(7.5, 2) and pow_abs2(gamma(3.5), 0.3) and relu(-2) are constants now; the division by +0 is not finite: kept
    assert de.count_nodes(s) == 7 and de.string_tree(s, ops).count("relu") == 0
    relu_leaf = s.children[1].children[1]
    assert relu_leaf.degree == 0 and relu_leaf.val == 0.0 and not np.signbit(relu_leaf.val)
    after, _ = api.eval_tree_array(s, X, ops, eval_context=api.EvalContext(early_exit=False))
    assert np.all(np.isposinf(after)) and np.all(np.isposinf(before))
    # a tree that stays finite: the simplified tree gives the bits of the original (device-side folding == host fold)
    t2 = de.Node(1, de.Node(2, x1, de.Node(1, de.Node(1, c(2.5)), de.Node(4, c(7.5), c(2.0)))), de.Node(4, de.Node(3, c(3.5)), c(0.3)))
    b2, ok2 = api.eval_tree_array(t2.copy(), X, ops)
    s2 = simplify.simplify_tree(t2.copy(), ops, np.float32)
    a2, oka = api.eval_tree_array(s2, X, ops)
    assert ok2 and oka and de.count_nodes(s2) == 5
    np.testing.assert_array_equal(a2, b2)
