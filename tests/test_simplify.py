"""Host-side tree preprocessing: the known answers of test/test_simplification.jl:89-137 (the `repr` strings
compared modulo brackets and blanks, as the reference's `≈` on strings does), plus value preservation."""
import numpy as np

import dynamicexpressions_jl_amd as de
from oracle import oracle

OPS = de.OperatorEnum(binary_operators=("+", "-", "*", "/"), unary_operators=("cos", "sin"))
PLUS, MINUS = 1, 2


def same(a: str, b: str) -> bool:  # Base.:≈(::String, ::String) of the reference test: brackets and blanks stripped
    strip = lambda s: "".join(ch for ch in s if ch not in "() ")
    return strip(a) == strip(b)


def C(v):
    return de.Node(val=v)


def X(i):
    return de.Node(feature=i)


def test_simplify_tree_known_answers():
    t = de.Node(1, C(0.0))  # unary operator applied to constant => constant (:94-97)
    assert same(de.string_tree(t, OPS), "cos(0.0)")
    assert same(de.string_tree(de.simplify_tree(t, OPS), OPS), "1.0")
    t = de.Node(1, C(float("nan")))  # except when the result is a NaN (:99-102)
    assert same(de.string_tree(de.simplify_tree(t, OPS), OPS), "cos(nan)")
    t = de.Node(PLUS, de.Node(3, C(2.0), C(4.0)), X(1))  # a constant pair below a variable node folds, the rest stays
    assert same(de.string_tree(de.simplify_tree(t, OPS), OPS), "8.0 + x1")
    t = de.Node(4, C(1.0), C(0.0))  # 1/0 = Inf is not a valid value: unchanged (combine_children!, :123-125)
    assert same(de.string_tree(de.simplify_tree(t, OPS), OPS), "1.0 / 0.0")


def test_combine_operators_known_answers():
    # the same as above, but inside a binary tree (:104-108)
    t = de.Node(PLUS, de.Node(1, de.Node(PLUS, de.Node(PLUS, C(0.1), C(0.2)), C(0.2))), C(2.0))
    assert same(de.string_tree(t, OPS), "(cos((0.1 + 0.2) + 0.2) + 2.0)")
    assert same(de.string_tree(de.combine_operators(t, OPS), OPS), "(cos(0.4 + 0.1) + 2.0)")
    # left is constant (:110-113)
    t = de.Node(PLUS, C(0.5), de.Node(PLUS, C(0.2), X(1)))
    assert same(de.string_tree(de.combine_operators(t, OPS), OPS), "(x1 + 0.7)")
    # (const - (const - var)) => (var - const) (:115-118)
    t = de.Node(MINUS, C(0.5), de.Node(MINUS, C(0.2), X(1)))
    assert same(de.string_tree(t, OPS), "(0.5 - (0.2 - x1))")
    assert same(de.string_tree(de.combine_operators(t, OPS), OPS), "(x1 - -0.3)")
    # ((const - var) - const) => (const - var) (:120-123)
    t = de.Node(MINUS, de.Node(MINUS, C(0.5), X(1)), C(0.2))
    assert same(de.string_tree(de.combine_operators(t, OPS), OPS), "(0.3 - x1)")
    # (const - (var - const)) => (const - var) (:125-128)
    t = de.Node(MINUS, C(0.5), de.Node(MINUS, X(1), C(0.2)))
    assert same(de.string_tree(de.combine_operators(t, OPS), OPS), "(0.7 - x1)")
    # ((var - const) - const) => (var - const) (:130-133)
    t = de.Node(MINUS, de.Node(MINUS, X(1), C(0.2)), C(0.6))
    assert same(de.string_tree(de.combine_operators(t, OPS), OPS), "(x1 - 0.8)")


def test_rewrites_shorten_and_preserve_values():
    """Random trees: simplify_tree is value-identical for IEEE-exact operators (it performs the same scalar operations
    the evaluation would), combine_operators only re-associates constants (test_simplification.jl:78-86 checks by
    evaluation with a loose tolerance; here the oracle evaluates both)."""
    ops = de.OperatorEnum(binary_operators=("+", "-", "*", "/"), unary_operators=("neg", "square"))
    rng = de.synth.Xoshiro256ss(5)
    Xm = de.synth.random_X(3, 50, seed=2, dtype=np.float64)
    shorter = 0
    for i in range(200):
        t = de.synth.gen_random_tree_fixed_size(5 + i % 20, ops, 3, rng, np.float64)
        y0, ok0 = oracle.eval_tree_array(*de.flatten(t, ops, np.float64), Xm)
        n0 = de.count_nodes(t)
        t1 = de.simplify_tree(t.copy(), ops)
        y1, ok1 = oracle.eval_tree_array(*de.flatten(t1, ops, np.float64), Xm)
        assert de.count_nodes(t1) <= n0
        shorter += de.count_nodes(t1) < n0
        if ok0 and ok1:
            np.testing.assert_array_equal(y0, y1)
        t2 = de.combine_operators(t1.copy(), ops)
        assert de.count_nodes(t2) <= de.count_nodes(t1)
        y2, ok2 = oracle.eval_tree_array(*de.flatten(t2, ops, np.float64), Xm)
        if ok0 and ok2:
            np.testing.assert_allclose(y2, y0, rtol=1e-9, atol=1e-9 * np.abs(y0).max())
    assert shorter > 20
