"""Pins the CPU oracle against every known-answer the reference's own tests/docs hold for
the hot path (tests/golden/reference_known_answers.json).  CPU only."""
import os
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from helpers import case_X, case_options, case_tree, check_values, load_golden
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = load_golden()
MODE = {"variable": oracle.GRAD_VARIABLE, "constant": oracle.GRAD_CONSTANT, "both": oracle.GRAD_BOTH}


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_known_answer(case):
    tree, ops = case_tree(case)
    dt = np.dtype(case["dtype"])
    tape, consts = de.flatten(tree, ops, dt)
    X = case_X(case)
    exp = case["expect"]
    opts = case_options(case)
    if case["kind"] in ("eval", "flag"):
        y, ok = oracle.eval_tree_array(tape, consts, X, opts)
        assert ok == exp["ok"], f"{case['name']} ({case['cite']})"
        if ok and "y" in exp:
            check_values(y, case)
        # the per-element flavour of the validity test must agree on every reference case
        y2, ok2 = oracle.eval_tree_array(tape, consts, X, opts, elementwise=True)
        assert ok2 == exp["ok"]
    elif case["kind"] == "param":
        y, ok = oracle.eval_tree_array_parametric(
            tape, consts, X, np.asarray(exp["params"], dtype=dt), exp["classes"], 1, opts)
        assert ok == exp["ok"]
        check_values(y, case)
    elif case["kind"] == "grad":
        y, g, ok = oracle.eval_grad_tree_array(tape, consts, X, MODE[exp["mode"]])
        assert ok == exp["ok"], f"{case['name']} ({case['cite']})"
        if ok:
            if "y" in exp:
                check_values(y, case)
            if "grad" in exp:
                check_values(g, case, "grad")
            for row, vals in exp.get("grad_rows", {}).items():
                c2 = dict(case, expect=dict(exp, g_=vals))
                check_values(g[int(row)], c2, "g_")
            # eval_diff_tree_array per feature equals the gradient rows (test_derivatives.jl:84-92)
            if exp["mode"] == "variable":
                for f in range(X.shape[0]):
                    _, d, okd = oracle.eval_diff_tree_array(tape, consts, X, f)
                    assert okd
                    np.testing.assert_array_equal(d, g[f])
    else:
        raise AssertionError(case["kind"])


def test_constant_gradient_row_order():
    """index_constant_nodes: constants numbered depth-first, left to right
    (test/test_derivatives.jl:146-170, src/NodeUtils.jl:184-201)."""
    ops = de.OperatorEnum(binary_operators=("+", "*", "-", "/"), unary_operators=("cos",))
    x1 = de.Node(feature=1)
    # (c0 * x1) + cos(c1 - (x1 / c2))
    tree = de.Node(1, de.Node(2, de.Node(val=2.0), x1),
                   de.Node(1, de.Node(3, de.Node(val=3.0), de.Node(4, x1, de.Node(val=5.0)))))
    tape, consts = de.flatten(tree, ops, np.float64)
    assert list(consts) == [2.0, 3.0, 5.0]
    vals, refs = de.get_scalar_constants(tree)
    assert list(vals) == [2.0, 3.0, 5.0]
    X = np.asfortranarray(np.array([[0.7, 1.3]]))
    _, g, ok = oracle.eval_grad_tree_array(tape, consts, X, oracle.GRAD_CONSTANT)
    assert ok
    x = X[0]
    np.testing.assert_allclose(g[0], x, rtol=1e-15)
    np.testing.assert_allclose(g[1], -np.sin(3.0 - x / 5.0), rtol=1e-14)
    np.testing.assert_allclose(g[2], -np.sin(3.0 - x / 5.0) * (x / 25.0), rtol=1e-14)
    # :both mode: features first, then constants (src/EvaluateDerivative.jl:220)
    _, gb, ok = oracle.eval_grad_tree_array(tape, consts, X, oracle.GRAD_BOTH)
    np.testing.assert_array_equal(gb[1:], g)


def test_sum_overflow_quirk_is_the_only_flag_difference():
    """is_valid_array tests isfinite(sum(x)) (src/ValueInterface.jl:9): a finite array whose
    T-precision sum overflows is reported incomplete; the per-element test says complete."""
    ops = de.OperatorEnum(binary_operators=("+", "*"), unary_operators=("cos",))
    tree = de.Node(2, de.Node(feature=1), de.Node(val=1.0))
    tape, consts = de.flatten(tree, ops, np.float32)
    big = np.float32(3e38)
    X = np.asfortranarray(np.full((1, 16), big, dtype=np.float32))
    _, ok_ref = oracle.eval_tree_array(tape, consts, X)
    _, ok_el = oracle.eval_tree_array(tape, consts, X, elementwise=True)
    assert ok_ref is False and ok_el is True


def test_unfused_leaf_is_validity_tested_but_fused_leaf_is_not():
    """Which leaves get tested depends on the fused dispatch (src/Evaluate.jl:488-651):
    1/x1 is deg2_l0_r0 (operands untested) while cos(x1)+(1/x1) tests x1 through the
    unfused cos(x1) child."""
    ops = de.OperatorEnum(binary_operators=("+", "/"), unary_operators=("cos",))
    x1 = de.Node(feature=1)
    X = np.asfortranarray(np.array([[np.inf, 2.0]]))
    t1 = de.Node(2, de.Node(val=1.0), x1)
    tape, consts = de.flatten(t1, ops, np.float64)
    y, ok = oracle.eval_tree_array(tape, consts, X)
    assert ok and y[0] == 0.0
    t2 = de.Node(1, de.Node(1, x1), de.Node(2, de.Node(val=1.0), x1))
    tape, consts = de.flatten(t2, ops, np.float64)
    _, ok = oracle.eval_tree_array(tape, consts, X)
    assert not ok
    # with fusion disabled (>15 operators of the degree) every leaf is tested
    _, ok = oracle.eval_tree_array(*de.flatten(t1, ops, np.float64), X, oracle.OPT_EARLY_EXIT)
    assert not ok


def test_oracle_is_clean_under_address_and_ub_sanitizers():
    """`make -C oracle asan`: the oracle's entry points under ASan + UBSan (oracle/selftest.c: 4000 random tapes over every
    opcode through eval / grad / diff / parametric, malformed tapes, two known answers).  The oracle decides what every
    parity test expects, so a memory error in it would surface as a wrong expectation, not as a crash."""
    import shutil
    import subprocess
    if not shutil.which("gcc") and not shutil.which("cc"):
        pytest.skip("no C compiler")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle selftest OK" in r.stdout
