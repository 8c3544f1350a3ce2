#!/usr/bin/env python3
"""Transcribes the reference's known-answer / golden cases for the hot path into
``reference_known_answers.json`` (SURVEY.md §8c lists them).

Every case carries the reference file:line it comes from.  A case is DATA: a tree (as an
S-expression over Julia operator names), the operator lists, the input matrix, and the
expected output / flag / gradient the reference's own test (or doc) asserts.  Where the
reference asserts equality with a closed-form Julia expression (e.g. ``cos(2.1)+sin(1.0)``)
the expected numbers are produced here by evaluating that closed form with Python's IEEE
double math (``math``/numpy) — which is also what pins them independently of the oracle.
Where the reference draws X from Julia's MersenneTwister (not reproducible without Julia)
X is redrawn from numpy's PCG64 and the closed form re-evaluated: the reference test asserts
the closed form, not the particular stream.

Run:  python tests/golden/make_reference_known_answers.py
"""
import json
import math
import os

import numpy as np

INF, NAN = float("inf"), float("nan")
X_ = lambda i: ["x", i]  # noqa: E731
P_ = lambda i: ["p", i]  # noqa: E731

cases = []


def case(name, cite, tree, X, unary=(), binary=(), ternary=(), dtype="float64", kind="eval",
         options=None, **expect):
    X = np.asarray(X, dtype=np.float64)
    cases.append(dict(name=name, cite=cite, kind=kind, dtype=dtype, unary=list(unary),
                      binary=list(binary), ternary=list(ternary), tree=tree,
                      X=X.tolist(), options=options or {}, expect=expect))


rng = np.random.Generator(np.random.PCG64(20260927))
pi = math.pi

# ---------------------------------------------------------------- README / config C1
X = rng.standard_normal((2, 100))
case("readme_x1_cos_x2_minus_3.2", "README.md:30-39 (BASELINE config 1)",
     ["*", X_(1), ["cos", ["-", X_(2), 3.2]]], X, unary=["cos"], binary=["+", "-", "*"],
     y=(X[0] * np.cos(X[1] - 3.2)).tolist(), ok=True, rtol=1e-14, atol=0)

# ---------------------------------------------------------------- test_evaluation.jl:9-92
# 24 closures, one per fused-kernel branch; operators (+,*,/,-),(cos,sin); X 3x100;
# reference tolerance: abs(err)/N < 1e-6 with N=100.
X3 = rng.standard_normal((3, 100))
x1, x2, x3 = X3
B4, U2 = ["+", "*", "/", "-"], ["cos", "sin"]
closures = [
    ("deg2_l0_r0:x1*x2", ["*", X_(1), X_(2)], x1 * x2),
    ("deg2_l0_r0:x1*3", ["*", X_(1), 3.0], x1 * 3.0),
    ("deg2_l0_r0:3*x2", ["*", 3.0, X_(2)], 3.0 * x2),
    ("deg2_l0_r0:3*6", ["*", 3.0, 6.0], np.full(100, 18.0)),
    ("deg2_l0:x1*sin(x2)", ["*", X_(1), ["sin", X_(2)]], x1 * np.sin(x2)),
    ("deg2_l0:3*sin(x2)", ["*", 3.0, ["sin", X_(2)]], 3.0 * np.sin(x2)),
    ("deg2_r0:sin(x1)*x2", ["*", ["sin", X_(1)], X_(2)], np.sin(x1) * x2),
    ("deg2_r0:sin(x1)*3", ["*", ["sin", X_(1)], 3.0], np.sin(x1) * 3.0),
    ("branch0:(x1*x2)+x3", ["+", ["*", X_(1), X_(2)], X_(3)], (x1 * x2) + x3),
    ("branch0:(3*x2)+x3", ["+", ["*", 3.0, X_(2)], X_(3)], (3.0 * x2) + x3),
    ("branch0:(x1*3)+x3", ["+", ["*", X_(1), 3.0], X_(3)], (x1 * 3.0) + x3),
    ("branch0:(x1*x2)+3", ["+", ["*", X_(1), X_(2)], 3.0], (x1 * x2) + 3.0),
    ("branch0:x1+(x2*x3)", ["+", X_(1), ["*", X_(2), X_(3)]], x1 + (x2 * x3)),
    ("branch0:3+(x2*x3)", ["+", 3.0, ["*", X_(2), X_(3)]], 3.0 + (x2 * x3)),
    ("branch0:x1+(3*x3)", ["+", X_(1), ["*", 3.0, X_(3)]], x1 + (3.0 * x3)),
    ("branch0:x1+(x2*3)", ["+", X_(1), ["*", X_(2), 3.0]], x1 + (x2 * 3.0)),
    ("deg1_l2:cos(x1*x2)", ["cos", ["*", X_(1), X_(2)]], np.cos(x1 * x2)),
    ("deg1_l2:cos(x1*3)", ["cos", ["*", X_(1), 3.0]], np.cos(x1 * 3.0)),
    ("deg1_l2:cos(3*x2)", ["cos", ["*", 3.0, X_(2)]], np.cos(3.0 * x2)),
    ("deg1_l2:cos(3*-0.5)", ["cos", ["*", 3.0, -0.5]], np.full(100, math.cos(3.0 * -0.5))),
    ("deg1_l1:cos(sin(x1))", ["cos", ["sin", X_(1)]], np.cos(np.sin(x1))),
    ("deg1_l1:cos(sin(3))", ["cos", ["sin", 3.0]], np.full(100, math.cos(math.sin(3.0)))),
    ("else:(sin(cos(sin(cos(x1)*x3)*3)*-0.5)+2)*5",
     ["*", ["+", ["sin", ["*", ["cos", ["*", ["sin", ["*", ["cos", X_(1)], X_(3)]], 3.0]], -0.5]], 2.0], 5.0],
     (np.sin(np.cos(np.sin(np.cos(x1) * x3) * 3.0) * -0.5) + 2.0) * 5.0),
]
for nm, tr, y in closures:
    for dt in ("float64", "float32"):
        case(f"test_evaluation:{nm}:{dt}", "test/test_evaluation.jl:9-92", tr, X3, unary=U2, binary=B4,
             dtype=dt, y=np.asarray(y).tolist(), ok=True, rtol=0, atol=1e-6 * 100 if dt == "float64" else 1e-4)

# ---------------------------------------------------------------- test_evaluation.jl:137-197
# Fused branch preserves early exit.  finite_min -> min (errors on non-finite input in the
# reference, i.e. it must never be CALLED with one; here only the flag is asserted).
for i, (tr, Xc) in enumerate([
    (["min", ["/", X_(1), X_(2)], X_(3)], [[1.0], [0.0], [2.0]]),
    (["min", X_(1), ["/", X_(2), X_(3)]], [[2.0], [1.0], [0.0]]),
    (["min", ["/", X_(1), X_(2)], X_(3)], [[1.0], [1.0], [INF]]),
    (["min", X_(1), ["/", X_(2), X_(3)]], [[INF], [1.0], [1.0]]),
]):
    case(f"fused_branch_early_exit:{i}", "test/test_evaluation.jl:151-163", tr, Xc, binary=["min", "/"],
         kind="flag", ok=False)
for i, tr in enumerate([
    ["min", ["/", INF, X_(2)], X_(3)],
    ["min", ["/", X_(1), INF], X_(3)],
    ["min", ["/", X_(1), X_(2)], INF],
]):
    case(f"fused_branch_inf_const:{i}", "test/test_evaluation.jl:165-178", tr, np.ones((3, 1)),
         binary=["min", "/"], kind="flag", ok=False)
Xr = [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]]
case("fused_branch_right_no_early_exit", "test/test_evaluation.jl:180-196",
     ["+", X_(1), ["*", X_(2), X_(3)]], Xr, binary=["+", "*"], options={"early_exit": False},
     y=[1.0 + 3.0 * 5.0, 2.0 + 4.0 * 6.0], ok=True, rtol=0, atol=0)

# ---------------------------------------------------------------- test_evaluation.jl:199-247
case("branch:cos(cos(3))", "test/test_evaluation.jl:216-222", ["cos", ["cos", 3.0]], [[0.0]],
     unary=["cos", "sin"], binary=B4, y=[math.cos(math.cos(3.0))], ok=True, rtol=1e-15, atol=0)
case("branch:3+4", "test/test_evaluation.jl:224-230", ["+", 3.0, 4.0], [[0.0]], unary=["cos", "sin"],
     binary=B4, y=[7.0], ok=True, rtol=0, atol=0)
case("branch:cos(3+4)", "test/test_evaluation.jl:232-238", ["cos", ["+", 3.0, 4.0]], [[0.0]],
     unary=["cos", "sin"], binary=B4, y=[math.cos(7.0)], ok=True, rtol=1e-15, atol=0)
Xn = rng.standard_normal((3, 10))
case("nan_presence:sin(x1/0)", "test/test_evaluation.jl:240-246", ["sin", ["/", X_(1), 0.0]], Xn,
     unary=["cos", "sin"], binary=["+", "-", "*", "/"], kind="flag", ok=False)

# ---------------------------------------------------------------- test_evaluation.jl:352-387
for dt, fmax in (("float32", float(np.finfo(np.float32).max)), ("float64", float(np.finfo(np.float64).max))):
    quad = ["/", ["-", ["neg", X_(2)], ["sqrt", ["-", ["^", X_(2), 2.0], ["*", ["*", 4.0, X_(1)], X_(3)]]]],
            ["*", 2.0, X_(3)]]
    Xq = [[-1.0, -1.0], [1.0, fmax], [1.0, 1.0]]
    case(f"disable_early_exit:quadratic_root:early:{dt}", "test/test_evaluation.jl:367-383", quad, Xq,
         unary=["neg", "sqrt"], binary=["-", "*", "/", "^"], dtype=dt, kind="flag", ok=False)
    case(f"disable_early_exit:quadratic_root:noexit:{dt}", "test/test_evaluation.jl:384-386", quad, Xq,
         unary=["neg", "sqrt"], binary=["-", "*", "/", "^"], dtype=dt, options={"early_exit": False},
         y=[-1.618033988749895, NAN], y_nonfinite_idx=[1], ok=True,
         rtol=1e-6 if dt == "float32" else 1e-14, atol=0)

# ---------------------------------------------------------------- test_nan_detection.jl:6-33
for dt in ("float32", "float64"):
    X100 = np.ones((1, 10)) * 100
    case(f"nan_detection:exp4:{dt}", "test/test_nan_detection.jl:8-13",
         ["exp", ["exp", ["exp", ["exp", ["+", X_(1), 1.0]]]]], X100, unary=["cos", "sin", "exp"],
         binary=["+", "*", "/", "-"], dtype=dt, kind="flag", ok=False)
    case(f"nan_detection:div0:{dt}", "test/test_nan_detection.jl:15-19", ["cos", ["/", X_(1), 0.0]], X100,
         unary=["cos", "sin", "exp"], binary=["+", "*", "/", "-"], dtype=dt, kind="flag", ok=False)
    case(f"nan_detection:inf_const:{dt}", "test/test_nan_detection.jl:21-25", ["cos", ["+", X_(1), INF]], X100,
         unary=["cos", "sin", "exp"], binary=["+", "*", "/", "-"], dtype=dt, kind="flag", ok=False)
    case(f"nan_detection:nan_const:{dt}", "test/test_nan_detection.jl:26-29", ["cos", ["+", X_(1), NAN]], X100,
         unary=["cos", "sin", "exp"], binary=["+", "*", "/", "-"], dtype=dt, kind="flag", ok=False)

# ---------------------------------------------------------------- test_buffered_evaluation.jl:147-159
case("buffered:1/0", "test/test_buffered_evaluation.jl:147-159", ["/", 1.0, 0.0], rng.random((2, 10)),
     unary=["sin"], binary=["+", "/", "*"], kind="flag", ok=False)

# ---------------------------------------------------------------- test_expressions.jl:472-492
ops_new = dict(unary=["sin", "cos"], binary=["+", "-", "*", "/", "max", "min", "rem"])
# X = [1.0 2.0; 3.0 1.0]' in Julia => feature rows x1 = [1, 3], x2 = [2, 1]
case("new_binary:max", "test/test_expressions.jl:473-476", ["max", X_(1), X_(2)], [[1.0, 3.0], [2.0, 1.0]],
     **ops_new, y=[2.0, 3.0], ok=True, rtol=0, atol=0)
case("new_binary:min", "test/test_expressions.jl:478-479", ["min", X_(1), X_(2)], [[1.0, 3.0], [2.0, 1.0]],
     **ops_new, y=[1.0, 1.0], ok=True, rtol=0, atol=0)
# X = [5.0 7.0; 3.0 2.0]' => x1 = [5, 3], x2 = [7, 2]
case("new_binary:rem", "test/test_expressions.jl:481-484", ["rem", X_(1), X_(2)], [[5.0, 3.0], [7.0, 2.0]],
     **ops_new, y=[5.0, 1.0], ok=True, rtol=0, atol=0)
case("new_binary:rem_const", "test/test_expressions.jl:490-493", ["rem", X_(1), 2.0], [[5.0, 7.0]],
     **ops_new, y=[1.0, 1.0], ok=True, rtol=0, atol=0)

# ---------------------------------------------------------------- test_expression_math.jl
f_tree = ["-", ["*", X_(1), X_(1)], ["cos", ["+", ["*", 2.5, X_(2)], -0.5]]]
g_tree = ["exp", ["neg", ["*", X_(2), X_(2)]]]
em = dict(unary=["neg", "cos", "exp"], binary=["+", "-", "*", "/"], dtype="float32")
f32 = np.float32


def f_closed(x, y):
    x, y = f32(x), f32(y)
    return float(f32(x * x) - f32(math.cos(f32(f32(f32(2.5) * y) + f32(-0.5)))))


for nm, xy in (("zero", (0.0, 0.0)), ("large", (1e5, 1e5)), ("small", (1e-5, 1e-5)), ("neg", (-1.0, -1.0))):
    case(f"expression_math:f:{nm}", "test/test_expression_math.jl:27-42", f_tree, [[xy[0]], [xy[1]]], **em,
         y=[f_closed(*xy)], ok=True, rtol=2e-4, atol=1e-6)
for nm, yv, ex in (("zero", 0.0, 1.0), ("large", 1e5, 0.0), ("small", 1e-5, math.exp(-1e-10)), ("neg", -1.0, math.exp(-1.0))):
    case(f"expression_math:g:{nm}", "test/test_expression_math.jl:28-42", g_tree, [[0.0], [yv]], **em,
         y=[ex], ok=True, rtol=2e-4, atol=1e-7)
case("expression_math:f:nan_input", "test/test_expression_math.jl:45", f_tree, [[NAN], [1.0]], **em,
     kind="flag", ok=False)
# (test_expression_math.jl:145-178 defines its own safe_sqrt(x<0)=0 closure: not a table operator, skipped)

# ---------------------------------------------------------------- test_initial_errors.jl:87, Parse.jl:40-58
case("initial_errors:cos(2.1*x1)+sin(x2)", "test/test_initial_errors.jl:84-87 (Bumper path)",
     ["+", ["cos", ["*", X_(1), 2.1]], ["sin", X_(2)]], np.ones((2, 10)), unary=["cos", "sin"],
     binary=["+", "-", "*", "/"], y=[math.cos(2.1) + math.sin(1.0)] * 10, ok=True, rtol=1e-15, atol=0)
case("initial_errors:bumper_checks", "test/test_initial_errors.jl:84-87",
     ["+", ["cos", ["*", X_(1), 2.1]], ["sin", X_(2)]], np.ones((2, 10)), unary=["cos", "sin"],
     binary=["+", "-", "*", "/"], options={"bumper": True}, y=[math.cos(2.1) + math.sin(1.0)] * 10,
     ok=True, rtol=1e-15, atol=0)
# my_custom_op(x, y) = x + y^3 ; ex = my_custom_op(x, sin(y) + 0.3) at ones(2,1) -> 2.487286478935302
case("parse_docstring:x+(sin(y)+0.3)^3", "src/Parse.jl:40-58",
     ["+", X_(1), ["cube", ["+", ["sin", X_(2)], 0.3]]], np.ones((2, 1)), unary=["sin", "cube"],
     binary=["+", "-", "*"], y=[2.487286478935302], ok=True, rtol=2e-16 * 4, atol=0)

# ---------------------------------------------------------------- gradients
Xd = [[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]
case("docs_eval:grad 0.5*x1+cos(x2-0.2)", "docs/src/eval.md:166-217",
     ["+", ["*", 0.5, X_(1)], ["cos", ["-", X_(2), 0.2]]], Xd, unary=["cos", "sin"],
     binary=["+", "-", "*", "/"], kind="grad", mode="variable",
     grad=[[0.5, 0.5, 0.5], [0.611858, 0.996165, 0.464602]], ok=True, rtol=0, atol=5e-7)
case("zygote_wrapper:d(x^2) at 2", "test/test_zygote_gradient_wrapper.jl:23-28", ["square", X_(1)], [[2.0]],
     unary=["square"], binary=["*"], kind="grad", mode="variable", grad=[[4.0]], ok=True, rtol=0, atol=0)
case("zygote_wrapper:d(x*y) at (2,3)", "test/test_zygote_gradient_wrapper.jl:30-32", ["*", X_(1), X_(2)],
     [[2.0], [3.0]], unary=["square"], binary=["*"], kind="grad", mode="variable", grad=[[3.0], [2.0]],
     ok=True, rtol=0, atol=0)
case("initial_errors:dsin(1)=cos(1)", "test/test_initial_errors.jl:75", ["sin", X_(1)], [[1.0]], unary=["sin"],
     binary=["+"], kind="grad", mode="variable", grad=[[math.cos(1.0)]], ok=True, rtol=1e-15, atol=0)
Xe = rng.random((2, 10)) + 1
inner = 2.0 * Xe[0] + np.exp(Xe[1] + 5.0)
case("expressions:sin(2x1+exp(x2+5)) d/dx1", "test/test_expressions.jl:55-74",
     ["sin", ["+", ["*", 2.0, X_(1)], ["exp", ["+", X_(2), 5.0]]]], Xe, unary=["sin", "cos", "exp"],
     binary=["+", "-", "*", "/"], kind="grad", mode="variable", y=np.sin(inner).tolist(),
     grad_rows={"0": (2.0 * np.cos(inner)).tolist()}, ok=True, rtol=1e-8, atol=0)
Xg = rng.random((3, 100)) * 5
case("derivatives:d(3.2*x1)/dc = x1", "test/test_derivatives.jl:108-115", ["*", 3.2, X_(1)], Xg,
     unary=["custom_cos", "exp", "sin"], binary=["+", "*", "-", "/", "pow_abs2"], kind="grad",
     mode="constant", grad=[Xg[0].tolist()], ok=True, rtol=1e-15, atol=0)
# equation1/2 of test_derivatives.jl:14-15 with closed-form gradients (the reference compares
# against Zygote of the closed form at rtol=0.1)
x1, x2, x3 = Xg
case("derivatives:equation1", "test/test_derivatives.jl:14,58-104",
     ["+", ["+", ["+", X_(1), X_(2)], X_(3)], 3.2], Xg, unary=["custom_cos", "exp", "sin"],
     binary=["+", "*", "-", "/", "pow_abs2"], kind="grad", mode="variable",
     y=(x1 + x2 + x3 + 3.2).tolist(), grad=np.ones((3, 100)).tolist(), ok=True, rtol=1e-14, atol=0)
pa = np.exp(x2 * np.log(np.abs(x1)))
eq2 = pa + x3 + np.cos(1.0 + x3) ** 2 + 3.0 / x1
g1 = pa * x2 / np.abs(x1) * np.sign(x1) - 3.0 / x1 ** 2
g2 = pa * np.log(np.abs(x1))
g3 = 1.0 + 2.0 * np.cos(1.0 + x3) * -np.sin(1.0 + x3)
case("derivatives:equation2", "test/test_derivatives.jl:15,58-104",
     ["+", ["+", ["+", ["pow_abs2", X_(1), X_(2)], X_(3)], ["custom_cos", ["+", 1.0, X_(3)]]], ["/", 3.0, X_(1)]],
     Xg, unary=["custom_cos", "exp", "sin"], binary=["+", "*", "-", "/", "pow_abs2"], kind="grad",
     mode="variable", y=eq2.tolist(), grad=[g1.tolist(), g2.tolist(), g3.tolist()], ok=True,
     rtol=1e-9, atol=1e-9)
# constant gradient of equation5 (test_derivatives.jl:117-141): c1=2.1f0, c2=-3.2f0
c1, c2 = float(f32(2.1)), float(f32(-3.2))
gc1 = 2.0 * np.cos(c1 + x3) * -np.sin(c1 + x3)
gc2 = 1.0 / x1
case("derivatives:equation5_constants", "test/test_derivatives.jl:117-141",
     ["+", ["+", ["+", ["pow_abs2", X_(1), X_(2)], X_(3)], ["custom_cos", ["+", c1, X_(3)]]], ["/", c2, X_(1)]],
     Xg, unary=["custom_cos", "exp", "sin"], binary=["+", "*", "-", "/", "pow_abs2"], kind="grad",
     mode="constant", grad=[gc1.tolist(), gc2.tolist()], ok=True, rtol=1e-9, atol=1e-9)
# test_chainrules.jl:31-83: d/dconstants of sin(x1*3.2-0.9)+0.2*x2-x3
Xc = rng.standard_normal((3, 20))
x1, x2, x3 = Xc
case("chainrules:constant_grad", "test/test_chainrules.jl:31-83",
     ["-", ["+", ["sin", ["-", ["*", X_(1), 3.2], 0.9]], ["*", 0.2, X_(2)]], X_(3)], Xc, unary=["sin", "cos"],
     binary=["+", "-", "*", "/"], kind="grad", mode="constant",
     grad=[(np.cos(x1 * 3.2 - 0.9) * x1).tolist(), (-np.cos(x1 * 3.2 - 0.9)).tolist(), x2.tolist()],
     ok=True, rtol=1e-12, atol=1e-14)
# test_undefined_derivatives.jl:11-19 / test_chainrules.jl:116-131: NaN forward => not ok
case("undefined_derivatives:safe_log(-1)", "test/test_undefined_derivatives.jl:5-19", ["safe_log", X_(1)],
     np.zeros((3, 1)) - 1, unary=["safe_log", "cos"], binary=["+", "*", "-", "/"], kind="grad",
     mode="variable", ok=False)
# undefined gradient (Zygote `nothing`) becomes zero: relu-like branch, test_chainrules.jl:90-127
case("chainrules:undefined_grad_is_zero", "test/test_chainrules.jl:88-127 (undefined_grad_op)",
     ["relu", ["+", X_(1), 0.5]], [[-1.0]], unary=["sin", "relu"], binary=["+", "*", "-"], kind="grad",
     mode="variable", y=[0.0], grad=[[0.0]], ok=True, rtol=0, atol=0)

# ---------------------------------------------------------------- parametric
Xp = [[0.0, pi / 2, pi, 3 * pi / 2, 2 * pi]]
for cl, ex in (([1, 1, 1, 1, 1], [1.0, 2.0, 1.0, 0.0, 1.0]), ([1, 2, 2, 3, 1], [1.0, 3.0, 2.0, 2.0, 1.0])):
    case(f"parametric:sin(x)+p:{cl}", "test/test_parametric_expression.jl:72-94", ["+", ["sin", X_(1)], P_(1)],
         Xp, unary=["sin"], binary=["+", "-"], kind="param", params=[[1.0, 2.0, 3.0]], classes=cl,
         y=ex, ok=True, rtol=0, atol=1e-15)
Xp2 = [[0.0, pi / 2, pi, 1.2], [0.0, 0.0, 1.5, 0.1]]
case("parametric:2params_2vars", "test/test_parametric_expression.jl:99-128",
     ["+", ["+", ["sin", X_(1)], X_(2)], ["*", P_(1), P_(2)]], Xp2, unary=["sin"], binary=["+", "-", "*"],
     kind="param", params=[[1.0, 1.0, 0.8], [2.0, 3.0, 5.0]], classes=[1, 1, 2, 3],
     y=[math.sin(0.0) + 0.0 + 1.0 * 2.0, math.sin(pi / 2) + 0.0 + 1.0 * 2.0,
        math.sin(pi) + 1.5 + 1.0 * 3.0, math.sin(1.2) + 0.1 + 0.8 * 5.0], ok=True, rtol=1e-15, atol=0)
Xo = [[11, 12, 13, 14, 15, 16, 17, 18, 19.0], [21, 22, 23, 24, 25, 26, 27, 28, 29]]
clo = [1, 2, 3, 3, 2, 1, 1, 2, 3]
for nm, pr in (("init", [[0.0] * 3, [0.0] * 3]), ("true", [[-0.2, 0.2, 0.3], [1.4, 0.5, -0.9]])):
    yy = [(Xo[0][i] * pr[1][clo[i] - 1] + Xo[1][i]) + pr[0][clo[i] - 1] for i in range(9)]
    case(f"parametric:(x*p2)+y+p1:{nm}", "test/test_parametric_expression.jl:130-183 (exact ==)",
         ["+", ["+", ["*", X_(1), P_(2)], X_(2)], P_(1)], Xo, unary=["sin"], binary=["+", "-", "*", "/"],
         kind="param", params=pr, classes=clo, y=yy, ok=True, rtol=0, atol=0)

# ---------------------------------------------------------------- n-ary structure
# adapted from test/test_n_arity_nodes.jl:146-248: same tree SHAPES, the custom closures
# (x*y-z, x^2+y) replaced by table operators of the same arity (fma, clamp).
Xt = rng.standard_normal((3, 10))
case("narity:fma(x1,x2,0.5)", "test/test_n_arity_nodes.jl:172-176 (adapted)", ["fma", X_(1), X_(2), 0.5], Xt,
     unary=["sin"], binary=["+"], ternary=["fma", "clamp"], y=(Xt[0] * Xt[1] + 0.5).tolist(), ok=True,
     rtol=1e-15, atol=1e-16)
case("narity:nested", "test/test_n_arity_nodes.jl:178-189 (adapted)",
     ["fma", ["sin", X_(1)], X_(2), ["+", X_(3), 0.5]], Xt, unary=["sin"], binary=["+"],
     ternary=["fma", "clamp"], y=(np.sin(Xt[0]) * Xt[1] + (Xt[2] + 0.5)).tolist(), ok=True, rtol=1e-15,
     atol=1e-16)
case("narity:const_nested", "test/test_n_arity_nodes.jl:214-248 (adapted)",
     ["fma", ["sin", 0.5], 1.5, ["+", 0.5, 2.5]], np.zeros((1, 1)), unary=["sin"], binary=["+"],
     ternary=["fma", "clamp"], y=[math.sin(0.5) * 1.5 + 3.0], ok=True, rtol=1e-15, atol=0)
case("narity:clamp", "test/test_supposition_consistency.jl:20 (clamp in operator set)",
     ["clamp", X_(1), -0.5, 0.5], Xt, unary=["sin"], binary=["+"], ternary=["fma", "clamp"],
     y=np.clip(Xt[0], -0.5, 0.5).tolist(), ok=True, rtol=0, atol=0)

# ---------------------------------------------------------------- test_tree_construction.jl:24-79
# sub(abs(3*unaop(x1))^2.0, -1.2) over the unary operators of the test, |err|/N < 1e-6
Xu = rng.standard_normal((5, 100)) / 3
Xu = Xu + np.sign(Xu) * 0.1
Xpos = rng.random((5, 100)) / 3 + 0.1
unas = {
    "cos": np.cos, "exp": np.exp, "safe_log": np.log, "safe_log2": np.log2, "safe_log10": np.log10,
    "safe_sqrt": np.sqrt, "relu": lambda v: np.where(v < 0, 0.0, v), "safe_acosh": np.arccosh,
    "gamma": np.vectorize(math.gamma),
}
for un, fn in unas.items():
    Xc_ = Xpos + 1.0 if un == "safe_acosh" else (Xpos if un.startswith("safe_") else Xu)
    yt = np.abs(3.0 * fn(Xc_[0])) ** 2.0 - (-1.2)
    case(f"tree_construction:{un}", "test/test_tree_construction.jl:24-79",
         ["-", ["^", ["abs", ["*", 3.0, [un, X_(1)]]], 2.0], -1.2], Xc_, unary=[un, "abs"],
         binary=["+", "*", "^", "/", "-"], y=yt.tolist(), ok=True, rtol=0,
         atol=3e-2 * 100 if un == "gamma" else 1e-6 * 100)

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.json")
with open(out, "w") as fh:
    json.dump({"generator": "tests/golden/make_reference_known_answers.py",
               "reference": "SymbolicML/DynamicExpressions.jl v2.9.2", "cases": cases}, fh, indent=0)
print(f"wrote {len(cases)} cases to {out}")
