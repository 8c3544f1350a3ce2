"""Host-side tree preprocessing before flattening: ``simplify_tree`` and
``combine_operators`` — the two rewrites of the reference's SimplifyModule (src/Simplify.jl:25-138).

They shorten tapes; they are NOT needed for speed on constant subtrees (the device program folds those itself,
value-identically, csrc/de_lower.cpp) — ``simplify_tree`` makes the fold permanent in the tree, as a search loop
does after mutation, and ``combine_operators`` merges constants across nested ``+``/``*``/``-`` (which changes
rounding, exactly like the reference's).  Scalar arithmetic happens in the tree's element type with numpy
(IEEE-exact for ``+ - * /``; numpy's libm instead of Julia's for transcendentals: results within an ulp or so).
"""
from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .node import Node
from .operators import OperatorEnum

_UNARY: Dict[str, Callable] = {
    "cos": np.cos, "sin": np.sin, "tan": np.tan, "exp": np.exp, "log": np.log, "log2": np.log2, "log10": np.log10,
    "log1p": np.log1p, "sqrt": np.sqrt, "cbrt": np.cbrt, "abs": np.abs, "sinh": np.sinh, "cosh": np.cosh,
    "tanh": np.tanh, "atan": np.arctan, "asinh": np.arcsinh, "acosh": np.arccosh, "neg": np.negative,
    "-": np.negative, "square": lambda x: x * x, "cube": lambda x: (x * x) * x, "floor": np.floor,
    "ceil": np.ceil, "sign": np.sign, "round": np.rint, "inv": lambda x: type(x)(1) / x,
    "relu": lambda x: x if not (x < 0) else type(x)(0),  # +0 for negative arguments (and NaN stays NaN), like the device
    "safe_log": lambda x: np.log(x) if x > 0 else type(x)(np.nan), "safe_sqrt": lambda x: np.sqrt(x) if x >= 0 else type(x)(np.nan),
    "safe_log2": lambda x: np.log2(x) if x > 0 else type(x)(np.nan), "safe_log10": lambda x: np.log10(x) if x > 0 else type(x)(np.nan),
    "safe_log1p": lambda x: np.log1p(x) if x > -1 else type(x)(np.nan), "safe_acosh": lambda x: np.arccosh(x) if x >= 1 else type(x)(np.nan),
    "asin": np.arcsin, "acos": np.arccos, "atanh": np.arctanh, "exp2": np.exp2,
}
_BINARY: Dict[str, Callable] = {
    "+": lambda a, b: a + b, "-": lambda a, b: a - b, "sub": lambda a, b: a - b, "*": lambda a, b: a * b,
    "/": lambda a, b: a / b, "max": max, "min": min, "^": lambda a, b: np.power(a, b), "pow": lambda a, b: np.power(a, b),
    "greater": lambda a, b: type(a)(1) if a > b else type(a)(0), "rem": np.fmod,
}
_COMMUTATIVE = ("+", "*")  # is_commutative, src/Simplify.jl:15-17
_SUBTRACTION = ("-",)      # is_subtraction, :19-20


def _is_const(n: Node) -> bool:  # is_node_constant, src/NodeUtils.jl:37
    return n.degree == 0 and n.constant and not getattr(n, "is_parameter", False)


def _scalar(name: str, degree: int) -> Optional[Callable]:
    return (_UNARY if degree == 1 else _BINARY if degree == 2 else {}).get(name)


def _set_leaf_const(n: Node, val: float) -> None:  # set_node!(p, constant leaf)
    n.degree, n.constant, n.val, n.feature, n.op, n.children = 0, True, float(val), 0, 0, ()


def _device_values(tree: Node, operators: OperatorEnum, dtype) -> Optional[Dict[int, float]]:
    """Value of every feature-free operator node, computed by the device library itself in ONE launch (a population of the
    constant subtrees on a single sample): the bits the device's own constant folding would produce, for every opcode.
    None when no GPU / library is available (the numpy tables are used then)."""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
        from . import api
        api.library()
    except Exception:
        return None
    nodes = []

    def walk(n: Node) -> bool:
        const = n.degree == 0 and _is_const(n) if n.degree == 0 else all([walk(c) for c in n.children])
        if n.degree > 0 and const:
            nodes.append(n)
        return const
    walk(tree)
    if not nodes:
        return {}
    pop = api.Population([n for n in nodes], operators, dtype, n_features=1, eval_context=api.EvalContext(early_exit=False))
    try:
        out, _ = pop.eval(np.zeros((1, 1), dtype=dtype, order="F"))
    finally:
        pop.close()
    return {id(n): float(out[k, 0]) for k, n in enumerate(nodes)}


def simplify_tree(tree: Node, operators: OperatorEnum, dtype=np.float64, use_device: Optional[bool] = None) -> Node:
    """``simplify_tree!`` (src/Simplify.jl:131-136, ``combine_children!`` :118-129): bottom-up, an operator whose
    children are all constants becomes the constant it evaluates to — unless a child or the result is not finite
    (``cos(NaN)`` stays).  In place; returns the tree.

    The constants are evaluated ON THE DEVICE when one is available (``use_device`` None/True): one launch over all
    feature-free subtrees, so a folded constant carries exactly the bits the device program's own folding would give and
    every operator of include/de_opcodes.h folds, as in the reference (any Julia function).  Without a GPU
    (``use_device=False``, or none visible) the numpy scalar tables above are used: operators outside them stay unfolded and
    transcendentals may differ from the device by an ulp."""
    if use_device is not False:
        vals = _device_values(tree, operators, dtype)
        if vals is None and use_device:
            raise RuntimeError("simplify_tree(use_device=True): no MI355X / libde_hip.so available")
        if vals is not None:
            return _simplify_with(tree, vals)
    return _simplify_numpy(tree, operators, dtype)


def _simplify_with(tree: Node, vals: Dict[int, float]) -> Node:
    for c in tree.children:
        _simplify_with(c, vals)
    if tree.degree >= 1 and all(_is_const(c) and np.isfinite(c.val) for c in tree.children) and id(tree) in vals:
        if np.isfinite(vals[id(tree)]):
            _set_leaf_const(tree, vals[id(tree)])
    return tree


def _simplify_numpy(tree: Node, operators: OperatorEnum, dtype=np.float64) -> Node:
    dt = np.dtype(dtype).type
    for c in tree.children:
        _simplify_numpy(c, operators, dtype)
    if tree.degree in (1, 2) and all(_is_const(c) for c in tree.children):
        f = _scalar(operators.ops[tree.degree - 1][tree.op - 1], tree.degree)
        vals = [dt(c.val) for c in tree.children]
        if f is not None and all(np.isfinite(v) for v in vals):
            with np.errstate(all="ignore"):
                out = dt(f(*vals))
            if np.isfinite(out):
                _set_leaf_const(tree, out)
    return tree


def combine_operators(tree: Node, operators: OperatorEnum, dtype=np.float64) -> Node:
    """``combine_operators`` (src/Simplify.jl:25-116), binary trees only: ``((const + var) + const) => (const + var)``
    for commutative ``+``/``*`` and the four nested-subtraction patterns.  Returns the (possibly different) root."""
    dt = np.dtype(dtype).type
    if tree.degree == 0:
        return tree
    tree.children = tuple(combine_operators(c, operators, dtype) for c in tree.children)
    if tree.degree != 2:
        return tree
    name = operators.ops[1][tree.op - 1]
    l, r = tree.children
    top_const = _is_const(l) or _is_const(r)
    if name in _COMMUTATIVE and top_const:
        f = _BINARY[name]
        if _is_const(l):  # constant to the right (:44-48)
            l, r = r, l
            tree.children = (l, r)
        top = dt(r.val)
        below = l
        if below.degree == 2 and below.op == tree.op:
            bl, br = below.children
            if _is_const(bl):
                bl.val = float(f(dt(bl.val), top))
                tree = below
            elif _is_const(br):
                br.val = float(f(dt(br.val), top))
                tree = below
    if tree.degree == 2 and operators.ops[1][tree.op - 1] in _SUBTRACTION and \
            (_is_const(tree.children[0]) or _is_const(tree.children[1])):
        l, r = tree.children
        if _is_const(l):
            if r.degree == 2 and r.op == tree.op:
                rl, rr = r.children
                if _is_const(rl):      # (const - (const - var)) => (var - const)   (:76-83)
                    l.val = float(dt(rl.val) - dt(l.val))
                    tree.children = (rr, l)
                elif _is_const(rr):    # (const - (var - const)) => (const - var)   (:84-90)
                    l.val = float(dt(l.val) + dt(rr.val))
                    tree.children = (l, rl)
        else:
            if l.degree == 2 and l.op == tree.op:
                ll, lr = l.children
                if _is_const(ll):      # ((const - var) - const) => (const - var)   (:95-102)
                    r.val = float(dt(ll.val) - dt(r.val))
                    tree.children = (r, lr)
                elif _is_const(lr):    # ((var - const) - const) => (var - const)   (:103-109)
                    r.val = float(dt(r.val) + dt(lr.val))
                    tree.children = (ll, r)
    return tree
