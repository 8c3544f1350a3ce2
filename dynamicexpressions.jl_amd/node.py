"""Host-side mirror of ``Node{T,D}`` / ``ParametricNode`` and the tape flattener.

This is the host half of the boundary: the tree data structure stays on
the host exactly as in the reference (src/Node.jl:74-90, 1-based ``feature`` and ``op``);
``flatten`` turns it into the post-order tape + constant pool that crosses the C ABI
(``de_tape_node_t`` in include/de_hip.h).  The Julia version of this file is
``julia/DynamicExpressionsHIPExt.jl``; this Python twin exists because the build image has no
Julia, and drives the parity tests.
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from .operators import OperatorEnum

LEAF_CONST, LEAF_FEATURE, LEAF_PARAM = 0, 1, 2

TAPE_DTYPE = np.dtype([("degree", np.uint8), ("op", np.uint8), ("arg", np.uint16)])
assert TAPE_DTYPE.itemsize == 4


class Node:
    """``Node{T,D}`` (src/Node.jl:74-90).

    ``Node(val=3.0)`` constant leaf, ``Node(feature=1)`` variable leaf x1 (1-based),
    ``Node(op=i, children=(l, r))`` operator node with 1-based operator index ``i`` into
    ``operators[degree]``.  ``Node(i, l)`` / ``Node(i, l, r)`` are the positional forms
    used throughout the reference tests.
    """

    __slots__ = ("degree", "constant", "val", "feature", "op", "children")

    def __init__(self, *args, val=None, feature=None, op=None, children=None):
        if args:
            op, children = args[0], tuple(args[1:])
        self.constant = False
        self.val = 0.0
        self.feature = 0
        self.op = 0
        self.children: Tuple["Node", ...] = ()
        if children:
            if op is None:
                raise ValueError("operator node needs `op`")
            self.degree = len(children)
            self.op = int(op)
            self.children = tuple(children)
        elif val is not None:
            self.degree = 0
            self.constant = True
            self.val = float(val)
        elif feature is not None:
            self.degree = 0
            self.feature = int(feature)
            if self.feature < 1:
                raise ValueError("feature indices are 1-based")
        else:
            raise ValueError("Node needs val, feature or op+children")

    # get_child / get_children, src/Node.jl:203-215 (1-based)
    def get_child(self, i: int) -> "Node":
        return self.children[i - 1]

    @property
    def l(self) -> "Node":
        return self.children[0]

    @property
    def r(self) -> "Node":
        return self.children[1]

    def __iter__(self) -> Iterator["Node"]:
        """Depth-first pre-order traversal (any/foreach order, src/base.jl)."""
        stack = [self]
        while stack:
            n = stack.pop()
            yield n
            stack.extend(reversed(n.children))

    def copy(self) -> "Node":
        import copy as _copy

        n = _copy.copy(self)
        n.children = tuple(c.copy() for c in self.children)
        return n


class ParametricNode(Node):
    """``ParametricNode{T,D}`` (src/ParametricExpression.jl:52-74): a leaf may also be a
    per-class parameter (``is_parameter``, 1-based ``parameter`` row)."""

    __slots__ = ("is_parameter", "parameter")

    def __init__(self, *args, parameter=None, **kw):
        self.is_parameter = False
        self.parameter = 0
        if parameter is not None:
            super().__init__(feature=1)
            self.feature = 0
            self.is_parameter = True
            self.parameter = int(parameter)
            if self.parameter < 1:
                raise ValueError("parameter indices are 1-based")
        else:
            super().__init__(*args, **kw)


def count_nodes(tree: Node) -> int:  # src/base.jl:271-280
    return sum(1 for _ in tree)


def count_depth(tree: Node) -> int:  # src/NodeUtils.jl:25-29 (leaf = 1)
    # iterative: (node, depth)
    best, stack = 0, [(tree, 1)]
    while stack:
        n, d = stack.pop()
        best = max(best, d)
        stack.extend((c, d + 1) for c in n.children)
    return best


def is_node_constant(n: Node) -> bool:  # src/NodeUtils.jl:37
    return n.degree == 0 and n.constant


def count_constant_nodes(tree: Node) -> int:  # src/NodeUtils.jl:43-51
    return sum(1 for n in tree if is_node_constant(n))


def postorder(tree: Node) -> List[Node]:
    """Children left-to-right, then the node (tree_mapreduce order, src/base.jl:123-158)."""
    out: List[Node] = []
    stack: List[Tuple[Node, int]] = [(tree, 0)]
    while stack:
        n, i = stack.pop()
        if i < n.degree:
            stack.append((n, i + 1))
            stack.append((n.children[i], 0))
        else:
            out.append(n)
    return out


def get_scalar_constants(tree: Node) -> Tuple[np.ndarray, List[Node]]:
    """Constants in depth-first left-to-right order (src/NodeUtils.jl:99-120) — the same
    order as index_constant_nodes (:184-201), i.e. the constant-gradient row order."""
    refs = [n for n in postorder(tree) if is_node_constant(n)]
    return np.array([n.val for n in refs], dtype=np.float64), refs


def set_scalar_constants(tree: Node, constants: Sequence[float], refs: List[Node]) -> None:
    for n, v in zip(refs, constants):  # src/NodeUtils.jl:130-143
        n.val = float(v)


def max_feature(tree: Node) -> int:
    return max((n.feature for n in tree if n.degree == 0 and not n.constant
                and not getattr(n, "is_parameter", False)), default=0)


def flatten(tree: Node, operators: OperatorEnum, dtype=np.float32) -> Tuple[np.ndarray, np.ndarray]:
    """Tree -> (tape, consts).

    tape: post-order ``de_tape_node_t`` records; consts: the tree's constant pool in
    depth-first left-to-right order.  Leaf indices become 0-based here.  Raises
    UnsupportedOperatorError for functions without an opcode and ValueError for a degree
    the OperatorEnum has no operators for (get_op, src/Evaluate.jl:408-419).
    """
    po = postorder(tree)
    tape = np.zeros(len(po), dtype=TAPE_DTYPE)
    consts: List[float] = []
    for i, n in enumerate(po):
        if n.degree == 0:
            if n.constant:
                tape[i] = (0, LEAF_CONST, len(consts))
                consts.append(n.val)
            elif getattr(n, "is_parameter", False):
                tape[i] = (0, LEAF_PARAM, n.parameter - 1)
            else:
                tape[i] = (0, LEAF_FEATURE, n.feature - 1)
        else:
            tape[i] = (n.degree, operators.opcode(n.degree, n.op), 0)
    if len(consts) > 65535:
        raise ValueError("more than 65535 constants in one tree")
    return tape, np.asarray(consts, dtype=dtype)


def flatten_population(trees: Sequence[Node], operators: OperatorEnum, dtype=np.float32):
    """Population -> (nodes, node_offsets, consts, const_offsets): the arguments of
    ``de_program_create``."""
    tapes, pools = [], []
    for t in trees:
        tp, cs = flatten(t, operators, dtype)
        tapes.append(tp)
        pools.append(cs)
    node_offsets = np.zeros(len(trees) + 1, dtype=np.int64)
    const_offsets = np.zeros(len(trees) + 1, dtype=np.int64)
    if trees:
        np.cumsum([len(t) for t in tapes], out=node_offsets[1:])
        np.cumsum([len(c) for c in pools], out=const_offsets[1:])
    nodes = np.concatenate(tapes) if tapes else np.zeros(0, dtype=TAPE_DTYPE)
    consts = np.concatenate(pools).astype(dtype) if pools else np.zeros(0, dtype=dtype)
    return nodes, node_offsets, consts, const_offsets


def string_tree(tree: Node, operators: OperatorEnum) -> str:
    """Minimal printer (src/Strings.jl:158-199) for test diagnostics."""
    if tree.degree == 0:
        if tree.constant:
            return repr(tree.val)
        if getattr(tree, "is_parameter", False):
            return f"p{tree.parameter}"
        return f"x{tree.feature}"
    name = operators.ops[tree.degree - 1][tree.op - 1]
    args = [string_tree(c, operators) for c in tree.children]
    if tree.degree == 2 and name in "+-*/^":
        return f"({args[0]} {name} {args[1]})"
    return f"{name}({', '.join(args)})"
