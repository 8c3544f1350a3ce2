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

LEAF_CONST, LEAF_FEATURE, LEAF_PARAM, LEAF_SHARED = 0, 1, 2, 3
OP_SHARE = 0xFE  # DE_OP_SHARE (include/de_opcodes.h): "the subtree just emitted is shared subtree `arg`"

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


class GraphNode(Node):
    """``GraphNode{T,D}`` (src/Node.jl:138-166): exactly ``Node``, with the assumption that some nodes are SHARED — the
    same object used as a child in several places.  ``preserve_sharing`` (src/Node.jl:341-342) changes what the
    constant bookkeeping means: a shared constant is ONE constant (``count_constant_nodes`` with ``f_on_shared``,
    src/NodeUtils.jl:43-51; one gradient row, the sum over its occurrences), and ``copy`` keeps the graph structure.
    Evaluation is unchanged in value — the reference evaluates a shared node once per parent — so the device may, and
    does, evaluate it once per tape (``flatten_graph``, ``de_program_create_cse``)."""

    __slots__ = ()

    def copy(self, _memo=None) -> "GraphNode":
        import copy as _copy

        memo = {} if _memo is None else _memo
        if id(self) in memo:
            return memo[id(self)]
        n = _copy.copy(self)
        memo[id(self)] = n
        n.children = tuple(c.copy(memo) if isinstance(c, GraphNode) else c.copy() for c in self.children)
        return n


def break_sharing(tree: Node) -> Node:
    """``copy(tree; break_sharing=Val(true))`` (src/base.jl:37-46) as a plain ``Node`` tree: every occurrence its own node."""
    if tree.degree == 0:
        n = Node(val=tree.val) if tree.constant else Node(feature=tree.feature)
        return n
    return Node(tree.op, *[break_sharing(c) for c in tree.children])


def preserve_sharing(tree: Node) -> bool:  # src/Node.jl:341-342
    return isinstance(tree, GraphNode)


def _unique_postorder(tree: Node) -> List[Node]:
    """Post-order with every shared node visited once, at its first occurrence (tree_mapreduce with an id_map,
    src/base.jl:83-120)."""
    seen, out = set(), []
    for n in postorder(tree):
        if id(n) not in seen:
            seen.add(id(n))
            out.append(n)
    return out


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


def count_constant_nodes(tree: Node) -> int:  # src/NodeUtils.jl:43-51 (a shared constant counts once: f_on_shared)
    nodes = _unique_postorder(tree) if preserve_sharing(tree) else tree
    return sum(1 for n in nodes if is_node_constant(n))


def postorder(tree: Node) -> List[Node]:
    """Children left-to-right, then the node (tree_mapreduce order, src/base.jl:123-158)."""
    # node, then its subtrees RIGHT to left (children pushed left to right, the right one is popped first): reversed, that is
    # left subtree, right subtree, node
    out: List[Node] = []
    stack: List[Node] = [tree]
    pop, push, more = stack.pop, out.append, stack.extend
    while stack:
        n = pop()
        push(n)
        if n.degree:
            more(n.children)
    out.reverse()
    return out


def get_scalar_constants(tree: Node) -> Tuple[np.ndarray, List[Node]]:
    """Constants in depth-first left-to-right order (src/NodeUtils.jl:99-120) — the same
    order as index_constant_nodes (:184-201), i.e. the constant-gradient row order."""
    refs = [n for n in (_unique_postorder(tree) if preserve_sharing(tree) else postorder(tree)) if is_node_constant(n)]
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
    deg: List[int] = []
    op: List[int] = []
    arg: List[int] = []
    consts: List[float] = []
    _flatten_into(tree, operators, deg, op, arg, consts, {})
    return _tape_of(deg, op, arg), np.asarray(consts, dtype=dtype)


def _flatten_into(tree: Node, operators: OperatorEnum, deg: List[int], op: List[int], arg: List[int], consts: List[float],
                  opcache: dict) -> int:
    """Appends the tape of ``tree`` (three parallel lists: one numpy conversion per POPULATION instead of one structured-array
    store per node: 10 -> 3 us per tree) and its constants; returns the number of constants.  Constant slots are numbered per tree."""
    c0 = len(consts)
    for n in postorder(tree):
        d = n.degree
        if d == 0:
            deg.append(0)
            if n.constant:
                op.append(LEAF_CONST)
                arg.append(len(consts) - c0)
                consts.append(n.val)
            elif getattr(n, "is_parameter", False):
                op.append(LEAF_PARAM)
                arg.append(n.parameter - 1)
            else:
                op.append(LEAF_FEATURE)
                arg.append(n.feature - 1)
        else:
            key = (d, n.op)
            code = opcache.get(key)
            if code is None:
                code = opcache[key] = operators.opcode(d, n.op)
            deg.append(d)
            op.append(code)
            arg.append(0)
    nc = len(consts) - c0
    if nc > 65535:
        raise ValueError("more than 65535 constants in one tree")
    return nc


def _tape_of(deg: List[int], op: List[int], arg: List[int]) -> np.ndarray:
    tape = np.zeros(len(deg), dtype=TAPE_DTYPE)
    if deg:
        tape["degree"] = deg
        tape["op"] = op
        tape["arg"] = np.asarray(arg, dtype=np.int64)  # (range-checked by the cast below: a feature index > 65535 must not wrap silently)
        if max(arg) > 65535 or min(arg) < 0:
            raise ValueError("leaf index outside 0..65535")
    return tape


def flatten_graph(tree: Node, operators: OperatorEnum, dtype=np.float32):
    """GraphNode tree -> (expanded tape, consts, cse tape | None, occurrence_of).

    * the EXPANDED tape is what ``flatten`` gives (the reference's evaluation order: a shared node once per parent), one
      constant slot per OCCURRENCE of a constant leaf;
    * ``occurrence_of[s]`` = the index of the unique constant (order of ``get_scalar_constants``) slot ``s`` is an
      occurrence of: the host keeps the occurrence slots of a shared constant equal and sums their gradient rows
      (``Population`` does both), which is what the reference's shared ``NodeIndex`` row amounts to;
    * the CSE tape restates the tree with every shared, non-constant operator subtree ONCE (``DE_OP_SHARE`` after its
      first occurrence, ``DE_LEAF_SHARED`` afterwards; constant leaves keep the slot of their first occurrence), or None
      when nothing can be shared (shared leaves and constant subtrees are cheaper re-read / folded than stored; a shared
      subtree that is the root or a direct child of a ternary operator is left expanded, include/de_hip.h)."""
    tape, consts = flatten(tree, operators, dtype)
    # unique constants in order of first occurrence; slot -> unique index
    uniq, occ = {}, []
    for n in postorder(tree):
        if is_node_constant(n):
            occ.append(uniq.setdefault(id(n), len(uniq)))
    occurrence_of = np.asarray(occ, dtype=np.int64)
    # which operator subtrees occur more than once?
    count, is_const_sub = {}, {}
    for n in postorder(tree):
        count[id(n)] = count.get(id(n), 0) + 1
        is_const_sub[id(n)] = is_node_constant(n) if n.degree == 0 else all(is_const_sub[id(c)] for c in n.children)
    # a node reached only through an already-shared ancestor is not shared in its own right
    parents = {}
    for n in _unique_postorder(tree):
        for c in n.children:
            parents.setdefault(id(c), set()).add(id(n))
    shareable = {k for k, c in count.items() if c > 1}
    cse: List[Tuple[int, int, int]] = []
    defined = {}
    slot = [0]

    def emit(n: Node, under_ternary: bool) -> None:
        key = id(n)
        if key in defined:
            cse.append((0, LEAF_SHARED, defined[key]))
            skip(n)
            return
        if n.degree == 0:
            if n.constant:
                cse.append((0, LEAF_CONST, slot[0]))
                slot[0] += 1
            elif getattr(n, "is_parameter", False):
                cse.append((0, LEAF_PARAM, n.parameter - 1))
            else:
                cse.append((0, LEAF_FEATURE, n.feature - 1))
            return
        for c in n.children:
            emit(c, n.degree == 3)
        cse.append((n.degree, operators.opcode(n.degree, n.op), 0))
        if (key in shareable and not is_const_sub[key] and not under_ternary and n is not tree and len(defined) < 12
                and multi_use(n)):
            defined[key] = len(defined)
            cse.append((1, OP_SHARE, defined[key]))

    def skip(n: Node) -> None:  # a later occurrence: its constant leaves still own slots in the expanded numbering
        for m in postorder(n):
            if is_node_constant(m):
                slot[0] += 1

    def multi_use(n: Node) -> bool:
        # used again OUTSIDE the occurrences already covered by a shared ancestor: more than one distinct parent, or the
        # same parent twice
        ps = parents.get(id(n), set())
        if len(ps) > 1:
            return True
        (p_id,) = tuple(ps) if ps else (None,)
        for q in _unique_postorder(tree):
            if id(q) == p_id:
                return sum(1 for c in q.children if c is n) > 1
        return False

    emit(tree, False)
    if not defined:
        return tape, consts, None, occurrence_of
    cse_tape = np.zeros(len(cse), dtype=TAPE_DTYPE)
    for i, rec in enumerate(cse):
        cse_tape[i] = rec
    return tape, consts, cse_tape, occurrence_of


def flatten_population(trees: Sequence[Node], operators: OperatorEnum, dtype=np.float32):
    """Population -> (nodes, node_offsets, consts, const_offsets): the arguments of
    ``de_program_create``."""
    deg: List[int] = []
    op: List[int] = []
    arg: List[int] = []
    consts: List[float] = []
    opcache: dict = {}
    node_offsets = np.zeros(len(trees) + 1, dtype=np.int64)
    const_offsets = np.zeros(len(trees) + 1, dtype=np.int64)
    for k, t in enumerate(trees):
        _flatten_into(t, operators, deg, op, arg, consts, opcache)
        node_offsets[k + 1] = len(deg)
        const_offsets[k + 1] = len(consts)
    return _tape_of(deg, op, arg), node_offsets, np.asarray(consts, dtype=dtype), const_offsets


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
