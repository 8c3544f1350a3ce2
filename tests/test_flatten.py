"""The host flattener (node.py: the Python twin of the Julia shim's `flatten!`): post-order tape + per-tree constant pool, the arguments of
de_program_create.  The fast population path (three lists per population, a tuple-free post-order walk) must give exactly what a plain
recursive restatement of tree_mapreduce's order (src/base.jl:123-158: children left to right, then the node) gives."""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd.node import LEAF_CONST, LEAF_FEATURE, LEAF_PARAM, TAPE_DTYPE, postorder


def slow_flatten(tree, ops, dtype):
    rows, consts = [], []

    def walk(n):
        for c in n.children:
            walk(c)
        if n.degree == 0:
            if n.constant:
                rows.append((0, LEAF_CONST, len(consts)))
                consts.append(n.val)
            elif getattr(n, "is_parameter", False):
                rows.append((0, LEAF_PARAM, n.parameter - 1))
            else:
                rows.append((0, LEAF_FEATURE, n.feature - 1))
        else:
            rows.append((n.degree, ops.opcode(n.degree, n.op), 0))
    walk(tree)
    return np.array(rows, dtype=TAPE_DTYPE), np.asarray(consts, dtype=dtype)


@pytest.mark.parametrize("kind", ["bench", "parametric", "wide"])
def test_population_flattener_is_the_recursive_restatement(kind):
    if kind == "parametric":
        ops = de.synth.BENCH_OPERATORS
        trees = de.synth.random_population(300, seed=5, node_type=de.ParametricNode, nparams=8)
    elif kind == "wide":
        ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min"), unary_operators=("cos", "exp", "sin", "abs", "safe_log"))
        rng = de.synth.Xoshiro256ss(9)
        trees = [de.synth.gen_random_tree_fixed_size(1 + i % 40, ops, 7, rng, np.float64) for i in range(300)]
    else:
        ops = de.synth.BENCH_OPERATORS
        trees = de.synth.random_population(300, seed=4)
    for dtype in (np.float32, np.float64):
        nodes, noff, consts, coff = de.flatten_population(trees, ops, dtype)
        assert nodes.dtype == TAPE_DTYPE and consts.dtype == dtype
        assert noff[0] == 0 and coff[0] == 0 and noff[-1] == len(nodes) and coff[-1] == len(consts)
        for k, t in enumerate(trees):
            tape, cs = slow_flatten(t, ops, dtype)
            assert np.array_equal(nodes[noff[k]:noff[k + 1]], tape)
            assert np.array_equal(consts[coff[k]:coff[k + 1]].view(np.uint8), cs.view(np.uint8))
            one_tape, one_cs = de.flatten(t, ops, dtype)
            assert np.array_equal(one_tape, tape) and np.array_equal(one_cs.view(np.uint8), cs.view(np.uint8))
            assert [id(n) for n in postorder(t)] == _ids_postorder(t)


def _ids_postorder(tree):
    out = []

    def walk(n):
        for c in n.children:
            walk(c)
        out.append(id(n))
    walk(tree)
    return out


def test_empty_population_and_single_leaves():
    ops = de.synth.BENCH_OPERATORS
    nodes, noff, consts, coff = de.flatten_population([], ops, np.float32)
    assert len(nodes) == 0 and len(consts) == 0 and list(noff) == [0] and list(coff) == [0]
    trees = [de.Node(val=1.5), de.Node(feature=3), de.Node(val=-0.0)]
    nodes, noff, consts, coff = de.flatten_population(trees, ops, np.float64)
    assert list(noff) == [0, 1, 2, 3] and list(coff) == [0, 1, 1, 2]
    assert [tuple(r) for r in nodes] == [(0, LEAF_CONST, 0), (0, LEAF_FEATURE, 2), (0, LEAF_CONST, 0)]
    assert np.signbit(consts[1]) and consts[0] == 1.5


def test_a_feature_index_beyond_the_tape_field_is_refused():
    ops = de.synth.BENCH_OPERATORS
    with pytest.raises(ValueError):
        de.flatten(de.Node(feature=70000), ops, np.float32)
