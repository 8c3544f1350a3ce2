"""CPU unit tests of the HOST logic: tape flattening, the C lowering (de_lower_tape makes no HIP
call) and the reference-dispatch annotation that decides the `ok` flag.  The lowered program is
executed by tests/prog_interp.py (a numpy model of the device machine) and compared with the
oracle — values AND flags — on the golden cases and on seeded random trees."""
import os
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from helpers import case_X, case_options, case_tree, load_golden, value_tolerance
from oracle import oracle
import prog_interp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [c for c in load_golden() if c["kind"] in ("eval", "flag", "param")]


def run_lowered(tree, ops, X, options, params=None, classes=None):
    dt = X.dtype
    tape, consts = de.flatten(tree, ops, dt)
    P = 0 if params is None else params.shape[0]
    words, meta = api.lower_tape(tape, consts, X.shape[0], P, options, dt)
    cls0 = None if classes is None else np.asarray(classes) - 1
    out, ok = prog_interp.run(words, X, bool(options & 1), params, cls0, meta["host_ok_eval"])
    return out, ok, words, meta, tape, consts


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_lowered_program_matches_oracle_on_golden(case):
    tree, ops = case_tree(case)
    X = case_X(case)
    opts = case_options(case)
    exp = case["expect"]
    if case["kind"] == "param":
        params = np.asarray(exp["params"], dtype=X.dtype)
        out, ok, *_ = run_lowered(tree, ops, X, opts, params, exp["classes"])
    else:
        out, ok, *_ = run_lowered(tree, ops, X, opts)
    assert ok == exp["ok"], case["name"]
    if ok and "y" in exp:
        want = np.asarray(exp["y"], dtype=np.float64)
        m = np.isfinite(want)
        tol = max(exp.get("atol", 0), 1e-12) + max(exp.get("rtol", 0), 4e-7 if X.dtype == np.float32 else 1e-14) * np.abs(want[m])
        assert np.all(np.abs(out[m].astype(np.float64) - want[m]) <= tol)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("options", [7, 1, 6, 0, 9])
def test_random_trees_values_and_flags(dtype, options):
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "pow_abs2"),
                          unary_operators=("cos", "exp", "safe_log", "neg", "square"))
    rng = de.synth.Xoshiro256ss(1234 + options)
    g = np.random.Generator(np.random.PCG64(7))
    n_bad = 0
    for it in range(150):
        tree = de.synth.gen_random_tree_fixed_size(3 + it % 25, ops, 4, rng, dtype)
        X = np.asfortranarray(g.standard_normal((4, 33)).astype(dtype))
        if it % 5 == 0:  # non-finite inputs exercise the leaf-test rules
            X[g.integers(4), g.integers(33)] = [np.inf, -np.inf, np.nan][it % 3]
        if it % 7 == 0:
            for n in tree:
                if n.degree == 0 and n.constant and g.random() < 0.3:
                    n.val = float("inf")
        out, ok, words, meta, tape, consts = run_lowered(tree, ops, X, options)
        y, ok_ref = oracle.eval_tree_array(tape, consts, X, options, elementwise=True)
        assert ok == ok_ref, (de.string_tree(tree, ops), options)
        n_bad += not ok
        if ok:
            y64, _ = oracle.eval_tree_array(tape, consts.astype(np.float64), X.astype(np.float64), options, True)
            m = np.isfinite(y)  # (only early_exit=false leaves non-finite samples in an ok tree)
            assert np.array_equal(np.isfinite(out), m), de.string_tree(tree, ops)
            tol = value_tolerance(y[m], y64[m], dtype) + (1e-6 if dtype == np.float32 else 0)
            assert np.all(np.abs(out[m].astype(np.float64) - y[m]) <= tol), de.string_tree(tree, ops)
        elif not (options & 1):
            pass
        # every tree needs few spill slots and never more instructions than nodes
        assert len(words) <= len(tape) and meta["n_slots"] <= 4
    assert 5 < n_bad < 140


def test_no_early_exit_values_match_oracle_including_inf_injection():
    """early_exit=false: finite samples keep their values, the fused deg1 kernels write Inf
    where the inner value is non-finite (src/Evaluate.jl:722,787) — the oracle restates that and
    the lowered program must reproduce it bit for bit on IEEE-exact operators."""
    ops = de.OperatorEnum(binary_operators=("+", "-", "*", "/"), unary_operators=("neg", "abs"))
    x1, x2 = de.Node(feature=1), de.Node(feature=2)
    X = np.asfortranarray(np.array([[1.0, 0.0, np.inf, 2.0], [0.0, 0.0, 1.0, 4.0]]))
    trees = [
        de.Node(1, de.Node(4, x1, x2)),            # neg(x1/x2): deg1_l2_ll0_lr0 -> Inf at 1/0, 0/0
        de.Node(2, de.Node(1, x1)),                # abs(neg(x1)): deg1_l1_ll0 -> Inf at inf
        de.Node(1, de.Node(2, de.Node(4, x1, x2))),  # neg(abs(x1/x2)): only the inner pair is fused
    ]
    for t in trees:
        out, ok, words, meta, tape, consts = run_lowered(t, ops, X, 6)
        y, ok_ref = oracle.eval_tree_array(tape, consts, X, 6)
        assert ok and ok_ref
        np.testing.assert_array_equal(out, y)


def test_tape_validation_errors():
    T = de.node.TAPE_DTYPE
    ok_tape = np.array([(0, 1, 0), (0, 1, 1), (2, 64, 0)], dtype=T)
    api.lower_tape(ok_tape, [], 2)
    with pytest.raises(ValueError):  # operator without operands
        api.lower_tape(np.array([(2, 64, 0)], dtype=T), [], 2)
    with pytest.raises(ValueError):  # two roots
        api.lower_tape(np.array([(0, 1, 0), (0, 1, 1)], dtype=T), [], 2)
    with pytest.raises(ValueError):  # feature out of range
        api.lower_tape(np.array([(0, 1, 5)], dtype=T), [], 2)
    with pytest.raises(ValueError):  # constant slot out of range
        api.lower_tape(np.array([(0, 0, 0)], dtype=T), [], 2)
    with pytest.raises(de.UnsupportedOperatorError):  # binary opcode used as unary
        api.lower_tape(np.array([(0, 1, 0), (1, 64, 0)], dtype=T), [], 2)
    with pytest.raises(de.UnsupportedOperatorError):
        ops = de.OperatorEnum(unary_operators=("my_custom_op",))
        de.flatten(de.Node(1, de.Node(feature=1)), ops)
    with pytest.raises(ValueError):  # get_op: no operators of this degree (src/Evaluate.jl:408-419)
        de.flatten(de.Node(1, de.Node(feature=1)), de.OperatorEnum(binary_operators=("+",)))


def test_deep_and_wide_trees_spill_slots():
    ops = de.OperatorEnum(binary_operators=("+", "*"), unary_operators=("cos",))

    def full(d):
        return de.Node(feature=1) if d == 0 else de.Node(1 + d % 2, full(d - 1), full(d - 1))

    t = full(6)  # complete binary tree: Strahler number 7 -> 5 spill slots (leaf pairs need none)
    X = np.asfortranarray(np.array([[1.5, -0.25, 3.0]]))
    out, ok, words, meta, tape, consts = run_lowered(t, ops, X, 7)
    y, _ = oracle.eval_tree_array(tape, consts, X)
    np.testing.assert_array_equal(out, y)
    assert meta["n_slots"] == 5
    chain = de.Node(feature=1)
    for _ in range(500):
        chain = de.Node(1, chain)
    out, ok, words, meta, tape, consts = run_lowered(chain, ops, X, 7)
    assert meta["n_slots"] == 0 and len(words) == 500


# ---- superinstruction pass (csrc/de_bind.h fuse_tree): fused form must be a pure regrouping ------
BOP = dict(LOAD_ROW=0, LOAD_CONST=1, PUSH=2, CHECK_ROW=3, CHECK_ACC=4, BIN=5, UN=29, COUNT=48)
TOP = dict(LOADROW=48, LOADCONST_PUSH=52, UNROW=53, BINROWC=77, BIN2=89, COUNT=137)
MIRROR = {0: 0, 1: 2, 2: 1, 3: 3, 4: 5, 5: 4}


def unfuse(fused):
    """Expand fused instruction words back into the bound instruction stream they stand for."""
    out = []
    for top, arg, lo, hi in (tuple(int(v) for v in r) for r in fused):
        row = arg & 0xFFFFFF
        d = arg >> 24
        d = d - 256 if d >= 128 else d
        push_row = row + d
        if top < BOP["COUNT"]:
            out.append((top, arg, lo, hi))
        elif top < TOP["LOADCONST_PUSH"]:
            v = top - TOP["LOADROW"]
            if v & 2: out.append((BOP["PUSH"], push_row, 0, 0))
            if v & 1: out.append((BOP["CHECK_ROW"], row, 0, 0))
            out.append((BOP["LOAD_ROW"], row, 0, 0))
        elif top == TOP["LOADCONST_PUSH"]:
            out.append((BOP["PUSH"], arg, 0, 0))
            out.append((BOP["LOAD_CONST"], None, lo, hi))
        elif top < TOP["BINROWC"]:
            v = top - TOP["UNROW"]
            chk, push, outc, k = v & 1, (v >> 1) & 1, (v >> 2) & 1, v >> 3
            if push: out.append((BOP["PUSH"], push_row, 0, 0))
            if chk: out.append((BOP["CHECK_ROW"], row, 0, 0))
            out.append((BOP["UN"] + 4 * k + 2 + outc, row, 0, 0))
        elif top < TOP["BIN2"]:
            v = top - TOP["BINROWC"]
            out.append((BOP["CHECK_ROW"], row, 0, 0))
            out.append((BOP["BIN"] + 4 * (v >> 1) + (v & 1), row, 0, 0))
        else:
            v = top - TOP["BIN2"]
            push, outc, cst, k = v & 1, (v >> 1) & 1, (v >> 2) & 1, v >> 3
            if push: out.append((BOP["PUSH"], push_row, 0, 0))
            if cst:  # either LOAD_ROW a; BIN:const  or  LOAD_CONST c; BIN:row (mirrored) — both listed
                out.append(("bin2c", k, outc, row, lo, hi))
            else:
                lo_s = lo - (1 << 32) if lo >= (1 << 31) else lo
                out.append((BOP["LOAD_ROW"], row, 0, 0))
                out.append((BOP["BIN"] + 4 * k + outc, row + lo_s, 0, 0))
    return out


def same_stream(bound, expanded):
    """Compare, resolving the two spellings of a row-constant pair."""
    bound = [tuple(int(v) for v in r) for r in bound]
    i = 0
    for e in expanded:
        if e[0] == "bin2c":
            _, k, outc, row, lo, hi = e
            a, b = bound[i], bound[i + 1]
            as_row_first = (a[0] == BOP["LOAD_ROW"] and a[1] & 0xFFFFFF == row and
                            b[0] == BOP["BIN"] + 4 * k + 2 + outc and (b[2], b[3]) == (lo, hi))
            as_const_first = (a[0] == BOP["LOAD_CONST"] and (a[2], a[3]) == (lo, hi) and
                              b[0] == BOP["BIN"] + 4 * MIRROR[k] + outc and b[1] & 0xFFFFFF == row)
            if not (as_row_first or as_const_first):
                return False
            i += 2
            continue
        b = bound[i]
        if e[0] != b[0] or (e[1] is not None and e[1] != b[1]) or (e[2], e[3]) != (b[2], b[3]):
            return False
        i += 1
    return i == len(bound)


@pytest.mark.parametrize("options", [7, 1, 6, 0, 9])
def test_fused_program_is_a_regrouping_of_the_bound_program(options):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(300, seed=0xDE02)
    ops2 = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "pow_abs2"),
                           unary_operators=("cos", "exp", "sin", "safe_log", "neg", "square"))
    rng = de.synth.Xoshiro256ss(4)
    trees2 = [de.synth.gen_random_tree_fixed_size(3 + i % 30, ops2, 7, rng, np.float32) for i in range(300)]
    n_bound = n_fused = 0
    for tr, op, F in [(t, ops, 5) for t in trees] + [(t, ops2, 7) for t in trees2]:
        tape, consts = de.flatten(tr, op, np.float32)
        bound = api.lower_tape_stage(tape, consts, F, 2, options=options)
        fused = api.lower_tape_stage(tape, consts, F, 3, options=options)
        assert len(fused) <= len(bound)
        assert all(int(r[0]) < TOP["COUNT"] for r in fused)
        assert same_stream(bound, unfuse(fused)), de.string_tree(tr, op)
        n_bound += len(bound)
        n_fused += len(fused)
    if options == 7:
        assert n_fused < 0.85 * n_bound  # the pass pays: > 15 % fewer dispatches on random trees


def test_shared_subtrees_are_evaluated_like_their_expansion():
    """A GraphNode-style DAG (the same node object under two parents, src/Node.jl:138-166): the reference's
    `_eval_tree_array` ignores sharing and evaluates the shared node once per parent, so the
    flattener expands it; tape, constants and values equal those of the deep copy (`copy_node(...; break_sharing)`)."""
    ops = de.OperatorEnum(binary_operators=("+", "*"), unary_operators=("cos",))
    s = de.Node(1, de.Node(2, de.Node(feature=1), de.Node(val=0.75)))  # cos(x1 * 0.75)
    dag = de.Node(1, s, de.Node(2, s, s))                               # s + s * s
    tree = dag.copy()                                                   # sharing broken
    assert tree.children[0] is not tree.children[1].children[0]
    assert de.node.count_nodes(dag) == de.node.count_nodes(tree) == 14
    assert de.node.count_constant_nodes(dag) == 3
    tape_d, c_d = de.flatten(dag, ops, np.float64)
    tape_t, c_t = de.flatten(tree, ops, np.float64)
    np.testing.assert_array_equal(tape_d, tape_t)
    np.testing.assert_array_equal(c_d, c_t)
    X = np.asfortranarray(np.linspace(-2, 2, 33, dtype=np.float64)[None, :])
    y, ok = oracle.eval_tree_array(tape_d, c_d, X)
    c = np.cos(X[0] * 0.75)
    assert ok
    np.testing.assert_allclose(y, c + c * c, rtol=1e-15)


def test_valu_slot_table_layout():
    """tools/valu_slots.py restates the handler-id layout of csrc/de_bind.h in Python to name the handler functions in
    the disassembly; every id the fused (stage 3) program of the bench population dispatches must resolve to a handler of
    the shipped code object with a VALU slot count, and the committed table must agree with the current build."""
    import importlib.util
    import json
    obj = os.path.join(ROOT, "dynamicexpressions.jl_amd", "csrc", "_obj", "irp_de_kernels", "k.out")
    if not os.path.exists(obj):
        pytest.skip("device code object not built here (csrc/_obj is a build artefact)")
    spec = importlib.util.spec_from_file_location("valu_slots", os.path.join(ROOT, "tools", "valu_slots.py"))
    vs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vs)
    slots, counts = vs.table(obj, "float")
    assert len(slots) == counts["TOPX_COUNT"]  # every id has a function in the code object
    ops = de.synth.BENCH_OPERATORS
    used = set()
    for tree in de.synth.random_population(300, seed=0xDE02):
        tape, consts = de.flatten(tree, ops, np.float32)
        w = api.lower_tape_stage(tape, consts, 5, 3)
        used.update(int(v) for v in w[:, 0])
    assert used and all(u in slots for u in used)
    # acc + row: the LDS address add (SGPR operand: half rate) + two v_pk_add_f32 per plane (csrc/de_kernels.h DE_TG: 2 planes, or 1 in an A/B build)
    assert any(slots[5]["valu_cycles"] == pytest.approx((1 + 2 * planes) * 3.7) for planes in (1, 2))
    committed = os.path.join(ROOT, "profiles", "valu_slots.json")
    if os.path.exists(committed):
        with open(committed) as fh:
            tab = json.load(fh)["handlers"]
        stale = [k for k in used if tab.get(str(k), {}).get("valu_cycles") != slots[k]["valu_cycles"]]
        assert not stale, f"profiles/valu_slots.json is stale for handlers {stale}: rerun tools/valu_slots.py"


def random_graph(rng, ops, n_nodes, n_features, n_shares, dtype=np.float64):
    """A random GraphNode DAG: a random tree in which `n_shares` leaves are replaced by references to operator subtrees
    that already exist elsewhere in the tree (never an ancestor: the result stays acyclic)."""
    G = de.GraphNode
    tree = de.synth.gen_random_tree_fixed_size(n_nodes, ops, n_features, rng, dtype, node_type=G)
    for _ in range(n_shares):
        nodes = list(de.postorder(tree))
        inner = [n for n in nodes if n.degree > 0 and n is not tree]
        parents = [n for n in nodes if n.degree > 0]
        if not inner or not parents:
            break
        target = inner[rng.randint(len(inner)) - 1]
        below = {id(m) for m in de.postorder(target)}
        # a parent outside the target's subtree whose child slot holds a leaf
        cands = [(p, k) for p in parents if id(p) not in below for k, c in enumerate(p.children) if c.degree == 0]
        cands = [(p, k) for p, k in cands if not _is_ancestor(p, target)]
        if not cands:
            continue
        p, k = cands[rng.randint(len(cands)) - 1]
        ch = list(p.children)
        ch[k] = target
        p.children = tuple(ch)
    return tree


def _is_ancestor(p, target):
    """Would making `target` a child of `p` create a cycle?  (p reachable from target)"""
    return any(m is p for m in de.postorder(target))


@pytest.mark.parametrize("options", [7, 6, 1, 0, 15])
def test_cse_tapes_of_graph_nodes_lower_to_the_values_and_flags_of_the_expanded_tree(options):
    """GraphNode sharing (src/Node.jl:138-166): the CSE tape (shared subtree once + DE_OP_SHARE, then
    DE_LEAF_SHARED) must lower to a program that gives, on the numpy model of the accumulator machine, exactly the values
    AND the flag the oracle gives for the EXPANDED tree — the reference evaluates a shared node once per parent — with
    fewer instructions."""
    ops = de.OperatorEnum(binary_operators=("+", "-", "*", "/"), unary_operators=("cos", "exp", "safe_log", "square"))
    rng = de.synth.Xoshiro256ss(4242 + options)
    X = np.asfortranarray(de.synth.random_X(3, 64, seed=9, dtype=np.float64))
    X[1, 7] = np.inf
    n_cse = saved = 0
    for it in range(300):
        g = random_graph(rng, ops, 6 + it % 22, 3, 1 + it % 4)
        tape, consts, cse, occ = de.flatten_graph(g, ops, np.float64)
        # constants: one slot per occurrence in the expanded tape; a shared constant node owns several
        assert len(consts) == len(occ) and de.count_constant_nodes(g) == (len(np.unique(occ)) if len(occ) else 0)
        y, ok = oracle.eval_tree_array(tape, consts, X, options, elementwise=True)
        w_exp, _ = api.lower_tape(tape, consts, 3, 0, options, np.float64)
        if cse is None:
            continue
        n_cse += 1
        w_cse, meta = api.lower_tape(cse, consts, 3, 0, options, np.float64)
        out, ok_c = prog_interp.run(w_cse, X, bool(options & 1), host_ok=meta["host_ok_eval"])
        ref, ok_e = prog_interp.run(w_exp, X, bool(options & 1), host_ok=meta["host_ok_eval"])
        assert ok_c == ok == ok_e, (it, de.string_tree(g, ops))  # the reference's flag (oracle, expanded tree)
        if ok or not (options & 1):
            # the same machine model, the same libm: the CSE program reproduces the expanded program bit for bit ...
            m = ~(np.isnan(ref) & np.isnan(out))
            np.testing.assert_array_equal(out[m], ref[m], err_msg=de.string_tree(g, ops))
            # (the expanded program against the oracle's values is the subject of the tests above)
        assert len(w_cse) <= len(w_exp)
        saved += len(w_exp) - len(w_cse)
    assert n_cse > 100 and saved > 2 * n_cse  # sharing really shortens the programs


def test_cse_tape_validation():
    G = de.GraphNode
    ops = de.OperatorEnum(binary_operators=("+", "*"), unary_operators=("cos",))
    T = de.node.TAPE_DTYPE
    c = np.zeros(0)
    ok_tape = np.array([(0, 1, 0), (1, 20, 0), (1, 0xFE, 0), (0, 3, 0), (2, 64, 0)], dtype=T)   # cos(x1){0} + {0}
    w, _ = api.lower_tape(ok_tape, c, 1, 0, 7, np.float64)
    assert len(w) == 2  # cos(row) [pushed to the share row by the next instruction], + share row
    for bad in ([(0, 3, 0), (1, 20, 0)],                                   # reference before definition
                [(0, 1, 0), (1, 0xFE, 0), (1, 20, 0)],                      # a leaf cannot be shared
                [(0, 1, 0), (1, 20, 0), (1, 0xFE, 1), (0, 3, 1), (2, 64, 0)],  # ids must start at 0
                [(0, 1, 0), (1, 20, 0), (1, 0xFE, 0)]):                     # shared root
        with pytest.raises(ValueError):
            api.lower_tape(np.array(bad, dtype=T), c, 1, 0, 7, np.float64)
    # a plain (non-CSE) program must keep rejecting the markers
    x = G(feature=1)
    s = G(1, x)
    tree = G(1, s, s)
    assert de.flatten_graph(tree, ops, np.float64)[2] is not None
