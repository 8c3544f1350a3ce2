"""Round 6: strict flags (the certificate behind every evaluation), the cached certificate program, the unpatched toolchain build."""
import os
import subprocess
import sys

import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    return _api


def _flags_of_oracle(trees, ops, X, dtype):
    el, sm = [], []
    for t in trees:
        tape, consts = de.flatten(t, ops, dtype)
        el.append(oracle.eval_tree_array(tape, consts, X, 7, elementwise=True)[1])
        sm.append(oracle.eval_tree_array(tape, consts, X, 7, elementwise=False)[1])
    return np.array(el), np.array(sm)


def test_strict_flags_lists_the_trees_whose_flag_is_not_provably_the_references(api):
    """EvalContext(strict_flags=True): Population.eval runs de_eval_sum_certificate behind the evaluation and
    `uncertified` lists the trees whose element-wise flag is not provably the reference's isfinite(sum(x)) flag (src/ValueInterface.jl:9).
    Outside that list `ok` IS the oracle's SUM flavour — the reference's own bit; the one-tree sugar raises UncertifiedFlag for a tree
    in it (the caller keeps the CPU path)."""
    dtype = np.float32
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*"), unary_operators=("cos", "exp"))
    N = 4096
    big = 3e34   # N * big overflows, big does not
    x1 = de.Node(feature=1)
    mul, div, add = ops.index("*", 2), ops.index("/", 2), ops.index("+", 2)
    quirk = de.Node(div, de.Node(mul, de.Node(add, x1, de.Node(val=2.5)), de.Node(val=big)), de.Node(val=big))
    trees = [quirk] + de.synth.random_population(80, seed=0xC0DF, dtype=dtype)
    g = np.random.Generator(np.random.PCG64(4))
    X = np.asfortranarray(np.abs(g.standard_normal((5, N))).astype(dtype) + dtype(0.5))
    el, sm = _flags_of_oracle(trees, ops, X, dtype)
    assert el[0] and not sm[0]
    pop = api.Population(trees, ops, dtype, n_features=5, eval_context=api.EvalContext(strict_flags=True))
    _, ok = pop.eval(X)
    ok = np.asarray(ok, dtype=bool)
    unc = set(int(i) for i in pop.uncertified)
    assert 0 in unc, "the constructed quirk tree cannot be certified"
    cert = np.array([i not in unc for i in range(len(trees))])
    assert np.array_equal(ok[cert], sm[cert]), "every certified flag is the reference's (sum flavour) flag"
    assert np.array_equal(ok, el)
    assert cert.mean() > 0.9
    print(f"[strict flags] {len(unc)} of {len(trees)} trees uncertified; quirk trees in the oracle: {int((el != sm).sum())}")
    pop.close()
    # the sugar: a certified tree evaluates, the quirk tree raises
    y, k = api.eval_tree_array(trees[1], X, ops, eval_context=api.EvalContext(strict_flags=True))
    assert k == sm[1]
    with pytest.raises(api.UncertifiedFlag):
        api.eval_tree_array(quirk, X, ops, eval_context=api.EvalContext(strict_flags=True))
    # torch inputs take the same path
    import torch
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda().t()
    pop = api.Population(trees, ops, dtype, n_features=5, eval_context=api.EvalContext(strict_flags=True))
    _, okd = pop.eval(Xd)
    assert np.array_equal(okd.cpu().numpy().astype(bool), el) and 0 in set(int(i) for i in pop.uncertified)
    pop.close()


def test_certificate_program_is_cached_per_constants_generation(api):
    """the certificate program is built once per set of constants — a second call reuses it (same answers), and
    de_program_set_consts invalidates it (the largest |constant operand| of a tree is part of the certificate)."""
    dtype = np.float32
    ops = de.synth.BENCH_OPERATORS
    x1 = de.Node(feature=1)
    add, mul = ops.index("+", 2), ops.index("*", 2)
    t0 = de.Node(mul, de.Node(add, x1, de.Node(val=1.5)), de.Node(val=2.0))
    trees = [t0] + de.synth.random_population(40, seed=0xCE27, dtype=dtype)
    X = np.asfortranarray(np.abs(np.random.Generator(np.random.PCG64(5)).standard_normal((5, 2048))).astype(dtype) + dtype(0.5))
    pop = api.Population(trees, ops, dtype, n_features=5)
    a = pop.sum_certificate(X)
    b = pop.sum_certificate(X)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    assert a[1][0] and a[2][0] < 100
    consts = np.concatenate([de.flatten(t, ops, dtype)[1] for t in trees]).astype(dtype)
    c2 = consts.copy()
    c2[1] = 3e36  # t0's second constant: N * |x * 3e36| overflows Float32 although every element is finite
    pop.set_constants(c2)
    ok2, cert2, mx2 = pop.sum_certificate(X)
    assert ok2[0] and not cert2[0] and mx2[0] >= 3e36, (ok2[0], cert2[0], mx2[0])
    pop.set_constants(consts)
    c = pop.sum_certificate(X)
    for u, v in zip(a, c):
        assert np.array_equal(u, v)
    pop.close()


def test_plain_build_without_the_assembly_and_object_passes_passes_the_parity_tests():
    """the toolchain escape hatch is PROVEN, not asserted.  `DE_PLAIN_BUILD=1 bash csrc/build.sh` (no asmopt.py peephole
    pass, no asmpatch.py instruction-word rewrite; made by __graft_entry__.build() as csrc/libde_hip_plain.so) must pass the golden known
    answers, the random-population comparison with the oracle, the bit-identity test of the IEEE-exact operators and a gradient + fused-loss
    file through DE_HIP_LIB — and return the SAME BITS as the shipped library on a population that exits early."""
    plain = os.path.join(ROOT, "dynamicexpressions.jl_amd", "csrc", "libde_hip_plain.so")
    assert os.path.exists(plain), "csrc/libde_hip_plain.so is missing: run __graft_entry__.build()"
    env = dict(os.environ, DE_HIP_LIB=plain)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_eval.py"), os.path.join(ROOT, "tests", "test_gpu_grad.py"),
                        "-k", "golden or random_population or ieee_exact or chunk_and_tile or gradient_modes or bit_identical"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, f"parity tests against the plain build failed:\n{tail}\n{r.stderr[-2000:]}"
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 100, tail
    print(f"[plain build] {m.group(1)} parity tests passed against csrc/libde_hip_plain.so (DE_ASMOPT=0 DE_NO_ASMPATCH=1)")
    # same bits as the shipped library (rows of complete trees, flags, fused loss), each library in a process of its own
    code = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, %r)
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
trees = de.synth.random_population(300, seed=0x91A1)
X = torch.from_numpy(np.ascontiguousarray(de.synth.random_X(5, 70001, seed=6).T)).cuda().t()
pop = api.Population(trees, de.synth.BENCH_OPERATORS, np.float32, n_features=5)
out, ok = pop.eval(X)
torch.cuda.synchronize()
ok = ok.cpu().numpy(); out = out.cpu().numpy()
y = torch.from_numpy(de.synth.random_X(1, 70001, seed=7)[0].copy()).cuda()
loss, ok2 = pop.eval_loss(X, y)
h = hashlib.sha256(); h.update(ok.tobytes()); h.update(out[ok.astype(bool)].tobytes()); h.update(np.asarray(loss.cpu())[ok.astype(bool)].tobytes())
print("HASH", h.hexdigest(), int(ok.sum()))
""" % ROOT
    hashes = []
    for lib in (None, plain):
        e = dict(os.environ)
        e.pop("DE_HIP_LIB", None)
        if lib:
            e["DE_HIP_LIB"] = lib
        rr = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert rr.returncode == 0, rr.stderr[-2000:]
        hashes.append([ln for ln in rr.stdout.splitlines() if ln.startswith("HASH")][0])
    assert hashes[0] == hashes[1], hashes
    assert 0 < int(hashes[0].split()[2]) < 300


def _fold_routes(api, trees, ops, dtype, X, monkeypatch, consts=None):
    """rows / flags / stream hash of the population under the three folding routes of constant subtrees: default (IEEE-exact subtrees on the
    host, the others by de_fold_kernel), everything by de_fold_kernel, everything through the auxiliary program of rounds 1-5"""
    res = {}
    for name, env in (("default", {}), ("kernel", {"DE_NO_HOST_FOLD": "1"}), ("aux", {"DE_NO_HOST_FOLD": "1", "DE_NO_KERNEL_FOLD": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pop = api.Population(trees, ops, dtype, n_features=X.shape[0])
        if consts is not None:
            pop.set_constants(consts)
        out, ok = pop.eval(X)
        res[name] = (np.asarray(out), np.asarray(ok, dtype=bool), pop.stream_hash())
        pop.close()
        for k in env:
            monkeypatch.delenv(k)
    return res


def _same_bits(a, b, dtype, what):
    u = np.uint32 if np.dtype(dtype) == np.float32 else np.uint64
    assert np.array_equal(a[1], b[1]), f"{what}: flags"
    fin = np.isfinite(a[0]) | np.isfinite(b[0])       # (NaN sign / payload is not compared: x86 0/0 = -NaN, gfx950 +NaN)
    assert np.array_equal(a[0].view(u)[fin], b[0].view(u)[fin]), f"{what}: rows"
    assert np.array_equal(np.isnan(a[0]), np.isnan(b[0])), f"{what}: NaN positions"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_host_and_kernel_folded_constant_subtrees_have_the_bits_of_the_auxiliary_program(api, dtype, monkeypatch):
    """Round 6: constant subtrees made of + - * / only are folded ON THE HOST at de_program_create /
    de_program_set_consts, the others by de_fold_kernel (one thread per subtree, the operators' own device code) — the auxiliary PROGRAM
    of rounds 1-5 (a second population lowered, bound, threaded and evaluated with N = 1) only remains for turbo programs.  Same bits by
    construction — checked: rows and flags under the three routes on trees whose constant subtrees overflow, underflow into subnormals,
    divide by zero and produce NaN; then again after new constants."""
    ops = de.synth.BENCH_OPERATORS
    rng = np.random.Generator(np.random.PCG64(17))
    add, sub, mul, div = (ops.index(s, 2) for s in "+-*/")
    cos, exp = ops.index("cos", 1), ops.index("exp", 1)
    tiny, huge = (1e-30, 1e30) if dtype == np.float32 else (1e-200, 1e200)
    pool = [0.0, -0.0, 1.0, -2.5, 3.0, tiny, -tiny, huge, -huge, tiny * 1e-8, 0.1, 7.0, 1.0 / 3.0, 1e5, 88.0, -104.0]

    def const_subtree(depth, unary=False):
        if depth == 0 or rng.random() < 0.3:
            return de.Node(val=float(pool[rng.integers(len(pool))]) if rng.random() < 0.7 else float(rng.standard_normal()))
        if unary and rng.random() < 0.4:
            return de.Node([cos, exp][rng.integers(2)], const_subtree(depth - 1, unary))
        return de.Node([add, sub, mul, div][rng.integers(4)], const_subtree(depth - 1, unary), const_subtree(depth - 1, unary))

    trees = []
    for k in range(300):
        x = de.Node(feature=int(rng.integers(1, 6)))
        c = const_subtree(int(rng.integers(1, 4)))
        if c.degree == 0:
            c = de.Node(mul, c, de.Node(val=2.0))
        inner = const_subtree(3, unary=True) if k % 2 == 0 else const_subtree(2)   # (a subtree with cos / exp cannot be folded on the host)
        trees.append(de.Node([add, mul, sub, div][k % 4], de.Node(add, x, inner), c))
    X = np.asfortranarray(rng.standard_normal((5, 700)).astype(dtype))
    r = _fold_routes(api, trees, ops, dtype, X, monkeypatch)
    assert r["default"][1].any() and not r["default"][1].all()
    _same_bits(r["default"], r["aux"], dtype, "host + kernel against the auxiliary program")
    _same_bits(r["kernel"], r["aux"], dtype, "kernel against the auxiliary program")
    assert len({r[k][2] for k in r}) == 3, "the three programs differ in WHERE they fold (the hash covers the fold tables)"
    # ... and through de_program_set_consts
    consts = np.concatenate([de.flatten(t, ops, dtype)[1] for t in trees]).astype(dtype)
    c2 = (consts * dtype(1.5) + dtype(0.25)).astype(dtype)
    r2 = _fold_routes(api, trees, ops, dtype, X, monkeypatch, consts=c2)
    _same_bits(r2["default"], r2["aux"], dtype, "after set_constants: host + kernel against the auxiliary program")
    _same_bits(r2["kernel"], r2["aux"], dtype, "after set_constants: kernel against the auxiliary program")
    print(f"[constant folds {np.dtype(dtype).name}] {int(r['default'][1].sum())} of {len(trees)} trees complete; rows and flags bit-equal under host + kernel folding, "
          "kernel folding and the auxiliary program, before and after set_constants")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_kernel_folded_subtrees_of_every_operator_have_the_bits_of_the_auxiliary_program(api, dtype, monkeypatch):
    """de_fold_kernel evaluates EVERY operator of the table (hot ones through the flat-switch kernel's code, the others through cold_op /
    cold_op3): constant subtrees over the wide operator set + the ternary operators, against the auxiliary program, bit for bit; a subtree
    too deep for the kernel's stack (DE_FOLD_STACK = 16) stays with the auxiliary program and still evaluates."""
    import fuzzlib as FZ
    ops = de.OperatorEnum(binary_operators=FZ.OPS_WIDE.ops[1] + ("mod", "rem", "greater"), unary_operators=FZ.OPS_WIDE.ops[0] + ("sign", "round", "floor", "ceil", "inv", "sqrt", "cbrt", "exp2", "log", "log2", "log10", "log1p", "tan", "sinh", "cosh", "asin", "acos", "asinh", "acosh", "atanh", "safe_log2", "safe_log10", "safe_log1p", "safe_acosh", "gamma"),
                          ternary_operators=("fma", "clamp", "+", "max"))
    rng = de.synth.Xoshiro256ss(0xF01D)
    g = np.random.Generator(np.random.PCG64(23))
    trees = []
    add = ops.index("+", 2)
    for k in range(400):
        sub = FZ.gen_mixed_arity_tree(rng, ops, 1, dtype, 3 + k % 3)
        # make the subtree constant: every feature leaf becomes a constant
        def constify(n):
            if n.degree == 0:
                return n if n.constant else de.Node(val=float(g.standard_normal() * (10.0 if k % 7 == 0 else 1.0)))
            return de.Node(n.op, *[constify(c) for c in n.children])
        trees.append(de.Node(add, de.Node(feature=1 + k % 3), constify(sub)))
    # one subtree deeper than the kernel's stack: a right-leaning chain of 20 additions
    deep = de.Node(val=1.0)
    for i in range(20):
        deep = de.Node(add, de.Node(val=float(i)), de.Node(ops.index("cos", 1), deep)) if i % 2 else de.Node(add, de.Node(val=float(i)), deep)
    left = de.Node(val=0.5)
    for i in range(19):   # left operands pile up: stack depth 20
        left = de.Node(add, de.Node(ops.index("cos", 1), de.Node(val=float(i))), left)
    trees.append(de.Node(add, de.Node(feature=1), deep))
    trees.append(de.Node(add, de.Node(feature=2), left))
    X = np.asfortranarray(g.standard_normal((3, 300)).astype(dtype))
    r = _fold_routes(api, trees, ops, dtype, X, monkeypatch)
    assert r["default"][1].sum() > 50
    _same_bits(r["default"], r["aux"], dtype, "wide operator set: host + kernel against the auxiliary program")
    _same_bits(r["kernel"], r["aux"], dtype, "wide operator set: kernel against the auxiliary program")
    print(f"[constant folds, every operator, {np.dtype(dtype).name}] {int(r['default'][1].sum())} of {len(trees)} trees complete; bit-equal to the auxiliary program")


def test_ctx_trim_releases_the_retained_buffers_and_the_context_stays_usable(api):
    """de_ctx_trim: the parked host vectors and recycled device buffers of destroyed programs are freed; programs created
    afterwards build, byte for byte, what a fresh context builds."""
    import torch
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(3000, seed=0x7219)
    ctx = api.Context(0)
    X = np.asfortranarray(de.synth.random_X(5, 500, seed=8))
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
        out0, ok0 = pop.eval(X)
        h0 = pop.stream_hash()
        pop.close()
    ctx.trim()
    pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
    out1, ok1 = pop.eval(X)
    assert pop.stream_hash() == h0 and np.array_equal(ok0, ok1)
    assert np.array_equal(np.asarray(out0)[ok0].view(np.uint32), np.asarray(out1)[ok1].view(np.uint32))
    pop.close()
    ctx.trim()
    ctx.close()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)


# ---- wave groups (csrc/de_api_program.cpp choose_waves, de_kernels.hip KArgs::var_stride) ----------------------------------------------------
def _wave_case(kind, dtype, g):
    """(trees, operators, X, params, classes, P): a population whose one-wave workgroup is short of resident waves."""
    ops = de.synth.BENCH_OPERATORS
    if kind == "parametric":
        F, P, C = 5, 8, 11
        trees = de.synth.random_population(330, seed=0x3A7E, dtype=dtype, node_type=de.ParametricNode, nparams=P)
    else:  # a wide feature matrix
        F, P, C = 20, 0, 0
        trees = de.synth.random_population(330, seed=0x3A7F, dtype=dtype, nfeatures=F)
    N = 150_001  # >= 512 sample tiles (priority tiles, probe launch, compaction of the live trees) and a ragged last tile
    X = np.asfortranarray((g.standard_normal((F, N)) * 1.5).astype(dtype))
    params = np.asfortranarray((g.standard_normal((P, C)) * 2).astype(dtype)) if P else None
    classes = g.integers(1, C + 1, N).astype(np.int64) if P else None
    return trees, ops, X, params, classes, P


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["parametric", "wide X"])
def test_wave_groups_have_the_bits_and_flags_of_one_wave_workgroups(api, kind, dtype, monkeypatch):
    """A population with staged parameter rows, or with many features, runs 2 or 4 waves per workgroup on one sample tile (X and the parameter
    rows staged once; a chunk, spill-slot rows, a live-tree list and a stream variant per wave).  Values, flags, fused losses: the bits of the
    one-wave workgroups of rounds 1-5 (DE_EVAL_WAVES=1) — with the early exit's probe launch and compaction, and without (full evaluation);
    and after de_program_set_consts (every variant of the stream carries the new immediates)."""
    g = np.random.Generator(np.random.PCG64(99))
    trees, ops, X, params, classes, P = _wave_case(kind, dtype, g)
    y = g.standard_normal(X.shape[1]).astype(dtype)
    kw = dict(params=params, classes=classes) if P else {}
    results = {}
    for waves in ("1", "2", "4", "8", None):
        if waves is None:
            monkeypatch.delenv("DE_EVAL_WAVES", raising=False)
        else:
            monkeypatch.setenv("DE_EVAL_WAVES", waves)
        got = []
        for ec in (api.EvalContext(), api.EvalContext(early_exit=False)):
            pop = api.Population(trees, ops, dtype, n_features=X.shape[0], n_params=P, eval_context=ec)
            pop.verify()  # (the sanitizer walks every variant of the stream)
            assert pop.meta(0)["waves"] == (int(waves) if waves else 4)  # (8 parameter rows / 20 features: the rule stops at four waves)
            out, ok = pop.eval(X, **kw)
            loss, ok_l = pop.eval_loss(X, y, **kw)
            got.append((np.asarray(out), np.asarray(ok), np.asarray(loss), np.asarray(ok_l)))
            if ec.early_exit:  # new constants: patched in place in every variant of the stream
                consts = np.concatenate([de.flatten(t, ops, dtype)[1] for t in trees]).astype(dtype)
                pop.set_constants(consts * dtype(1.25) - dtype(0.5))
                out2, ok2 = pop.eval(X, **kw)
                got.append((np.asarray(out2), np.asarray(ok2), None, None))
            pop.close()
        results[waves] = got
    ref = results["1"]
    assert ref[0][1].sum() > 100 and (ref[0][1] == 0).sum() > 30, "the case needs complete and incomplete trees"
    for waves, got in results.items():
        if waves == "1":
            continue
        assert len(got) == len(ref)
        for (o, k, l, kl), (ro, rk, rl, rkl) in zip(got, ref):
            assert np.array_equal(k, rk), f"waves {waves}: flags"
            live = rk != 0
            assert o[live].tobytes() == ro[live].tobytes(), f"waves {waves}: values of the complete trees"
            if l is not None:
                assert np.array_equal(kl, rkl) and l[rkl != 0].tobytes() == rl[rkl != 0].tobytes(), f"waves {waves}: fused loss"


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("kind", ["graph", "ternary", "turbo"])
def test_wave_groups_with_shared_rows_ternary_operators_and_turbo(api, kind, dtype, monkeypatch):
    """The slot rows a stream variant moves are not only spill slots: a GraphNode's persistent rows (one evaluation of a shared subtree,
    several readers), the two popped operands of a ternary operator; and `turbo` programs take the same route with their own handlers.
    20 features make every one of these populations run four waves per workgroup: bits and flags of DE_EVAL_WAVES=1, and the oracle's
    flags on the way."""
    import fuzzlib as FZ
    from test_lowering import random_graph
    F, N = 20, 70_001
    g = np.random.Generator(np.random.PCG64(321))
    rng = de.synth.Xoshiro256ss(4711)
    ec = api.EvalContext(turbo=True) if kind == "turbo" else api.EvalContext()
    if kind == "turbo" and dtype == np.float64:
        pytest.skip("turbo is a Float32 mode")
    if kind == "graph":
        ops = de.OperatorEnum(binary_operators=("+", "-", "*", "/"), unary_operators=("cos", "exp", "safe_log", "square"))
        trees = [random_graph(rng, ops, 6 + i % 24, F, 1 + i % 4, dtype) for i in range(300)]
        assert sum(de.flatten_graph(t, ops, dtype)[2] is not None for t in trees) > 100
    elif kind == "ternary":
        ops = de.OperatorEnum(unary_operators=("abs", "cos", "exp"), binary_operators=("+", "-", "*", "/"), ternary_operators=("fma", "clamp", "+", "max"))
        trees = [FZ.gen_mixed_arity_tree(rng, ops, F, dtype, 12) for _ in range(400)]
        trees = [t for t in trees if 3 <= de.count_nodes(t) <= 200][:300]
        assert sum(1 for t in trees for n in de.node.postorder(t) if n.degree == 3) > 100
    else:
        ops = de.synth.BENCH_OPERATORS
        trees = de.synth.random_population(300, seed=0x7B0, dtype=dtype, nfeatures=F)
    X = np.asfortranarray((g.standard_normal((F, N)) * 1.2).astype(dtype))
    got = {}
    for waves in ("1", None):
        if waves is None:
            monkeypatch.delenv("DE_EVAL_WAVES", raising=False)
        else:
            monkeypatch.setenv("DE_EVAL_WAVES", waves)
        pop = api.Population(trees, ops, dtype, n_features=F, eval_context=ec)
        pop.verify()
        out, ok = pop.eval(X)
        got[waves] = (np.asarray(out), np.asarray(ok), pop.meta(0)["waves"])
        pop.close()
    (o1, k1, w1), (o4, k4, w4) = got["1"], got[None]
    assert w1 == 1 and w4 in (2, 4, 8), f"the population was meant to run in wave groups: {w1} {w4}"
    print(f"[wave groups, {kind}] {w4} waves per workgroup")
    assert np.array_equal(k1, k4)
    live = k1 != 0
    assert live.sum() > 50
    assert o1[live].tobytes() == o4[live].tobytes()
    if kind != "turbo":  # the flags are the oracle's (element-wise flavour) on a prefix the oracle finishes quickly
        n = 4096
        pop = api.Population(trees, ops, dtype, n_features=F, eval_context=ec)
        _, kk = pop.eval(X[:, :n])
        pop.close()
        want = []
        for t in trees:
            tape, consts = de.flatten(de.break_sharing(t) if kind == "graph" else t, ops, dtype)
            want.append(oracle.eval_tree_array(tape, consts, X[:, :n], 7, elementwise=True)[1])
        assert np.array_equal(np.asarray(kk, dtype=bool), np.array(want, dtype=bool))


@pytest.mark.parametrize("F", [40, 60, 120])
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_wide_feature_matrices_run_the_threaded_kernel_in_wave_groups(api, dtype, F):
    """Up to round 6 a feature matrix of more than 35 rows went to the flat-switch kernel's `direct` variant (features gathered from global
    memory: 14 - 20 ms per 10^6 samples x 1000 trees against 1.4 ms at 30 features).  The threaded kernel's rows are a quarter as long and a
    wave group shares them: it stages X up to ~140 rows.  Parity with the oracle (values by the tolerance model, flags exactly), the fused
    loss — refused for such programs before — against the values, and the same bits under DE_EVAL_WAVES=1."""
    from test_gpu_eval import compare_population
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(120, seed=0xF00 + F, dtype=dtype, nfeatures=F)
    g = np.random.Generator(np.random.PCG64(F))
    X = np.asfortranarray((g.standard_normal((F, 2500)) * 1.3).astype(dtype))
    for ec in (api.EvalContext(), api.EvalContext(early_exit=False)):
        compare_population(api, trees, ops, X, dtype, eval_context=ec, min_ok=30, label=f"wide X, F = {F}, ee={ec.early_exit}")
    pop = api.Population(trees, ops, dtype, n_features=F)
    out, ok = pop.eval(X)
    assert pop.ctx.last_kernel_name() == "de_eval_threaded_kernel" and pop.meta(0)["waves"] == (4 if F == 40 else 8), (pop.ctx.last_kernel_name(), pop.meta(0))
    y = g.standard_normal(X.shape[1]).astype(dtype)
    loss, ok_l = pop.eval_loss(X, y)
    assert np.array_equal(np.asarray(ok_l), np.asarray(ok))
    live = np.asarray(ok, dtype=bool)
    want = ((np.asarray(out)[live].astype(np.float64) - y.astype(np.float64)) ** 2).sum(axis=1)
    got = np.asarray(loss)[live].astype(np.float64)
    inr = want < (1e36 if dtype == np.float32 else 1e300)  # (a square beyond the element type's range is Inf on the device, finite in this Float64 sum)
    assert inr.sum() > 20
    np.testing.assert_allclose(got[inr], want[inr], rtol=2e-4 if dtype == np.float32 else 1e-11)
    pop.close()
    os.environ["DE_EVAL_WAVES"] = "1"
    try:
        pop1 = api.Population(trees, ops, dtype, n_features=F)
        out1, ok1 = pop1.eval(X)
        pop1.close()
    finally:
        del os.environ["DE_EVAL_WAVES"]
    assert np.array_equal(np.asarray(ok1), np.asarray(ok)) and np.asarray(out1)[live].tobytes() == np.asarray(out)[live].tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_gradient_kernels_share_the_leaf_rows_of_a_wide_feature_matrix(api, dtype, monkeypatch):
    """From 16 leaf rows on the four waves of a forward-dual workgroup run different trees on the SAME samples: X is staged once per workgroup
    instead of once per wave, every wave keeps its own slot rows and reads the stream variant that names them (csrc/de_api_grad.cpp
    ensure_grad_threaded, GradArgs::gt_share).  Jacobians, fused losses and loss gradients have the bits of one copy per wave
    (DE_GRAD_SHARE=0) — one and two samples per lane, a ragged last tile, new constants —, the parametric case too when forced; and the
    Jacobians stay inside the oracle's bounds."""
    from test_gpu_grad import grad_compare
    ops = de.synth.BENCH_OPERATORS
    g = np.random.Generator(np.random.PCG64(2718))

    def run(trees, F, P, N, share, new_consts=False):
        monkeypatch.setenv("DE_GRAD_SHARE", share)
        X = np.asfortranarray((np.random.Generator(np.random.PCG64(N)).standard_normal((F, N)) * 1.3).astype(dtype))
        y = np.cos(np.arange(N)).astype(dtype)
        gg = np.random.Generator(np.random.PCG64(N + 1))
        kw = dict(params=np.asfortranarray((gg.standard_normal((P, 7)) * 2).astype(dtype)), classes=gg.integers(1, 8, N).astype(np.int64)) if P else {}
        pop = api.Population(trees, ops, dtype, n_features=F, n_params=P)
        res = []
        for variable in ((False, "both") if P else (False, True)):
            out, grads, ok = pop.eval_grad(X, variable=variable, **kw)
            res.append((np.asarray(ok), [np.asarray(a) for a in grads], np.asarray(out)))
            l, dl, okl = pop.eval_loss_grad(X, y, variable=variable, **kw)
            res.append((np.asarray(okl), [np.asarray(a) for a in dl], np.asarray(l)))
        if new_consts:  # (the immediates of every variant are patched in place)
            consts = np.concatenate([de.flatten(t, ops, dtype)[1] for t in trees]).astype(dtype)
            pop.set_constants(consts * dtype(0.75) + dtype(0.25))
            l, dl, okl = pop.eval_loss_grad(X, y, variable=False, **kw)
            res.append((np.asarray(okl), [np.asarray(a) for a in dl], np.asarray(l)))
        pop.close()
        return res

    cases = [("wide X", de.synth.random_population(200, seed=0x51A, dtype=dtype, nfeatures=24), 24, 0),
             ("parametric", de.synth.random_population(200, seed=0x51B, dtype=dtype, node_type=de.ParametricNode, nparams=8), 5, 8)]
    for name, trees, F, P in cases:
        for N in (70_001, 777):  # two samples per lane (Float32, >= 65536 samples) and one; ragged last tiles
            a, b = run(trees, F, P, N, "0", new_consts=True), run(trees, F, P, N, "1", new_consts=True)
            assert len(a) == len(b) == 5
            for (k0, g0, o0), (k1, g1, o1) in zip(a, b):
                assert np.array_equal(k0, k1), f"{name} N={N}: flags"
                live = k0 != 0
                assert live.sum() > 50
                assert o0[live].tobytes() == o1[live].tobytes(), f"{name} N={N}: values / losses"
                assert all(x.tobytes() == y_.tobytes() for x, y_, l in zip(g0, g1, live) if l), f"{name} N={N}: gradient rows"
    monkeypatch.delenv("DE_GRAD_SHARE")  # the rule: 24 leaf rows share
    X = np.asfortranarray((g.standard_normal((24, 1500)) * 1.2).astype(dtype))
    for mode in ("constant", "variable"):
        assert grad_compare(api, cases[0][1][:80], ops, X, dtype, mode) > 20
