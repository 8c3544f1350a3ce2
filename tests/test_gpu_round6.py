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


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_host_folded_constant_subtrees_have_the_bits_of_the_device_folded_ones(api, dtype, monkeypatch):
    """Round 6: constant subtrees made of + - * / only are folded ON THE HOST at de_program_create /
    de_program_set_consts, the others through the auxiliary device program.  Same bits by construction — checked: rows, flags and the
    lowered streams' immediates against DE_NO_HOST_FOLD=1 (everything on the device) on trees whose constant subtrees overflow, underflow
    into subnormals, divide by zero and produce NaN; then again after new constants."""
    ops = de.synth.BENCH_OPERATORS
    rng = np.random.Generator(np.random.PCG64(17))
    add, sub, mul, div = (ops.index(s, 2) for s in "+-*/")
    cos = ops.index("cos", 1)
    tiny, huge = (1e-30, 1e30) if dtype == np.float32 else (1e-200, 1e200)
    pool = [0.0, -0.0, 1.0, -2.5, 3.0, tiny, -tiny, huge, -huge, tiny * 1e-8, 0.1, 7.0, 1.0 / 3.0]

    def const_subtree(depth):
        if depth == 0 or rng.random() < 0.3:
            return de.Node(val=float(pool[rng.integers(len(pool))]) if rng.random() < 0.7 else float(rng.standard_normal()))
        return de.Node([add, sub, mul, div][rng.integers(4)], const_subtree(depth - 1), const_subtree(depth - 1))

    trees = []
    for k in range(300):
        x = de.Node(feature=int(rng.integers(1, 6)))
        c = const_subtree(int(rng.integers(1, 4)))
        if c.degree == 0:
            c = de.Node(mul, c, de.Node(val=2.0))
        inner = de.Node(cos, const_subtree(2)) if k % 5 == 0 else const_subtree(2)   # (a cos(...) subtree goes to the device)
        trees.append(de.Node([add, mul, sub, div][k % 4], de.Node(add, x, inner), c))
    X = np.asfortranarray(rng.standard_normal((5, 700)).astype(dtype))

    def run(consts=None):
        pop = api.Population(trees, ops, dtype, n_features=5)
        if consts is not None:
            pop.set_constants(consts)
        out, ok = pop.eval(X)
        dumps = [pop.dump(t).copy() for t in range(0, len(trees), 7)]
        h = pop.stream_hash()
        pop.close()
        return np.asarray(out), np.asarray(ok, dtype=bool), dumps, h

    host = run()
    monkeypatch.setenv("DE_NO_HOST_FOLD", "1")
    dev = run()
    monkeypatch.delenv("DE_NO_HOST_FOLD")
    assert np.array_equal(host[1], dev[1]), "flags"
    assert host[1].any() and not host[1].all()
    u = np.uint32 if dtype == np.float32 else np.uint64
    fin = np.isfinite(dev[0]) | np.isfinite(host[0])       # (NaN sign / payload is not compared: x86 0/0 = -NaN, gfx950 +NaN)
    assert np.array_equal(host[0].view(u)[fin], dev[0].view(u)[fin]), "rows"
    assert np.array_equal(np.isnan(host[0]), np.isnan(dev[0]))
    assert host[3] != dev[3], "the two programs differ in WHERE they fold (the hash covers the host-fold tables)"
    # ... and through de_program_set_consts
    consts = np.concatenate([de.flatten(t, ops, dtype)[1] for t in trees]).astype(dtype)
    c2 = (consts * dtype(1.5) + dtype(0.25)).astype(dtype)
    host2 = run(c2)
    monkeypatch.setenv("DE_NO_HOST_FOLD", "1")
    dev2 = run(c2)
    monkeypatch.delenv("DE_NO_HOST_FOLD")
    assert np.array_equal(host2[1], dev2[1])
    fin = np.isfinite(dev2[0]) | np.isfinite(host2[0])
    assert np.array_equal(host2[0].view(u)[fin], dev2[0].view(u)[fin])
    print(f"[host folds {np.dtype(dtype).name}] {int(host[1].sum())} of {len(trees)} trees complete; rows and flags bit-equal to the device-folded program, before and after set_constants")
