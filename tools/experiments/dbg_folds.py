"""Which folding route disagrees with the auxiliary program, on which tree (debug helper of tests/test_gpu_round6.py)."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
dtype = np.float32
ops = de.synth.BENCH_OPERATORS
rng = np.random.Generator(np.random.PCG64(17))
add, sub, mul, div = (ops.index(s, 2) for s in "+-*/")
cos, exp = ops.index("cos", 1), ops.index("exp", 1)
tiny, huge = (1e-30, 1e30)
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
    inner = const_subtree(3, unary=True) if k % 2 == 0 else const_subtree(2)
    trees.append(de.Node([add, mul, sub, div][k % 4], de.Node(add, x, inner), c))
X = np.asfortranarray(rng.standard_normal((5, 700)).astype(dtype))
res = {}
for name, env in (("default", {}), ("kernel", {"DE_NO_HOST_FOLD": "1"}), ("aux", {"DE_NO_HOST_FOLD": "1", "DE_NO_KERNEL_FOLD": "1"})):
    os.environ.update(env)
    pop = api.Population(trees, ops, dtype, n_features=5)
    out, ok = pop.eval(X)
    res[name] = (np.asarray(out), np.asarray(ok, dtype=bool), [pop.dump(t).copy() for t in range(len(trees))])
    pop.close()
    for k in env: del os.environ[k]
for name in ("default", "kernel"):
    a, b = res[name], res["aux"]
    bad = [t for t in range(len(trees)) if a[1][t] != b[1][t] or not np.array_equal(a[0][t].view(np.uint32)[np.isfinite(a[0][t]) | np.isfinite(b[0][t])], b[0][t].view(np.uint32)[np.isfinite(a[0][t]) | np.isfinite(b[0][t])])]
    print(name, "differs from aux on", len(bad), "trees", bad[:10])
    for t in bad[:4]:
        print("  tree", t, de.string_tree(trees[t], ops), "ok", a[1][t], b[1][t], "first vals", a[0][t][:3], b[0][t][:3])
