import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
import fuzzlib as FZ
from helpers import parity_tolerance
seed0, rep = 31, 2
rng = de.synth.Xoshiro256ss(seed0 * 1000 + rep)
for ops, F in ((FZ.OPS_HOT, 5), (FZ.OPS_WIDE, 3), (FZ.OPS_HOT, 2)):
    for dtype in (np.float32, np.float64):
        trees = FZ.random_trees(rng, ops, F, dtype, 400, 33, rep)
        if ops is FZ.OPS_WIDE and dtype == np.float32:
            tree, tops, tF = trees[394], ops, F
print(de.string_tree(tree, tops))
g = np.random.Generator(np.random.PCG64(seed0 + rep))
N = int(g.integers(1, 1500))
X = np.asfortranarray((g.standard_normal((tF, N)) * g.choice([0.1, 1, 10])).astype(np.float32))
ec = api.EvalContext(early_exit=False)
pop = api.Population([tree], tops, np.float32, n_features=tF, eval_context=ec)
out, ok = pop.eval(X)
tape, consts = de.flatten(tree, tops, np.float32)
opts = ec.option_bits(tops) & 15
y, oke = oracle.eval_tree_array(tape, consts, X, opts, elementwise=True)
tol = parity_tolerance(tree, tops, X, np.float32, opts)
fo, fg = np.isfinite(y), np.isfinite(out[0])
bad = np.nonzero((fo != fg) & np.isfinite(tol))[0]
print("N", N, "mismatching finite patterns at", bad[:5])
for j in bad[:3]:
    print(j, "x", X[:, j], "gpu", out[0][j], "oracle", y[j], "tol", tol[j])
    # inner value before the four squarings: evaluate the subtree below the outermost square(square(neg(square(.))))
    sub = tree.l.l.l.l  # relu(...)
    ts, cs = de.flatten(sub, tops, np.float32)
    ys, _ = oracle.eval_tree_array(ts, cs, X[:, j:j+1].copy(order='F'), opts, elementwise=True)
    ps = api.Population([sub], tops, np.float32, n_features=tF, eval_context=ec); og, _ = ps.eval(np.asfortranarray(X[:, j:j+1])); ps.close()
    print("   relu(...) oracle", repr(ys[0]), "gpu", repr(og[0][0]), " ^8 ->", float(ys[0])**8, float(og[0][0])**8)

# walk down the tree at the first mismatching sample: where do the device and the oracle part?
j = int(bad[0])
Xj = np.asfortranarray(np.repeat(X[:, j:j+1], 4, axis=1))
def both(sub, ctx):
    ts, cs = de.flatten(sub, tops, np.float32)
    yo, _ = oracle.eval_tree_array(ts, cs, Xj, ctx.option_bits(tops) & 15, elementwise=True)
    ps = api.Population([sub], tops, np.float32, n_features=tF, eval_context=ctx); og, _ = ps.eval(Xj); ps.close()
    return yo[0], og[0][0]
node, depth = tree, 0
while True:
    for name, ctx in (("early_exit=False", api.EvalContext(early_exit=False)), ("default", api.EvalContext())):
        o, gq = both(node, ctx)
        print("  " * depth, name, "oracle", repr(o), "gpu", repr(gq), "|", de.string_tree(node, tops)[:90])
    if node.degree == 0:
        break
    # follow the child that holds the features
    kids = list(node.children)
    node = max(kids, key=lambda c: de.synth.count_nodes(c))
    depth += 1
