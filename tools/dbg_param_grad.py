import sys; sys.path.insert(0, '/root/repo')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
ops = de.synth.BENCH_OPERATORS
P, F, C, N = 8, 5, 16, 1500
trees = de.synth.random_population(60, seed=0xDE05, nfeatures=F, node_type=de.ParametricNode, nparams=P)
g = np.random.Generator(np.random.PCG64(5))
params = np.asfortranarray(g.standard_normal((P, C)).astype(np.float32))
classes = g.integers(1, C + 1, N).astype(np.int64)
X = de.synth.random_X(F, N, seed=7)
pop = api.Population(trees, ops, np.float32, n_features=F, n_params=P)
outg, grads, okg = pop.eval_grad(X, False, params, classes)
for t, tree in enumerate(trees):
    tape, consts = de.flatten(tree, ops, np.float32)
    t2, PX = oracle.parametric_to_plain(tape, X, params, classes)
    yg, gg, okg_el = oracle.eval_grad_tree_array(t2, consts, PX, oracle.GRAD_CONSTANT, elementwise=True)
    if okg_el and gg.size:
        sc = np.max(np.abs(gg)) + 1e-30
        okm = np.abs(grads[t] - gg) <= 1e-4 * np.abs(gg) + 1e-6 * sc
        if okm.mean() < 0.98:
            bad = np.argwhere(~okm)
            print(t, okm.mean(), de.string_tree(tree, ops)[:150], "rows bad:", np.unique(bad[:,0]), "n_c", len(consts))
            k, j = bad[0]
            print("  sample", j, "gpu", grads[t][k, j], "ref", gg[k, j], "y gpu", outg[t][j], "ref", yg[j])
print("---- single-tree populations")
for t in (52, 55, 10):
    p1 = api.Population([trees[t]], ops, np.float32, n_features=F, n_params=P)
    o1, g1, k1 = p1.eval_grad(X, False, params, classes)
    tape, consts = de.flatten(trees[t], ops, np.float32)
    t2, PX = oracle.parametric_to_plain(tape, X, params, classes)
    yg, gg, _ = oracle.eval_grad_tree_array(t2, consts, PX, oracle.GRAD_CONSTANT, elementwise=True)
    print(t, "alone: y max err", np.nanmax(np.abs(o1[0]-yg)), "in-pop: ", np.nanmax(np.abs(outg[t]-yg)))
    w = p1.dump(0)
    print("  n_instr", len(w))
import ctypes as C
for t in (51, 52):
    n = api.library().de_program_dump(pop._h, t, None, 0, 0)
    print("tree", t, "generic words", n, "n_consts", pop.n_consts[t], "cum consts", pop.n_consts[:t].sum())
