import sys; sys.path.insert(0, '/root/repo')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
ops = de.synth.BENCH_OPERATORS  # bin: + - / *   una: cos exp
P, F, C, N = 8, 5, 16, 64
g = np.random.Generator(np.random.PCG64(5))
params = np.asfortranarray(g.standard_normal((P, C)).astype(np.float32))
classes = g.integers(1, C + 1, N).astype(np.int64)
X = de.synth.random_X(F, N, seed=7)
PN = de.ParametricNode
def p(i): return PN(parameter=i)
def x(i): return PN(feature=i)
def c(v): return PN(val=v)
cases = {
 "p1": p(1), "p1+x1": PN(1, p(1), x(1)), "x1*p2": PN(4, x(1), p(2)), "cos(p5)": PN(1, p(5)), "exp(p5)/p5": PN(3, PN(2, p(5)), p(5)),
 "p5*p1": PN(4, p(5), p(1)), "c-p4": PN(2, c(0.5), p(4)), "exp(c)-p4": PN(2, PN(2, c(-0.1)), p(4)),
 "cos(p5*p1)-(exp(c)-p4)": PN(2, PN(1, PN(4, p(5), p(1))), PN(2, PN(2, c(-0.1)), p(4))),
}
for name, tree in cases.items():
    pop = api.Population([tree], ops, np.float32, n_features=F, n_params=P)
    o, gr, ok = pop.eval_grad(X, "both", params, classes)
    oe, oke = pop.eval(X, params, classes)
    tape, consts = de.flatten(tree, ops, np.float32)
    t2, PX = oracle.parametric_to_plain(tape, X, params, classes)
    y, gg, okr = oracle.eval_grad_tree_array(t2, consts, PX, oracle.GRAD_BOTH, elementwise=True)
    print(f"{name:28s} y err {np.max(np.abs(o[0]-y)):.3g} (eval err {np.max(np.abs(oe[0]-y)):.3g}) grad err {np.max(np.abs(gr[0]-gg)):.3g} ok {bool(ok[0])}/{okr}")
    if np.max(np.abs(o[0]-y)) > 1e-3:
        w = np.zeros(400, np.uint32); n = api.library().de_program_dump(pop._h, 0, w.ctypes.data, 400, 0)
        print("   generic:", [(hex(a), hex(b)) for a, b in w[:n].reshape(-1,4)[:, :2]])
