import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from test_lowering import random_graph
OPS = de.OperatorEnum(binary_operators=("+", "-", "*", "/"), unary_operators=("cos", "exp", "safe_log", "square"))
dtype=np.float32
rng = de.synth.Xoshiro256ss(95)
graphs = [random_graph(rng, OPS, 8 + i % 22, 4, 1 + i % 4, dtype) for i in range(200)]
expanded = [de.break_sharing(g) for g in graphs]
X = de.synth.random_X(4, 1300, seed=7, dtype=dtype)
y = np.cos(np.arange(X.shape[1])).astype(dtype)
os.environ["DE_LOSS_GRAD_REVERSE"]="1"
pg = api.Population(graphs, OPS, dtype, n_features=4)
pe = api.Population(expanded, OPS, dtype, n_features=4)
lg, dg, kg = pg.eval_loss_grad(X, y, variable=True)
print(pg.ctx.last_kernel_name())
le, dee, ke = pe.eval_loss_grad(X, y, variable=True)
os.environ["DE_LOSS_GRAD_REVERSE"]="0"
lf, dfw, kf = pg.eval_loss_grad(X, y, variable=True)
lf2, dfw2, kf2 = pe.eval_loss_grad(X, y, variable=True)
print("graph-rev vs exp-rev flag diffs", np.nonzero(kg!=ke)[0], "graph-rev vs graph-fwd", np.nonzero(kg!=kf)[0], "exp-rev vs exp-fwd", np.nonzero(ke!=kf2)[0])
for t in np.nonzero(kg!=ke)[0][:6]:
    print(t, kg[t], ke[t], kf[t], de.string_tree(expanded[t], OPS)[:200])
    print("  dg", dg[t], "de", dee[t], "df", dfw[t])
for t in (1, 2, 3, 5):
    print(t, kg[t], ke[t], kf[t], de.string_tree(expanded[t], OPS)[:300])
    print("  graph-rev", dg[t], "\n  exp-rev  ", dee[t], "\n  graph-fwd", dfw[t], "\n  exp-fwd  ", dfw2[t])
os.environ["DE_LOSS_GRAD_REVERSE"]="1"; os.environ["DE_REV_NO_FUSE"]="1"
pg2 = api.Population(graphs, OPS, dtype, n_features=4)
l2, d2, k2 = pg2.eval_loss_grad(X, y, variable=True)
print("nofuse graph-rev tree1", d2[1])
