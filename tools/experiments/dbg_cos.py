import os, sys
sys.path.insert(0, '.')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
cos, add, mul = ops.index("cos", 1), ops.index("+", 2), ops.index("*", 2)
for c in (-0.4562351009524229, -2.5, 1.0, 0.3):
    cf = np.float32(c)
    X = np.asfortranarray(np.full((5, 300), cf, dtype=np.float32))
    X[1] = 0.0
    t_const = de.Node(add, de.Node(feature=2), de.Node(cos, de.Node(val=float(c))))       # x2 (= 0) + cos(c): folded subtree cos(c)
    t_feat = de.Node(cos, de.Node(feature=1))                                                # cos(x1), x1 = c
    t_acc = de.Node(cos, de.Node(add, de.Node(feature=1), de.Node(feature=2)))               # cos(x1 + x2): accumulator form
    out = {}
    for name, env in (("kernel", {}), ("aux", {"DE_NO_KERNEL_FOLD": "1"}), ("nofold", {"DE_NO_FOLD": "1"})):
        os.environ.update(env)
        pop = api.Population([t_const, t_feat, t_acc], ops, np.float32, n_features=5)
        o, ok = pop.eval(X)
        out[name] = np.asarray(o)[:, 0].view(np.uint32)
        pop.close()
        for k in env: del os.environ[k]
    print(c, {k: [hex(int(v)) for v in out[k]] for k in out}, "numpy f64->f32", hex(int(np.float32(np.cos(np.float64(cf))).view(np.uint32))))
