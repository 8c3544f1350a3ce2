#!/usr/bin/env python3
"""Shared leaf rows in the forward-dual gradient kernels (DE_GRAD_SHARE = 0 | 1, csrc/de_api_grad.cpp ensure_grad_threaded): the four waves of a
workgroup on the SAME samples with different trees.  Bit-equality of Jacobians, fused losses and loss gradients against one copy per wave."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402

ops = de.synth.BENCH_OPERATORS
fails = 0


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


for dtype in (np.float32, np.float64):
    for kind in ("parametric", "wide X", "narrow X"):
        g = np.random.Generator(np.random.PCG64(7))
        if kind == "parametric":
            F, P, C = 5, 8, 9
            trees = de.synth.random_population(300, seed=0x6A5, dtype=dtype, node_type=de.ParametricNode, nparams=P)
        else:
            F, P, C = (24 if kind == "wide X" else 5), 0, 0
            trees = de.synth.random_population(300, seed=0x6A6, dtype=dtype, nfeatures=F)
        for N in (70_001, 513):
            X = np.asfortranarray((g.standard_normal((F, N)) * 1.3).astype(dtype))
            y = g.standard_normal(N).astype(dtype)
            kw = dict(params=np.asfortranarray((g.standard_normal((P, C)) * 2).astype(dtype)), classes=g.integers(1, C + 1, N).astype(np.int64)) if P else {}
            res = {}
            for share in ("0", "1"):
                os.environ["DE_GRAD_SHARE"] = share
                pop = api.Population(trees, ops, dtype, n_features=F, n_params=P)
                r = []
                for variable in ((False, "both", True) if P else (False, True)):
                    out, grads, ok = pop.eval_grad(X, variable=variable, **kw)
                    r.append((np.asarray(ok), [np.asarray(gr) for gr in grads], np.asarray(out)))
                    l, dl, okl = pop.eval_loss_grad(X, y, variable=variable, **kw)
                    r.append((np.asarray(okl), [np.asarray(d) for d in dl], np.asarray(l)))
                res[share] = r
                pop.close()
            for (k0, g0, o0), (k1, g1, o1) in zip(res["0"], res["1"]):
                okeq = np.array_equal(k0, k1)
                live = k0 != 0
                geq = all(same(a, b) for a, b, l in zip(g0, g1, live) if l)
                oeq = same(o0[live], o1[live])
                fails += 0 if (okeq and geq and oeq) else 1
                print(f"{np.dtype(dtype).name} {kind} N {N}: flags {okeq} complete {int(live.sum())} rows {geq} values {oeq}", flush=True)
print("FAILS", fails)
