#!/usr/bin/env python3
"""Trace VALUE findings of the command-line fuzzers that tools/trace_findings.py does not re-run: `hot<seed>` (tests/fuzz/fuzz_hot.py: the
hot operator set, the four evaluation contexts) and `gpu<seed>ee0` (tests/fuzz/fuzz_gpu.py with EvalContext(early_exit=False)).  For every
value beyond the tolerance model: device, oracle and the 200-bit value of the same tree at the same sample — who is the outlier, in units of
the tolerance the test applied.     gpurun -- 'python tools/trace_value_findings.py hot54 gpu55ee0 > gpurun_out/value_findings.jsonl'"""
import json
import sys

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402
from oracle import oracle  # noqa: E402
import fuzzlib as FZ  # noqa: E402
from helpers import parity_tolerance  # noqa: E402
import trace_findings as TF  # noqa: E402

mpf = TF.mpf
STATS = dict(trees=0, values=0)


def check(trees, ops, X, dtype, ec, label, out):
    pop = api.Population(trees, ops, dtype, n_features=X.shape[0], eval_context=ec)
    o, ok = pop.eval(X)
    opts = ec.option_bits(ops)
    for t, tree in enumerate(trees):
        tape, consts = de.flatten(tree, ops, dtype)
        y, ok_el = oracle.eval_tree_array(tape, consts, X, opts, elementwise=True)
        if not ok_el or not ok[t]:
            continue
        tol = parity_tolerance(tree, ops, X, dtype, opts)
        m = np.isfinite(y) & np.isfinite(o[t]) & np.isfinite(tol)
        err = np.abs(o[t].astype(np.float64) - y.astype(np.float64))
        with np.errstate(invalid="ignore", divide="ignore"):
            ratio = np.where(m & (tol > 0), err / np.where(tol > 0, tol, 1), 0)
        STATS["trees"] += 1
        STATS["values"] += int(m.sum())
        if ratio.max() <= 1.0:
            continue
        j = int(np.argmax(ratio))
        leaf = lambda n: mpf(float(dtype(n.val))) if n.constant else mpf(float(X[n.feature - 1, j]))  # noqa: E731
        try:
            d = TF.truth(tree, ops, leaf, 0, lambda n: None).v
            eg, eo = float(abs(mpf(float(o[t][j])) - d)), float(abs(mpf(float(y[j])) - d))
        except Exception as e:  # noqa: BLE001
            d, eg, eo = None, None, str(e)
        rec = dict(kind="value", fuzz=label, dtype=np.dtype(dtype).name, options=int(opts), tree=de.string_tree(tree, ops)[:260], sample=j,
                   x=[float(v) for v in X[:, j]], err_over_tol=float(ratio[j]), tol=float(tol[j]), gpu=float(o[t][j]), oracle=float(y[j]),
                   truth=None if d is None else float(d), gpu_err_over_tol=None if eg is None else eg / float(tol[j]),
                   oracle_err_over_tol=eo / float(tol[j]) if isinstance(eo, float) else eo, samples_beyond=int((ratio > 1.0).sum()))
        out.append(rec)
        print(json.dumps(rec), flush=True)
    pop.close()


def hot(seed0, out):
    ops = de.synth.BENCH_OPERATORS
    for rep in range(5):
        rng = de.synth.Xoshiro256ss(seed0 * 31 + rep)
        for dtype in (np.float32, np.float64):
            F = 1 + (seed0 + rep) % 7
            trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 3 + rep) % 40, ops, F, rng, dtype) for i in range(300)]
            g = np.random.Generator(np.random.PCG64(seed0 * 7 + rep))
            N = int(g.choice([1, 2, 63, 64, 65, 1023, 1024, 1025, 2047, 3000, 5121]))
            X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.5, 1, 3])).astype(dtype))
            for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False), api.EvalContext(bumper=True)):
                check(trees, ops, X, dtype, ec, f"fuzz_hot {seed0} rep {rep}", out)


def gpu_ee0(seed0, out):
    for rep in range(6):
        rng = de.synth.Xoshiro256ss(seed0 * 1000 + rep)
        for ops, F in ((FZ.OPS_HOT, 5), (FZ.OPS_WIDE, 3), (FZ.OPS_HOT, 2)):
            for dtype in (np.float32, np.float64):
                trees = FZ.random_trees(rng, ops, F, dtype, 400, 33, rep)
                g = np.random.Generator(np.random.PCG64(seed0 + rep))
                N = int(g.integers(1, 1500))
                X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.1, 1, 10])).astype(dtype))
                if rep % 2:
                    X[0, N // 2] = np.inf
                check(trees, ops, X, dtype, api.EvalContext(early_exit=False), f"fuzz_gpu {seed0} rep {rep} early_exit=False", out)


if __name__ == "__main__":
    found = []
    for w in sys.argv[1:]:
        if w.startswith("hot"):
            hot(int(w[3:]), found)
        elif w.startswith("gpu") and w.endswith("ee0"):
            gpu_ee0(int(w[3:-3]), found)
    print(json.dumps(dict(summary=True, flavours=sys.argv[1:], findings=len(found), **STATS,
                          device_is_the_outlier=sum(1 for r in found if isinstance(r.get("gpu_err_over_tol"), float) and isinstance(r.get("oracle_err_over_tol"), float)
                                                    and r["gpu_err_over_tol"] > 1.0 and r["gpu_err_over_tol"] > r["oracle_err_over_tol"]))))
