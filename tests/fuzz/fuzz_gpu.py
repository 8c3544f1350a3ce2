"""Differential fuzz on the GPU box with a seed from the command line: many random trees, wide operator set, all option
modes, both dtypes, eval + gradients against the oracle — every difference classified by tests/fuzzlib.py (ill-conditioned
samples are counted, the run goes on to the end; `python tests/fuzz/fuzz_gpu.py 21`).  The fixed-seed gate lives in
tests/test_gpu_fuzz.py."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
import fuzzlib as FZ

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tot = FZ.Findings()
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
            for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False), api.EvalContext(bumper=True)):
                tot.add(FZ.fuzz_eval(api, trees, ops, X, dtype, ec, label=f"rep {rep}"))
            for mode in ("variable", "constant", "both"):
                tot.add(FZ.fuzz_grad(api, trees[:150], ops, X, dtype, mode, label=f"rep {rep}"))
            print("done", rep, F, dtype.__name__, N, tot.summary(), flush=True)
for r in tot.real:
    print("REAL:", r)
print("fuzz finished:", tot.summary())
sys.exit(1 if tot.real else 0)
