"""Fuzz of the hot operator set only (+ - * / cos exp): larger N (several tiles, ragged tails), all option
modes, both dtypes, eval + fused loss; flags must match the oracle, values within the parity tolerance."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
import test_gpu_eval as TE
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
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
            TE.compare_population(api, trees, ops, X, dtype, eval_context=ec, min_ok=0)
        print("ok", rep, dtype.__name__, F, N, flush=True)
print("hot fuzz passed")
