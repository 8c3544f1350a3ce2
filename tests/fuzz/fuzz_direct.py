"""Fuzz the wide-X 'direct' variant (flat-switch kernel, features gathered from global memory) and the
DE_EVAL_THREADED=0 / DE_NO_FOLD / DE_NO_FUSE fallbacks against the oracle."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
import test_gpu_eval as TE
ops_wide = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2"),
                           unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin", "safe_sqrt", "atan", "relu"))
F = int(os.environ.get("FUZZ_F", "60"))
for rep in range(3):
    rng = de.synth.Xoshiro256ss(500 + rep)
    for ops in (de.synth.BENCH_OPERATORS, ops_wide):
        for dtype in (np.float32, np.float64):
            trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 7 + rep) % 33, ops, F, rng, dtype) for i in range(250)]
            g = np.random.Generator(np.random.PCG64(rep))
            N = int(g.integers(1, 3000))
            X = np.asfortranarray(g.standard_normal((F, N)).astype(dtype))
            for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False), api.EvalContext(bumper=True)):
                TE.compare_population(api, trees, ops, X, dtype, eval_context=ec, min_ok=0)
            print("ok", rep, dtype.__name__, N, flush=True)
print("direct/fallback fuzz passed")
