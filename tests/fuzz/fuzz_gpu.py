"""One-off differential fuzz on the GPU box: many random trees, wide operator set, all option modes, both
dtypes, eval + gradients against the oracle (reuses the comparison helpers of the GPU test-suite)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
import test_gpu_eval as TE
import test_gpu_grad as TG

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
# continuous operators only: at a discontinuity (greater, mod, rem, round, sign at equality) a 1-ulp difference of an
# operand legitimately flips the result and the relative-noise tolerance model cannot see it
ops_wide = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "min", "pow_abs2", "^"),
                           unary_operators=("cos", "exp", "safe_log", "neg", "square", "cube", "abs", "tanh", "sin",
                                            "safe_sqrt", "atan", "relu"))
ops_hot = de.synth.BENCH_OPERATORS
tot = 0
for rep in range(6):
    rng = de.synth.Xoshiro256ss(seed0 * 1000 + rep)
    for ops, F in ((ops_hot, 5), (ops_wide, 3), (ops_hot, 2)):
        for dtype in (np.float32, np.float64):
            trees = [de.synth.gen_random_tree_fixed_size(1 + (i * 7 + rep) % 33, ops, F, rng, dtype) for i in range(400)]
            g = np.random.Generator(np.random.PCG64(seed0 + rep))
            N = int(g.integers(1, 1500))
            X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.1, 1, 10])).astype(dtype))
            if rep % 2:
                X[0, N // 2] = np.inf
            for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False), api.EvalContext(bumper=True)):
                TE.compare_population(api, trees, ops, X, dtype, eval_context=ec, min_ok=0)
            if ops is ops_hot and dtype == np.float32:  # (the f64 branch of grad_compare has no conditioning model)
                for mode in ("variable", "constant", "both"):
                    TG.grad_compare(api, trees[:150], ops, X, dtype, mode)
            tot += len(trees)
            print("ok", rep, F, dtype.__name__, N, flush=True)
print("fuzz passed, trees:", tot)
