"""Do the Float32 operators keep subnormal RESULTS (exp, ^, pow_abs2, *, /)?  Prints device values next to numpy's."""
import sys
sys.path.insert(0, '.')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.OperatorEnum(binary_operators=("^", "pow_abs2", "*", "/"), unary_operators=("exp", "square"))
C = np.array([0.6, 0.9, 1.4, 1.6, 2.4, 2.6, 3.6, 100.3])  # results c * 2^-149: the last bits of the subnormal range
a = (np.log(C) - 149 * np.log(2.0)).astype(np.float32)
X = np.asfortranarray(np.stack([a, np.full_like(a, 0.5), np.full_like(a, 1e-30)]))
x1, x2, x3 = (de.Node(feature=i) for i in (1, 2, 3))
cases = {
    "exp(x1)": (de.Node(1, x1), np.exp(a.astype(np.float64))),
    "0.5 ^ (x1 * -1.4427)": (de.Node(1, x2, de.Node(3, x1, de.Node(val=-1.4426950408889634))), np.exp(a.astype(np.float64))),
    "pow_abs2(0.5, x1 * -1.4427)": (de.Node(2, x2, de.Node(3, x1, de.Node(val=-1.4426950408889634))), np.exp(a.astype(np.float64))),
    "x3 * 1e-10": (de.Node(3, x3, de.Node(val=1e-10)), np.full(len(a), 1e-40)),
    "x3 / 1e12": (de.Node(4, x3, de.Node(val=1e12)), np.full(len(a), 1e-42)),
}
for name, (tree, ref) in cases.items():
    for ec in (api.EvalContext(), api.EvalContext(use_fused=False)):
        y, ok = api.eval_tree_array(tree, X, ops, eval_context=ec)
        print(f"{name:32s} fused={ec.use_fused} gpu/2^-149 {y.astype(np.float64) * 2.0**149}  ref/2^-149 {np.round(ref * 2.0**149, 2)}")
