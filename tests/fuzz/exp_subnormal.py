"""Do exp / ^ / pow_abs2 round results c * 2^emin (the last bits of the subnormal range) like the oracle (glibc = correctly
rounded there)?  OCML's Float32 expf/powf returned 0 for c in (1/2, 1) — DESIGN.md §5.  Prints device and oracle values in
units of the smallest subnormal."""
import sys
sys.path.insert(0, '.')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
from oracle import oracle
ops = de.OperatorEnum(binary_operators=("^", "pow_abs2", "*"), unary_operators=("exp",))
C = np.array([0.4, 0.6, 0.9, 1.4, 1.6, 2.4, 2.6, 3.6, 100.3])
for dtype, emin in ((np.float32, -149), (np.float64, -1074)):
    a = (np.log(C) + emin * np.log(2.0)).astype(dtype)
    X = np.asfortranarray(np.stack([a, np.full_like(a, 0.5)]))
    x1, x2 = de.Node(feature=1), de.Node(feature=2)
    expo = de.Node(3, x1, de.Node(val=-1.4426950408889634))
    for name, tree in (("exp(x1)", de.Node(1, x1)), ("0.5 ^ (x1 * -log2 e)", de.Node(1, x2, expo)),
                       ("pow_abs2(0.5, x1 * -log2 e)", de.Node(2, x2, expo))):
        y, ok = api.eval_tree_array(tree, X, ops)
        tape, consts = de.flatten(tree, ops, dtype)
        yo, _ = oracle.eval_tree_array(tape, consts, X, api.EvalContext().option_bits(ops), elementwise=True)
        u = float(np.ldexp(1.0, emin))
        print(f"{dtype.__name__} {name:28s} gpu {y / u}  oracle {yo / u}")
