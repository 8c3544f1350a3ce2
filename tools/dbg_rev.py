"""Debug aid: reverse-accumulation loss gradient vs the forward Jacobian for the exact-operator test trees."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*"), unary_operators=("neg", "square", "abs"))
rng = de.synth.Xoshiro256ss(17)
import inspect, test_gpu_loss
src = inspect.getsource(test_gpu_loss.test_fused_loss_grad_exact_operators_vs_oracle)
print(src[:1500])
