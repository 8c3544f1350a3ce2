"""CPU pin of the Float64 gamma algorithm (csrc/de_device_ops.h de_gamma_f64): tools/fit/gamma_proto.py restates it operation by
operation in Python floats (IEEE double, round to nearest, no contraction — what the device code computes; its two_prod uses Dekker's
split where the device uses an FMA: the same exact error term) and is held against mpmath here.  The device itself is measured in
tests/test_gpu_ulp_f64.py (0.75 ulp)."""
import importlib.util
import math
import os
import random

import pytest

mpmath = pytest.importorskip("mpmath")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _proto():
    src = open(os.path.join(ROOT, "tools", "fit", "gamma_proto.py")).read().split("random.seed(1)")[0]  # the functions, not the sweep
    ns = {}
    exec(compile(src, "gamma_proto", "exec"), ns)
    return ns


def test_gamma_restatement_is_within_one_ulp():
    ns = _proto()
    rng = random.Random(7)
    pts = [rng.uniform(0.05, 30) for _ in range(500)] + [rng.uniform(-5.9, -0.1) for _ in range(300)] + [rng.uniform(30, 171.6) for _ in range(200)]
    pts += [rng.uniform(-170, -6) for _ in range(200)] + [10.0 ** rng.uniform(-300, -1) for _ in range(50)] + [float(k) for k in range(1, 172, 7)]
    pts += [-k + s * 2.0 ** -e for k in range(0, 12) for e in (10, 30, 50) for s in (1, -1) if (-k + s * 2.0 ** -e) != round(-k + s * 2.0 ** -e)]
    worst = 0.0
    for x in pts:
        e = ns["ulp_err"](x)
        if e is not None:
            worst = max(worst, e)
    assert worst < 0.85, worst
    # exact factorials up to 22! (exactly representable) come out exact or within an ulp, and the pole neighbourhood keeps its sign
    for n in range(1, 23):
        g = ns["gamma_f64"](float(n))
        assert abs(g - math.factorial(n - 1)) <= math.ulp(float(math.factorial(n - 1)))
    assert ns["gamma_f64"](-1.0 + 2.0 ** -30) < 0 < ns["gamma_f64"](-1.0 - 2.0 ** -30) or ns["gamma_f64"](-1.0 + 2.0 ** -30) * ns["gamma_f64"](-1.0 - 2.0 ** -30) < 0
