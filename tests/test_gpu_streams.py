"""The independent launches of a gradient / reverse call run on the caller's stream and two side streams (csrc/de_grad_kernels.hip ForkJoin);
DE_GRAD_STREAMS=1 puts them all on the caller's stream.  The switch is read once per process, so each setting runs in a process of its own:
the same bits in every output (Jacobians of the three modes, fused loss gradients forward and reverse, the by-class pullback)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, json, sys
import numpy as np, torch
sys.path.insert(0, %r)
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
trees = de.synth.random_population(300, seed=0x57A3)
N = 2**18 + 19
g = torch.Generator(device="cuda").manual_seed(2)
X = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()
y = torch.randn(N, generator=g, device="cuda", dtype=torch.float32)
out = {}
def h(*ts):
    m = hashlib.sha256()
    for t in ts:
        m.update(torch.nan_to_num(t.float(), nan=12345.0).contiguous().cpu().numpy().tobytes())
    return m.hexdigest()
pop = api.Population(trees, ops, np.float32, n_features=5)
s = torch.cuda.Stream()
with torch.cuda.stream(s):            # a non-default caller stream: the side streams fork from and join into THIS one
    for variable in (True, False, "both"):
        yv, gr, ok = pop.eval_grad(X, variable)
        okb = ok.bool()
        out["grad_%%s" %% variable] = h(yv[okb], ok, *[gr[t] for t in range(len(trees)) if bool(okb[t])])
    import os
    for rev in ("0", "1"):
        os.environ["DE_LOSS_GRAD_REVERSE"] = rev
        l, d, ok = pop.eval_loss_grad(X, y, variable=False)
        okb = ok.bool()
        out["lossgrad_rev%%s" %% rev] = h(l[okb], ok, *[d[t] for t in range(len(trees)) if bool(okb[t])])
    s.synchronize()
print(json.dumps(out))
""" % ROOT


def _run(n_streams):
    env = dict(os.environ, DE_GRAD_STREAMS=str(n_streams))
    env.pop("DE_LOSS_GRAD_REVERSE", None)
    o = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert o.returncode == 0, o.stderr[-2000:]
    return json.loads([l for l in o.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_side_streams_change_no_bit():
    a, b = _run(1), _run(3)
    assert a.keys() == b.keys() and len(a) == 5
    assert a == b
