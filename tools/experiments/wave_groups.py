#!/usr/bin/env python3
"""Wave groups (DE_EVAL_WAVES = 1 | 2 | 4, csrc/de_api_program.cpp choose_waves): parametric evaluation with W waves per workgroup sharing
the staged X / parameter rows.  Bit-equality of values, flags and fused losses against one-wave workgroups, and the time of each.

    gpurun -- 'timeout 600 python tools/experiments/wave_groups.py'
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402

lib = api.library()
ctx = api.Context(0)
dev = "cuda"


PLAIN = "plain" in sys.argv  # plain (non-parametric) programs
ENV = "DE_EVAL_WAVES"


def run(trees, ops, dtype, N, n_cls, per_sample, waves, steps=0, loss=False, P=8, F=5, seed=3):
    if waves is None:
        os.environ.pop(ENV, None)
    else:
        os.environ[ENV] = str(waves)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    X = (torch.randn((N, F), generator=g, device=dev, dtype=tdt) * 1.5).t()  # [F, N] column-major
    params = torch.randn((n_cls, P), generator=g, device=dev, dtype=tdt)
    classes = torch.arange(1, N + 1, device=dev, dtype=torch.int32) if per_sample else torch.randint(1, n_cls + 1, (N,), generator=g, device=dev, dtype=torch.int32)
    pop = api.Population(trees, ops, dtype, n_features=F, ctx=ctx, n_params=0 if PLAIN else P)
    pa = api.ParamArgs()
    pa.params, pa.ld_params, pa.n_classes = params.data_ptr(), P, n_cls
    pa.classes, pa.classes_is_i64, pa.class_base = classes.data_ptr(), 0, 1
    par = None if PLAIN else ctypes.byref(pa)
    out = torch.zeros((len(trees), N), device=dev, dtype=tdt)
    ok = torch.zeros(len(trees), device=dev, dtype=torch.uint8)
    y = torch.randn(N, generator=g, device=dev, dtype=tdt)
    lossv = torch.zeros(len(trees), device=dev, dtype=tdt)

    def step():
        if loss:
            ctx.check(lib.de_eval_loss(ctx._h, pop._h, X.data_ptr(), N, F, par, y.data_ptr(), None, 0, lossv.data_ptr(), ok.data_ptr()))
        else:
            ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, par, out.data_ptr(), N, ok.data_ptr()))
    step()
    ctx.synchronize()
    ms = None
    if steps:
        for _ in range(30):
            step()
        ctx.synchronize()
        ctx.timing_ring(steps)
        for _ in range(steps):
            step()
        ctx.synchronize()
        v = [t for t in ctx.timing_read() if t is not None]
        ctx.timing_ring(0)
        ms = float(np.mean(v))
    okh = ok.cpu().numpy().copy()
    res = (lossv if loss else out).cpu().numpy().copy()
    del pop
    return okh, res, ms


def same(a, b, okh):
    a = a[okh != 0]
    b = b[okh != 0]
    return a.tobytes() == b.tobytes()


fails = 0
ops = de.synth.BENCH_OPERATORS
for dtype in (() if 'timeonly' in sys.argv else (np.float32, np.float64)):
    for (n_trees, N, n_cls, per_sample, P) in ((1000, 100_003, 16, False, 8), (300, 50_000, 50_000, True, 8), (7, 3_000, 5, False, 3), (1, 1_000_000, 16, False, 8),
                                               (2000, 300_000, 16, False, 16)):
        trees = de.synth.random_population(n_trees, seed=0xC5 + n_trees) if PLAIN else de.synth.random_population(n_trees, seed=0xC5 + n_trees, node_type=de.ParametricNode, nparams=P)
        for loss in (False, True):
            ok1, r1, _ = run(trees, ops, dtype, N, n_cls, per_sample, 1, loss=loss, P=P)
            for w in (2, 4, None):
                okw, rw, _ = run(trees, ops, dtype, N, n_cls, per_sample, w, loss=loss, P=P)
                good = (ok1 == okw).all() and same(r1, rw, ok1)
                fails += 0 if good else 1
                print(f"{np.dtype(dtype).name} trees {n_trees} N {N} classes {n_cls} P {P} loss {loss} waves {w}: flags equal {(ok1 == okw).all()} "
                      f"complete {int(ok1.sum())}, complete rows bit-equal {same(r1, rw, ok1)}", flush=True)
print("FAILS", fails, flush=True)

# timing: the C5N shape (1000 trees, 10^6 samples, 8 per-sample parameters) and 16 classes
if PLAIN:  # the headline population at 10^6 and 10^7 samples
    trees = de.synth.random_population(1000, seed=0xDE02)
    for N in (10**6, 10**7):
        for w in (1, 2, 4):
            for loss in (False, True):
                _, _, ms = run(trees, ops, np.float32, N, 16, False, w, steps=20, loss=loss)
                print(f"TIME plain N {N} waves {w} loss {loss}: {ms:.3f} ms", flush=True)
    # wide feature matrices: the X rows are what cuts the occupancy
    for F in (12, 20, 30):
        trees = de.synth.random_population(1000, seed=0xDE02 + F, nfeatures=F)
        ref = None
        for w in (1, 2, 4, None):
            okh, res, ms = run(trees, ops, np.float32, 10**6, 16, False, w, steps=20, F=F)
            if ref is None:
                ref = (okh, res)
            print(f"TIME plain F {F} N 1000000 waves {w}: {ms:.3f} ms; flags equal {(ref[0] == okh).all()}, complete rows bit-equal {same(ref[1], res, okh)}", flush=True)
    raise SystemExit(0)
trees = de.synth.random_population(1000, seed=0xDE05, node_type=de.ParametricNode, nparams=8)
for (n_cls, per_sample) in ((10**6, True), (16, False)):
    for w in (1, 2, 4, None):
        for loss in (False, True):
            _, _, ms = run(trees, ops, np.float32, 10**6, n_cls, per_sample, w, steps=40, loss=loss)
            print(f"TIME classes {n_cls} waves {w} loss {loss}: {ms:.3f} ms", flush=True)
