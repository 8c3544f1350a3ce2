#!/usr/bin/env python3
"""In-situ cost of one dispatch of the eval kernel, per handler family (gpurun): synthetic populations whose trees are a
chain of n identical operators, timed at several n; the slope of time over n is the SIMD time one wavefront-dispatch
of that handler really takes (all overheads included), to set beside its VALU issue cycles in profiles/valu_slots.json.

    python tools/exp_dispatch_cost.py [--turbo] [-o gpurun_out/dispatch_cost.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402

ops = de.synth.BENCH_OPERATORS
B = {n: i + 1 for i, n in enumerate(ops.binops)}
U = {n: i + 1 for i, n in enumerate(ops.unaops)}
N = de.Node


def chain_binary(op, n, const=False):
    """((x1 op x2) op x3) op ... : n dispatches `acc = acc op row` (or `acc op const`) after one load."""
    t = N(feature=1)
    for i in range(n):
        leaf = N(val=1.0 + 0.001 * i) if const else N(feature=2 + i % 4)
        t = N(B[op], t, leaf)
    return t


def chain_unary(names, n):
    """f1(f2(f1(...x1))): n dispatches `acc = f(acc)`."""
    t = N(feature=1)
    for i in range(n):
        t = N(U[names[i % len(names)]], t)
    return t


def time_pop(trees, X, turbo, steps=6):
    pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(turbo=turbo))
    for _ in range(2):
        pop.eval(X)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        out = pop.eval(X)
    b.record()
    torch.cuda.synchronize()
    lib = api.library()
    n = lib.de_program_dump(pop._h, 0, None, 0, 3)
    pop.close()
    del out
    return a.elapsed_time(b) / steps, int(n) // 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--turbo", action="store_true")
    ap.add_argument("-o", "--out", default=os.path.join(ROOT, "gpurun_out", "dispatch_cost.json"))
    a = ap.parse_args()
    n_trees, n_samp = 500, 2_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    X = (torch.rand((n_samp, 5), generator=g, device="cuda", dtype=torch.float32) + 0.5).t()  # [0.5, 1.5): every chain stays finite
    cases = {
        "+ row": lambda n: chain_binary("+", n),
        "* row": lambda n: chain_binary("*", n),
        "/ row": lambda n: chain_binary("/", n),
        "* const": lambda n: chain_binary("*", n, True),
        "/ const": lambda n: chain_binary("/", n, True),
        "cos acc": lambda n: chain_unary(["cos"], n),
        "exp(cos) acc": lambda n: chain_unary(["cos", "exp"], n),
    }
    clock, simds = 2.0e9, 1024
    tree_waves = n_trees * n_samp / 256
    res = {}
    ms, nd = time_pop([N(feature=1)] * n_trees, X, a.turbo)
    res["leaf only"] = dict(ms=ms, dispatches=nd, simd_cycles_per_tree_wave=ms * 1e-3 * clock * simds / tree_waves)
    print("leaf only", res["leaf only"], flush=True)
    for name, mk in cases.items():
        pts = []
        for n in (8, 16, 32):
            ms, nd = time_pop([mk(n)] * n_trees, X, a.turbo)
            pts.append((n, nd, ms))
        (n0, d0, m0), (n1, d1, m1) = pts[0], pts[-1]
        slope_ms = (m1 - m0) / (d1 - d0)
        cyc = slope_ms * 1e-3 * clock * simds / tree_waves
        icpt = (m0 - slope_ms * d0) * 1e-3 * clock * simds / tree_waves
        res[name] = dict(points=pts, simd_cycles_per_dispatch=cyc, intercept_cycles=icpt)
        print(f"{name:14s} {pts}  -> {cyc:7.1f} SIMD cycles per wave-dispatch (at 2.0 GHz), intercept {icpt:7.1f}", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(turbo=a.turbo, n_trees=n_trees, n_samples=n_samp, clock_ghz=2.0, cases=res), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
