#!/usr/bin/env python3
"""What the reference's early exit is worth on the bench population (gpurun).

The reference stops evaluating a tree at the first non-finite intermediate array (`@return_on_nonfinite_array`,
src/Evaluate.jl:26-32); a kernel that runs every tree to its end on every sample spends its time on values nobody
may read (when ok == false only the flag is contractual).  This measures, on the headline workload,
  * the share of incomplete trees and what they cost (whole population vs. its complete / incomplete halves),
  * how early an incomplete tree shows: the fraction of them already flagged by the first k sample tiles.

    python tools/exp_early_exit.py [-o gpurun_out/early_exit.json] [--N 10000000]
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


def timed(pop, X, out, ok, steps=5):
    lib, ctx = api.library(), pop.ctx
    N = X.shape[1]
    for _ in range(2):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), out.shape[1], ok.data_ptr()))
    torch.cuda.synchronize()
    ms = []
    for _ in range(steps):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), out.shape[1], ok.data_ptr()))
        torch.cuda.synchronize()
        ms.append(ctx.last_kernel_ms())
    return float(np.median(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--out", default=os.path.join(ROOT, "gpurun_out", "early_exit.json"))
    ap.add_argument("--N", type=int, default=10**7)
    ap.add_argument("--turbo", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(1000, seed=0xDE02)
    N = a.N
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
    ec = api.EvalContext(turbo=True) if a.turbo else None
    res = {"N": N, "turbo": a.turbo}

    def run(sub, tag):
        pop = api.Population(sub, ops, np.float32, n_features=5, eval_context=ec)
        out = torch.empty((len(sub), N), device=dev, dtype=torch.float32)
        ok = torch.empty(len(sub), device=dev, dtype=torch.uint8)
        ms = timed(pop, X, out, ok)
        flags = ok.cpu().numpy().astype(bool)
        nodes = sum(de.count_nodes(t) for t in sub)
        res[tag] = dict(trees=len(sub), nodes=nodes, ms=ms, complete=int(flags.sum()))
        print(tag, res[tag], flush=True)
        pop.close()
        del out
        return flags

    flags = run(trees, "all")
    comp = [t for t, f in zip(trees, flags) if f]
    inc = [t for t, f in zip(trees, flags) if not f]
    run(comp, "complete_only")
    run(inc, "incomplete_only")
    # how early does an incomplete tree show?  flags over the first k tiles of 512 samples
    pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=ec)
    lib, ctx = api.library(), pop.ctx
    det = {}
    for k in (1, 8, 22, 176, 1024, 4096):
        n = min(512 * k, N)
        out = torch.empty((len(trees), n), device=dev, dtype=torch.float32)
        ok = torch.empty(len(trees), device=dev, dtype=torch.uint8)
        Xs = X[:, :n]
        ctx.check(lib.de_eval(ctx._h, pop._h, Xs.data_ptr(), n, 5, None, out.data_ptr(), n, ok.data_ptr()))
        torch.cuda.synchronize()
        f = ok.cpu().numpy().astype(bool)
        det[str(k)] = dict(samples=n, flagged=int((~f).sum()), of_incomplete=int((~flags).sum()),
                           flagged_but_complete_at_full_N=int(((~f) & flags).sum()))
        print("first", k, "tiles:", det[str(k)], flush=True)
        del out
    res["detected_by_first_tiles"] = det
    # per incomplete tree: share of 512-sample tiles with a non-finite FINAL value (lower bound of the tiles that flag it)
    n = min(N, 2 ** 20)
    out = torch.empty((len(trees), n), device=dev, dtype=torch.float32)
    ok = torch.empty(len(trees), device=dev, dtype=torch.uint8)
    Xs = X[:, :n]
    ctx.check(lib.de_eval(ctx._h, pop._h, Xs.data_ptr(), n, 5, None, out.data_ptr(), n, ok.data_ptr()))
    torch.cuda.synchronize()
    bad = (~torch.isfinite(out)).view(len(trees), n // 512, 512).any(dim=2).float().mean(dim=1).cpu().numpy()
    f = ok.cpu().numpy().astype(bool)
    share = bad[~f]
    res["tile_share_nonfinite_final_value"] = dict(samples=n, trees=int((~f).sum()),
                                                   quantiles={q: float(np.quantile(share, q)) for q in (0.05, 0.25, 0.5, 0.75, 0.95)},
                                                   mean=float(share.mean()))
    print(res["tile_share_nonfinite_final_value"])
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
