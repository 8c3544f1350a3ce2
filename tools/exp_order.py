#!/usr/bin/env python3
"""Which sample tiles flag the incomplete trees, and would a cheap pre-pass over X find them early?  (gpurun)
Full evaluation of the headline population -> per (incomplete tree, 256-sample tile): does the tile hold a non-finite value ->
for a tile ORDER, the share of (tree, tile) pairs evaluated before the tree's first flagged tile is reached (what the early exit
cannot skip).  Orders: as launched (sequential), random, by the tile's largest |x| (descending), by its largest x.  -> gpurun_out/order.json"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api

dev = torch.device("cuda", 0)
ops = de.synth.BENCH_OPERATORS
trees = de.synth.random_population(1000, seed=0xDE02)
N = 10**7
TILE = 256
n_tiles = (N + TILE - 1) // TILE
res = {}
for seed in (1, 2, 3):
    g = torch.Generator(device=dev).manual_seed(seed)
    X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
    pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(full_eval=True))
    out, ok = pop.eval(X)
    f = ok.cpu().numpy().astype(bool) if hasattr(ok, "cpu") else np.asarray(ok).astype(bool)
    inc = np.nonzero(~f)[0]
    pad = n_tiles * TILE - N
    bad = torch.zeros((len(inc), n_tiles), dtype=torch.bool, device=dev)
    for k, t in enumerate(inc):
        b = ~torch.isfinite(out[int(t)])
        if pad:
            b = torch.cat([b, torch.zeros(pad, dtype=torch.bool, device=dev)])
        bad[k] = b.view(n_tiles, TILE).any(dim=1)
    del out
    Xp = X if not pad else torch.cat([X, torch.zeros((5, pad), device=dev)], dim=1)
    amax = Xp.abs().view(5, n_tiles, TILE).amax(dim=2).amax(dim=0)   # largest |x| of the tile
    smax = Xp.view(5, n_tiles, TILE).amax(dim=2).amax(dim=0)         # largest x
    smin = -Xp.view(5, n_tiles, TILE).amin(dim=2).amin(dim=0)

    def share(order):
        # position (in `order`) of each tree's first flagged tile
        b = bad[:, order]
        first = torch.where(b.any(dim=1), b.float().argmax(dim=1), torch.tensor(n_tiles, device=dev))
        return float(first.double().mean().item() / n_tiles), first

    r = {}
    seq = torch.arange(n_tiles, device=dev)
    r["sequential"], first_seq = share(seq)
    gen = torch.Generator(device=dev).manual_seed(100 + seed)
    r["random orders"] = [share(torch.randperm(n_tiles, device=dev, generator=gen))[0] for _ in range(5)]
    r["by largest |x| of the tile"] = share(torch.argsort(amax, descending=True))[0]
    r["by largest x"] = share(torch.argsort(smax, descending=True))[0]
    r["by smallest x"] = share(torch.argsort(smin, descending=True))[0]
    # the K most extreme tiles first, the rest as launched
    for K in (64, 512, 4096):
        top = torch.argsort(amax, descending=True)[:K]
        mask = torch.ones(n_tiles, dtype=torch.bool, device=dev)
        mask[top] = False
        r[f"the {K} tiles with the largest |x| first, then sequential"] = share(torch.cat([top, seq[mask]]))[0]
    r["incomplete"] = int(len(inc))
    r["tiles_flagging_share_quantiles"] = [float(q) for q in torch.quantile(bad.float().mean(dim=1), torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], device=dev))]
    late = first_seq > 1000
    r["trees first flagged after tile 1000 (sequential)"] = int(late.sum().item())
    r["... their share of the sequential cost"] = float(first_seq[late].double().sum().item() / first_seq.double().sum().item())
    res[f"seed {seed}"] = r
    print(seed, json.dumps(r), flush=True)
    pop.close()
    del X, Xp, bad
    torch.cuda.empty_cache()
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "order.json"), "w"), indent=1)
