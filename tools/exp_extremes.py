#!/usr/bin/env python3
"""Do the samples with extreme feature values flag the incomplete trees?  (gpurun)  The headline population on (a) the first k tiles,
(b) k random tiles, (c) the k tiles holding the largest |x_f| / largest x_f / smallest x_f of each feature: how many of the trees that
are incomplete on all 10^7 samples are already flagged.  -> gpurun_out/extremes.json"""
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
N, TILE, F = 10**7, 256, 5
res = {}
for seed in (1, 2):
    g = torch.Generator(device=dev).manual_seed(seed)
    X = torch.randn((N, F), generator=g, device=dev, dtype=torch.float32).t().contiguous()  # [F, N] row-major here (ld = N)
    Xs = X.t().contiguous().t()  # [F, N] with features contiguous per sample (the API's column-major [F, N])
    pop = api.Population(trees, ops, np.float32, n_features=F)
    _, ok = pop.eval(Xs)
    full = ok.cpu().numpy().astype(bool) if hasattr(ok, "cpu") else np.asarray(ok).astype(bool)
    n_tiles = N // TILE
    Xt = X[:, : n_tiles * TILE].view(F, n_tiles, TILE)

    def flagged(tiles):
        tiles = torch.as_tensor(tiles, device=dev).long()
        sub = Xt[:, tiles, :].reshape(F, -1)
        Xsub = sub.t().contiguous().t()
        _, ok2 = pop.eval(Xsub)
        o = ok2.cpu().numpy().astype(bool) if hasattr(ok2, "cpu") else np.asarray(ok2).astype(bool)
        return int((~o & ~full).sum())

    r = {"incomplete on all samples": int((~full).sum())}
    rng = np.random.default_rng(seed)
    for k in (16, 64, 256, 1024):
        per = max(1, k // (3 * F))
        ext = set()
        for f in range(F):
            hi = Xt[f].amax(dim=1); lo = Xt[f].amin(dim=1); ab = Xt[f].abs().amin(dim=1)
            ext |= set(torch.topk(hi, per).indices.tolist()) | set(torch.topk(-lo, per).indices.tolist()) | set(torch.topk(-ab, per).indices.tolist())
        ext = sorted(ext)[:k]
        r[f"{k} tiles"] = {"first": flagged(list(range(k))), "random": flagged(rng.choice(n_tiles, k, replace=False).tolist()),
                           "extremes (largest, smallest, closest to 0 per feature)": flagged(ext), "n_extreme_tiles": len(ext)}
    res[f"seed {seed}"] = r
    print(seed, json.dumps(r), flush=True)
    pop.close()
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "extremes.json"), "w"), indent=1)
