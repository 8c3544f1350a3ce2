#!/usr/bin/env python3
"""NOTE: the DE_GRID_* rows need the experimental launcher of round 3 (measured and removed: DESIGN.md section 4.3,
profiles/r3_launch_order.json); without it every row is the one-workgroup-per-pair launch.
How many (tree, tile) pairs of the INCOMPLETE trees a launch still evaluates, by launch order (gpurun): the output is pre-filled
with a sentinel NaN and every 256-sample tile of every row that no longer holds it was evaluated.  -> gpurun_out/walk2.json"""
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
lib = api.library()
N = 10**7
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
pop = api.Population(trees, ops, np.float32, n_features=5)
out = torch.empty((1000, N), device=dev, dtype=torch.float32)
ok = torch.empty(1000, device=dev, dtype=torch.uint8)
SENT = 0x7FC12345
n_tiles = (N + 255) // 256
res = {}
for tag, env in [("one workgroup per pair", {"DE_GRID_LIFE": 1}), ("counters, life 4", {"DE_GRID_LIFE": 4}), ("static 87048", {"DE_GRID_STATIC": 87048}),
                 ("static 89096", {"DE_GRID_STATIC": 89096}), ("static 140936", {"DE_GRID_STATIC": 140936}), ("static 147080", {"DE_GRID_STATIC": 147080}),
                 ("one workgroup per pair, every flag access at agent scope", {"DE_GRID_LIFE": 1, "DE_SKIP_PROTOCOL": 1}),
                 ("static 89096, agent scope", {"DE_GRID_STATIC": 89096, "DE_SKIP_PROTOCOL": 1}),
                 ("static 87048, agent scope", {"DE_GRID_STATIC": 87048, "DE_SKIP_PROTOCOL": 1})]:
    for k in ("DE_GRID_LIFE", "DE_GRID_STATIC", "DE_SKIP_PROTOCOL"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    ms = []
    for i in range(3):
        out.view(torch.int32).fill_(SENT)
        pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        torch.cuda.synchronize()
        ms.append(pop.ctx.last_kernel_ms())
    f = ok.cpu().numpy().astype(bool)
    first = out.view(torch.int32)[:, ::256][:, :n_tiles]  # first sample of every tile
    ev = (first != SENT)
    per_tree = ev.sum(dim=1).cpu().numpy()
    inc = ~f
    by_xcd = [int(ev[torch.from_numpy(inc).to(dev)][:, x::8].sum().item()) for x in range(8)]
    r = dict(ms=float(np.median(ms)), incomplete=int(inc.sum()), tiles=n_tiles,
             evaluated_tile_share_of_incomplete_trees=float(per_tree[inc].sum() / (inc.sum() * n_tiles)),
             evaluated_tiles_complete_trees_share=float(per_tree[f].sum() / (f.sum() * n_tiles)),
             by_xcd=by_xcd, worst_trees=sorted((per_tree[inc] / n_tiles).tolist())[-12:])
    res[tag] = r
    print(tag, json.dumps(r), flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "walk2.json"), "w"), indent=1)
