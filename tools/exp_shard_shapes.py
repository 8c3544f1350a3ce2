#!/usr/bin/env python3
"""Per-rank launch shapes of the multi-GPU runs, measured on ONE GPU (no 8-GPU node was available to the builder):
rank 0's shard of the headline population under strong scaling (1000 / 500 / 250 / 125 trees x 10^7 samples) and of BASELINE config 4
(1250 of 10000 trees), exact mode, default path / dataset declared (de_ctx_declare_dataset: no per-call pass over X) / priority tiles off.
Projected strong-scaling efficiency = T(1 rank) / (world x T(rank 0's shard)): what the kernels allow, before any xGMI effect (the only
exchange is one all_gather of n_trees flag bytes per step).

    gpurun -- 'python tools/exp_shard_shapes.py > gpurun_out/shard_shapes.json'"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api, dist as dedist

N = 10**7
dev = torch.device("cuda", 0)
ops = de.synth.BENCH_OPERATORS
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()
lib = api.library()
ctx = api.Context(0)


def timed(trees, steps=30, warmup=3):
    pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
    out = torch.empty((len(trees), N), device=dev, dtype=torch.float32)
    ok = torch.empty(len(trees), device=dev, dtype=torch.uint8)
    def step():
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    live = pop.last_live_trees()
    complete = int(ok.sum().item())
    pop.close()
    del out
    return ms, live, complete


res = dict(N=N, note="one MI355X, rank 0's shard; UNMEASURED on a multi-GPU node", shapes=[])
full = de.synth.random_population(1000, seed=0xDE02)
c4 = de.synth.random_population(10000, seed=0xDE04)
cases = [("headline", full, w) for w in (1, 2, 4, 8)] + [("C4", c4, 8)]
base = {}
for name, popu, world in cases:
    trees = [popu[i] for i in dedist.shard_indices(len(popu), 0, world)]
    row = dict(workload=name, world=world, trees_this_rank=len(trees))
    ctx.declare_dataset(None)
    row["ms_default"], row["live_after_probe"], row["complete"] = timed(trees)
    ctx.declare_dataset(X)
    row["ms_dataset_declared"], _, _ = timed(trees)
    ctx.declare_dataset(None)
    os.environ["DE_NO_PRIO_TILES"] = "1"
    row["ms_no_priority_tiles"], _, _ = timed(trees)
    del os.environ["DE_NO_PRIO_TILES"]
    if name == "headline":
        if world == 1:
            base = dict(row)
        row["projected_strong_scaling_efficiency"] = base["ms_default"] / (world * row["ms_default"])
        row["projected_strong_scaling_efficiency_dataset_declared"] = base["ms_dataset_declared"] / (world * row["ms_dataset_declared"])
    row["us_per_complete_tree"] = 1e3 * row["ms_default"] / max(row["complete"], 1)
    res["shapes"].append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
# ---- the exchange of a step, as far as one GPU can run it (round 5; UNMEASURED on a multi-GPU node) -------------------------------
# (a) the pack + unpack launches of de_dist_gather_flags for a simulated world (de_dist_reorder_selftest: device time of ONE rank's
#     share, no collective); (b) the world-size-1 gather through the C ABI, wall clock around call + synchronise (a device copy).
import ctypes as C  # noqa: E402
import time  # noqa: E402
lib = api.library()
ex = []
for n_trees, world in ((1000, 2), (1000, 4), (1000, 8), (10000, 8)):
    flags = (np.arange(n_trees) % 3 != 0).astype(np.uint8)
    outf = np.zeros(n_trees, dtype=np.uint8)
    ms = C.c_float(0)
    best = 1e9
    for _ in range(5):
        ctx.check(lib.de_dist_reorder_selftest(ctx._h, flags.ctypes.data, n_trees, world, outf.ctypes.data, C.byref(ms)))
        best = min(best, float(ms.value))
    assert np.array_equal(outf, flags)
    ex.append(dict(n_trees=n_trees, world=world, pack_plus_unpack_us=1e3 * best))
comm = dedist.Comm(ctx, 0, 1, b"")
okd = torch.ones(1000, device=dev, dtype=torch.uint8)
for _ in range(3):
    comm.gather_flags(okd, 1000)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    comm.gather_flags(okd, 1000)
torch.cuda.synchronize()
res["exchange"] = dict(note="pack + unpack = the library's own launches around ncclAllGather (3 stream operations per exchange since round 5, 11 before at 8 ranks); "
                            "the collective itself (1000 bytes per rank, latency-bound over xGMI) is NOT in these numbers: no multi-GPU node",
                       simulated_world=ex, world_size_1_gather_us_per_call=1e6 * (time.perf_counter() - t0) / 50)
comm.close()
print(json.dumps(res, indent=1))
