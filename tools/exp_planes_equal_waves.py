#!/usr/bin/env python3
"""Round 5: what would two planes per lane buy at UNCHANGED occupancy?  A proxy that needs no new kernel: a population over ONE feature
whose trees need <= 1 spill slot has 2 LDS rows per workgroup, so the two-plane build (DE_TG=2: 80 + 16 registers) is REGISTER-limited
at 5 waves per SIMD — what a workgroup of 4 waves sharing one staged X tile would reach with 5 features — and the one-plane build, padded
with DE_EXTRA_LDS_ROWS to the 5.25 waves of the shipped headline launch, is its like-for-like partner.
   gpurun -- 'for l in default variants/libde_hip_tg2.so; do for e in 0 5; do DE_HIP_LIB_SEL=$l DE_EXTRA_LDS_ROWS=$e python tools/exp_planes_equal_waves.py; done; done'
(DE_HIP_LIB_SEL=default | path of a variant library)"""
import json
import os
import sys
import time

import numpy as np
import torch

sel = os.environ.get("DE_HIP_LIB_SEL", "default")
if sel != "default":
    os.environ["DE_HIP_LIB"] = os.path.abspath(sel)
sys.path.insert(0, '.')
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api

N = 10**7
F = int(os.environ.get("EXP_F", "1"))
NT = int(os.environ.get("EXP_TREES", "600"))
dev = torch.device("cuda", 0)
ops = de.synth.BENCH_OPERATORS
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((N, F), generator=g, device=dev, dtype=torch.float32).t()
lib = api.library()
ctx = api.Context(0)
cand = de.synth.random_population(4000, seed=0xDE0C, nfeatures=F)


def slots_of(tree):
    w = np.zeros(4, dtype=np.uint32)
    p = api.Population([tree], ops, np.float32, n_features=F, ctx=ctx)
    lib.de_program_dump(p._h, 0, w.ctypes.data, 4, 1)
    p.close()
    return int(w[0])


def run(trees, steps=10, warmup=2):
    pop = api.Population(trees, ops, np.float32, n_features=F, ctx=ctx)
    out = torch.empty((len(trees), N), device=dev, dtype=torch.float32)
    ok = torch.empty(len(trees), device=dev, dtype=torch.uint8)

    def step():
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, None, out.data_ptr(), N, ok.data_ptr()))
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    okh = ok.cpu().numpy().astype(bool)
    chk = float(out[okh.nonzero()[0][:50]].double().sum().item()) if okh.any() else 0.0
    pop.close()
    del out
    return ms, okh, chk


_, okc, _ = run(cand[:1500], steps=1, warmup=0)
complete = [t for t, k in zip(cand[:1500], okc) if k]
max_slots = int(os.environ.get("EXP_MAX_SLOTS", "1"))
sel_trees = [t for t in complete if slots_of(t) <= max_slots][:NT]
ms, okh, chk = run(sel_trees)
print(json.dumps({"lib": sel, "extra_lds_rows": int(os.environ.get("DE_EXTRA_LDS_ROWS", "0")), "F": F, "trees": len(sel_trees),
                  "max_slots": max_slots, "all_complete": bool(okh.all()), "ms_per_step": round(ms, 3),
                  "us_per_tree": round(1e3 * ms / max(len(sel_trees), 1), 3), "checksum": chk}))
