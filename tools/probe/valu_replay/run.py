#!/usr/bin/env python3
"""VALU replay of the threaded eval kernel — runs on the GPU box (gpurun), see gen.py for what is replayed.

    gpurun -- 'python tools/probe/valu_replay/run.py'      ->  gpurun_out/valu_replay/result.json  (-> profiles/r6_valu_replay.md)

Population: bench.py's `complete` workload (1000 COMPLETE trees, seed 0xDE0C, rejection-sampled on the bench's X at N = 10^7:
complete_flags.json from phase_a_flags.py).  The REAL kernel is timed on it (nothing exits early: every tree-sample is executed), then
the four replay variants on its dispatch histogram, at the real launch's occupancy (5 one-wave workgroups per SIMD by LDS) and with as
many passes over the histogram as the real launch has sample tiles."""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
so = os.path.join(HERE, "libvalu_replay.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "replay_gen.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-shared", "-fPIC", "-w", "-o", so, os.path.join(HERE, "replay_gen.hip")], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402

N = 10**7
ops = de.synth.BENCH_OPERATORS
flags = json.load(open(os.path.join(HERE, "complete_flags.json")))["complete"]
cand = de.synth.random_population(3000, seed=0xDE0C)
trees = [t for t, k in zip(cand, flags) if k][:1000]
assert len(trees) == 1000
table = json.load(open(os.path.join(HERE, "handlers.json")))
slot_of = {int(k): v["slot"] for k, v in table["handlers"].items()}
valu_of = {int(k): v["valu_insts"] for k, v in table["handlers"].items()}

ctx = api.Context(0)
lib = api.library()
pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
counts = np.zeros(table["slots"], dtype=np.uint32)
unknown = 0
for t in range(len(trees)):
    n = lib.de_program_dump(pop._h, t, None, 0, 3)
    w = np.zeros(int(n), dtype=np.uint32)
    lib.de_program_dump(pop._h, t, w.ctypes.data, w.size, 3)
    for k in w.reshape(-1, 4)[:, 0]:
        if int(k) in slot_of:
            counts[slot_of[int(k)]] += 1
        else:
            unknown += 1
total = int(counts.sum())
valu_per_pass = sum(int(counts[slot_of[h]]) * valu_of[h] for h in slot_of) + len(trees) * table["tree_end_valu_insts"]
print(f"histogram: {total} dispatches of {len(trees)} trees ({total / len(trees):.2f} per tree), {unknown} without a replay body, "
      f"{valu_per_pass} VALU instructions per pass ({valu_per_pass / len(trees):.1f} per tree-wave)", flush=True)

# ---- the real kernel on this population
Xh = de.synth.random_X(5, N, seed=1, dtype=np.float32)
X = torch.from_numpy(np.ascontiguousarray(Xh.T)).cuda().t()
out = torch.empty((len(trees), N), device="cuda", dtype=torch.float32)
ok = torch.empty(len(trees), device="cuda", dtype=torch.uint8)
for _ in range(3):
    ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
ctx.synchronize()
ctx.timing_ring(10)
for _ in range(10):
    ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
ctx.synchronize()
real_ms = [v for v in ctx.timing_read() if v is not None]
ctx.timing_ring(0)
assert int(ok.sum().item()) == len(trees), "the population must be complete"
plan = pop.plan(N)
n_tiles = (N + plan["tile"] - 1) // plan["tile"]
print(f"real kernel: {np.mean(real_ms):.3f} ms (min {min(real_ms):.3f}) for {n_tiles} sample tiles x {len(trees)} trees, plan {plan}", flush=True)
del out
torch.cuda.empty_cache()

# ---- the replay
# Launch shape = the real launch's: MANY short one-wave workgroups (the real kernel: 39063 tiles x 16 chunks), so that the CUs stay filled to
# what the LDS allows and the tail is a few per cent — a workgroup runs 1 / DIV of the population's histogram (counts / DIV, rounded; the
# time is scaled by the VALU instructions really replayed), `n_tiles x DIV` workgroups.  Resident workgroups per CU are READ from the
# runtime (hipOccupancyMaxActiveBlocksPerMultiprocessor) for every LDS size.
rl = C.CDLL(so)
rl.valu_replay_run.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
rl.valu_replay_occupancy.argtypes = [C.c_int, C.c_uint32]
DIV = 8
c_div = np.maximum(np.rint(counts / DIV), (counts > 0)).astype(np.uint32)
trees_div = len(trees) // DIV
total_div = int(c_div.sum())
valu_div = sum(int(c_div[slot_of[h]]) * valu_of[h] for h in slot_of) + trees_div * table["tree_end_valu_insts"]
scale = (valu_per_pass / DIV) / valu_div  # rounding of the histogram
cd = torch.from_numpy(c_div.view(np.int32)).cuda()
res = {"population": "bench.py `complete` (1000 complete trees, seed 0xDE0C)", "dispatches_per_tree": total / len(trees), "valu_insts_per_tree_wave": valu_per_pass / len(trees),
       "real_kernel_ms": float(np.mean(real_ms)), "real_kernel_ms_min": float(min(real_ms)), "n_tiles": int(n_tiles), "workgroups": int(n_tiles * DIV),
       "histogram_share_per_workgroup": f"1/{DIV}", "variants": {}}
names = {0: "V0 VALU only", 1: "V1 = V0 + the handlers' scalar ALU instructions", 2: "V2 = V1 + LDS operand reads / spill writes",
         3: "V3 = V2 + per-dispatch store bookkeeping (3 scalar instructions, a branch), no store", 4: "V4 = V3 + one 1 KB non-temporal output store per tree"}
ms = C.c_float(0)
blocks = n_tiles * DIV
os.makedirs(os.path.join(ROOT, "gpurun_out", "valu_replay"), exist_ok=True)
json.dump({"counts": [int(v) for v in c_div], "blocks": int(blocks), "trees_div": int(trees_div), "total_div": int(total_div), "scale": float(scale)},
          open(os.path.join(ROOT, "gpurun_out", "valu_replay", "counts_div.json"), "w"))
stride = (blocks * 1024 + 4095) // 4096 * 4096
buf = torch.empty(trees_div * stride + 4096, device="cuda", dtype=torch.uint8)
for lds in (4096, 6144, 7168, 8192, 9216, 12288, 16384):
    for v in range(5):
        occ = rl.valu_replay_occupancy(v, lds)
        best = None
        for rep in range(3):
            rc = rl.valu_replay_run(v, cd.data_ptr(), 1, trees_div, total_div, buf.data_ptr(), stride, blocks, lds, C.byref(ms))
            assert rc == 0, rc
            best = ms.value if best is None else min(best, ms.value)
        scaled = best * scale
        res["variants"].setdefault(names[v], {})[f"LDS {lds} B per workgroup"] = {"resident_workgroups_per_cu": occ, "waves_per_simd": occ / 4, "ms": scaled}
        print(f"{names[v]:90s} LDS {lds:6d} B: {occ:3d} workgroups / CU = {occ / 4:5.2f} waves / SIMD: {scaled:7.3f} ms ({scaled / np.mean(real_ms):.3f} of the real kernel)", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "valu_replay"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "valu_replay", "result.json"), "w"), indent=1)
