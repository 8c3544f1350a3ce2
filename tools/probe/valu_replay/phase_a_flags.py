"""Phase A of the VALU replay (tools/probe/valu_replay/README.md), on the GPU box: which candidates of bench.py's `complete` population
(seed 0xDE0C, 3000 candidates) come out complete on the bench's X at N = 10^7 -> gpurun_out/valu_replay/complete_flags.json."""
import json, os, sys
sys.path.insert(0, '.')
import numpy as np
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
N = 10**7
ops = de.synth.BENCH_OPERATORS
Xh = de.synth.random_X(5, N, seed=1, dtype=np.float32)
X = torch.from_numpy(np.ascontiguousarray(Xh.T)).cuda().t()
cand = de.synth.random_population(3000, seed=0xDE0C)
out = torch.empty((1000, N), device="cuda", dtype=torch.float32)
flags = []
ctx = api.Context(0)
lib = api.library()
for b in range(0, 3000, 1000):
    pop = api.Population(cand[b:b + 1000], ops, np.float32, n_features=5, ctx=ctx)
    ok = torch.empty(1000, device="cuda", dtype=torch.uint8)
    ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
    torch.cuda.synchronize()
    flags += [int(v) for v in ok.cpu().numpy()]
    pop.close()
os.makedirs("gpurun_out/valu_replay", exist_ok=True)
json.dump({"seed": 0xDE0C, "n_candidates": 3000, "N": N, "complete": flags}, open("gpurun_out/valu_replay/complete_flags.json", "w"))
print(sum(flags), "of 3000 complete")
