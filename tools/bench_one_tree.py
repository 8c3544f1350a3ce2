"""The reference's own call shape at the sizes a search evaluates ONE tree on: de_eval_tree_array (create + eval + synchronise + destroy in
one call) for a 20-node tree x N rows, device pointers.  DE_DEBUG_TIMING=1 prints the phases of the creation.
    gpurun -- 'python tools/bench_one_tree.py'"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
ctx = api.Context(0)
lib = api.library()
trees = de.synth.random_population(64, seed=0xDE02)
tapes = [de.flatten(t, ops, np.float32) for t in trees]
for N in (1000, 100000, 10**7):
    X = torch.from_numpy(np.ascontiguousarray(np.asarray(de.synth.random_X(5, N, seed=1, dtype=np.float32)).T)).cuda()
    out = torch.empty(N, device="cuda", dtype=torch.float32)
    ok = torch.zeros(1, device="cuda", dtype=torch.uint8)
    torch.cuda.synchronize()
    best = 1e9
    tot = []
    for rep in range(5):
        t0 = time.perf_counter()
        for tape, consts in tapes:
            rc = lib.de_eval_tree_array(ctx._h, 0, tape.ctypes.data, len(tape), consts.ctypes.data if len(consts) else None, len(consts),
                                        X.data_ptr(), 5, N, 7, out.data_ptr(), ok.data_ptr())
            assert rc == 0, lib.de_last_error(ctx._h)
        tot.append(1e3 * (time.perf_counter() - t0) / len(tapes))
    # the same trees as ONE population, program kept: what the call costs without the creation
    pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
    out2 = torch.empty((len(trees), N), device="cuda", dtype=torch.float32)
    ok2 = torch.zeros(len(trees), device="cuda", dtype=torch.uint8)
    ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out2.data_ptr(), N, ok2.data_ptr())); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out2.data_ptr(), N, ok2.data_ptr()))
    ctx.synchronize()
    t_pop = 1e3 * (time.perf_counter() - t0) / 5 / len(trees)
    pop.close()
    print(f"N {N}: de_eval_tree_array {min(tot):.3f} ms per tree (one call per tree: create + eval + sync + destroy); the same 64 trees as one kept population: {t_pop:.4f} ms per tree", flush=True)
