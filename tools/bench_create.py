"""Cost of de_program_create (host lowering + uploads) vs Python-side flattening, of the first de_eval of a fresh program and of
de_program_destroy — what a search loop pays per GENERATION (new trees every call), next to the kernel it then runs.
DE_DEBUG_TIMING=1 prints the phases of the creation (stderr).     gpurun -- 'DE_DEBUG_TIMING=1 python tools/bench_create.py'"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
ctx = api.Context(0)
lib = api.library()
NROWS = 1000
X = torch.from_numpy(np.ascontiguousarray(np.asarray(de.synth.random_X(5, NROWS, seed=1, dtype=np.float32)).T)).cuda()
torch.cuda.synchronize()
import os
for nt in [int(float(v)) for v in os.environ.get('NTREES', '1e3,1e4').split(',')]:
    trees = de.synth.random_population(nt, seed=0xDE02)
    out = torch.empty((nt, NROWS), device="cuda", dtype=torch.float32)
    ok = torch.empty(nt, device="cuda", dtype=torch.uint8)
    yv = torch.randn(NROWS, device="cuda", dtype=torch.float32)
    lossv = torch.empty(nt, device="cuda", dtype=torch.float32)
    dlossv = torch.empty(nt * 20 + 16, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tape, noff, consts, coff = de.flatten_population(trees, ops, np.float32)
    t1 = time.perf_counter()
    best = None
    for rep in range(6):
        h = C.c_void_p()
        t2 = time.perf_counter()
        rc = lib.de_program_create(ctx._h, 0, tape.ctypes.data, noff.ctypes.data, nt, consts.ctypes.data, coff.ctypes.data, 5, 0, 7, C.byref(h))
        t3 = time.perf_counter()
        assert rc == 0
        ev = []
        for k in range(3):
            ta = time.perf_counter()
            ctx.check(lib.de_eval(ctx._h, h, X.data_ptr(), NROWS, 5, None, out.data_ptr(), NROWS, ok.data_ptr()))
            ctx.synchronize()
            ev.append(1e3 * (time.perf_counter() - ta))
        lg = []
        for k in range(3):  # the optimiser callback on the fresh program: fused loss + d loss / d constants (its gradient program is lowered on first use)
            ta = time.perf_counter()
            ctx.check(lib.de_eval_loss_grad(ctx._h, h, X.data_ptr(), NROWS, 5, None, 1, yv.data_ptr(), None, 0, lossv.data_ptr(), dlossv.data_ptr(), None, ok.data_ptr()))
            ctx.synchronize()
            lg.append(1e3 * (time.perf_counter() - ta))
        t4 = time.perf_counter()
        lib.de_program_destroy(h)
        t5 = time.perf_counter()
        cur = (1e3 * (t3 - t2), ev[0], ev[1], lg[0], lg[1], 1e3 * (t5 - t4))
        best = cur if best is None else tuple(min(a, b) for a, b in zip(best, cur))
        print(f"n_trees {nt}: python flatten {t1 - t0:.3f} s, de_program_create {1e3 * (t3 - t2):.2f} ms ({1e6 * (t3 - t2) / nt:.2f} us/tree), "
              f"de_eval x {NROWS} rows (call + sync) 1st {ev[0]:.3f} 2nd {ev[1]:.3f} 3rd {ev[2]:.3f} ms, de_eval_loss_grad 1st {lg[0]:.3f} 2nd {lg[1]:.3f} 3rd {lg[2]:.3f} ms, de_program_destroy {1e3 * (t5 - t4):.2f} ms", flush=True)
    print(f"n_trees {nt} BEST OF 6 (the box's host is shared: single runs scatter by 2 x): de_program_create {best[0]:.2f} ms ({1e3 * best[0] / nt:.2f} us/tree), "
          f"de_eval 1st {best[1]:.3f} / steady {best[2]:.3f} ms, de_eval_loss_grad 1st {best[3]:.3f} / steady {best[4]:.3f} ms, de_program_destroy {best[5]:.2f} ms", flush=True)
