#!/usr/bin/env python3
"""What the stream variants of a wave group cost at de_program_create / de_program_set_consts (10^4 parametric trees, 8 parameters)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dynamicexpressions_jl_amd as de  # noqa: E402
from dynamicexpressions_jl_amd import api  # noqa: E402

ops = de.synth.BENCH_OPERATORS
ctx = api.Context(0)
lib = api.library()
import ctypes as C  # noqa: E402
for kind in ("parametric", "wide X (F = 20)"):
    if kind == "parametric":
        trees = de.synth.random_population(10_000, seed=0xDE05, node_type=de.ParametricNode, nparams=8)
        F, P = 5, 8
    else:
        trees = de.synth.random_population(10_000, seed=0xDE06, nfeatures=20)
        F, P = 20, 0
    nodes, noff, consts, coff = de.flatten_population(trees, ops, np.float32)
    for waves in ("1", "2", "4"):
        os.environ["DE_EVAL_WAVES"] = waves
        best_c, best_s = 1e9, 1e9
        for rep in range(6):
            h = C.c_void_p()
            t0 = time.perf_counter()
            ctx.check(lib.de_program_create(ctx._h, 0, nodes.ctypes.data, noff.ctypes.data, len(trees), consts.ctypes.data, coff.ctypes.data, F, P,
                                            api.EvalContext().option_bits(ops), C.byref(h)))
            t1 = time.perf_counter()
            c2 = (consts * np.float32(1.01)).astype(np.float32)
            ctx.check(lib.de_program_set_consts(h, c2.ctypes.data))
            t2 = time.perf_counter()
            lib.de_program_destroy(h)
            if rep:
                best_c, best_s = min(best_c, t1 - t0), min(best_s, t2 - t1)
        print(f"{kind}: DE_EVAL_WAVES={waves}: de_program_create {best_c * 1e3:.2f} ms, de_program_set_consts {best_s * 1e3:.2f} ms (10^4 trees, best of 5)", flush=True)
