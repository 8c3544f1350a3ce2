"""Cost of de_program_create (host lowering + uploads) vs Python-side flattening."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
ctx = api.Context(0)
lib = api.library()
for nt in (1000, 10000):
    trees = de.synth.random_population(nt, seed=0xDE02)
    t0 = time.perf_counter()
    tape, noff, consts, coff = de.flatten_population(trees, ops, np.float32)
    t1 = time.perf_counter()
    for rep in range(3):
        h = C.c_void_p()
        t2 = time.perf_counter()
        rc = lib.de_program_create(ctx._h, 0, tape.ctypes.data, noff.ctypes.data, nt, consts.ctypes.data, coff.ctypes.data, 5, 0, 7, C.byref(h))
        t3 = time.perf_counter()
        assert rc == 0
        lib.de_program_destroy(h)
        print(f"n_trees {nt}: python flatten {t1 - t0:.3f} s, de_program_create {1e3 * (t3 - t2):.1f} ms ({1e6 * (t3 - t2) / nt:.1f} us/tree)")
