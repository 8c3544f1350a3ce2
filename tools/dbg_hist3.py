"""Fused-program (stage 3) dispatch histogram of the bench population by handler id -> gpurun_out/hist3.json"""
import collections, json, sys
sys.path.insert(0, '.')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
trees = de.synth.random_population(1000, seed=0xDE02)
pop = api.Population(trees, ops, np.float32, n_features=5)
lib = api.library()
c = collections.Counter()
for t in range(1000):
    n = lib.de_program_dump(pop._h, t, None, 0, 3)
    w = np.zeros(int(n), dtype=np.uint32)
    lib.de_program_dump(pop._h, t, w.ctypes.data, w.size, 3)
    c.update(int(v) for v in w.reshape(-1, 4)[:, 0])
json.dump({str(k): v for k, v in c.items()}, open('gpurun_out/hist3.json', 'w'))
print(sum(c.values()))
