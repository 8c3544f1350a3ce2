"""Histogram of the bound (folded) eval program of the bench population: unigrams/bigrams/trigrams."""
import collections
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api

ops = de.synth.BENCH_OPERATORS
trees = de.synth.random_population(1000, seed=0xDE02)
pop = api.Population(trees, ops, np.float32, n_features=5)
lib = api.library()
BIN = ["ADD", "SUB", "RSUB", "MUL", "DIV", "RDIV"]
UN = ["COS", "EXP", "SIN"]


def name(b, arg):
    if b == 0: return "LOAD_ROW" + ("S" if (arg & 0xFFFFFF) >= 5 else "F")
    if b == 1: return "LOAD_CONST"
    if b == 2: return "PUSH"
    if b == 3: return "CHECK_ROW"
    if b == 4: return "CHECK_ACC"
    if 5 <= b < 29:
        k = b - 5
        s = "c" if k & 2 else ("S" if (arg & 0xFFFFFF) >= 5 else "F")
        return BIN[k // 4] + ":" + s + ("!" if k & 1 else "")
    if 29 <= b < 41:
        k = b - 29
        s = ("S" if (arg & 0xFFFFFF) >= 5 else "F") if k & 2 else "acc"
        return UN[k // 4] + ":" + s + ("!" if k & 1 else "")
    return f"op{b}"


uni, bi, tri = collections.Counter(), collections.Counter(), collections.Counter()
tot = 0
for t in range(1000):
    n = lib.de_program_dump(pop._h, t, None, 0, 2)
    w = np.zeros(int(n), dtype=np.uint32)
    lib.de_program_dump(pop._h, t, w.ctypes.data, w.size, 2)
    w = w.reshape(-1, 4)
    seq = [name(int(r[0]), int(r[1])) for r in w]
    tot += len(seq)
    uni.update(seq); bi.update(zip(seq, seq[1:])); tri.update(zip(seq, seq[1:], seq[2:]))
print("dispatches", tot)
for k, v in uni.most_common(45): print(f"{k:14s} {v:6d} {v/tot:.3f}")
print("--- bigrams")
for k, v in bi.most_common(40): print(k, v, f"{v/tot:.3f}")
print("--- trigrams")
for k, v in tri.most_common(15): print(k, v, f"{v/tot:.3f}")
