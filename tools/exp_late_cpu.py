#!/usr/bin/env python3
"""CPU study of the trees the probe launch misses (no GPU needed): for the headline population on the bench's PCG64 X, on which
512-sample tiles does every incomplete tree fail, which of them hold one of the 3 F priority tiles (largest / smallest / closest to
zero per feature, de_tile_extremes_kernel), and which other per-tile keys would have caught the late trees.

    python tools/exp_late_cpu.py [N] [workers]  ->  gpurun_out/late_cpu_<N>.json   (N = 10^7: ~6 min on 8 cores)

Float32 numpy evaluation of every node of a tree on every sample, no early exit: a sample fails a tree when ANY node value is
non-finite there (what the reference's per-node checks see, src/Evaluate.jl:16-32)."""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import dynamicexpressions_jl_amd as de

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**6
WORKERS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
SEED = int(os.environ.get("POP_SEED", str(0xDE02)), 0)
TILE = 512
OPS = de.synth.BENCH_OPERATORS
BIN = OPS.binops
UNA = OPS.unaops
_X = None


def X():
    global _X
    if _X is None:
        _X = np.asarray(de.synth.random_X(5, N, seed=1, dtype=np.float32))  # (5, N)
    return _X


def ev(node, Xm, bad):
    if node.degree == 0:
        if node.constant:
            return np.float32(node.val)
        return Xm[node.feature - 1]
    if node.degree == 1:
        a = ev(node.children[0], Xm, bad)
        name = UNA[node.op - 1]
        r = {"cos": np.cos, "exp": np.exp}[name](a)
    else:
        a = ev(node.children[0], Xm, bad)
        b = ev(node.children[1], Xm, bad)
        name = BIN[node.op - 1]
        r = {"+": np.add, "-": np.subtract, "*": np.multiply, "/": np.divide}[name](a, b)
    r = np.asarray(r, dtype=np.float32)
    if r.ndim == 0:
        if not np.isfinite(r):
            bad |= True
    else:
        bad |= ~np.isfinite(r)
    return r


def fail_tiles(t):
    tree = TREES[t]
    Xm = X()
    bad = np.zeros(N, dtype=bool)
    with np.errstate(all="ignore"):
        ev(tree, Xm, bad)
    nt = (N + TILE - 1) // TILE
    pad = np.zeros(nt * TILE, dtype=bool)
    pad[:N] = bad
    tiles = np.flatnonzero(pad.reshape(nt, TILE).any(axis=1))
    return t, int(bad.sum()), tiles.astype(np.int32)


TREES = de.synth.random_population(1000, seed=SEED)


def main():
    Xm = X()
    nt = (N + TILE - 1) // TILE
    # the 3 F priority tiles of the library
    prio = set()
    for f in range(5):
        prio.add(int(np.argmax(Xm[f])) // TILE)
        prio.add(int(np.argmin(Xm[f])) // TILE)
        prio.add(int(np.argmin(np.abs(Xm[f]))) // TILE)
    with ProcessPoolExecutor(WORKERS) as ex:
        res = list(ex.map(fail_tiles, range(len(TREES)), chunksize=4))
    incomplete = [(t, ns, tl) for t, ns, tl in res if ns > 0]
    caught = [t for t, ns, tl in incomplete if prio & set(tl.tolist())]
    late = [(t, ns, tl) for t, ns, tl in incomplete if not (prio & set(tl.tolist()))]
    # per-tile candidate keys (computed once): rank of a tile under each key = how early a key-ordered launch reaches it
    pad = np.full((5, nt * TILE), np.nan, dtype=np.float32)
    pad[:, :N] = Xm
    T = pad.reshape(5, nt, TILE)
    keys = {}
    for f in range(5):
        keys[f"max_x{f+1}"] = -np.nanmax(T[f], axis=1)
        keys[f"min_x{f+1}"] = np.nanmin(T[f], axis=1)
        keys[f"minabs_x{f+1}"] = np.nanmin(np.abs(T[f]), axis=1)
    keys["max_linf"] = -np.nanmax(np.abs(T), axis=(0, 2))
    keys["min_linf_abs"] = np.nanmin(np.abs(T), axis=(0, 2))
    ranks = {k: np.argsort(np.argsort(v, kind="stable"), kind="stable") for k, v in keys.items()}  # rank 0 = most extreme tile
    out_late = []
    for t, ns, tl in late:
        best = {k: int(r[tl].min()) for k, r in ranks.items()}
        kbest = min(best, key=best.get)
        out_late.append(dict(tree=t, failing_samples=ns, failing_tiles=int(len(tl)), expected_share_run=1.0 / (len(tl) + 1),
                             best_key=kbest, best_rank=best[kbest],
                             best_rank_per_feature_key=int(min(v for k, v in best.items() if "_x" in k))))
    # what the late trees cost in a launch whose tiles run in random order: E[share of tiles run before the first failing one] = 1 / (k + 1)
    cost = sum(o["expected_share_run"] for o in out_late)
    # ... and with the per-feature keys extended to the top-R tiles of each of the 3 F statistics
    by_R = {}
    for R in (1, 2, 4, 8, 16, 32, 64):
        sel = set()
        for k, r in ranks.items():
            if "_x" in k:
                sel |= set(np.flatnonzero(r < R).tolist())
        still = [o for (t, ns, tl), o in zip(late, out_late) if not (sel & set(tl.tolist()))]
        by_R[R] = dict(probe_tiles=len(sel), late=len(still), late_cost_tree_equivalents=round(sum(o["expected_share_run"] for o in still), 2))
    summary = dict(N=N, tile=TILE, n_tiles=nt, seed=hex(SEED), incomplete=len(incomplete), caught_by_probe=len(caught), late=len(late),
                   late_cost_tree_equivalents=round(cost, 2), top_R_per_statistic=by_R,
                   late_by_failing_tiles={"1": sum(o["failing_tiles"] == 1 for o in out_late), "2-3": sum(2 <= o["failing_tiles"] <= 3 for o in out_late),
                                          "4-15": sum(4 <= o["failing_tiles"] <= 15 for o in out_late), ">=16": sum(o["failing_tiles"] >= 16 for o in out_late)},
                   late_trees=sorted(out_late, key=lambda o: o["failing_tiles"]))
    print(json.dumps({k: v for k, v in summary.items() if k != "late_trees"}, indent=1))
    for o in summary["late_trees"][:80]:
        print(o)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(summary, open(os.path.join(ROOT, "gpurun_out", f"late_cpu_{N}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
