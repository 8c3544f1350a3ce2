#!/usr/bin/env python3
"""bench.py — the hot path of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of eval_tree_array over the whole population: every tree of this rank's
shard evaluated on every sample of X (already resident in HBM).  Workload at N=1 = the
configuration BASELINE.json's metric is quoted on: 1000 random depth<=15 20-node trees x
(5 features x 10^7 samples) Float32 (`--workload C2` gives the 10^6-sample config).  With N
GPUs the population is tree-sharded (weak scaling: 1000 trees per GPU, X replicated, no
data-path collective; one RCCL all_gather of the per-tree completion flags per step).
`--gpus N` with N > 1 and no launcher environment re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU), so the plain command
form produces the N-GPU line as well.  `--workload C4` is BASELINE config 4: 10 000 trees x
10^7 samples tree-sharded 8 ways (rank r owns trees {t : t mod 8 = r}, 1 250 trees and 50 GB of
output per GPU); with fewer than 8 ranks the job covers the first N shards (N=1: rank 0's shard).

Rank 0 prints ONE JSON line: metric node-evals/s (whole job), plus
  roofline     — the dominant kernel's ALGORITHMIC bytes / its average launch duration measured
                 with hipEvents on the launch stream inside the timed region, against 8 TB/s HBM3E.
                 The kernel stages each X tile once per chunk of K_eff trees, so per SURVEY.md §8d
                 the algorithmic figure is F*s/K_eff + s bytes per tree-sample (4.3 B at K_eff=63),
                 NOT the 24 B of a one-tree-per-pass design; `traffic` is the rocprofv3 PMC
                 measurement of the same launch (profiles/), and `single_tree_equivalent` restates
                 the rate in the north-star's 24 B/tree-sample accounting.  After X reuse the path
                 is issue-bound (VALU + scalar), not HBM-bound: see DESIGN.md §Roofline.
                 `roofline.valu` is the binding ceiling: VALU issue slots per tree-wavefront (the
                 population's dispatch histogram x the per-handler ISA slot counts of
                 profiles/valu_slots.json, tools/valu_slots.py) -> SIMD cycles at 2.4 GHz -> the
                 fraction of that floor the measured kernel time reaches.
  cpu_baseline — the CPU oracle (C restatement of the reference algorithm) timed on bounded samples
                 of the same workload on this box's host cores: one tree per task on ALL host cores
                 (`value`, `cores`) and on ONE thread (`single_thread`), plus the result of probing
                 for a `julia` binary (`julia_probe`; the reference itself cannot run on this box).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
F_FEATURES, ELEM = 5, 4
BYTES_PER_TREE_SAMPLE_SINGLE = (F_FEATURES + 1) * ELEM  # 24 B: one tree per pass over X

WORKLOADS = {
    "headline": dict(n_trees=1000, N=10**7, desc="1000 random depth<=15 20-node trees x (5 x 10^7) Float32"),
    "C2": dict(n_trees=1000, N=10**6, desc="1000 random depth<=15 20-node trees x (5 x 10^6) Float32"),
    "C3": dict(n_trees=1000, N=10**6, grad=True,
               desc="1000 random depth<=15 20-node trees x (5 x 10^6) Float32, eval_grad_tree_array(variable=true)"),
    "loss": dict(n_trees=1000, N=10**7, loss=True,
                 desc="1000 random depth<=15 20-node trees x (5 x 10^7) Float32, fused sum(abs2, tree(X) .- y)"),
    "lossgrad": dict(n_trees=1000, N=10**6, lossgrad=True,
                     desc="1000 random depth<=15 20-node trees x (5 x 10^6) Float32, fused loss + d loss/d constants "
                          "(optimiser callback, test/test_optim.jl:42-51)"),
    "C5": dict(n_trees=1000, N=10**6, parametric=True,
               desc="1000 random 20-node ParametricNode trees (8 parameters x 16 classes), 5 x 10^6 Float32: "
                    "eval_tree_array + eval_grad_tree_array(variable=false) per step"),
    "C5pb": dict(n_trees=1000, N=10**6, parametric=True, by_class=True, reverse_grad=True,
                 desc="[EvalContext(reverse_grad=true) = DE_OPT_REVERSE_GRAD: reverse accumulation, an opt-in since ABI 3] "
                      "1000 random 20-node ParametricNode trees (8 parameters x 16 classes, samples grouped by class), "
                      "5 x 10^6 Float32: eval_tree_array + the :both-mode pullback (dY = randn) with the parameter rows "
                      "reduced by class, fused (SURVEY.md §8d C5; src/ChainRules.jl:56-77, "
                      "test/test_parametric_expression.jl:326-372)"),
    "C4": dict(n_trees=1250, N=10**7, shards=8, seed=0xDE04,
               desc="10000 random depth<=15 20-node trees x (5 x 10^7) Float32, tree-sharded x8 (rank r: trees t = r mod 8; "
                    "1250 trees, 50 GB of output per GPU)"),
    "C5N": dict(n_trees=1000, N=10**6, parametric=True, per_sample=True,
                desc="1000 random 20-node ParametricNode trees, 8 PER-SAMPLE parameters (C = N classes, classes = 1:N), "
                     "5 x 10^6 Float32: eval_tree_array (SURVEY.md §8d C5 stress)"),
    "C5Ng": dict(n_trees=1000, N=10**6, parametric=True, per_sample=True, per_sample_grad=True,
                 desc="1000 random 20-node ParametricNode trees, 8 PER-SAMPLE parameters (C = N classes, classes = 1:N), "
                      "5 x 10^6 Float32: eval_tree_array + eval_grad_tree_array(variable=false) per step (BASELINE config 5 read literally)"),
    "complete": dict(n_trees=1000, N=10**7, complete=True,
                     desc="1000 COMPLETE random depth<=15 20-node trees (the headline's generator, seed 0xDE0C, rejection-sampled on this X: nothing exits "
                          "early) x (5 x 10^7) Float32 — the kernel-quality workload (the headline's `complete_only` leg as a workload of its own, for profiling)"),
    "tiny": dict(n_trees=64, N=10**5, desc="64 trees x (5 x 10^5) Float32 (plumbing)"),
}


def _oracle_timed(work_one, sizes, n_samples, threads, budget_s):
    """Whole trees of the workload on `threads` host threads for ~budget_s; returns (node-evals, trees, seconds).
    `work_one(i)` runs the oracle on tree i (the reference's call for this workload), `sizes[i]` = its node count."""
    from concurrent.futures import ThreadPoolExecutor
    deadline = [0.0]
    done = []

    def work(i):
        if time.perf_counter() > deadline[0]:
            return
        work_one(i)
        done.append(sizes[i])

    t0 = time.perf_counter()
    deadline[0] = t0 + budget_s
    if threads == 1:
        for i in range(len(sizes)):
            work(i)
    else:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(work, range(len(sizes))))
    dt = time.perf_counter() - t0
    return sum(done) * n_samples, len(done), dt


def julia_probe():
    """probe for the reference's own runtime and record the result (it is not in this image)."""
    import shutil
    import subprocess
    exe = shutil.which("julia")
    if not exe:
        return "julia: not found on PATH (the Julia reference cannot run on this box; baseline = C restatement)"
    try:
        v = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:  # pragma: no cover
        v = f"{type(e).__name__}: {e}"
    return f"{exe}: {v} (DynamicExpressions.jl itself is not installed: no network, no depot)"


def cpu_baseline(trees, ops, X_host, budget_s=11.0, budget_1t_s=7.0, kind="eval", params=None, classes=None):
    """Oracle (kind 'port') on bounded samples of the same workload: whole trees at N = 10^6 samples (arrays
    far larger than L2, like the reference at this scale), one tree per task — the reference evaluates one tree
    per call single-threaded and its callers parallelise a population over trees — (i) on ALL host cores of the
    box (ctypes releases the GIL inside the C call; `cores` = threads used) and (ii) on one thread.
    `kind`: "eval" = eval_tree_array (src/Evaluate.jl:279-309); "grad" = eval_grad_tree_array(variable=true)
    (src/EvaluateDerivative.jl:193-228; config 3); "param" = eval_tree_array of a ParametricExpression
    (src/ParametricExpression.jl:371-390) + the constant-mode gradient on the gathered [params; X] matrix — the
    reference's own reduction of the parametric case, built ONCE outside the timed region (the reference rebuilds
    it in every call: the baseline is the more favourable to the CPU) (config 5).  A step's node-evals are counted
    as on the device: nodes x samples per tree, whatever the step computes per node."""
    import dynamicexpressions_jl_amd as de
    from oracle import oracle
    tapes = [de.flatten(t, ops, np.float32) for t in trees]
    sizes = [len(tp) for tp, _ in tapes]
    N = X_host.shape[1]
    if kind == "eval":
        what_call = "eval_tree_array (C restatement of src/Evaluate.jl, early exit on"

        def work_one(i):
            oracle.eval_tree_array(tapes[i][0], tapes[i][1], X_host)
    elif kind == "grad":
        what_call = "eval_grad_tree_array(variable=true) (C restatement of src/EvaluateDerivative.jl, early exit on"

        def work_one(i):
            oracle.eval_grad_tree_array(tapes[i][0], tapes[i][1], X_host, oracle.GRAD_VARIABLE)
    elif kind == "param":
        what_call = ("eval_tree_array(::ParametricExpression) + eval_grad_tree_array(variable=false) on [params[:, classes]; X] "
                     "(C restatement of src/ParametricExpression.jl:371-390 / src/EvaluateDerivative.jl, early exit on")
        plain = [oracle.parametric_to_plain(tp, X_host[:, :1], params, classes[:1])[0] for tp, _ in tapes]  # the re-indexed tapes
        _, PX = oracle.parametric_to_plain(tapes[0][0], X_host, params, classes)  # [params[:, classes]; X], once

        def work_one(i):
            oracle.eval_tree_array_parametric(tapes[i][0], tapes[i][1], X_host, params, classes)
            oracle.eval_grad_tree_array(plain[i], tapes[i][1], PX, oracle.GRAD_CONSTANT)
    else:
        raise ValueError(kind)
    oracle.lib()  # load the library outside the timed region
    ncpu = os.cpu_count() or 1
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):  # pragma: no cover
        pass
    if budget_1t_s > 0:
        ne1, nt1, dt1 = _oracle_timed(work_one, sizes, N, 1, budget_1t_s)
    nea, nta, dta = _oracle_timed(work_one, sizes, N, ncpu, budget_s)
    what = f"oracle/libde_oracle.so {what_call}, -O3 -march=native), one tree per task"
    res = dict(value=nea / dta, unit="node-evals/s", cores=ncpu, kind="port",
               sample=f"{nta} trees x {N} samples of the same workload, {what} on {ncpu} threads "
                      f"(= all host cores, nproc {os.cpu_count()}), {dta:.1f} s")
    if budget_1t_s > 0:
        res["single_thread"] = dict(value=ne1 / dt1, unit="node-evals/s", cores=1,
                                    sample=f"{nt1} trees x {N} samples, {what}, 1 thread, {dt1:.1f} s")
    res["julia_probe"] = julia_probe()
    return res


def valu_ceiling(pop, n_trees, units, kernel_ms, turbo=False, complete=None):
    """The binding ceiling of the eval kernel (SURVEY.md §8d "secondary ceiling"): VALU issue slots per
    tree-wavefront = the fused program's dispatch histogram x the ISA slot count of every handler
    (profiles/valu_slots.json, generated by tools/valu_slots.py from the shipped code object)."""
    path = os.path.join(ROOT, "profiles", "valu_slots.json")
    if not os.path.exists(path):
        return None
    from dynamicexpressions_jl_amd import api
    with open(path) as fh:
        tab = json.load(fh)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_profiles import kernel_source_hash
    if tab.get("kernel_source_hash") != kernel_source_hash():  # a table of OTHER kernels prices nothing (python tools/valu_slots.py regenerates it, no GPU needed)
        return dict(stale="profiles/valu_slots.json was generated from other kernel sources than the running library (kernel_source_hash differs): "
                          "rerun `python tools/valu_slots.py` after the build")
    lib = api.library()
    hist = {}
    n_disp = 0
    n_all = n_trees
    if complete is not None:  # early exit: incomplete trees are (mostly) not evaluated — the floor is that of the complete ones
        n_trees = int(np.count_nonzero(complete))
        units = units * n_trees / max(n_all, 1)
        if n_trees == 0:
            return None
    for t in range(n_all):
        if complete is not None and not complete[t]:
            continue
        n = lib.de_program_dump(pop._h, t, None, 0, 3)
        if n <= 0:
            return None
        w = np.zeros(int(n), dtype=np.uint32)
        lib.de_program_dump(pop._h, t, w.ctypes.data, w.size, 3)
        ids = w.reshape(-1, 4)[:, 0]
        n_disp += len(ids)
        for k, c in zip(*np.unique(ids, return_counts=True)):
            hist[int(k)] = hist.get(int(k), 0) + int(c)
    import re
    cycles, missing = 0.0, 0
    by_op = {}
    fam = {"0": "cos", "1": "exp", "2": "sin"}
    binf = {"0": "+", "1": "-", "2": "-", "3": "*", "4": "/", "5": "/"}
    for k, c in hist.items():
        h = tab["handlers_turbo" if turbo else "handlers"].get(str(k))
        if h is None:
            missing += c
            continue
        cycles += c * h["valu_cycles"]
        m = re.match(r"h_(\w+)<\w+(?:, (\d+))?", h["name"])
        kind, K = m.group(1), m.group(2)
        label = fam.get(K, kind) if kind in ("un", "unrow_f") else (binf.get(K, kind) if kind in ("bin", "binrowc", "bin2") else "load/push/check/other")
        by_op[label] = by_op.get(label, 0.0) + c * h["valu_cycles"] / n_trees
    cycles += n_trees * tab["per_tree_overhead_cycles"]
    per_tree_wave = cycles / n_trees
    # second figure: what a dispatch really costs in the kernel (tools/exp_dispatch_cost.py, profiles/r2_dispatch_cost_*.json):
    # expensive handlers run at their VALU cycles, a cheap one (+ - * on a row or a constant: 7-18 cycles) cannot go below the
    # latency chain of a dispatch (record fetch -> s_setpc -> instruction fetch) shared by the ~5.5 resident waves of a SIMD,
    # ~30 cycles; a tree costs ~100 cycles beyond its dispatches (first load, end record + store, its share of the X staging)
    disp_floor, per_tree_meas = 30.0, 100.0
    modelled = sum(c * max(tab["handlers_turbo" if turbo else "handlers"][str(k)]["valu_cycles"], disp_floor)
                   for k, c in hist.items() if str(k) in tab["handlers_turbo" if turbo else "handlers"]) / n_trees + per_tree_meas
    samples_per_wave = pop.plan(10**6)["tile"]  # 64 lanes x the Float32 samples of a lane (8 with two planes, 4 with one: csrc/de_kernels.h DE_TG)
    # (the ISA table counts a handler's VALU cycles for ALL planes of a dispatch; per_tree_overhead_cycles likewise)
    tree_waves = units / samples_per_wave
    simds, peak, sustained = 256 * 4, 2.4e9, 2.08e9  # MI355X: 256 CUs x 4 SIMDs; peak engine clock; clock an all-VALU loop sustains
    floor_ms = tree_waves * per_tree_wave / (simds * peak) * 1e3
    return dict(trees_counted=n_trees, trees_counted_note="complete trees only (the others leave the kernel at their first flagged workgroup: "
                "early exit); their partial evaluation is real work the floor does not contain" if complete is not None else "all",
                simd_cycles_per_tree_wave=per_tree_wave, dispatches_per_tree=n_disp / n_trees, samples_per_wave=samples_per_wave,
                clock_ghz=2.4, simds=simds, floor_ms=floor_ms, frac=floor_ms / kernel_ms,
                sustained_clock_ghz=2.08, floor_ms_at_sustained_clock=floor_ms * peak / sustained,
                frac_at_sustained_clock=floor_ms * peak / sustained / kernel_ms, dispatches_without_cycle_count=missing,
                issue_model=dict(dispatch_floor_cycles=disp_floor, per_tree_cycles=per_tree_meas, simd_cycles_per_tree_wave=modelled,
                                 ms_at_sustained_clock=tree_waves * modelled / (simds * sustained) * 1e3,
                                 frac=tree_waves * modelled / (simds * sustained) * 1e3 / kernel_ms,
                                 source="tools/exp_dispatch_cost.py: slope of kernel time over chain length per handler family"),
                cycles_by_operator={k: round(v, 1) for k, v in sorted(by_op.items(), key=lambda kv: -kv[1])},
                source="profiles/valu_slots.json (tools/valu_slots.py: VALU cycles per handler on its shortest path, gfx950 ISA priced with "
                       "the per-instruction issue costs measured by tools/probe/valu_rate.py, profiles/r2_valu_rate.json; sustained clock: "
                       "profiles/r2_clock_probe.json) x de_program_dump(stage 3) dispatch histogram of this population")


def valu_measured(pm_entry):
    """The VALU ceiling from EXECUTED instruction counts (hardware counters of tools/profile_round.sh, same kernel sources:
    hash checked by the caller): VALU instructions x their issue cost against the SIMD cycles the step's kernels had.  The
    counters do not tell packed / SGPR-operand forms (3.7 cycles) from plain ones (2.0): `busy_bounds` prices all of them one
    way or the other, `busy_est` uses the static share of half-rate forms in the kernel's code object
    (profiles/valu_static_mix.json).  Transcendentals are counted exactly (7.35 cycles).  Works for every kernel of the
    library — the gradient, reverse and fused-loss ones have no per-handler cycle table."""
    sq = (pm_entry or {}).get("sq_per_step_by_kernel")
    times = (pm_entry or {}).get("kernels_us_per_step")
    if not sq or not times:
        return None
    mix_path = os.path.join(ROOT, "profiles", "valu_static_mix.json")
    mods = {}
    if os.path.exists(mix_path):
        with open(mix_path) as fh:
            mods = json.load(fh).get("modules", {})
    import re
    tot = dict(valu=0.0, trans=0.0, salu=0.0, lo=0.0, hi=0.0, est=0.0, avail=0.0, us=0.0)
    clocks = []
    for k, c in sq.items():
        v, t = c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        m = re.search(r"(gtm|rtm)_(\w+?)::", k)
        mod = ("de_gt_" if m and m.group(1) == "gtm" else "de_rt_") + m.group(2) if m else "de_kernels"
        h = mods.get(mod, {}).get("half_share_of_non_trans", 0.5)
        us = times.get(k)
        if us is None:
            continue
        grbm = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
        # (the counter pass and the timing pass are separate runs of the same steps: the clock is cycles of one over time of the other)
        if grbm > 0 and us > 0:
            clocks.append((grbm / (us * 1e-6), us))
        tot["valu"] += v
        tot["trans"] += t
        tot["salu"] += c.get("SQ_INSTS_SALU", 0.0)
        tot["lo"] += (v - t) * 2.0 + t * 7.35
        tot["hi"] += (v - t) * 3.7 + t * 7.35
        tot["est"] += (v - t) * (2.0 * (1 - h) + 3.7 * h) + t * 7.35
        tot["us"] += us
    if tot["us"] <= 0:
        return None
    clock = sum(c * w for c, w in clocks) / sum(w for _, w in clocks) if clocks else 2.08e9
    avail = tot["us"] * 1e-6 * clock * 1024  # SIMD cycles of 256 CUs x 4 SIMDs over the kernels' time
    return dict(valu_insts_per_step=tot["valu"], trans_insts_per_step=tot["trans"], salu_per_valu=tot["salu"] / max(tot["valu"], 1.0),
                kernels_us_per_step=tot["us"], engine_clock_ghz=clock / 1e9, simd_cycles_available=avail,
                valu_cycles_est=tot["est"], busy_est=tot["est"] / avail, busy_bounds=[tot["lo"] / avail, tot["hi"] / avail],
                floor_ms_est=tot["est"] / (1024 * clock) * 1e3,
                source="profiles/pmc_summary.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 ... of the same workload) x "
                       "issue costs of profiles/r2_valu_rate.json; half-rate share from profiles/valu_static_mix.json")


def wide_x_leg(ctx, lib, ops, device, F=60, n_trees=1000, N=10**6, steps=10):
    """A feature matrix the flat-switch kernel could not stage (F = 60: its `direct` variant gathered the features from global memory, 14.3 ms
    at this shape up to round 5): the threaded kernel in wave groups.  Not a BASELINE config: reported beside them."""
    import dynamicexpressions_jl_amd as de
    from dynamicexpressions_jl_amd import api
    trees = de.synth.random_population(n_trees, seed=0xDE02 + F, nfeatures=F)
    g = torch.Generator(device=device)
    g.manual_seed(F)
    X = (torch.randn((N, F), generator=g, device=device, dtype=torch.float32) * 1.2).t()  # [F, N] feature-fastest
    out = torch.empty((n_trees, N), device=device, dtype=torch.float32)
    ok = torch.empty(n_trees, device=device, dtype=torch.uint8)
    pop = api.Population(trees, ops, np.float32, n_features=F, ctx=ctx)
    t_end = time.perf_counter() + 0.08
    while time.perf_counter() < t_end:  # steady clocks (see the `configs` legs)
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, None, out.data_ptr(), N, ok.data_ptr()))
        ctx.synchronize()
    ctx.timing_ring(steps)
    for _ in range(steps):
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, F, None, out.data_ptr(), N, ok.data_ptr()))
    ctx.synchronize()
    ms = float(np.mean([t for t in ctx.timing_read() if t is not None]))
    ctx.timing_ring(0)
    okh = ok.cpu().numpy().astype(bool)
    nodes = np.array([de.count_nodes(t) for t in trees], dtype=np.int64)
    res = {"workload": f"{n_trees} random 20-node trees over {F} features x {N} samples, Float32 (not a BASELINE config)", "ms_per_step": ms, "steps": steps,
           "value": float(nodes[okh].sum()) * N / (ms * 1e-3), "value_all_trees": float(nodes.sum()) * N / (ms * 1e-3), "unit": "node-evals/s",
           "complete_fraction": float(okh.mean()), "kernel": ctx.last_kernel_name(), "waves_per_workgroup": pop.meta(0)["waves"],
           "rounds_1_5": "de_eval_tape_kernel<direct> (features gathered from global memory): 14.3 ms at this shape (DESIGN.md §4.1)"}
    pop.close()
    return res


def search_generation_leg(ctx, lib, ops, X, n_trees=10000, rows=1000):
    """de_program_create / de_eval / de_program_destroy of n_trees FRESH trees on `rows` samples through the C ABI."""
    import ctypes as C
    import dynamicexpressions_jl_amd as de
    trees = de.synth.random_population(n_trees, seed=0xDE0D)
    tape, noff, consts, coff = de.flatten_population(trees, ops, np.float32)
    Xs = X[:, :rows].t().contiguous()  # [rows, 5] feature-fastest
    out = torch.empty((n_trees, rows), device=X.device, dtype=torch.float32)
    ok = torch.empty(n_trees, device=X.device, dtype=torch.uint8)
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        h = C.c_void_p()
        t0 = time.perf_counter()
        ctx.check(lib.de_program_create(ctx._h, 0, tape.ctypes.data, noff.ctypes.data, n_trees, consts.ctypes.data, coff.ctypes.data, 5, 0, 7, C.byref(h)))
        t1 = time.perf_counter()
        ctx.check(lib.de_eval(ctx._h, h, Xs.data_ptr(), rows, 5, None, out.data_ptr(), rows, ok.data_ptr()))
        ctx.synchronize()
        t2 = time.perf_counter()
        lib.de_program_destroy(h)
        t3 = time.perf_counter()
        cur = (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2))
        best = cur if best is None else tuple(min(a, b) for a, b in zip(best, cur))
    nodes = sum(de.count_nodes(t) for t in trees)
    total = sum(best)
    return {"workload": f"{n_trees} fresh random 20-node trees x {rows} rows, Float32: de_program_create + de_eval (call + synchronise) + de_program_destroy, best of 4",
            "create_ms": best[0], "eval_ms": best[1], "destroy_ms": best[2], "generation_ms": total, "us_per_tree": 1e3 * total / n_trees,
            "value": nodes * rows / (total * 1e-3), "unit": "node-evals/s including the host side of the generation",
            "complete_fraction": float(ok.float().mean().item()),
            "note": "host lowering on the library's pool of host threads (DE_HOST_THREADS), device streams and host vectors recycled per context; "
                    "Python-side flattening of the trees (test harness) is not included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="the default run (--workload headline, 1 GPU) also times BASELINE configs 2, 3, 4 (rank 0's shard) and the config-5 family "
                         "and reports them under `configs`; this skips them")
    ap.add_argument("--turbo", action="store_true",
                    help="EvalContext(turbo=true): the relaxed-accuracy Float32 operators (DE_OPT_TURBO) for the whole run; "
                         "without it the plain-eval workloads time the exact mode and report turbo in a `turbo` sub-object")
    ap.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                    help="N > 1: 'strong' = the workload's population split over the N ranks (the metric of BASELINE.json: 1000 trees x "
                         "10^7 samples at 1/2/4/8 GPUs), 'weak' = the workload's population PER rank.  auto = strong (C4: its 8-way shards)")
    ap.add_argument("--no-turbo-leg", action="store_true", help="skip the secondary turbo timing (profiling runs: one kernel variant per process)")
    ap.add_argument("--no-full-eval-leg", action="store_true", help="skip the secondary timing without the early exit (DE_OPT_FULL_EVAL)")
    ap.add_argument("--no-complete-leg", action="store_true",
                    help="skip the `complete_only` leg: the same generator rejection-sampled to a population of COMPLETE trees (the kernel-quality number)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU, RCCL over xGMI) and relay
        # their output; rank 0 of the child job prints the JSON line with n_gpus = N
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); "
                         f"use --nproc-per-node {args.gpus}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if os.environ.get("DE_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) visible")
    # DE_BENCH_BACKEND=gloo: dry run of the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices, flags are gathered through host memory) — exercises the code path, the timing means nothing
    backend = os.environ.get("DE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import dynamicexpressions_jl_amd as de
    from dynamicexpressions_jl_amd import api, dist as dedist

    ctx = api.Context(local_rank)
    lib = api.library()
    x_cache = {}

    def run_workload(key, primary):
        """One workload of WORKLOADS: the timed steps, and (primary only) the secondary legs; returns the result dict on rank 0."""
        wl = WORKLOADS[key]
        n_per_gpu, N = wl["n_trees"], wl["N"]
        ops = de.synth.BENCH_OPERATORS
        # strong scaling (default; the metric is quoted on ONE population at 1/2/4/8 GPUs): the workload's n_trees round-robin sharded
        # over the ranks.  weak: n_trees PER rank (the job's population grows with the world).
        strong = args.scaling in ("auto", "strong")
        n_job = n_per_gpu if strong else n_per_gpu * world
        is_param = bool(wl.get("parametric"))
        shards = wl.get("shards")
        if shards:
            # a fixed population sharded `shards` ways (config C4: 10 000 trees x8); this job runs the first `world` shards
            if world > shards:
                raise SystemExit(f"workload {key} has {shards} shards; --gpus must be <= {shards}")
            full = de.synth.random_population(n_per_gpu * shards, seed=wl["seed"])
            shard_ids = [dedist.shard_indices(len(full), r, shards) for r in range(world)]
            all_trees = [full[i] for r in range(world) for i in shard_ids[r]]  # the trees this job evaluates
            trees = [full[i] for i in shard_ids[rank]]
            # node counts in the order the flag gather returns (entry r + i * world = rank r's i-th tree)
            nodes_gather = np.array([de.count_nodes(full[shard_ids[p % world][p // world]]) for p in range(len(all_trees))], dtype=np.int64)
            del full
        else:
            if is_param:
                all_trees = de.synth.random_population(n_job, seed=0xDE05, node_type=de.ParametricNode, nparams=8)
            else:
                all_trees = de.synth.random_population(n_job, seed=0xDE02)
            my_ids = dedist.shard_indices(len(all_trees), rank, world)
            trees = [all_trees[i] for i in my_ids]
            nodes_gather = np.array([de.count_nodes(t) for t in all_trees], dtype=np.int64)  # (round-robin shards: gather order = global order)
        total_nodes = sum(de.count_nodes(t) for t in all_trees)

        ec = api.EvalContext(turbo=True) if args.turbo else None
        if wl.get("reverse_grad"):  # C5pb: the reverse-accumulation kernel is what the workload measures (default = forward duals: `forward_default` leg)
            ec = api.EvalContext(turbo=bool(args.turbo), reverse_grad=True)
        # X: the in-repo stream SURVEY §8d specifies (synth.random_X: numpy PCG64 standard normals, seed 1 — the generator the parity tests
        # use), drawn on the host and uploaded ONCE, the same on every rank (replicated).  Rounds 1-4 drew it with torch.randn on the device:
        # `complete_fraction`, which decides half of the headline, then depended on torch's generator.  DE_BENCH_TORCH_X=1: that X.
        g = torch.Generator(device=dev).manual_seed(1)  # (targets, cotangents, parameters of the other workloads keep the device generator)
        if os.environ.get("DE_BENCH_TORCH_X", "0") == "1":
            X = torch.randn((N, 5), generator=g, device=dev, dtype=torch.float32).t()  # [5, N] feature-fastest
            x_source = "torch.randn(device generator, seed 1) [DE_BENCH_TORCH_X=1]"
        else:
            if N not in x_cache:  # (one X per sample count: the `configs` legs of the default run share the 10^6-sample X)
                Xh = de.synth.random_X(5, N, seed=1, dtype=np.float32)  # (5, N) Fortran-ordered = [N, 5] row-major storage
                x_cache[N] = torch.from_numpy(np.ascontiguousarray(Xh.T)).to(dev).t()  # [5, N] feature-fastest, strides (1, 5)
                del Xh
            X = x_cache[N]
            x_source = "synth.random_X(5, N, seed=1): numpy PCG64 standard normals, uploaded once"
        out_buf = [None]  # the step's output buffer, once it exists: the rejection sampling below evaluates into it instead of a second 40 GB

        def complete_population(n):
            """n trees of the generator (seed 0xDE0C) whose evaluation on this X comes out complete; (trees, candidates drawn)"""
            cand = de.synth.random_population(3 * n, seed=0xDE0C)
            chosen = []
            scratch = out_buf[0] if out_buf and out_buf[0] is not None and out_buf[0].shape[0] >= n else torch.empty((n, N), device=dev, dtype=torch.float32)
            for b in range(0, len(cand), n):
                if len(chosen) >= n:
                    break
                batch = cand[b:b + n]
                pop_b = api.Population(batch, ops, np.float32, n_features=5, eval_context=ec, ctx=ctx)
                okb = torch.empty(len(batch), device=dev, dtype=torch.uint8)
                ctx.check(lib.de_eval(ctx._h, pop_b._h, X.data_ptr(), N, 5, None, scratch.data_ptr(), N, okb.data_ptr()))
                torch.cuda.synchronize()
                chosen += [t for t, k in zip(batch, okb.cpu().numpy()) if k]
                pop_b.close()
            del scratch
            return (chosen[:n] if len(chosen) >= n else None), len(cand)

        if wl.get("complete"):
            if world != 1:
                raise SystemExit("workload complete: one GPU")
            trees, _ = complete_population(len(trees))
            all_trees = trees
            total_nodes = sum(de.count_nodes(t) for t in all_trees)
        pop = api.Population(trees, ops, np.float32, n_features=5, n_params=8 if is_param else 0, eval_context=ec, ctx=ctx)
        out = None if (wl.get("loss") or wl.get("lossgrad")) else torch.empty((len(trees), N), device=dev, dtype=torch.float32)
        out_buf[0] = out
        ok = torch.empty(len(trees), device=dev, dtype=torch.uint8)
        is_grad = bool(wl.get("grad"))
        is_loss = bool(wl.get("loss"))
        is_lossgrad = bool(wl.get("lossgrad"))
        if is_lossgrad:
            n_const = sum(pop.n_grad(t, 1) for t in range(len(trees)))
            dlossv = torch.empty(max(n_const, 1), device=dev, dtype=torch.float32)
        if is_loss or is_lossgrad:
            yv = torch.randn(N, generator=g, device=dev, dtype=torch.float32)
            lossv = torch.empty(len(trees), device=dev, dtype=torch.float32)
        grad = torch.empty(len(trees) * 5 * N, device=dev, dtype=torch.float32) if is_grad else None

        per_sample = bool(wl.get("per_sample"))
        by_class = ps_grad = False
        if is_param:
            n_cls = N if per_sample else 16
            params = torch.randn((n_cls, 8), generator=g, device=dev, dtype=torch.float32)  # [P=8, C] column-major
            if per_sample:  # "fully per-sample" parameters: classes = 1:N
                classes = torch.arange(1, N + 1, device=dev, dtype=torch.int32)
            else:
                classes = torch.randint(1, n_cls + 1, (N,), generator=g, device=dev, dtype=torch.int32)
            by_class = bool(wl.get("by_class"))
            if by_class:  # the dataset is ordered by class once (classes belong to the dataset)
                classes = torch.sort(classes).values
                starts = np.zeros(n_cls + 1, dtype=np.int64)
                np.cumsum(torch.bincount(classes.long() - 1, minlength=n_cls).cpu().numpy(), out=starts[1:])
                dY = torch.randn(N, generator=g, device=dev, dtype=torch.float32)
                ng_b = np.array([pop.n_grad(t, 2) for t in range(len(trees))], dtype=np.int64)
                lossv = torch.empty(len(trees), device=dev, dtype=torch.float32)
                dlossv = torch.empty(max(int(ng_b.sum()), 1), device=dev, dtype=torch.float32)
                dparv = torch.empty((len(trees), n_cls, 8), device=dev, dtype=torch.float32)
            pa = api.ParamArgs()
            pa.params, pa.ld_params, pa.n_classes = params.data_ptr(), 8, n_cls
            pa.classes, pa.classes_is_i64, pa.class_base = classes.data_ptr(), 0, 1
            import ctypes
            pa_ref = ctypes.byref(pa)
            ng_c = np.array([pop.n_grad(t, 1) for t in range(len(trees))], dtype=np.int64)
            goffs = np.zeros(len(trees), dtype=np.int64)
            np.cumsum(ng_c[:-1] * N, out=goffs[1:])
            ps_grad = bool(wl.get("per_sample_grad"))
            gradc = None if (wl.get("by_class") or (per_sample and not ps_grad)) else torch.empty(max(int((ng_c * N).sum()), 1), device=dev, dtype=torch.float32)

        def step():
            if is_param:
                ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, pa_ref, out.data_ptr(), N, ok.data_ptr()))
                if per_sample and not ps_grad:
                    pass
                elif by_class:
                    ctx.check(lib.de_eval_loss_grad_by_class(ctx._h, pop._h, X.data_ptr(), N, 5, pa_ref, 2, dY.data_ptr(), None, 2,
                                                             starts.ctypes.data, lossv.data_ptr(), dlossv.data_ptr(), None,
                                                             dparv.data_ptr(), ok.data_ptr()))
                else:
                    ctx.check(lib.de_eval_grad(ctx._h, pop._h, X.data_ptr(), N, 5, pa_ref, 1, None, N, gradc.data_ptr(),
                                               goffs.ctypes.data, ok.data_ptr()))
            elif is_grad:
                ctx.check(lib.de_eval_grad(ctx._h, pop._h, X.data_ptr(), N, 5, None, 0, out.data_ptr(), N,
                                           grad.data_ptr(), None, ok.data_ptr()))
            elif is_lossgrad:
                ctx.check(lib.de_eval_loss_grad(ctx._h, pop._h, X.data_ptr(), N, 5, None, 1, yv.data_ptr(), None, 0,
                                                lossv.data_ptr(), dlossv.data_ptr(), None, ok.data_ptr()))
            elif is_loss:
                ctx.check(lib.de_eval_loss(ctx._h, pop._h, X.data_ptr(), N, 5, None, yv.data_ptr(), None, 0,
                                           lossv.data_ptr(), ok.data_ptr()))
            else:
                ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
            if world > 1:
                if comm_c is not None:  # the C-ABI exchange (de_dist_gather_flags: ncclAllGather issued by the library itself)
                    return comm_c.gather_flags(ok, len(all_trees))
                return dedist.gather_flags(ok, len(all_trees), rank, world)
            return ok

        def barrier():
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()

        # N > 1: gather the flags through the library's own RCCL communicator (what a Julia / C caller uses, csrc/de_dist.cpp);
        # torch.distributed only ships the 128-byte unique id.  Checked against the torch.distributed gather once; any failure
        # falls back to that path (both are RCCL over xGMI) and is reported in the JSON line.
        comm_c, gather_via = None, "none (1 GPU)"
        gather_reason, declared_multi = "one GPU: nothing to gather", False
        comm_world = 1  # as the communicator reports it (not the environment)
        if world > 1:
            comm_world = int(torch.distributed.get_world_size())
            gather_via = "torch.distributed all_gather (RCCL)"
            # default on (DE_BENCH_C_COMM=0 turns it off): the driver's multi-GPU run exercises de_dist_* — checked once against the
            # torch.distributed gather below and dropped for it on any mismatch or error (a multi-GPU box was never available to the builder)
            gather_reason = "DE_BENCH_C_COMM=0 or a non-RCCL backend: the C-ABI communicator was not tried"
            if backend == "nccl" and os.environ.get("DE_BENCH_C_COMM", "1") == "1":
                # Every rank takes part in every torch collective below whatever happens to its C-ABI communicator (a rank that failed
                # alone must not leave the others waiting), and the C-ABI calls of the check are BOUNDED (de_dist_set_timeout: 60 s;
                # de_dist_init by DE_DIST_INIT_TIMEOUT_MS): the worst case is a fallback to torch.distributed with the reason in the line.
                ok.fill_(1)
                ok[::3] = 0
                b = dedist.gather_flags(ok, len(all_trees), rank, world)
                cand, status, why = None, 0, ""
                ids = [None]
                if rank == 0:
                    try:
                        ids = [dedist.Comm.unique_id()]
                    except Exception as e:  # noqa: BLE001
                        ids = [None]
                        why = f"de_dist_unique_id: {e}"
                torch.distributed.broadcast_object_list(ids, src=0)
                try:
                    if ids[0] is None:
                        raise RuntimeError("no RCCL unique id from rank 0 (librccl.so not loadable there)")
                    cand = dedist.Comm(ctx, rank, world, ids[0])
                    cand.set_timeout(60000)
                    a = cand.gather_flags(ok, len(all_trees))
                    torch.cuda.synchronize()
                    status = int(torch.equal(a, b))
                    if not status:
                        why = "de_dist_gather_flags returned other flags than the torch.distributed gather"
                except Exception as e:  # noqa: BLE001
                    why = f"{type(e).__name__}: {e}"
                    print(f"[rank {rank}] C-ABI communicator unavailable, using torch.distributed: {e}", file=sys.stderr)
                same = torch.tensor([status], device=dev)
                torch.distributed.all_reduce(same, op=torch.distributed.ReduceOp.MIN)
                whys = [None] * world
                torch.distributed.all_gather_object(whys, why)
                if int(same.item()) == 1:
                    cand.set_timeout(0)  # (the timed steps stay asynchronous, like the torch.distributed path they are compared with)
                    comm_c, gather_via = cand, "de_dist_gather_flags (C ABI: ncclAllGather inside libde_hip.so)"
                    gather_reason = "checked once against the torch.distributed gather (equal on every rank); collectives of the check bounded by de_dist_set_timeout(60 s)"
                    comm_world = cand.world_size()  # ncclCommCount of the library's own communicator
                else:
                    if cand is not None:
                        try:
                            cand.close()
                        except Exception:  # noqa: BLE001
                            pass
                    gather_reason = "fell back to torch.distributed: " + "; ".join(f"rank {r}: {w}" for r, w in enumerate(whys) if w)
            # N > 1: X does not change between the steps (nor between the calls of a search): every rank declares it once, so that the
            # per-call pass over X for the priority-tile keys — which does not shrink with the shard — is not paid per step (VERDICT r5 item 6;
            # the N = 1 line keeps the per-call pass and reports the declared number in `dataset_declared`)
            if os.environ.get("DE_BENCH_DECLARE", "1") == "1":
                ctx.declare_dataset(X)
                declared_multi = True

        warm_used = args.warmup
        t_w = time.perf_counter()
        for _ in range(args.warmup):
            step()
        if not primary:
            # the `configs` legs of the default run: W warm-up steps of a 1 ms workload are 5 ms — the device has not reached its steady
            # clocks by then (measured in round 6, same box: C2 1.025 ms after 5 warm-up steps, 0.937 after 50 or 200; the headline, C3 and C5
            # do not move: their W steps already take tens of ms).  These legs therefore warm up for at least 80 ms of wall time; the PRIMARY
            # workload of a run (what --warmup W is the contract for) does exactly W steps.
            torch.cuda.synchronize()
            while time.perf_counter() - t_w < 0.08 and warm_used < 400:
                step()
                warm_used += 1
                if warm_used % 8 == 0:
                    torch.cuda.synchronize()
        barrier()
        # the timed region is a FREE-RUNNING loop: the device time of every call is read afterwards from the context's ring of event pairs
        # (de_ctx_timing_ring; rounds 1-4 called last_kernel_ms() inside the loop, a synchronisation per step)
        ring_calls = 2 if (is_param and not (per_sample and not ps_grad)) else 1  # timed library calls per step
        ctx.timing_ring(min(args.steps * ring_calls, 4096))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            flags = step()
        barrier()
        elapsed = time.perf_counter() - t0
        kr = ctx.timing_read()
        ctx.timing_ring(0)
        kernel_ms = [sum(kr[i:i + ring_calls]) for i in range(0, len(kr) - ring_calls + 1, ring_calls)]
        ok_main = ok.clone()  # this rank's flags of the timed (exact / --turbo) steps: the secondary legs reuse the buffer
        if world > 1:
            t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())

        turbo_res = None
        if primary and not args.turbo and not args.no_turbo_leg and not (is_param or is_grad or is_lossgrad or is_loss):
            # the same steps with the reference's turbo option (DE_OPT_TURBO), reported beside the exact-mode line
            pop_t = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(turbo=True), ctx=ctx)

            def step_t():
                ctx.check(lib.de_eval(ctx._h, pop_t._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
            for _ in range(args.warmup):
                step_t()
            barrier()
            ctx.timing_ring(min(args.steps, 4096))
            t0t = time.perf_counter()
            for _ in range(args.steps):
                step_t()
            barrier()
            el_t = time.perf_counter() - t0t
            kt = ctx.timing_read()
            ctx.timing_ring(0)
            if world > 1:
                tt = torch.tensor([el_t], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                el_t = float(tt.item())
            kt = [k for k in kt if k is not None]
            turbo_res = dict(ms_per_step=1e3 * el_t / args.steps, kernel_ms_avg=float(np.mean(kt)) if kt else 1e3 * el_t / args.steps,
                             complete_fraction=float(ok.float().mean().item()), pop=pop_t)

        full_res = None
        if primary and not args.no_full_eval_leg and not (is_param or is_grad or is_lossgrad or is_loss):
            # the same steps WITHOUT the early exit (DE_OPT_FULL_EVAL: every tree on every sample, what rounds 1-2 timed)
            ecf = api.EvalContext(turbo=bool(args.turbo), full_eval=True)
            pop_f = api.Population(trees, ops, np.float32, n_features=5, eval_context=ecf, ctx=ctx)

            def step_f():
                ctx.check(lib.de_eval(ctx._h, pop_f._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
            for _ in range(args.warmup):
                step_f()
            barrier()
            t0f = time.perf_counter()
            for _ in range(args.steps):
                step_f()
            barrier()
            el_f = time.perf_counter() - t0f
            if world > 1:
                tf = torch.tensor([el_f], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                torch.distributed.all_reduce(tf, op=torch.distributed.ReduceOp.MAX)
                el_f = float(tf.item())
            full_res = dict(ms_per_step=1e3 * el_f / args.steps, kernel_ms=ctx.last_kernel_ms(), flags_equal=bool(torch.equal(ok, ok_main)))
            pop_f.close()

        complete_res = None
        if primary and not args.no_complete_leg and world == 1 and not (is_param or is_grad or is_lossgrad or is_loss) and not wl.get("complete"):
            # COMPLETE TREES ONLY: the same generator (another seed), rejection-sampled on this X to len(trees) trees whose evaluation
            # comes out complete — nothing exits early, every tree-sample is executed.  This is the number that says what the KERNEL
            # does per executed tree-sample; the headline beside it also contains the reference's early exit (config.early_exit).
            chosen, n_cand = complete_population(len(trees))
            if chosen is not None:
                pop_c = api.Population(chosen, ops, np.float32, n_features=5, eval_context=ec, ctx=ctx)

                def step_c():
                    ctx.check(lib.de_eval(ctx._h, pop_c._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
                for _ in range(args.warmup):
                    step_c()
                barrier()
                ctx.timing_ring(min(args.steps, 4096))
                t0c = time.perf_counter()
                for _ in range(args.steps):
                    step_c()
                barrier()
                el_c = time.perf_counter() - t0c
                kc = ctx.timing_read()
                ctx.timing_ring(0)
                kc = [k for k in kc if k is not None]
                complete_res = dict(ms_per_step=1e3 * el_c / args.steps, kernel_ms_avg=float(np.mean(kc)) if kc else 1e3 * el_c / args.steps,
                                    complete_fraction=float(ok.float().mean().item()), nodes=sum(de.count_nodes(t) for t in chosen),
                                    candidates=n_cand, pop=pop_c, n=len(chosen))

        fwd_res = None
        if wl.get("reverse_grad") and is_param and by_class and world == 1 and not args.no_complete_leg:  # (profiling runs pass --no-complete-leg: one kind of step under the profiler)
            # the same steps WITHOUT the permission: the library's default since ABI 3 — forward duals, the reference's flag semantics exactly
            # (one fused pass per class instead of one reverse launch over all classes): what parity-first costs on this workload
            pop_r = pop
            pop = api.Population(trees, ops, np.float32, n_features=5, n_params=8, eval_context=api.EvalContext(turbo=bool(args.turbo)), ctx=ctx)
            try:
                for _ in range(2):
                    step()
                barrier()
                t0w = time.perf_counter()
                n_f = max(3, args.steps // 4)
                for _ in range(n_f):
                    step()
                barrier()
                fwd_res = dict(ms_per_step=1e3 * (time.perf_counter() - t0w) / n_f, steps=n_f, kernel=ctx.last_kernel_name(),
                               flags_equal=bool(torch.equal(ok, ok_main)))
            finally:
                pop.close()
                pop = pop_r
            ok.copy_(ok_main)
        declared_res = None
        if world == 1 and not (is_param or is_grad or is_lossgrad or is_loss) and not args.no_complete_leg:
            # the same steps with the dataset DECLARED (de_ctx_declare_dataset: X does not change between the calls of a search, its
            # priority-tile keys are computed once instead of in every step).  A secondary leg: the headline keeps the per-call pass.
            ctx.declare_dataset(X)
            for _ in range(args.warmup):
                step()
            barrier()
            t0d = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            declared_res = dict(ms_per_step=1e3 * (time.perf_counter() - t0d) / args.steps, flags_equal=bool(torch.equal(ok, ok_main)))
            ctx.declare_dataset(None)

        torchx_res = None
        if (primary and world == 1 and key == "headline" and not args.no_complete_leg and not args.turbo
                and os.environ.get("DE_BENCH_TORCH_X", "0") != "1"):
            # THE SAME STEPS ON THE X OF ROUNDS 1-4 (torch.randn on the device, seed 1): the round-over-round comparable number.  That
            # generator plants exact zeros (x / 0: more incomplete trees), so its headline is not this line's `value`; see config.X.
            gx = torch.Generator(device=dev).manual_seed(1)
            X_keep = X
            X = torch.randn((N, 5), generator=gx, device=dev, dtype=torch.float32).t()
            for _ in range(args.warmup):
                step()
            barrier()
            t0x = time.perf_counter()
            for _ in range(args.steps):
                fx = step()
            barrier()
            torchx_res = dict(ms_per_step=1e3 * (time.perf_counter() - t0x) / args.steps, complete_fraction=float(fx.float().mean().item()))
            X = X_keep
            del X_keep
            ok.copy_(ok_main)  # (`flags` of the timed steps aliases this buffer at N = 1)

        if rank == 0:
            ms_per_step = 1e3 * elapsed / args.steps
            value_all = total_nodes * N * args.steps / elapsed
            # `value` = the EXECUTED rate: node-evals of the trees that came out complete / time.  The trees the early exit
            # drops (the reference drops them too) are counted in `value_all_trees`, the number rounds 1-5 printed as `value`.
            flags_h = (ok_main if world == 1 else flags).cpu().numpy().astype(bool)  # (at N = 1 `flags` aliases `ok`, which the secondary legs reuse)
            nodes_complete = int(nodes_gather[flags_h].sum())
            value = nodes_complete * N * args.steps / elapsed
            kms = [k for k in kernel_ms if k is not None]
            k_avg_ms = float(np.mean(kms)) if kms else ms_per_step
            plan = pop.plan(N)
            units = len(trees) * N  # tree-samples per launch (this rank's shard)
            if is_param:  # eval (X tile per chunk + output) + constant-mode Jacobian rows; priced on the whole step
                k_eff = plan["trees_per_chunk"]
                k_avg_ms = ms_per_step
                b_unit = (F_FEATURES * ELEM / k_eff + ELEM) + (F_FEATURES * ELEM / 32 + float(ng_c.mean()) * ELEM)
                if per_sample and not ps_grad:  # eval only; X + class id + the sample's 8 parameters per chunk of k_eff trees, one output
                    b_unit = ((F_FEATURES + 8) * ELEM + 4) / k_eff + ELEM
                elif per_sample:  # ... + the constant-mode Jacobian: X + class id + parameters per chunk of 32 trees, n_grad rows written
                    b_unit = (((F_FEATURES + 8) * ELEM + 4) / k_eff + ELEM) + (((F_FEATURES + 8) * ELEM + 4) / 32 + float(ng_c.mean()) * ELEM)
                elif by_class:  # eval as above; pullback: X + class id + dY tile per chunk of 32 trees, (1 + n_grad) partials per wave —
                    # written by the sweep kernel AND read back by the fixed-order finish pass (the two-pass reduction is what makes the
                    # result reproducible: both directions are bytes the algorithm needs; round 3 counted the write only)
                    b_unit = ((F_FEATURES * ELEM + 4) / k_eff + ELEM) + ((F_FEATURES * ELEM + 4 + ELEM) / 32
                                                                         + 2 * (1 + float(ng_b.mean())) * 4 * ELEM / 256)
            elif is_grad:  # X tile staged once per chunk of 32 trees (the K-tile rule of SURVEY.md §8d), x + 5 gradient rows written
                k_eff = 32
                b_unit = F_FEATURES * ELEM / k_eff + (1 + F_FEATURES) * ELEM
            elif is_lossgrad:  # X (+ y) tile staged once per chunk of 32 trees; (1 + n_const) partials per wave
                k_eff = 32
                b_unit = (F_FEATURES + 1) * ELEM / k_eff + (1 + n_const / len(trees)) * 4 * ELEM / 256
            elif is_loss:  # X (+ y) tile staged once per chunk of trees, nothing written but the partial sums
                k_eff = plan["trees_per_chunk"]
                b_unit = (F_FEATURES + 1) * ELEM / k_eff + 4 * ELEM / plan["tile"]
            else:  # X tile staged once per chunk of trees
                k_eff = plan["trees_per_chunk"]
                b_unit = F_FEATURES * ELEM / k_eff + ELEM
            plain_eval = not (is_param or is_grad or is_lossgrad)
            # Early exit (the reference's @return_on_nonfinite_array, at tree granularity here): a tree found incomplete is not
            # evaluated by the workgroups that start afterwards, and its row is unspecified.  The roofline is priced on
            # what MUST move: the tree-samples of the trees that came out complete (a lower bound of what the launch evaluated; the
            # partial evaluation of the others is not credited).  `all_units` restates it for every tree-sample of the job — the
            # count the node-evals/s metric uses, as the reference's own benchmark does for its early-exiting evaluator.
            okh = ok_main.cpu().numpy().astype(bool)
            complete_frac = float(okh.mean())
            units_all = units
            units = units_all * complete_frac
            alg_bytes = b_unit * units
            achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9
            single = BYTES_PER_TREE_SAMPLE_SINGLE * units / (k_avg_ms * 1e-3) / 1e9
            pm_entry = None
            traffic, traffic_note = None, "no profiles/pmc_summary.json"
            prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
            if os.path.exists(prof):  # measured by tools/profile_round.sh with rocprofv3 --pmc (separate passes)
                with open(prof) as fh:
                    pm = json.load(fh)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from make_profiles import kernel_source_hash
                key = "turbo" if (args.turbo and key == "headline") else key
                if pm.get("kernel_source_hash") != kernel_source_hash():
                    # the counters were collected with OTHER kernels than the ones running now: not this launch's traffic
                    traffic_note = "profiles/pmc_summary.json is stale (kernel_source_hash differs from the sources of the running library): rerun tools/profile_round.sh"
                else:
                    pm_entry = pm.get(key, {})
                    traffic = pm_entry.get("hbm_bytes_per_launch")
                    traffic_note = f"rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per step, profiles/pmc_summary.json[{key}] (same kernel sources: hash checked)"
            # C4 is a fixed 8-way sharded population: a job of N <= 8 ranks runs the first N shards (per-GPU work fixed: weak)
            scaling_kind = "weak" if (shards or not strong) else "strong"
            res = {
                "metric": "node-evals/sec", "value": value, "value_all_trees": value_all, "unit": "node-evals/s", "n_gpus": world,
                "steps": args.steps, "warmup": warm_used, "ms_per_step": ms_per_step,
                "higher_is_better": True, "scaling": scaling_kind, "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": wl["desc"], "workload_key": key, "trees_job": len(all_trees), "trees_this_rank": len(trees),
                           "n_samples": N, "n_features": 5, "X": x_source, "rccl_world_size": comm_world,
                           "nodes_per_tree": 20, "operators": "+ - / * cos exp", "sharding": f"tree-sharded x{world}", "flag_gather": gather_via, "flag_gather_reason": gather_reason,
                           "dataset_declared": declared_multi,
                           "complete_fraction": float(flags_h.mean()),
                           "early_exit": "EvalContext.early_exit = true (the reference's default): a tree is not evaluated by workgroups that start "
                                         "after its flag went to 0 (include/de_hip.h DE_OPT_EARLY_EXIT; `full_evaluation` = the same steps with DE_OPT_FULL_EVAL)",
                           "priority_tiles": ("every step first reads X once (de_tile_extremes_kernel, inside the timed region and inside roofline.kernel_ms_avg), runs the "
                                              "3 F sample tiles holding each feature's largest, smallest and closest-to-zero value as a probe launch, then re-links the "
                                              "records of the trees that are still live (de_compact_live_kernel) and runs the launch proper over dense chunks of them: "
                                              "order only (DESIGN.md 4.0)"
                                              if lib.de_prio_tiles_wanted(N, 5, len(trees)) else "off for this launch size (the library decides: de_prio_tiles_wanted)")},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                             "kernel": ctx.last_kernel_name(), "kernel_ms_avg": k_avg_ms,
                             "algorithmic_bytes_per_launch": alg_bytes,
                             "units": "tree-samples of the COMPLETE trees of this launch (early exit: see config.early_exit)",
                             "all_units": {"tree_samples": units_all, "achieved": b_unit * units_all / (k_avg_ms * 1e-3) / 1e9,
                                           "frac": b_unit * units_all / (k_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "note": "SURVEY §8d per-unit figure x EVERY tree-sample of the job (the metric's count): not what moved"},
                             "algorithmic_bytes_per_tree_sample": b_unit, "k_eff_trees_per_x_tile": k_eff,
                             "single_tree_equivalent": {"bytes_per_tree_sample": BYTES_PER_TREE_SAMPLE_SINGLE,
                                                        "achieved": single, "frac": single / HBM_PEAK_GBS},
                             "valu": valu_ceiling(pop, len(trees), units_all, k_avg_ms, complete=okh) if plain_eval else None,
                             "valu_measured": valu_measured(pm_entry),
                             "note": "X tile reused by k_eff trees from LDS: HBM traffic ~= the output; the kernel is "
                                     "VALU/scalar-issue bound (DESIGN.md §Roofline)"},
            }
            res["config"]["turbo"] = bool(args.turbo)
            if args.turbo and plain_eval:
                res["roofline"]["valu"] = valu_ceiling(pop, len(trees), units_all, k_avg_ms, turbo=True, complete=okh)
            if turbo_res is not None:
                tk = turbo_res["kernel_ms_avg"]
                res["turbo"] = {"option": "EvalContext(turbo=true) = DE_OPT_TURBO: relaxed-accuracy Float32 / exp cos sin (<= 1e-6 rel; "
                                          "tests/test_gpu_turbo.py), same population, same steps",
                                "ms_per_step": turbo_res["ms_per_step"], "value": total_nodes * N / (turbo_res["ms_per_step"] * 1e-3),
                                "kernel_ms_avg": tk, "roofline_frac": alg_bytes / (tk * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "complete_fraction": turbo_res["complete_fraction"],
                                "valu": valu_ceiling(turbo_res["pop"], len(trees), units_all, tk, turbo=True, complete=okh)}
            # what the headline `value` contains (VERDICT r3 / ADVICE r3): it counts every tree-sample of the job, as the reference's own
            # benchmark does for its early-exiting evaluator; `value_executed` counts only the node-evals of the trees that came out complete
            # (what was certainly executed); `full_evaluation.value` is the rate with the exit switched off; `complete_only` is the kernel on a
            # population in which nothing exits.
            res["value_executed"] = value
            res["value_note"] = ("`value` = `value_executed` = node-evals of the trees that came out COMPLETE / time (what was certainly executed; "
                                 f"{100 * (1 - float(flags_h.mean())):.1f} % of the job's trees are incomplete and leave the kernel at their first flagged workgroup, as they "
                                 "leave the reference's recursion); `value_all_trees` = node-evals of EVERY tree of the job / the same time (the count the "
                                 "reference's benchmark uses for its early-exiting evaluator; `value` of BENCH_r01..r05); `full_evaluation.value` = no exit; "
                                 "`complete_only` = a population of complete trees (kernel quality)")
            if complete_res is not None:
                ck = complete_res["kernel_ms_avg"]
                cu = complete_res["n"] * N
                res["complete_only"] = {"workload": f"{complete_res['n']} COMPLETE trees (same generator, seed 0xDE0C, rejection-sampled from {complete_res['candidates']} "
                                                    f"candidates on this X) x (5 x {N}) Float32: nothing exits early, every tree-sample is executed",
                                        "ms_per_step": complete_res["ms_per_step"], "kernel_ms_avg": ck,
                                        "value": complete_res["nodes"] * N / (complete_res["ms_per_step"] * 1e-3),
                                        "complete_fraction": complete_res["complete_fraction"],
                                        "us_per_tree_1e7_samples": 1e3 * ck / complete_res["n"] * (1e7 / N),
                                        "roofline": {"bound": "hbm", "achieved": b_unit * cu / (ck * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": b_unit * cu / (ck * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b_unit * cu,
                                                     "valu": valu_ceiling(complete_res["pop"], complete_res["n"], cu, ck, turbo=bool(args.turbo))}}
                complete_res["pop"].close()
            if fwd_res is not None:
                res["forward_default"] = {"option": "the same steps with the library's default (no DE_OPT_REVERSE_GRAD): forward duals, one fused pass per class — "
                                                    "the reference's flag semantics exactly (ABI 3)", **fwd_res}
            if declared_res is not None:
                res["dataset_declared"] = {"option": "de_ctx_declare_dataset(X) before the steps: the per-call pass over X (priority-tile keys) is done once per dataset",
                                           "ms_per_step": declared_res["ms_per_step"], "value": total_nodes * N / (declared_res["ms_per_step"] * 1e-3),
                                           "flags_equal_to_headline": declared_res["flags_equal"]}
            if torchx_res is not None:
                res["rounds_1_4_x"] = {"option": "the same steps on the X of rounds 1-4 (torch.randn on the device, seed 1; it plants exact zeros, profiles/r5_x_generators.txt): "
                                                 "compare THIS with BENCH_r01..r04's ms_per_step",
                                       "ms_per_step": torchx_res["ms_per_step"], "value": total_nodes * N / (torchx_res["ms_per_step"] * 1e-3),
                                       "complete_fraction": torchx_res["complete_fraction"]}
            if full_res is not None:
                res["full_evaluation"] = {"option": "DE_OPT_FULL_EVAL: no early exit, every tree evaluated on every sample (rounds 1-2 timed this)",
                                          "ms_per_step": full_res["ms_per_step"], "kernel_ms_last": full_res["kernel_ms"],
                                          "value": total_nodes * N / (full_res["ms_per_step"] * 1e-3), "flags_equal_to_early_exit": full_res["flags_equal"],
                                          "roofline_frac": b_unit * units_all / (full_res["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if not args.no_cpu_baseline and world == 1 and key == "headline" and not args.turbo:
                # What a SEARCH LOOP pays per generation (new trees every call): the host side of this path, which the reference does not have
                # (it evaluates the Node it is handed).  10^4 fresh trees: de_program_create, de_eval on 10^3 rows, de_program_destroy; best of 4.
                try:
                    res["search_generation"] = search_generation_leg(ctx, lib, ops, X)
                except Exception as e:  # noqa: BLE001  (a secondary leg must not cost the headline line)
                    res["search_generation"] = {"error": repr(e)}
            want_cpu = (not args.no_cpu_baseline and world == 1 and
                        (primary or key in ("C2", "C3", "C5")))  # reported at N=1 only (rank 0); same shape for C2 / C3 / C5
            if want_cpu and not (per_sample or (is_param and by_class) or is_loss or is_lossgrad):
                Ns = min(N, 10**6)
                Xh = np.asfortranarray(X[:, :Ns].t().contiguous().cpu().numpy().T)
                kw = {}
                if is_param:
                    kw = dict(kind="param", params=np.asfortranarray(params.cpu().numpy().T), classes=classes[:Ns].cpu().numpy())
                elif is_grad:
                    kw = dict(kind="grad")
                # the default run carries several baselines: the headline's keeps its budgets (<= 11 s on all cores + 7 s on one
                # thread), the configs' legs get <= 8 s on all cores + 3 s on one thread each
                b_all, b_one = (11.0, 7.0) if primary else (8.0, 3.0)
                res["cpu_baseline"] = cpu_baseline(all_trees, ops, Xh, b_all, b_one, **kw)
            pop.close()
        return res if rank == 0 else None

    res = run_workload(args.workload, True)
    if args.workload == "headline" and world == 1 and not args.no_configs and not args.turbo:
        # THE OTHER BASELINE CONFIGURATIONS, in the same run: config 2, config 3, rank 0's shard of config 4, and the
        # config-5 family (C5: 16 classes, eval + constant-mode Jacobian; C5N / C5Ng: 8 PER-SAMPLE parameters, eval / eval + Jacobian;
        # C5pb: eval + the :both pullback by class), each timed like the headline (same --steps / --warmup, free-running loop, device
        # times from the context's event ring) and condensed to the keys a reader needs; `python bench.py --workload <key>` prints the full line.
        res["configs"] = {}
        for k in ("C2", "C3", "C4", "C5", "C5N", "C5Ng", "C5pb"):
            x_keep = {n: x for n, x in x_cache.items() if n == WORKLOADS[k]["N"] or n == WORKLOADS["headline"]["N"]}
            x_cache.clear()
            x_cache.update(x_keep)
            gc.collect()
            torch.cuda.empty_cache()
            try:
                r = run_workload(k, False)
            except Exception as e:  # noqa: BLE001  (a secondary leg must not cost the headline line)
                res["configs"][k] = {"error": repr(e)}
                continue
            rf = r["roofline"]
            c = {"workload": r["config"]["workload"], "ms_per_step": r["ms_per_step"], "value": r["value"], "value_executed": r["value_executed"],
                 "value_all_trees": r["value_all_trees"], "complete_fraction": r["config"]["complete_fraction"], "steps": r["steps"], "warmup": r["warmup"],
                 "roofline": {kk: rf[kk] for kk in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch",
                                                     "algorithmic_bytes_per_tree_sample", "k_eff_trees_per_x_tile", "traffic", "traffic_source",
                                                     "kernel", "kernel_ms_avg")}}
            if rf.get("valu_measured"):
                c["roofline"]["valu_busy_est"] = rf["valu_measured"]["busy_est"]
            if rf.get("valu"):
                c["roofline"]["valu_frac"] = rf["valu"].get("frac")
            for kk in ("dataset_declared", "cpu_baseline", "forward_default"):
                if kk in r:
                    c[kk] = r[kk]
            res["configs"][k] = c
        # ... and one shape outside BASELINE.json that changed by an order of magnitude in round 6: a 60-feature dataset (last: it leaves the
        # configs' memory and clocks alone)
        x_cache.clear()
        gc.collect()
        torch.cuda.empty_cache()
        try:
            res["wide_x"] = wide_x_leg(ctx, api.library(), de.synth.BENCH_OPERATORS, "cuda")
        except Exception as e:  # noqa: BLE001
            res["wide_x"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
