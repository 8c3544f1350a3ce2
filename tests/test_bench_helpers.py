"""bench.py's CPU-side helpers (no GPU): the `cpu_baseline` legs of the three reference calls the default run times at the configs' shapes —
eval_tree_array (C2), eval_grad_tree_array(variable=true) (C3), the ParametricExpression eval + constant-mode gradient (C5) — on tiny inputs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dynamicexpressions_jl_amd as de  # noqa: E402


def test_cpu_baseline_kinds_time_the_reference_calls_of_the_configs():
    ops = de.synth.BENCH_OPERATORS
    X = de.synth.random_X(5, 4000, seed=1)
    trees = de.synth.random_population(24, seed=0xDE02)
    for kind in ("eval", "grad"):
        r = bench.cpu_baseline(trees, ops, X, 0.5, 0.25, kind=kind)
        assert r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1 and r["single_thread"]["cores"] == 1
        assert ("eval_grad_tree_array" in r["sample"]) == (kind == "grad")
    pt = de.synth.random_population(24, seed=0xDE05, node_type=de.ParametricNode, nparams=8)
    g = np.random.default_rng(0)
    params = np.asfortranarray(g.standard_normal((8, 16)).astype(np.float32))
    classes = g.integers(1, 17, 4000).astype(np.int32)
    r = bench.cpu_baseline(pt, ops, X, 0.5, 0.0, kind="param", params=params, classes=classes)
    assert r["value"] > 0 and "single_thread" not in r and "ParametricExpression" in r["sample"]


def test_default_run_lists_every_baseline_config():
    """the `configs` legs of the default run = BASELINE.json's configs 2-5 (config 4 as rank 0's shard, config 5 in its three readings)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'for k in ("C2", "C3", "C4", "C5", "C5N", "C5Ng", "C5pb")' in src
    for k in ("C2", "C3", "C4", "C5", "C5N", "C5Ng", "C5pb"):
        assert k in bench.WORKLOADS
    assert bench.WORKLOADS["C4"]["shards"] == 8 and bench.WORKLOADS["C4"]["n_trees"] * 8 == 10000
    assert bench.WORKLOADS["C5pb"].get("reverse_grad") is True  # (reverse accumulation is an opt-in since ABI 3: the workload that measures it says so)
