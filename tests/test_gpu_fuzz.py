"""The differential fuzzers as a regression GATE (`pytest -m gpu`).

Fixed seeds; every difference between the HIP path and the oracle is classified with the suite's tolerance models
(tests/fuzzlib.py): ill-conditioned samples / entries / flag freedoms are COUNTED and capped, a finding outside the model
fails the test — after the whole seed has run, with every finding listed.  Flavours = the command-line fuzzers of tests/fuzz/
(hot operator set with ragged N, wide operator set in all option modes, fall-back kernels, Jacobians in three modes,
ParametricExpression, fused loss gradient reverse-vs-forward), plus the mixed-arity random-tree property test of the reference
(test/test_supposition_consistency.jl:41-104: (abs, cos, exp), (+, -, *, /), (fma, clamp, +, max), <= 20 layers)."""
import os

import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
import fuzzlib as FZ

pytestmark = pytest.mark.gpu

ILL_VALUE_CAP = {"hot": 0.05, "wide": 0.20}   # share of compared samples / entries the model may class as ill-conditioned
ILL_FLAG_CAP = 0.03                           # share of trees whose FLAG differs through an ill-conditioned sample


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


def gate(f, kind, what):
    print(f"[fuzz gate {what}] {f.summary()}")
    assert not f.real, f"{len(f.real)} finding(s) outside the tolerance model [{what}]:\n  " + "\n  ".join(f.real[:12])
    assert f.ill_values <= ILL_VALUE_CAP[kind] * max(f.compared, 1), f"ill-conditioned share above the cap [{what}]: {f.summary()}"
    assert f.ill_flags <= max(2, ILL_FLAG_CAP * f.flag_checks), f"flag differences on ill-conditioned trees above the cap [{what}]: {f.summary()}"


def contexts(api):
    return (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False), api.EvalContext(bumper=True))


@pytest.mark.parametrize("seed", [201, 202, 203])
def test_fuzz_hot_operators_ragged_sizes(api, seed):
    tot = FZ.Findings()
    for rep in range(3):
        rng = de.synth.Xoshiro256ss(seed * 31 + rep)
        for dtype in (np.float32, np.float64):
            F = 1 + (seed + rep) % 7
            trees = FZ.random_trees(rng, FZ.OPS_HOT, F, dtype, 120, 40, rep)
            g = np.random.Generator(np.random.PCG64(seed * 7 + rep))
            N = int(g.choice([1, 2, 63, 64, 65, 511, 512, 513, 1023, 1025, 2047, 3000, 5121]))
            X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.5, 1, 3])).astype(dtype))
            for ec in contexts(api):
                tot.add(FZ.fuzz_eval(api, trees, FZ.OPS_HOT, X, dtype, ec, label=f"hot seed {seed} rep {rep} N={N}"))
    gate(tot, "hot", f"hot operators, seed {seed}")


@pytest.mark.parametrize("seed", [211, 212])
def test_fuzz_wide_operators_all_option_modes(api, seed):
    tot = FZ.Findings()
    for rep in range(2):
        rng = de.synth.Xoshiro256ss(seed * 1000 + rep)
        for ops, F in ((FZ.OPS_WIDE, 3), (FZ.OPS_HOT, 2)):
            for dtype in (np.float32, np.float64):
                trees = FZ.random_trees(rng, ops, F, dtype, 150, 33, rep)
                g = np.random.Generator(np.random.PCG64(seed + rep))
                N = int(g.integers(1, 1500))
                X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.1, 1, 10])).astype(dtype))
                if rep % 2:
                    X[0, N // 2] = np.inf
                for ec in contexts(api):
                    tot.add(FZ.fuzz_eval(api, trees, ops, X, dtype, ec, label=f"wide seed {seed} rep {rep} N={N}"))
    gate(tot, "wide", f"wide operator set, seed {seed}")


@pytest.mark.parametrize("env", [{"DE_EVAL_THREADED": "0"}, {"DE_NO_FUSE": "1"}, {"DE_NO_FOLD": "1"}, {"DE_X_VEC": "0"}])
def test_fuzz_fallback_kernels(api, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tot = FZ.Findings()
    rng = de.synth.Xoshiro256ss(221)
    for dtype in (np.float32, np.float64):
        trees = FZ.random_trees(rng, FZ.OPS_HOT, 5, dtype, 100, 30)
        g = np.random.Generator(np.random.PCG64(5))
        N = int(g.choice([700, 1025, 2050]))
        X = np.asfortranarray(g.standard_normal((5, N)).astype(dtype))
        for ec in contexts(api)[:2]:
            tot.add(FZ.fuzz_eval(api, trees, FZ.OPS_HOT, X, dtype, ec, label=f"fallback {env}"))
    gate(tot, "hot", f"fall-back kernels {env}")


@pytest.mark.parametrize("seed", [231, 232])
def test_fuzz_jacobians(api, seed):
    tot = FZ.Findings()
    for rep in range(2):
        rng = de.synth.Xoshiro256ss(seed * 100 + rep)
        for ops, F, kind in ((FZ.OPS_HOT, 5, "hot"), (FZ.OPS_WIDE, 3, "wide")):
            for dtype in (np.float32, np.float64):
                trees = FZ.random_trees(rng, ops, F, dtype, 90, 31, rep)
                g = np.random.Generator(np.random.PCG64(seed + rep))
                N = int(g.integers(1, 900))
                X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.3, 1, 4])).astype(dtype))
                for mode in ("variable", "constant", "both"):
                    tot.add(FZ.fuzz_grad(api, trees, ops, X, dtype, mode, label=f"grad seed {seed} rep {rep} N={N}"))
    gate(tot, "wide", f"Jacobians, seed {seed}")


@pytest.mark.parametrize("seed", [241])
def test_fuzz_parametric_expressions(api, seed):
    tot = FZ.Findings()
    for rep in range(3):
        rng = de.synth.Xoshiro256ss(seed * 77 + rep)
        for dtype in (np.float32, np.float64):
            P, F = 1 + rep % 3 * 3, 2 + rep
            trees = FZ.random_trees(rng, FZ.OPS_HOT, F, dtype, 90, 27, rep, de.ParametricNode, P)
            g = np.random.Generator(np.random.PCG64(seed * 10 + rep))
            N, C = int(g.integers(1, 1300)), int(g.integers(1, 9))
            X = np.asfortranarray(g.standard_normal((F, N)).astype(dtype))
            params = np.asfortranarray((g.standard_normal((P, C)) * 2).astype(dtype))
            classes = g.integers(1, C + 1, N).astype(np.int64)
            for ec in contexts(api)[:3]:
                tot.add(FZ.fuzz_eval(api, trees, FZ.OPS_HOT, X, dtype, ec, params, classes, label=f"param seed {seed} rep {rep}"))
            for mode in ("constant", "both", "variable"):
                tot.add(FZ.fuzz_grad(api, trees, FZ.OPS_HOT, X, dtype, mode, params, classes, label=f"param grad seed {seed} rep {rep}"))
    gate(tot, "wide", f"ParametricExpression, seed {seed}")


@pytest.mark.parametrize("case", ["DE_NO_PARAM_ROWS", "20 parameters"])
def test_fuzz_parametric_gather_fallback(api, case, monkeypatch):
    """The eval kernels stage <= 16 parameters as LDS rows (csrc/de_api.cpp rebind); beyond that, or with DE_NO_PARAM_ROWS=1, every use
    of a parameter gathers its samples' values (h_param, BOP_GEN_PARAM): the same differential run on that path."""
    if case == "DE_NO_PARAM_ROWS":
        monkeypatch.setenv("DE_NO_PARAM_ROWS", "1")
    P = 20 if case == "20 parameters" else 5
    tot = FZ.Findings()
    rng = de.synth.Xoshiro256ss(4242)
    for dtype in (np.float32, np.float64):
        trees = FZ.random_trees(rng, FZ.OPS_HOT, 3, dtype, 90, 27, 1, de.ParametricNode, P)
        g = np.random.Generator(np.random.PCG64(4243))
        N, C = 777, 6
        X = np.asfortranarray(g.standard_normal((3, N)).astype(dtype))
        params = np.asfortranarray((g.standard_normal((P, C)) * 2).astype(dtype))
        classes = g.integers(1, C + 1, N).astype(np.int64)
        for ec in contexts(api)[:3]:
            tot.add(FZ.fuzz_eval(api, trees, FZ.OPS_HOT, X, dtype, ec, params, classes, label=f"param gathers ({case})"))
    gate(tot, "wide", f"ParametricExpression, gathered parameters ({case})")


MIXED = de.OperatorEnum(unary_operators=("abs", "cos", "exp"), binary_operators=("+", "-", "*", "/"),
                        ternary_operators=("fma", "clamp", "+", "max"))  # test/test_supposition_consistency.jl:20


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mixed_arity_random_trees_match_the_oracle(api, dtype):
    """The reference's property test (test/test_supposition_consistency.jl:41-104): random trees over unary, binary AND
    ternary operators, up to 20 layers, small random batches — here against the oracle, in every option mode, on
    populations (the device's unit of work) and with a large batch as well."""
    from oracle import oracle
    tot = FZ.Findings()
    rng = de.synth.Xoshiro256ss(20250928)
    g = np.random.Generator(np.random.PCG64(17))
    n_ternary = 0
    for rep in range(4):
        trees = [FZ.gen_mixed_arity_tree(rng, MIXED, 5, dtype, 20) for _ in range(150)]
        trees = [t for t in trees if de.count_nodes(t) <= 400]
        n_ternary += sum(1 for t in trees for n in de.node.postorder(t) if n.degree == 3)
        N = [int(g.integers(1, 17)), 300, 1031, 16][rep]  # the reference draws batches of 1..16 (supposition_utils.jl:52-63)
        X = np.asfortranarray((g.standard_normal((5, N)) * g.choice([0.5, 1, 2])).astype(dtype))
        for ec in contexts(api):
            tot.add(FZ.fuzz_eval(api, trees, MIXED, X, dtype, ec, label=f"mixed arity rep {rep} N={N}"))
    assert n_ternary > 200, n_ternary
    gate(tot, "wide", f"mixed arity {np.dtype(dtype).name}")


def test_fuzz_loss_gradient_reverse_against_forward(api, monkeypatch):
    """Reverse accumulation against forward duals (tests/fuzz/fuzz_lossgrad.py with a fixed seed): same flags and losses, rows
    equal up to the conditioning of the row; the two documented divergences (rows whose paths cancel, overflow in
    one association only) are counted and capped."""
    bad = flagdiff = checked = 0
    findings = []
    for rep in range(2):
        rng = de.synth.Xoshiro256ss(251000 + rep)
        for ops, F, P in ((FZ.OPS_HOT, 5, 0), (FZ.OPS_WIDE, 3, 0), (FZ.OPS_HOT, 2, 3)):
            for dtype in (np.float32, np.float64):
                nt = de.ParametricNode if P else de.Node
                trees = FZ.random_trees(rng, ops, F, dtype, 100, 31, rep, nt, P)
                g = np.random.Generator(np.random.PCG64(2510 + rep))
                N = int(g.integers(1, 1500))
                X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.3, 1, 3])).astype(dtype))
                y = g.standard_normal(N).astype(dtype)
                w = (g.random(N) > 0.15).astype(dtype)
                kw = dict(params=np.asfortranarray(g.standard_normal((P, 4)).astype(dtype)), classes=g.integers(1, 5, N)) if P else {}
                pop = api.Population(trees, ops, dtype, n_features=F, n_params=P)
                for variable in (False, True, "both"):
                    for kind in ("L2", "pullback"):
                        monkeypatch.setenv("DE_LOSS_GRAD_REVERSE", "0")
                        lf, df, okf = pop.eval_loss_grad(X, y, weights=w, loss=kind, variable=variable, **kw)
                        monkeypatch.setenv("DE_LOSS_GRAD_REVERSE", "1")
                        lr, dr, okr = pop.eval_loss_grad(X, y, weights=w, loss=kind, variable=variable, **kw)
                        out, grads, okg = pop.eval_grad(X, variable, **kw)
                        eps = np.finfo(dtype).eps
                        for t in range(len(trees)):
                            if okf[t] != okr[t]:
                                flagdiff += 1
                                continue
                            if not okf[t]:
                                continue
                            checked += 1
                            g64 = np.asarray(grads[t], dtype=np.float64)
                            lp = y.astype(np.float64) if kind == "pullback" else 2 * (out[t].astype(np.float64) - y)
                            with np.errstate(all="ignore"):
                                mag = (np.abs(w * lp)[None, :] * np.abs(g64)).sum(axis=1)
                                err = np.abs(dr[t].astype(np.float64) - df[t].astype(np.float64))
                                M = mag.max(initial=0)
                                lim = 4096 * eps * np.where(mag > 1e-6 * M, mag, M) + 1e-3 * np.abs(df[t]) * (mag > 1e-6 * M) + 1e-30
                                fin = np.isfinite(df[t]) & np.isfinite(dr[t]) & (mag < 0.01 * np.finfo(dtype).max)
                                loss_off = kind == "L2" and np.isfinite(lf[t]) and lf[t] != lr[t] and abs(lf[t] - lr[t]) > 64 * eps * abs(lf[t])
                            if np.any((err > lim) & fin) or loss_off:
                                bad += 1
                                findings.append(f"{np.dtype(dtype).name} {variable} {kind} {de.string_tree(trees[t], ops)[:140]}")
                pop.close()
    print(f"[fuzz gate loss gradient] {checked} cases, {bad} rows beyond the row's conditioning, {flagdiff} flag differences")
    assert checked > 3000
    assert bad <= max(3, 0.002 * checked), findings[:10]
    assert flagdiff <= max(3, 0.002 * checked)
