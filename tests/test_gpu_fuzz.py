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

# Caps = what was OBSERVED, with a factor of two, not a free allowance (round 5's run of the gate,
# profiles/r5_pytest_gpu_8ulp.log: the largest ill-conditioned share of any flavour is 1.13 % (mixed arity), and NO flag differs anywhere):
ILL_VALUE_CAP = {"hot": 0.015, "wide": 0.025}  # share of compared samples / entries the model may class as ill-conditioned (0.05 / 0.20 until round 4)
ILL_FLAG_CAP_ABS = 2                           # trees whose FLAG differs through an ill-conditioned sample, per gate (3 % of the trees until round 4; observed: 0)


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
    assert f.ill_flags <= ILL_FLAG_CAP_ABS, f"flag differences on ill-conditioned trees above the cap [{what}]: {f.summary()}"


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


@pytest.mark.parametrize("case", ["DE_NO_PARAM_ROWS", "20 parameters", "DE_EVAL_WAVES=1", "DE_EVAL_WAVES=2"])
def test_fuzz_parametric_gather_fallback(api, case, monkeypatch):
    """The eval kernels stage <= 16 parameters as LDS rows (csrc/de_api_program.cpp rebind); beyond that, or with DE_NO_PARAM_ROWS=1, every use
    of a parameter gathers its samples' values (h_param, BOP_GEN_PARAM): the same differential run on that path.  Staged rows run in wave
    groups since round 6 (4 waves per workgroup by default here: the other fuzz tests): the one-wave and two-wave forms as well."""
    if case == "DE_NO_PARAM_ROWS":
        monkeypatch.setenv("DE_NO_PARAM_ROWS", "1")
    if case.startswith("DE_EVAL_WAVES"):
        monkeypatch.setenv("DE_EVAL_WAVES", case[-1])
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


# ---- the EXIT PATH against the oracle ------------------------------------------------------------------------
# Launches over >= 512 sample tiles and >= 96 trees run the priority tiles as a probe launch, compact the live trees and run the
# launch proper over dense chunks (csrc/de_kernels.hip); the fuzzers above use a few thousand samples and never get there.  Forced
# here on every launch (DE_PRIO_MIN_TILES = DE_PRIO_MIN_TREES = 1): the same differential runs — eval in every option mode, Jacobians
# in the three modes, ParametricExpression, the fused loss — against the ORACLE, not against the library's own full evaluation.
def _force_exit_path(monkeypatch):
    monkeypatch.setenv("DE_PRIO_MIN_TILES", "1")
    monkeypatch.setenv("DE_PRIO_MIN_TREES", "1")


@pytest.mark.parametrize("seed", [261, 262])
def test_fuzz_exit_path_forced_eval_and_jacobians(api, seed, monkeypatch):
    _force_exit_path(monkeypatch)
    tot, totg = FZ.Findings(), FZ.Findings()
    live = []
    for rep in range(2):
        rng = de.synth.Xoshiro256ss(seed * 53 + rep)
        for dtype in (np.float32, np.float64):
            F = 2 + (seed + rep) % 5
            trees = FZ.random_trees(rng, FZ.OPS_HOT, F, dtype, 140, 36, rep)
            g = np.random.Generator(np.random.PCG64(seed * 3 + rep))
            N = int(g.choice([2049, 4100, 6007]))  # 9-24 sample tiles: probe launch on <= 3 F of them, then the compacted launch
            X = np.asfortranarray((g.standard_normal((F, N)) * g.choice([0.5, 1, 3])).astype(dtype))
            for ec in contexts(api):
                tot.add(FZ.fuzz_eval(api, trees, FZ.OPS_HOT, X, dtype, ec, label=f"exit path seed {seed} rep {rep} N={N}"))
            for mode in ("variable", "constant", "both"):
                totg.add(FZ.fuzz_grad(api, trees, FZ.OPS_HOT, X, dtype, mode, label=f"exit path grad seed {seed} rep {rep} N={N}"))
            pop = api.Population(trees, FZ.OPS_HOT, dtype, n_features=F)
            pop.eval(X)
            live.append((pop.last_live_trees(), len(trees)))
            pop.close()
    print("live trees behind the probe launch:", live)
    assert all(0 <= n < m for n, m in live), live  # the launches really compacted (and flagged something)
    gate(tot, "hot", f"exit path forced, eval, seed {seed}")
    gate(totg, "wide", f"exit path forced, Jacobians, seed {seed}")


def test_fuzz_exit_path_forced_parametric_and_loss(api, monkeypatch):
    from oracle import oracle
    _force_exit_path(monkeypatch)
    tot = FZ.Findings()
    for rep in range(2):
        rng = de.synth.Xoshiro256ss(27100 + rep)
        for dtype in (np.float32, np.float64):
            P, F = 2 + rep * 3, 3
            trees = FZ.random_trees(rng, FZ.OPS_HOT, F, dtype, 120, 27, rep, de.ParametricNode, P)
            g = np.random.Generator(np.random.PCG64(2710 + rep))
            N, C = int(g.choice([2500, 5200])), int(g.integers(2, 9))
            X = np.asfortranarray(g.standard_normal((F, N)).astype(dtype))
            params = np.asfortranarray((g.standard_normal((P, C)) * 2).astype(dtype))
            classes = g.integers(1, C + 1, N).astype(np.int64)
            for ec in contexts(api)[:3]:
                tot.add(FZ.fuzz_eval(api, trees, FZ.OPS_HOT, X, dtype, ec, params, classes, label=f"exit path param rep {rep}"))
            for mode in ("constant", "both"):
                tot.add(FZ.fuzz_grad(api, trees, FZ.OPS_HOT, X, dtype, mode, params, classes, label=f"exit path param grad rep {rep}"))
    gate(tot, "wide", "exit path forced, ParametricExpression")
    # the fused loss: sum (tree(X) - y)^2 against the oracle's rows reduced in float64; NaN exactly where the oracle's flag is false
    for dtype in (np.float32, np.float64):
        rng = de.synth.Xoshiro256ss(27200)
        trees = FZ.random_trees(rng, FZ.OPS_HOT, 4, dtype, 150, 30)
        g = np.random.Generator(np.random.PCG64(2720))
        N = 4611
        X = np.asfortranarray(g.standard_normal((4, N)).astype(dtype))
        y = g.standard_normal(N).astype(dtype)
        pop = api.Population(trees, FZ.OPS_HOT, dtype, n_features=4)
        loss, ok = pop.eval_loss(X, y)
        n_live = pop.last_live_trees()
        pop.close()
        assert 0 <= n_live < len(trees)
        bad = []
        for t, tree in enumerate(trees):
            tape, consts = de.flatten(tree, FZ.OPS_HOT, dtype)
            yo, ok_o = oracle.eval_tree_array(tape, consts, X, elementwise=True)
            if bool(ok[t]) != ok_o:
                tol = FZ.parity_tolerance(tree, FZ.OPS_HOT, X, dtype)
                if not np.isinf(tol).any():
                    bad.append(f"FLAG tree {t}")
                continue
            if not ok_o:
                assert np.isnan(loss[t]), t
                continue
            tol = FZ.parity_tolerance(tree, FZ.OPS_HOT, X, dtype)
            if np.isinf(tol).any():
                continue
            e = yo.astype(np.float64) - y.astype(np.float64)
            ref = float(np.sum(e * e))
            bound = float(np.sum(2 * np.abs(e) * tol + tol * tol)) + (1e-5 if dtype == np.float32 else 1e-12) * ref
            if not np.isfinite(ref) or not np.isfinite(bound):  # a sum of finite squares that overflows the float64 reference or its bound: nothing to compare
                continue
            if not abs(float(loss[t]) - ref) <= bound:
                bad.append(f"LOSS tree {t}: {float(loss[t])} vs {ref} (bound {bound})")
        assert not bad, bad[:10]


def _oracle_rows(trees, ops, X, dtype):
    """Oracle rows and flags of a population, one tree per host thread (ctypes releases the GIL inside the C call)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    tapes = [de.flatten(t, ops, dtype) for t in trees]
    with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
        return list(ex.map(lambda tc: oracle.eval_tree_array(tc[0], tc[1], X, elementwise=True), tapes))


def _hand_made(ops):
    x1, x2 = de.Node(feature=1), de.Node(feature=2)
    B = {n_: i + 1 for i, n_ in enumerate(ops.binops)}
    U = {n_: i + 1 for i, n_ in enumerate(ops.unaops)}
    return [de.Node(U["exp"], de.Node(U["exp"], de.Node(B["*"], x1, de.Node(val=40.0)))),  # overflows on most tiles
            de.Node(B["/"], de.Node(val=1.0), de.Node(B["-"], x2, x2)),                     # 1 / 0 everywhere
            de.Node(U["exp"], de.Node(U["exp"], de.Node(B["*"], x1, de.Node(val=0.5)))),    # overflows for x1 > 8.97 only
            de.Node(B["+"], de.Node(U["cos"], x1), x2)]                                    # always complete


def _check_against_oracle(api, trees, ops, X, what):
    """early exit (default thresholds) == full evaluation == oracle: flags, and the rows of complete trees within the suite's
    per-sample tolerance model on a subset of the samples (the model costs ~16 float64 re-evaluations per tree) and within 1e-4
    relative on 99.9 % of ALL samples."""
    import torch
    dtype = np.float32
    n, N = len(trees), X.shape[1]
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda().t()
    res = {}
    for full in (True, False):
        pop = api.Population(trees, ops, dtype, n_features=X.shape[0], eval_context=api.EvalContext(full_eval=full))
        out, ok = pop.eval(Xd)
        torch.cuda.synchronize()
        res[full] = (out.cpu().numpy(), ok.cpu().numpy().astype(bool), pop.last_live_trees())
        pop.close()
    (of, kf, _), (oe, ke, n_live) = res[True], res[False]
    rows = _oracle_rows(trees, ops, X, dtype)
    ko = np.array([r[1] for r in rows])
    assert np.array_equal(kf, ke), what
    ill = 0
    for t in np.nonzero(ke != ko)[0]:  # a flag may legitimately differ where an overflow is a rounding away
        tol = FZ.parity_tolerance(trees[t], ops, X[:, ::max(1, N // 65536)], dtype)
        assert np.isinf(tol).any() or not np.isfinite(rows[t][0]).all(), f"{what}: flag of tree {t} differs from the oracle's"
        ill += 1
    print(f"[{what}] {ill} flag(s) differ from the oracle's on ill-conditioned trees (cap {ILL_FLAG_CAP_ABS})")
    assert ill <= ILL_FLAG_CAP_ABS, (what, ill)
    assert n_live >= 0, f"{what}: the launch did not take the priority-tile / compaction path"
    by_prio = (n - n_live) / max(1, int((~ke).sum()))
    print(f"[{what}] {n} trees, {int((~ke).sum())} incomplete, {n - n_live} of them flagged by the probe launch of the priority tiles ({100 * by_prio:.0f} %)")
    sub = np.random.Generator(np.random.PCG64(5)).choice(N, 4096, replace=False)
    for t in np.nonzero(ke & ko)[0]:
        assert np.array_equal(oe[t].view(np.uint32), of[t].view(np.uint32)), (what, t)
        y = rows[t][0]
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(y != 0, np.abs(oe[t] - y) / np.abs(y), np.abs(oe[t]))
        tol = FZ.parity_tolerance(trees[t], ops, X[:, sub], dtype)
        m = np.isfinite(tol)
        err = np.abs(oe[t][sub].astype(np.float64) - y[sub].astype(np.float64))
        assert not np.any(err[m] > tol[m]), (what, t, float(np.max(err[m] / tol[m])))
        if np.mean(m & (tol <= 1e-4 * np.abs(y[sub]))) > 0.99:  # a well-conditioned tree (by the model, on the subset): EVERY sample of the row
            assert np.mean(rel <= 1e-4) >= 0.995, (what, t, float(np.mean(rel <= 1e-4)))
    return by_prio


def test_exit_path_at_default_thresholds_matches_the_oracle(api):
    """2^20 + 77 samples x 132 trees with the library's own thresholds (4097 sample tiles: priority tiles, probe launch, compaction):
    flags and complete rows against the oracle."""
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(128, seed=0xEE21) + _hand_made(ops)
    X = de.synth.random_X(5, 2**20 + 77, seed=21)
    assert api.library().de_prio_tiles_wanted(X.shape[1], 5, len(trees)) == 1
    _check_against_oracle(api, trees, ops, np.asfortranarray(X), "randn X, 2^20 samples")


@pytest.mark.parametrize("kind", ["uniform_positive", "lognormal", "constant_column", "nan_inf_planted"])
def test_priority_tiles_on_non_gaussian_data(api, kind):
    """The priority tiles were tuned on randn features.  Other data: all-positive uniform, heavy-tailed, a constant
    column, NaN / Inf planted in X (the pre-pass ranks those first) — flags == full evaluation == oracle, complete rows within
    tolerance, and the share of incomplete trees the probe launch flags is reported."""
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(156, seed=0xEE22) + _hand_made(ops)
    N = 2**18 + 5  # 1025 sample tiles
    g = np.random.Generator(np.random.PCG64(0xEE23))
    if kind == "uniform_positive":
        X = g.uniform(0.5, 4.0, (5, N))
    elif kind == "lognormal":
        X = g.lognormal(0.0, 1.5, (5, N))
    elif kind == "constant_column":
        X = g.standard_normal((5, N))
        X[2, :] = 1.25
    else:
        X = g.standard_normal((5, N))
        X[0, N - 3] = np.nan
        X[3, 17] = np.inf
        X[4, N // 2] = -np.inf
    X = np.asfortranarray(X.astype(np.float32))
    assert api.library().de_prio_tiles_wanted(N, 5, len(trees)) == 1
    _check_against_oracle(api, trees, ops, X, f"{kind} X")
