"""Early exit at tree granularity (`pytest -m gpu`).

The reference stops evaluating a tree at the first non-finite intermediate array (`@return_on_nonfinite_array`,
src/Evaluate.jl:26-32; per-node `ok` in src/EvaluateDerivative.jl:230-243) and returns a partially evaluated buffer
(src/Evaluate.jl:350-351); SURVEY.md §8a: with ok == false only the flag is contractual.  The kernels do the same per
workgroup: a tree whose flag is already 0 when a workgroup starts is not evaluated on that workgroup's samples.  Contract
tested here, against the evaluate-everything mode (`EvalContext(full_eval=True)` = DE_OPT_FULL_EVAL):
  * the flags are identical;
  * every row / Jacobian / fused loss of a COMPLETE tree has the same bits;
  * rows of incomplete trees are really left alone (a sentinel survives in them) — the exit happens;
  * fused losses of incomplete trees are NaN in both modes;
  * early_exit=False never skips (its flags stay true)."""
import ctypes

import numpy as np
import pytest

import dynamicexpressions_jl_amd as de

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


def _population(n=192, seed=0xEE01):
    # random bench-style trees + hand-made ones that fail early, late (one sample in the last tile) and never
    trees = de.synth.random_population(n, seed=seed)
    x1, x2 = de.Node(feature=1), de.Node(feature=2)
    ops = de.synth.BENCH_OPERATORS
    B = {n_: i + 1 for i, n_ in enumerate(ops.binops)}
    U = {n_: i + 1 for i, n_ in enumerate(ops.unaops)}
    trees.append(de.Node(U["exp"], de.Node(U["exp"], de.Node(B["*"], x1, de.Node(val=40.0)))))   # overflows on most tiles
    trees.append(de.Node(B["/"], de.Node(val=1.0), de.Node(B["-"], x2, x2)))                      # 1 / 0 everywhere
    trees.append(de.Node(B["+"], de.Node(U["cos"], x1), x2))                                      # always complete
    return trees, ops


def _X(N, seed=3):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()


@pytest.mark.parametrize("turbo", [False, True])
def test_eval_flags_and_complete_rows_do_not_depend_on_the_exit(api, turbo):
    import torch
    trees, ops = _population()
    N = 3 * 2**19 + 77  # many more sample tiles than the chip runs at once, ragged tail
    Xd = _X(N)
    lib = api.library()
    res = {}
    for full in (True, False):
        pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(turbo=turbo, full_eval=full))
        out = torch.full((len(trees), N), 12345.0, device="cuda", dtype=torch.float32)
        ok = torch.empty(len(trees), device="cuda", dtype=torch.uint8)
        pop.ctx.use_torch_stream()
        pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, Xd.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        torch.cuda.synchronize()
        res[full] = (out, ok.bool())
        pop.close()
    (of, kf), (oe, ke) = res[True], res[False]
    assert torch.equal(kf, ke)
    assert 0 < int(kf.sum()) < len(trees)
    assert torch.equal(of[kf], oe[kf])                      # complete trees: the same bits
    assert int((of[kf] == 12345.0).sum()) == 0
    assert int((of[~kf] == 12345.0).sum()) == 0             # full evaluation writes every row ...
    # ... the early exit leaves an incomplete row alone from the second wave of workgroups on (the chip holds ~2800 workgroups
    # = ~700 of this launch's 3073 sample tiles at once: those start before any flag is known)
    left_alone = (oe[~ke] == 12345.0).float().mean(dim=1)
    assert float(left_alone.max()) > 0.6, left_alone
    assert float(left_alone.mean()) > 0.3, left_alone
    assert not bool(ke[-2]) and not bool(ke[-3]) and bool(ke[-1])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_priority_tiles_change_nothing_but_the_order(api, dtype, monkeypatch):
    """Launches over >= 512 sample tiles first run the tiles that hold each feature's largest, smallest and closest-to-zero value
    (csrc/de_kernels.hip de_tile_extremes_kernel: those samples flag most incomplete trees at once).  Forced here on a small launch
    (DE_PRIO_MIN_TILES=1): same flags and the same bits in every complete row as without them and as the evaluate-everything
    mode, for the eval, the fused loss and a parametric population; the tree that fails on ONE sample — the largest x1, placed in the
    last tile — is left alone almost everywhere although that tile is the last of the launch."""
    import torch
    trees, ops = _population(128, seed=0xEE07)
    x1 = de.Node(feature=1)
    U = {n_: i + 1 for i, n_ in enumerate(ops.unaops)}
    B = {n_: i + 1 for i, n_ in enumerate(ops.binops)}
    trees.append(de.Node(U["exp"], de.Node(U["exp"], de.Node(B["*"], x1, de.Node(val=0.5)))))  # overflows only for x1 > 8.97
    N = 2**19 + 131
    g = torch.Generator(device="cuda").manual_seed(11)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    Xd = torch.randn((N, 5), generator=g, device="cuda", dtype=tdt).clamp_(-5.0, 5.0)
    Xd[N - 7, 0] = 12.0 if dtype == np.float32 else 14.0   # (Float64: exp(exp(7)) overflows, exp(exp(2.5)) does not)
    Xd = Xd.t()
    y = torch.randn(N, generator=g, device="cuda", dtype=tdt)
    lib = api.library()
    res = {}
    for tag, env, full in (("full", {}, True), ("plain", {"DE_NO_PRIO_TILES": "1"}, False), ("prio", {"DE_PRIO_MIN_TILES": "1"}, False)):
        for k in ("DE_NO_PRIO_TILES", "DE_PRIO_MIN_TILES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pop = api.Population(trees, ops, dtype, n_features=5, eval_context=api.EvalContext(full_eval=full))
        out = torch.full((len(trees), N), 12345.0, device="cuda", dtype=tdt)
        ok = torch.empty(len(trees), device="cuda", dtype=torch.uint8)
        pop.ctx.use_torch_stream()
        pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, Xd.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        loss, lk = pop.eval_loss(Xd, y)
        gy, gg, gk = pop.eval_grad(Xd, True)                       # forward-mode kernel with the priority tiles in front
        l1, d1, k1 = pop.eval_loss_grad(Xd, y, variable=False)     # reverse / forward loss gradient
        torch.cuda.synchronize()
        res[tag] = (out, ok.bool(), loss, lk, gg, gk, l1, d1, k1)
        pop.close()
    kf = res["full"][1]
    assert not bool(kf[-1]) and 0 < int(kf.sum()) < len(trees)
    it = torch.int32 if dtype == np.float32 else torch.int64
    for tag in ("plain", "prio"):
        out, k, loss, lk, gg, gk, l1, d1, k1 = res[tag]
        assert torch.equal(k, kf) and torch.equal(lk, res["full"][3])
        assert torch.equal(out[kf], res["full"][0][kf])
        assert torch.equal(loss[kf].view(it), res["full"][2][kf].view(it)) and bool(torch.isnan(loss[~kf]).all())
        assert torch.equal(gk, res["full"][5]) and torch.equal(k1, res["full"][8])
        assert torch.equal(l1[k1].view(it), res["full"][6][k1].view(it))
        for t in range(len(trees)):
            if bool(gk[t]):
                assert torch.equal(gg[t].view(it), res["full"][4][t].view(it)), t
            if bool(k1[t]):
                assert torch.equal(d1[t].view(it), res["full"][7][t].view(it)), t
    alone = {tag: float((res[tag][0][-1] == 12345.0).float().mean()) for tag in ("plain", "prio")}
    print("row of the tree that fails on one sample of the last tile, share left alone:", alone)
    # found by the first workgroups (the chip holds a third of this launch at once: those start before any flag is known) / by the last ones
    assert alone["prio"] > 0.2 and alone["plain"] < 0.05, alone


def test_early_exit_false_evaluates_everything(api):
    import torch
    trees, ops = _population(64)
    N = 2**19
    Xd = _X(N)
    pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(early_exit=False))
    lib = api.library()
    out = torch.full((len(trees), N), 12345.0, device="cuda", dtype=torch.float32)
    ok = torch.empty(len(trees), device="cuda", dtype=torch.uint8)
    pop.ctx.use_torch_stream()
    pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, Xd.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
    torch.cuda.synchronize()
    assert bool(ok.bool().all())            # no validity tests, nothing to exit on (src/Evaluate.jl:305-308)
    assert int((out == 12345.0).sum()) == 0


@pytest.mark.parametrize("variable", [True, False, "both"])
def test_gradients_of_complete_trees_do_not_depend_on_the_exit(api, variable):
    import torch
    trees, ops = _population(96, seed=0xEE02)
    N = 2**18 + 5
    Xd = _X(N, seed=4)
    res = {}
    for full in (True, False):
        pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(full_eval=full))
        out, grads, ok = pop.eval_grad(Xd, variable=variable)
        torch.cuda.synchronize()
        res[full] = (out, grads, ok)
        pop.close()
    (of, gf, kf), (oe, ge, ke) = res[True], res[False]
    assert torch.equal(kf, ke) and 0 < int(kf.sum()) < len(trees)
    for t in range(len(trees)):
        if bool(kf[t]):
            assert torch.equal(of[t], oe[t]), t
            assert torch.equal(gf[t], ge[t]), t


@pytest.mark.parametrize("reverse", ["0", "1"])
def test_fused_losses_do_not_depend_on_the_exit(api, reverse, monkeypatch):
    import torch
    monkeypatch.setenv("DE_LOSS_GRAD_REVERSE", reverse)
    trees, ops = _population(96, seed=0xEE03)
    N = 2**18 + 5
    Xd = _X(N, seed=5)
    g = torch.Generator(device="cuda").manual_seed(9)
    y = torch.randn(N, generator=g, device="cuda", dtype=torch.float32)
    res = {}
    for full in (True, False):
        pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(full_eval=full))
        l0, k0 = pop.eval_loss(Xd, y)
        l1, d1, k1 = pop.eval_loss_grad(Xd, y, variable=False)
        torch.cuda.synchronize()
        res[full] = (l0, k0, l1, d1, k1)
        pop.close()
    (l0f, k0f, l1f, d1f, k1f), (l0e, k0e, l1e, d1e, k1e) = res[True], res[False]
    assert torch.equal(k0f, k0e) and torch.equal(k1f, k1e)
    assert torch.equal(l0f[k0f].view(torch.int32), l0e[k0f].view(torch.int32)) and bool(torch.isnan(l0e[~k0e]).all()) and bool(torch.isnan(l0f[~k0f]).all())
    assert torch.equal(l1f[k1f].view(torch.int32), l1e[k1f].view(torch.int32)) and bool(torch.isnan(l1e[~k1e]).all())
    for t in range(len(trees)):
        if bool(k1f[t]):  # bit patterns: a sum of finite entries may overflow to Inf - Inf = NaN without touching the flag
            assert torch.equal(d1f[t].view(torch.int32), d1e[t].view(torch.int32)), t
        else:
            assert bool(torch.isnan(d1e[t]).all()), t


def test_host_buffers_and_the_flat_switch_kernel(api, monkeypatch):
    # numpy (host, staged) inputs through the fall-back kernel: same contract
    monkeypatch.setenv("DE_EVAL_THREADED", "0")
    trees, ops = _population(40, seed=0xEE04)
    N = 2**17 + 3
    X = np.asfortranarray(_X(N, seed=6).cpu().numpy())
    res = {}
    for full in (True, False):
        pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(full_eval=full))
        out, ok = pop.eval(X)
        res[full] = (out, ok)
        pop.close()
    (of, kf), (oe, ke) = res[True], res[False]
    assert np.array_equal(kf, ke) and 0 < kf.sum() < len(trees)
    assert np.array_equal(of[kf].view(np.uint32), oe[kf].view(np.uint32))


@pytest.mark.parametrize("tpc", ["16", "128"])
def test_float64_parametric_and_sub_chunks(api, tpc, monkeypatch):
    """The same contract for Float64 (32-bit handler addresses in the records: the walk rebuilds them from the program counter's
    high half), for a ParametricExpression population (the class-row variant of the kernel) and with DE_EVAL_TPC != 64 (more
    than 64 trees per workgroup run as 64-tree sub-chunks with a fresh mask each; fewer: more workgroups per tile)."""
    import torch
    monkeypatch.setenv("DE_EVAL_TPC", tpc)
    ops = de.synth.BENCH_OPERATORS
    N = 2**18 + 3
    g = torch.Generator(device="cuda").manual_seed(7)
    X = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float64).t()
    plain = de.synth.random_population(150, seed=0xEE05, dtype=np.float64)
    par = de.synth.random_population(150, seed=0xEE06, dtype=np.float64, node_type=de.ParametricNode, nparams=3)
    params = torch.randn((4, 3), generator=g, device="cuda", dtype=torch.float64).t()  # [P=3, C=4], column-major
    classes = torch.randint(1, 5, (N,), generator=g, device="cuda", dtype=torch.int32)
    for trees, kw, P in ((plain, {}, 0), (par, dict(params=params, classes=classes), 3)):
        res = {}
        for full in (True, False):
            pop = api.Population(trees, ops, np.float64, n_features=5, n_params=P, eval_context=api.EvalContext(full_eval=full))
            out, ok = pop.eval(X, **kw)
            torch.cuda.synchronize()
            res[full] = (out, ok)
            pop.close()
        (of, kf), (oe, ke) = res[True], res[False]
        assert torch.equal(kf, ke) and 0 < int(kf.sum()) < len(trees)
        assert torch.equal(of[kf].view(torch.int64), oe[kf].view(torch.int64))


@pytest.mark.parametrize("dtype,turbo", [(np.float32, False), (np.float32, True), (np.float64, False)])
def test_compaction_of_the_live_trees_changes_nothing_but_the_order(api, dtype, turbo, monkeypatch):
    """Behind the probe launch of the priority tiles one workgroup re-links the records of the trees whose flag is still 1 into a second
    stream (csrc/de_kernels.hip de_compact_live_kernel) and the launch proper runs dense chunks over it.  Same flags and the same bits
    in every complete row as without the compaction (DE_COMPACT=0: the launch proper walks past flagged trees) and as the
    evaluate-everything mode — eval, fused loss, a parametric population — and a tree the probe launch flags is now left alone on
    EVERY tile of the launch proper (only the <= 15 priority tiles ever touch its row)."""
    import torch
    trees, ops = _population(300, seed=0xEE11)  # 5 chunks of the plain launch, ~2-3 compact ones
    n = len(trees)
    N = 2**19 + 131  # 2049 sample tiles: priority tiles by default
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    it = torch.int32 if dtype == np.float32 else torch.int64
    g = torch.Generator(device="cuda").manual_seed(21)
    Xd = torch.randn((N, 5), generator=g, device="cuda", dtype=tdt).t()
    y = torch.randn(N, generator=g, device="cuda", dtype=tdt)
    par = de.synth.random_population(200, seed=0xEE12, dtype=dtype, node_type=de.ParametricNode, nparams=3)
    params = torch.randn((4, 3), generator=g, device="cuda", dtype=tdt).t()
    classes = torch.randint(1, 5, (N,), generator=g, device="cuda", dtype=torch.int32)
    lib = api.library()
    res = {}
    for tag, env, full in (("full", {}, True), ("walk", {"DE_COMPACT": "0"}, False), ("compact", {"DE_COMPACT": "1"}, False)):
        monkeypatch.delenv("DE_COMPACT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = api.EvalContext(turbo=turbo, full_eval=full)
        pop = api.Population(trees, ops, dtype, n_features=5, eval_context=ctx)
        out = torch.full((n, N), 12345.0, device="cuda", dtype=tdt)
        ok = torch.empty(n, device="cuda", dtype=torch.uint8)
        pop.ctx.use_torch_stream()
        pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, Xd.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
        loss, lk = pop.eval_loss(Xd, y)
        ppop = api.Population(par, ops, dtype, n_features=5, n_params=3, eval_context=ctx)
        pout, pok = ppop.eval(Xd, params=params, classes=classes)
        torch.cuda.synchronize()
        res[tag] = (out, ok.bool(), loss, lk, pout, pok)
        pop.close()
        ppop.close()
    of, kf, lf, lkf, pf, pkf = res["full"]
    assert 0 < int(kf.sum()) < n and 0 < int(pkf.sum()) < len(par)
    for tag in ("walk", "compact"):
        out, k, loss, lk, pout, pok = res[tag]
        assert torch.equal(k, kf) and torch.equal(lk, lkf) and torch.equal(pok, pkf), tag
        assert torch.equal(out[kf].view(it), of[kf].view(it)), tag
        assert int((out[kf] == 12345.0).sum()) == 0, tag
        assert torch.equal(loss[kf].view(it), lf[kf].view(it)) and bool(torch.isnan(loss[~kf]).all()), tag
        assert torch.equal(pout[pkf].view(it), pf[pkf].view(it)), tag
    # 1 / (x2 - x2): flagged by the very first workgroup of the probe launch.  Compacted launch: its row is only ever touched by the
    # priority tiles (<= 15 x 256 samples); walking launch: the same here (flag known to every workgroup), so equal shares are fine
    alone = {tag: float((res[tag][0][n - 2] == 12345.0).float().mean()) for tag in ("walk", "compact")}
    print("row of the tree that fails everywhere, share left alone:", alone)
    assert alone["compact"] >= 1.0 - 15 * 512 / N - 1e-9, alone  # (<= 15 priority tiles of <= 512 samples)


def test_declared_dataset_skips_the_pre_pass_and_changes_nothing(api):
    """`de_ctx_declare_dataset`: the priority-tile keys of an X that does not change between calls are computed once; eval, fused loss and
    the gradient entry points then skip their own pass over X.  Same flags, same bits; a different X (or a withdrawn declaration) takes
    the per-call pass again."""
    import torch
    trees, ops = _population(200, seed=0xEE31)
    N = 2**18 + 9
    Xa, Xb = _X(N, seed=31), _X(N, seed=32)
    y = torch.randn(N, device="cuda", dtype=torch.float32)
    pop = api.Population(trees, ops, np.float32, n_features=5)
    ref = {}
    for tag, X in (("a", Xa), ("b", Xb)):
        ref[tag] = (pop.eval(X), pop.eval_loss(X, y), pop.eval_loss_grad(X, y, variable=False), pop.last_live_trees())
    torch.cuda.synchronize()
    pop.ctx.declare_dataset(Xa)
    for rep in range(2):
        for tag, X in (("a", Xa), ("b", Xb)):  # b: not the declared matrix -> its own pre-pass
            (o, k), (l, lk), (l1, d1, k1) = pop.eval(X), pop.eval_loss(X, y), pop.eval_loss_grad(X, y, variable=False)
            torch.cuda.synchronize()
            (ro, rk), (rl, rlk), (rl1, rd1, rk1), _ = ref[tag]
            assert torch.equal(k, rk) and torch.equal(lk, rlk) and torch.equal(k1, rk1)
            assert torch.equal(o[k.bool()], ro[rk.bool()])
            assert torch.equal(l[lk.bool()].view(torch.int32), rl[rlk.bool()].view(torch.int32))
            for t in range(len(trees)):
                if bool(k1[t]):
                    assert torch.equal(d1[t].view(torch.int32), rd1[t].view(torch.int32)), t
    pop.eval(Xa)
    assert pop.last_live_trees() == ref["a"][3] >= 0
    pop.ctx.declare_dataset(None)
    o, k = pop.eval(Xa)
    assert torch.equal(k, ref["a"][0][1])
    pop.close()


@pytest.mark.parametrize("compact", ["1", "0"])
def test_chain_lengths_around_the_sentinel_bit(api, compact, monkeypatch):
    """A chain of tail calls runs <= 63 trees and finds its end — and a skipped successor — through ONE bit of the skip mask (the sentinel
    behind the last tree's bit, HTREE_END_TAIL / h_tree_skip of csrc/de_kernels.hip).  Populations of 1 ... 130 trees over 8193 sample
    tiles (chains of 1, 2, 62, 63 trees; 64 and more: several chunks; DE_EVAL_TPC=200: sub-chunks of 63 + 63 + 4) with the incomplete
    trees — 1 / (x2 - x2): flagged by the probe launch of the priority tiles — at the first, the last, the last two, every other, all
    but one and all positions; with the live trees re-linked into a dense stream (default) and with the walking launch (DE_COMPACT=0):
    same flags and the same bits in every complete row as the evaluate-everything mode."""
    import torch
    ops = de.synth.BENCH_OPERATORS
    B = {n_: i + 1 for i, n_ in enumerate(ops.binops)}
    U = {n_: i + 1 for i, n_ in enumerate(ops.unaops)}
    x1, x2 = de.Node(feature=1), de.Node(feature=2)
    bad = lambda: de.Node(B["/"], de.Node(val=1.0), de.Node(B["-"], x2, x2))
    good = lambda i: de.Node(B["+"], de.Node(U["cos"], de.Node(B["*"], x1, de.Node(val=0.5 + 0.01 * i))), x2)
    N = 2**21 + 77
    Xd = _X(N, seed=9)
    lib = api.library()
    monkeypatch.setenv("DE_PRIO_MIN_TILES", "1")
    monkeypatch.setenv("DE_PRIO_MIN_TREES", "1")
    monkeypatch.setenv("DE_COMPACT", compact)
    patterns = {
        "none": lambda n: set(), "first": lambda n: {0}, "last": lambda n: {n - 1}, "last2": lambda n: {n - 1, n - 2} & set(range(n)),
        "alternate": lambda n: set(range(0, n, 2)), "all_but_first": lambda n: set(range(1, n)), "all_but_last": lambda n: set(range(n - 1)),
        "all": lambda n: set(range(n)),
    }
    checked = 0
    for n, tpc in ((1, None), (2, None), (62, None), (63, None), (64, None), (65, None), (126, None), (127, None), (130, "200")):
        if tpc:
            monkeypatch.setenv("DE_EVAL_TPC", tpc)
        else:
            monkeypatch.delenv("DE_EVAL_TPC", raising=False)
        for name, pat in patterns.items():
            badset = pat(n)
            trees = [bad() if i in badset else good(i) for i in range(n)]
            res = {}
            for full in (True, False):
                pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(full_eval=full))
                out = torch.full((n, N), 12345.0, device="cuda", dtype=torch.float32)
                ok = torch.empty(n, device="cuda", dtype=torch.uint8)
                pop.ctx.use_torch_stream()
                pop.ctx.check(lib.de_eval(pop.ctx._h, pop._h, Xd.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
                torch.cuda.synchronize()
                res[full] = (out, ok.bool().cpu())
                pop.close()
            (of, kf), (oe, ke) = res[True], res[False]
            want = torch.tensor([i not in badset for i in range(n)])
            assert torch.equal(kf, want) and torch.equal(ke, want), (n, name, compact)
            if bool(want.any()):
                idx = want.nonzero().flatten().cuda()
                assert torch.equal(of[idx], oe[idx]), (n, name, compact)
                assert int((oe[idx] == 12345.0).sum()) == 0, (n, name, compact)   # every complete row written on every tile
            del of, oe, res
            checked += 1
    assert checked == 72
