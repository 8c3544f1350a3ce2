"""BASELINE.json configurations 4 and 5 at FULL size on one MI355X, through size-independent properties
(an oracle run at these sizes would take hours): run with `pytest -m gpu`.

* C4 — 10 000 random trees x (5 x 10^7) Float32, tree-sharded over 8 GPUs: one GPU's shard,
  rank 0 of 8 = trees {t : t mod 8 = 0} = 1 250 trees, 50 GB of output, is evaluated exactly as `bench.py
  --workload C4` does.
* C5 — ParametricExpression population with 8 parameters, 10^6 samples: eval + constant-mode gradient with
  C = 16 classes (parameter table in LDS) and the fully per-sample stress C = N, classes = 1:N (table in global
  memory), SURVEY.md §8d.
Properties: head and ragged tail of every tree equal a separate small launch on those columns (tiling
independence) AND the oracle on a sub-sample of trees; the flag of the full run is the AND of the flags of a
sample split; two launches give the same bits."""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from helpers import grad_tolerance, parity_tolerance
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


def _cols(Xd, a, b):
    return Xd[:, a:b].t().contiguous().t()


def test_full_size_properties_config_C4_one_gpu_shard(api):
    import torch
    from dynamicexpressions_jl_amd import dist as dedist
    ops = de.synth.BENCH_OPERATORS
    full = de.synth.random_population(10000, seed=0xDE04)
    ids = dedist.shard_indices(len(full), 0, 8)
    assert len(ids) == 1250 and ids[:3] == [0, 8, 16]
    trees = [full[i] for i in ids]
    N = 10**7
    g = torch.Generator(device="cuda").manual_seed(1)
    Xd = torch.randn((N, 5), generator=g, device="cuda", dtype=torch.float32).t()
    pop = api.Population(trees, ops, np.float32, n_features=5)
    out, ok = pop.eval(Xd)  # 1250 x 10^7 x 4 B = 50 GB
    assert out.shape == (1250, N)
    H, T = 2048, 1000  # N is not a multiple of the 1024-sample tile: the tail is ragged
    head, ok_h = pop.eval(Xd[:, :H])
    tail, ok_t = pop.eval(_cols(Xd, N - T, N))
    torch.cuda.synchronize()
    okc = ok.cpu().numpy().astype(bool)
    complete = ok
    assert torch.equal(out[:, :H][complete], head[complete])
    assert torch.equal(out[:, N - T:][complete], tail[complete])
    # flag of the full run = AND over an 8-way sample split
    acc = torch.ones_like(ok)
    for i in range(8):
        acc &= pop.eval(_cols(Xd, i * (N // 8), (i + 1) * (N // 8)))[1]
    assert torch.equal(ok, acc)
    # run-to-run bit equality through a per-tree checksum (in chunks: a second 50 GB buffer is fine, a float64 copy is not)
    s1 = torch.stack([torch.nan_to_num(out[t0:t0 + 50]).double().sum(1) for t0 in range(0, 1250, 50)]).reshape(-1)
    del out
    out2, ok2 = pop.eval(Xd)
    s2 = torch.stack([torch.nan_to_num(out2[t0:t0 + 50]).double().sum(1) for t0 in range(0, 1250, 50)]).reshape(-1)
    assert torch.equal(ok, ok2)
    assert torch.equal(s1[complete], s2[complete])
    assert 100 < int(okc.sum()) < 1250
    # the oracle on the head columns of every 25th tree anchors the values themselves
    Xh = np.asfortranarray(Xd[:, :H].cpu().numpy())
    head_np = head.cpu().numpy()
    okh = ok_h.cpu().numpy().astype(bool)
    n_cmp = 0
    for t in range(0, 1250, 25):
        tape, consts = de.flatten(trees[t], ops, np.float32)
        y, ok_el = oracle.eval_tree_array(tape, consts, Xh, elementwise=True)
        assert bool(okh[t]) == ok_el
        if ok_el:
            tol = parity_tolerance(trees[t], ops, Xh, np.float32)
            m = np.isfinite(tol)
            assert np.all(np.abs(head_np[t].astype(np.float64) - y)[m] <= tol[m]), de.string_tree(trees[t], ops)
            n_cmp += 1
    assert n_cmp >= 5
    pop.close()


@pytest.mark.parametrize("per_sample", [False, True], ids=["C=16", "C=N"])
def test_full_size_properties_config_C5(api, per_sample):
    import torch
    ops = de.synth.BENCH_OPERATORS
    P, F, N, n_trees = 8, 5, 10**6, 1000
    trees = de.synth.random_population(n_trees, seed=0xDE05, nfeatures=F, node_type=de.ParametricNode, nparams=P)
    g = torch.Generator(device="cuda").manual_seed(1)
    Xd = torch.randn((N, F), generator=g, device="cuda", dtype=torch.float32).t()
    C = N if per_sample else 16
    params = torch.randn((C, P), generator=g, device="cuda", dtype=torch.float32).t()  # [P, C], parameter index fastest
    if per_sample:
        classes = torch.arange(1, N + 1, device="cuda", dtype=torch.int32)  # "fully per-sample": classes = 1:N
    else:
        classes = torch.randint(1, C + 1, (N,), generator=g, device="cuda", dtype=torch.int32)
    pop = api.Population(trees, ops, np.float32, n_features=F, n_params=P)
    out, ok = pop.eval(Xd, params, classes)
    H, T = 2048, 1000
    head, ok_h = pop.eval(Xd[:, :H], params, classes[:H])
    tail, ok_t = pop.eval(_cols(Xd, N - T, N), params, classes[N - T:])
    torch.cuda.synchronize()
    assert torch.equal(out[:, :H][ok], head[ok])
    assert torch.equal(out[:, N - T:][ok], tail[ok])
    acc = torch.ones_like(ok)
    for i in range(4):
        a, b = i * (N // 4), (i + 1) * (N // 4)
        acc &= pop.eval(_cols(Xd, a, b), params, classes[a:b])[1]
    assert torch.equal(ok, acc)
    out2, ok2 = pop.eval(Xd, params, classes)
    assert torch.equal(ok, ok2) and torch.equal(torch.nan_to_num(out[ok]), torch.nan_to_num(out2[ok]))
    assert 50 < int(ok.sum()) < n_trees
    # the other half of C5: the constant-mode Jacobian at full size; its head equals a small launch, bit for bit
    sub = list(range(0, n_trees, 10))  # 100 trees: ~4 constants each x 10^6 samples
    popg = api.Population([trees[t] for t in sub], ops, np.float32, n_features=F, n_params=P)
    og, grads, okg = popg.eval_grad(Xd, False, params, classes)
    ogh, grads_h, okg_h = popg.eval_grad(Xd[:, :H], False, params, classes[:H])
    torch.cuda.synchronize()
    for k in range(len(sub)):
        if bool(okg[k]):
            assert torch.equal(grads[k][:, :H], grads_h[k])
            assert torch.equal(og[k, :H], ogh[k])
    # oracle anchor on the head columns (the reference's own formulation: parameters gathered above X)
    Xh = np.asfortranarray(Xd[:, :H].cpu().numpy())
    ph = np.asfortranarray(params.cpu().numpy()[:, :H] if per_sample else params.cpu().numpy())
    ch = classes[:H].cpu().numpy().astype(np.int64)
    head_np, okh = head.cpu().numpy(), ok_h.cpu().numpy().astype(bool)
    n_cmp = n_ent = n_ill = 0
    for k, t in enumerate(sub[:40]):
        tape, consts = de.flatten(trees[t], ops, np.float32)
        y, ok_el = oracle.eval_tree_array_parametric(tape, consts, Xh, ph, ch.astype(np.int32), 1, elementwise=True)
        assert bool(okh[t]) == ok_el, de.string_tree(trees[t], ops)
        if ok_el:
            tol = parity_tolerance(trees[t], ops, Xh, np.float32, 7, ph, ch - 1)
            m = np.isfinite(tol)
            assert np.all(np.abs(head_np[t].astype(np.float64) - y)[m] <= tol[m]), de.string_tree(trees[t], ops)
            n_cmp += 1
        t2, PX = oracle.parametric_to_plain(tape, Xh, ph, ch)
        yg, gg, okg_el = oracle.eval_grad_tree_array(t2, consts, PX, oracle.GRAD_CONSTANT, elementwise=True)
        assert bool(okg_h[k]) == okg_el
        if okg_el and gg.size:
            tolg = grad_tolerance(trees[t], ops, Xh, np.float32, "constant", ph, ch, 1)
            err = np.abs(grads_h[k].cpu().numpy().astype(np.float64) - gg.astype(np.float64))
            assert not (err > tolg).any(), de.string_tree(trees[t], ops)
            n_ent += tolg.size
            n_ill += int(np.isinf(tolg).sum())
    assert n_cmp >= 3
    assert n_ill <= 0.05 * max(n_ent, 1)
    pop.close()
    popg.close()
