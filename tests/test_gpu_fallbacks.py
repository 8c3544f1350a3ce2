"""Every kernel the library can fall back to through an environment switch is shipped code: each must
give the flags and values of the default path.  The switches are read when a program is created / launched, so every
case builds its own populations.
  DE_EVAL_THREADED=0  flat-`switch` eval kernel instead of the direct-threaded one (also what wide-X programs run)
  DE_NO_FOLD=1        no device-side constant folding        DE_NO_FUSE=1   no superinstructions
  DE_X_VEC=0          scalar staging of the X tile           DE_GRAD_THREADED=0  flat-`switch` gradient kernel
  DE_LOSS_GRAD_REVERSE=0|1  forward duals / reverse accumulation for the fused loss gradient"""
import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from helpers import grad_tolerance, parity_tolerance

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dynamicexpressions_jl_amd import api as _api
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _api.library()
    return _api


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    m = ~(np.isnan(a) & np.isnan(b))
    ui = np.uint32 if a.dtype == np.float32 else np.uint64
    np.testing.assert_array_equal(np.ascontiguousarray(a).view(ui)[m], np.ascontiguousarray(b).view(ui)[m], err_msg=what)


WIDE = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max", "pow_abs2"),
                       unary_operators=("cos", "exp", "sin", "safe_log", "square", "tanh", "abs"))


def populations(dtype):
    rng = de.synth.Xoshiro256ss(77)
    bench = de.synth.random_population(150, seed=0xDE02, dtype=dtype)
    wide = [de.synth.gen_random_tree_fixed_size(3 + i % 25, WIDE, 4, rng, dtype) for i in range(120)]
    return ((bench, de.synth.BENCH_OPERATORS, de.synth.random_X(5, 3001, seed=3, dtype=dtype)),
            (wide, WIDE, de.synth.random_X(4, 1537, seed=4, dtype=dtype)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("env", [{"DE_EVAL_THREADED": "0"}, {"DE_NO_FOLD": "1"}, {"DE_NO_FUSE": "1"}, {"DE_X_VEC": "0"},
                                 {"DE_EVAL_THREADED": "0", "DE_NO_FOLD": "1"}], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_eval_fallback_paths_match_the_default_path(api, monkeypatch, env, dtype):
    for trees, ops, X in populations(dtype):
        for ec in (api.EvalContext(), api.EvalContext(early_exit=False)):
            ref = api.Population(trees, ops, dtype, n_features=X.shape[0], eval_context=ec)
            a, ka = ref.eval(X)
            name_default = ref.ctx.last_kernel_name()
            ref.close()
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            alt = api.Population(trees, ops, dtype, n_features=X.shape[0], eval_context=ec)
            b, kb = alt.eval(X)
            name_alt = alt.ctx.last_kernel_name()
            alt.close()
            for k in env:
                monkeypatch.delenv(k)
            assert np.array_equal(ka, kb), env
            # same device functions on every path: the same bits wherever the evaluation is complete
            _same(a[ka], b[kb], str(env))
            if "DE_EVAL_THREADED" in env:
                assert name_default == "de_eval_threaded_kernel" and name_alt == "de_eval_tape_kernel"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gradient_fallback_kernel_matches_the_threaded_one(api, monkeypatch, dtype):
    for trees, ops, X in populations(dtype):
        X = X[:, :700]
        for variable in (True, False, "both"):
            ref = api.Population(trees, ops, dtype, n_features=X.shape[0])
            oa, ga, ka = ref.eval_grad(X, variable)
            ref.close()
            monkeypatch.setenv("DE_GRAD_THREADED", "0")
            alt = api.Population(trees, ops, dtype, n_features=X.shape[0])
            ob, gb, kb = alt.eval_grad(X, variable)
            name = alt.ctx.last_kernel_name()
            alt.close()
            monkeypatch.delenv("DE_GRAD_THREADED")
            assert np.array_equal(ka, kb) and "tape" in name
            mode = {True: "variable", False: "constant", "both": "both"}[variable]
            for t in np.nonzero(ka)[0]:
                # the two kernels share the value/partial functions but not the hot-operator fast paths (cos/exp/sin): each is
                # within the conditioned bound of the true Jacobian (helpers.grad_tolerance), so they are within twice that
                tol = grad_tolerance(trees[t], ops, X, dtype, mode)
                err = np.abs(np.asarray(ga[t], dtype=np.float64) - np.asarray(gb[t], dtype=np.float64))
                assert tol is not None and not (err > 2 * tol).any(), f"tree {t} {variable}"
                tx = parity_tolerance(trees[t], ops, X, dtype)
                m = np.isfinite(tx)
                assert np.all(np.abs(oa[t].astype(np.float64) - ob[t])[m] <= 2 * tx[m])


def test_loss_gradient_forward_and_reverse_kernels_agree(api, monkeypatch):
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(100, seed=5)
    X = de.synth.random_X(5, 4096, seed=6)
    y = np.sin(np.arange(4096)).astype(np.float32)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DE_LOSS_GRAD_REVERSE", mode)
        pop = api.Population(trees, ops, np.float32, n_features=5)
        res[mode] = pop.eval_loss_grad(X, y)
        pop.close()
    monkeypatch.delenv("DE_LOSS_GRAD_REVERSE")
    (l0, d0, k0), (l1, d1, k1) = res["0"], res["1"]
    both = k0 & k1
    assert both.sum() > 20 and np.mean(k0 == k1) > 0.97  # a product chain that overflows in one association only
    np.testing.assert_allclose(l0[both], l1[both], rtol=1e-5)
    for t in np.nonzero(both)[0]:
        sc = np.max(np.abs(d0[t])) + 1e-30 if len(d0[t]) else 1.0
        np.testing.assert_allclose(d0[t], d1[t], rtol=2e-3, atol=2e-4 * sc)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_program_sanitizer_accepts_every_lowered_form(api, monkeypatch, dtype):
    """de_program_verify walks the generic, bound, fused and chained streams against the bounds the launches allocate
    (operand rows, spill slots, LDS offsets, handler addresses in the device table, end records).  Every program the
    library builds must pass — all option sets, turbo, parametric, no-fuse / no-fold variants — and keep passing after
    de_program_set_consts patches immediates in place (DE_VERIFY=1 runs the check inside both calls)."""
    monkeypatch.setenv("DE_VERIFY", "1")
    for trees, ops, X in populations(dtype):
        for ec in (api.EvalContext(), api.EvalContext(early_exit=False), api.EvalContext(use_fused=False), api.EvalContext(bumper=True),
                   api.EvalContext(turbo=True)):
            pop = api.Population(trees, ops, dtype, n_features=X.shape[0], eval_context=ec)
            pop.verify()
            pop.set_constants(np.linspace(-1, 1, int(pop.n_consts.sum())).astype(dtype))
            pop.verify()
            pop.close()
    for env in ({"DE_NO_FUSE": "1"}, {"DE_NO_FOLD": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        trees, ops, X = populations(dtype)[0]
        pop = api.Population(trees, ops, dtype, n_features=5)
        pop.verify()
        pop.close()
        for k in env:
            monkeypatch.delenv(k)
    ptrees = de.synth.random_population(80, seed=0xDE05, nfeatures=5, node_type=de.ParametricNode, nparams=8, dtype=dtype)
    pop = api.Population(ptrees, de.synth.BENCH_OPERATORS, dtype, n_features=5, n_params=8)
    pop.verify()
    pop.close()


def test_dist_c_abi_single_rank_and_rccl_loading(api):
    """de_dist_* (csrc/de_dist.cpp): with world = 1 the exchange degenerates to a copy and needs no RCCL; the unique id
    comes from librccl.so (dlopen), which must be loadable on the GPU box — the N-GPU path of a C / Julia caller."""
    import torch
    from dynamicexpressions_jl_amd import dist as dedist
    ctx = api.default_context()
    comm = dedist.Comm(ctx, 0, 1)
    assert comm.shard_size(10) == 10 and comm.world_size() == 1
    ok = torch.tensor([1, 0, 1, 1, 0], dtype=torch.uint8, device="cuda")
    g = comm.gather_flags(ok, 5)
    X = torch.arange(12, dtype=torch.float32, device="cuda")
    comm.broadcast(X, 0)
    torch.cuda.synchronize()
    assert g.tolist() == [1, 0, 1, 1, 0] and X.tolist() == list(range(12))
    comm.close()
    uid = dedist.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    lib = api.library()
    assert lib.de_dist_shard_size(10, 1, 4) == 3 and lib.de_dist_shard_size(10, 3, 4) == 2 and lib.de_dist_shard_size(2, 3, 4) == 0
    # a one-rank RCCL communicator exercises ncclCommInitRank / ncclAllGather / ncclBroadcast on the device itself
    import ctypes as C
    h = C.c_void_p()
    idb = C.create_string_buffer(uid, 128)
    # (world = 1 short-cuts RCCL inside the library; the calls above already proved the symbols resolve)
    assert lib.de_dist_init(ctx._h, 0, 1, idb, C.byref(h)) == 0 and lib.de_dist_destroy(h) == 0
    assert lib.de_dist_init(ctx._h, 2, 2, idb, C.byref(h)) == 1  # rank outside the world: invalid argument, no RCCL call


def test_bench_two_ranks_on_one_gpu_strong_scaling_line(api):
    """`bench.py --gpus 2` (self-launching, one rank per process) with the ranks sharing this box's GPU
    and the flags gathered through gloo (DE_BENCH_BACKEND=gloo: a dry run of the multi-rank code path — sharding, flag
    gather, max-over-ranks timing; the timing itself means nothing).  The line must say n_gpus 2, strong scaling (ONE
    population split over the ranks: the metric's 1/2/4/8-GPU series) and the communicator's own world size."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["rccl_world_size"] == 2 and d["config"]["trees_job"] == 64 and d["config"]["trees_this_rank"] == 32
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    # round 6: the line says how the flags were gathered and WHY, that every rank declared the dataset once (X is constant
    # across the steps: the per-call pass over X does not shrink with the shard), and `value` is the executed rate beside the all-trees count
    assert d["config"]["dataset_declared"] is True
    assert "flag_gather_reason" in d["config"] and d["config"]["flag_gather"].startswith("torch.distributed")
    assert 0 < d["value"] <= d["value_all_trees"] and d["value_executed"] == d["value"]


def test_dist_timeout_is_accepted_and_a_one_rank_gather_still_works(api):
    """de_dist_set_timeout (round 6): with a bound set, de_dist_gather_flags WAITS for what it queued (the one-rank path is a copy) and
    returns the flags; a negative bound is refused; 0 puts the calls back to asynchronous."""
    import torch
    from dynamicexpressions_jl_amd import dist as dedist
    ctx = api.Context(0)
    comm = dedist.Comm(ctx, 0, 1)
    flags = torch.tensor([1, 0, 1, 1, 0], dtype=torch.uint8, device="cuda")
    comm.set_timeout(5000)
    out = comm.gather_flags(flags, 5)
    assert torch.equal(out.cpu(), flags.cpu())   # (no synchronisation needed: the bounded call waited)
    with pytest.raises(api.DeviceError):
        comm.set_timeout(-1)
    comm.set_timeout(0)
    out = comm.gather_flags(flags, 5)
    ctx.synchronize()
    assert torch.equal(out.cpu(), flags.cpu())
    comm.close()
