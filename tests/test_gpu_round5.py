"""Round 5 additions behind the C ABI: the flag exchange's pack / unpack launches (csrc/de_dist.cpp, no RCCL needed for the re-ordering
itself), the context's ring of timing events (a free-running loop reads the device time of every call afterwards), de_ctx_device, and
de_program_verify with another device current."""
import ctypes as C

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


@pytest.mark.parametrize("n_trees,world", [(1, 1), (7, 2), (1000, 8), (1001, 8), (10000, 8), (5, 8), (64, 3)])
def test_flag_exchange_reordering_is_the_identity_on_global_order(api, n_trees, world):
    """rank r owns trees r, r + world, ...: pack (pad with 1) -> [all_gather] -> unpack must reproduce the global flag array"""
    lib, ctx = api.library(), api.Context(0)
    rng = np.random.default_rng(n_trees * 31 + world)
    flags = (rng.random(n_trees) < 0.6).astype(np.uint8)
    out = np.full(n_trees, 7, dtype=np.uint8)
    ms = C.c_float(-1.0)
    ctx.check(lib.de_dist_reorder_selftest(ctx._h, flags.ctypes.data, n_trees, world, out.ctypes.data, C.byref(ms)))
    np.testing.assert_array_equal(out, flags)
    assert 0.0 <= ms.value < 5.0, ms.value  # two tiny launches
    assert lib.de_ctx_device(ctx._h) == 0
    ctx.close()


def test_world_size_one_gather_through_the_c_abi_and_its_time(api):
    import torch
    from dynamicexpressions_jl_amd import dist as dedist
    ctx = api.Context(0)
    comm = dedist.Comm(ctx, 0, 1, b"")
    ok = torch.tensor([1, 0, 1, 1, 0], device="cuda", dtype=torch.uint8)
    got = comm.gather_flags(ok, 5)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ok.cpu())
    comm.close()
    ctx.close()


def test_timing_ring_gives_every_call_of_a_free_running_loop(api):
    import torch
    lib, ctx = api.library(), api.Context(0)
    trees = de.synth.random_population(64, seed=0xDE02)
    pop = api.Population(trees, de.synth.BENCH_OPERATORS, np.float32, n_features=5, ctx=ctx)
    N = 50_000
    X = torch.from_numpy(np.ascontiguousarray(de.synth.random_X(5, N, seed=3).T)).cuda().t()
    out = torch.empty((64, N), device="cuda", dtype=torch.float32)
    ok = torch.empty(64, device="cuda", dtype=torch.uint8)

    def step():
        ctx.check(lib.de_eval(ctx._h, pop._h, X.data_ptr(), N, 5, None, out.data_ptr(), N, ok.data_ptr()))
    step()
    one = ctx.last_kernel_ms()
    ctx.timing_ring(8)
    for _ in range(5):
        step()
    ms = ctx.timing_read()
    assert len(ms) == 5 and all(0.0 < m < 50 * max(one, 0.01) for m in ms), (ms, one)
    assert ctx.timing_read() == []            # the ring restarts
    for _ in range(11):                        # more calls than slots: the last 8
        step()
    assert abs(ctx.last_kernel_ms() - 0) >= 0  # still answers (the most recent pair)
    assert len(ctx.timing_read()) == 8
    ctx.timing_ring(0)
    step()
    assert ctx.last_kernel_ms() > 0.0
    pop.close()
    ctx.close()


def test_program_verify_runs_on_the_programs_own_device(api):
    """(one-GPU box: the device is the same, the call path — hipSetDevice first — is what runs)"""
    ctx = api.Context(0)
    pop = api.Population(de.synth.random_population(40, seed=5), de.synth.BENCH_OPERATORS, np.float32, n_features=5, ctx=ctx)
    pop.verify()
    pop.close()
    ctx.close()
