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


def _flags_of_oracle(trees, ops, X, dtype):
    from oracle import oracle
    el, sm = [], []
    for t in trees:
        tape, consts = de.flatten(t, ops, dtype)
        el.append(oracle.eval_tree_array(tape, consts, X, 7, elementwise=True)[1])
        sm.append(oracle.eval_tree_array(tape, consts, X, 7, elementwise=False)[1])
    return np.array(el), np.array(sm)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sum_certificate_closes_the_isfinite_sum_quirk(api, dtype):
    """src/ValueInterface.jl:9: the reference tests isfinite(sum(x)), the kernels every element.  de_eval_sum_certificate says for which
    trees the two PROVABLY agree; on the others (finite elements, overflowing sum — constructed here) the caller re-derives the flag.
    The oracle has both flavours."""
    ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*"), unary_operators=("cos", "exp"))
    N = 4096
    big = 3e34 if dtype == np.float32 else 1e305   # N * big overflows, big does not
    x1, x2 = de.Node(feature=1), de.Node(feature=2)
    mul, div, add = ops.index("*", 2), ops.index("/", 2), ops.index("+", 2)
    cos = ops.index("cos", 1)
    quirk_inner = de.Node(div, de.Node(mul, de.Node(add, x1, de.Node(val=2.5)), de.Node(val=big)), de.Node(val=big))  # ((x1 + 2.5) * big) / big: the inner product's SUM overflows
    quirk_const = de.Node(mul, de.Node(cos, x1), de.Node(div, de.Node(val=1.0), de.Node(add, x2, de.Node(val=big))))  # a constant leaf array of N copies of `big` is summed
    plain = de.synth.random_population(60, seed=0xC0DE, dtype=dtype)
    trees = [quirk_inner, quirk_const] + plain
    g = np.random.Generator(np.random.PCG64(3))
    X = np.asfortranarray(np.abs(g.standard_normal((5, N))).astype(dtype) + dtype(0.5))
    el, sm = _flags_of_oracle(trees, ops, X, dtype)
    assert el[0] and not sm[0], "the constructed tree must show the quirk in the oracle itself"
    pop = api.Population(trees, ops, dtype, n_features=5)
    _, ok_eval = pop.eval(X)
    ok, cert, mx = pop.sum_certificate(X)
    assert np.array_equal(ok, np.asarray(ok_eval, dtype=bool)), "the certificate pass computes the flags de_eval computes"
    assert np.array_equal(ok, el), "element-wise flags = the oracle's element-wise flavour"
    assert not cert[0] and mx[0] >= big * 2.9, (cert[0], mx[0])       # not certified: the sum overflows although every element is finite
    # soundness: wherever the certificate is given, the reference's flag (sum flavour) IS the device's
    assert np.array_equal(ok[cert], sm[cert]), "certified trees must carry the reference's own flag"
    # usefulness: ordinary trees are certified (all of the incomplete ones, and the complete ones far from floatmax / N)
    assert cert[2:].mean() > 0.9, cert[2:].mean()
    # a program without early_exit sums nothing
    pop2 = api.Population(trees, ops, dtype, n_features=5, eval_context=api.EvalContext(early_exit=False))
    _, c2, _ = pop2.sum_certificate(X)
    assert c2.all()
    pop.close(); pop2.close()


@pytest.mark.parametrize("scale", [1.0, 1e18, 1e33])
def test_sum_certificate_is_sound_on_random_trees_near_the_overflow(api, scale):
    """Soundness on data that DOES come near floatmax / N: wherever the certificate is given, the oracle's sum flavour equals the device flag
    (300 random trees over X scaled into the 1e18 / 1e33 range, Float32)."""
    dtype = np.float32
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(300, seed=0x5EED, dtype=dtype, node_count=12)
    g = np.random.Generator(np.random.PCG64(int(np.log10(scale)) + 11))
    X = np.asfortranarray((g.standard_normal((5, 3000)) * scale).astype(dtype))
    el, sm = _flags_of_oracle(trees, ops, X, dtype)
    pop = api.Population(trees, ops, dtype, n_features=5)
    ok, cert, mx = pop.sum_certificate(X)
    pop.close()
    assert np.array_equal(ok, el)
    assert np.array_equal(ok[cert], sm[cert])
    print(f"[sum certificate, X scale {scale:g}] {int(cert.sum())} of {len(trees)} trees certified, {int((el != sm).sum())} show the quirk in the oracle, "
          f"{int(((el != sm) & cert).sum())} of those certified (must be 0)")
    assert not ((el != sm) & cert).any()


def test_forward_duals_are_the_default_and_reverse_accumulation_is_an_opt_in(api):
    """ABI 3: a population wide enough for reverse accumulation (>= 8 gradient rows per tree) runs FORWARD duals by
    default — the reference's flag semantics (src/EvaluateDerivative.jl:230-243,340-365) — bit for bit what DE_LOSS_GRAD_REVERSE=0 and
    round 5's DE_OPT_FORWARD_GRAD give; EvalContext(reverse_grad=True) (DE_OPT_REVERSE_GRAD) is the permission to use reverse accumulation,
    bit for bit what DE_LOSS_GRAD_REVERSE=1 gives (it associates the products leaf-wards and may flip `ok` where a product
    chain overflows in one association only); DE_OPT_FORWARD_GRAD wins over DE_OPT_REVERSE_GRAD."""
    import os
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(60, seed=0xF0, node_count=45, max_depth=40)  # ~11 constants per tree: reverse where it is allowed
    g = np.random.Generator(np.random.PCG64(9))
    X = np.asfortranarray(g.standard_normal((5, 5000)).astype(np.float32))
    y = g.standard_normal(5000).astype(np.float32)

    def run(ec=None, env=None):
        if env is not None:
            os.environ["DE_LOSS_GRAD_REVERSE"] = env
        try:
            pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=ec)
            l, d, ok = pop.eval_loss_grad(X, y)
            name = pop.ctx.last_kernel_name()
            pop.close()
        finally:
            os.environ.pop("DE_LOSS_GRAD_REVERSE", None)
        return np.asarray(l), [np.asarray(a) for a in d], np.asarray(ok), name

    def same(a, b):
        assert np.array_equal(a[2], b[2])
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        for u, v in zip(a[1], b[1]):
            assert np.array_equal(u.view(np.uint32), v.view(np.uint32))

    dflt = run()
    assert dflt[3] != "de_rev_threaded_kernel", dflt[3]
    same(dflt, run(env="0"))
    same(dflt, run(api.EvalContext(forward_grad=True)))
    rev = run(api.EvalContext(reverse_grad=True))
    assert rev[3] == "de_rev_threaded_kernel", rev[3]
    same(rev, run(env="1"))
    both = run(api.EvalContext(forward_grad=True, reverse_grad=True))
    assert both[3] != "de_rev_threaded_kernel"
    same(dflt, both)
    # the two agree to rounding where both are complete (the tolerance tests of tests/test_gpu_loss.py bound them properly)
    okb = dflt[2].astype(bool) & rev[2].astype(bool)
    assert okb.sum() > 10
    assert np.allclose(dflt[0][okb], rev[0][okb], rtol=1e-4)


def test_tail_split_changes_nothing_but_the_order(api, monkeypatch):
    """The last chunk of a small launch runs as four short sub-chunks (csrc/de_kernels.hip `tail_split`): which workgroup evaluates which
    trees — rows and flags are bit for bit those of the unsplit launch, with and without the early exit, on a ragged last tile."""
    import torch
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(200, seed=0x7A11)
    N = 40_003
    X = torch.from_numpy(np.ascontiguousarray(de.synth.random_X(5, N, seed=5).T)).cuda().t()
    res = {}
    for split in ("1", "4", "7"):
        monkeypatch.setenv("DE_TAIL_SPLIT", split)
        for full in (False, True):
            pop = api.Population(trees, ops, np.float32, n_features=5, eval_context=api.EvalContext(full_eval=full))
            out, ok = pop.eval(X)
            torch.cuda.synchronize()
            res[(split, full)] = (out.cpu().numpy(), ok.cpu().numpy())
            pop.close()
    ref_o, ref_k = res[("1", True)]
    assert ref_k.any() and not ref_k.all()
    for key, (o, k) in res.items():
        assert np.array_equal(k, ref_k), key
        assert np.array_equal(o[ref_k].view(np.uint32), ref_o[ref_k].view(np.uint32)), key


def test_sum_certificate_on_a_parametric_population_and_device_inputs(api):
    """The certificate pass takes what de_eval takes: a ParametricExpression population (parameters staged as rows), X on the device, a ragged
    sample count; its flags are de_eval's."""
    import torch
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(80, seed=0xDE05, node_type=de.ParametricNode, nparams=8, dtype=np.float32)
    g = np.random.Generator(np.random.PCG64(21))
    N, C = 3001, 16
    X = torch.from_numpy(np.ascontiguousarray(g.standard_normal((N, 5)).astype(np.float32))).cuda().t()
    params = torch.from_numpy(np.ascontiguousarray(g.standard_normal((C, 8)).astype(np.float32))).cuda().t()
    classes = torch.from_numpy(g.integers(1, C + 1, N).astype(np.int32)).cuda()
    pop = api.Population(trees, ops, np.float32, n_features=5, n_params=8)
    _, ok_eval = pop.eval(X, params, classes)
    ok, cert, mx = pop.sum_certificate(X, params, classes)
    torch.cuda.synchronize()
    assert np.array_equal(ok, ok_eval.cpu().numpy().astype(bool))
    assert cert[~ok].all() and cert.mean() > 0.9 and np.isfinite(mx[ok]).all()
    pop.close()


def test_a_share_definition_nobody_references_is_a_bad_tape(api):
    """a DE_OP_SHARE whose subtree no DE_LEAF_SHARED names would leave a persistent row without a reader (the reverse sweep's
    POPADD would add a row nothing wrote): the lowering refuses the tape; the same tape WITH its reference is accepted."""
    import ctypes as C
    from dynamicexpressions_jl_amd.node import TAPE_DTYPE, LEAF_FEATURE, LEAF_SHARED, OP_SHARE
    lib, ctx = api.library(), api.Context(0)
    cos = lib.de_opcode_by_name(b"cos", 1)
    add = lib.de_opcode_by_name(b"+", 2)
    expanded = np.array([(0, LEAF_FEATURE, 0), (1, cos, 0), (0, LEAF_FEATURE, 0), (1, cos, 0), (2, add, 0)], dtype=TAPE_DTYPE)  # cos(x1) + cos(x1)
    good = np.array([(0, LEAF_FEATURE, 0), (1, cos, 0), (1, OP_SHARE, 0), (0, LEAF_SHARED, 0), (2, add, 0)], dtype=TAPE_DTYPE)
    bad = np.array([(0, LEAF_FEATURE, 0), (1, cos, 0), (1, OP_SHARE, 0), (0, LEAF_FEATURE, 0), (1, cos, 0), (2, add, 0)], dtype=TAPE_DTYPE)  # defined, never read
    noff = np.array([0, len(expanded)], dtype=np.int64)
    coff = np.zeros(2, dtype=np.int64)
    for cse, want_ok in ((good, True), (bad, False)):
        cse_off = np.array([0, len(cse)], dtype=np.int64)
        h = C.c_void_p()
        rc = lib.de_program_create_cse(ctx._h, 0, expanded.ctypes.data, noff.ctypes.data, cse.ctypes.data, cse_off.ctypes.data, 1, None, coff.ctypes.data,
                                       1, 0, 7, C.byref(h))
        assert (rc == 0) == want_ok, (rc, lib.de_last_error(ctx._h))
        if rc == 0:
            lib.de_program_destroy(h)
        else:
            assert b"DE_OP_SHARE" in lib.de_last_error(ctx._h)
    ctx.close()


def test_contexts_on_two_devices_with_the_other_device_current(api):
    """ABI version 2 (d): one process may hold contexts on several GPUs; handler addresses belong to ONE device's copy of the code object.
    Needs two devices (the round's boxes have one: skipped there) — verify, eval and gradient of a program on device d with device 1 - d
    current."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one device visible")
    ops = de.synth.BENCH_OPERATORS
    trees = de.synth.random_population(40, seed=11)
    Xh = de.synth.random_X(5, 3000, seed=2)
    ref = None
    for d in (0, 1):
        ctx = api.Context(d)
        pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
        torch.cuda.set_device(1 - d)
        pop.verify()
        out, ok = pop.eval(Xh)
        _, grads, okg = pop.eval_grad(Xh, variable=True)
        res = (np.asarray(out), np.asarray(ok), [np.asarray(g) for g in grads])
        if ref is None:
            ref = res
        else:
            assert np.array_equal(res[1], ref[1])
            assert np.array_equal(res[0][ref[1]].view(np.uint32), ref[0][ref[1]].view(np.uint32))
        pop.close()
        ctx.close()
    torch.cuda.set_device(0)


@pytest.mark.parametrize("kind", ["plain", "parametric", "graph"])
def test_program_creation_on_host_threads_builds_what_one_thread_builds(api, monkeypatch, kind):
    """de_program_create runs every per-tree pass (lowering, merge, bind, superinstructions, record chaining, constant sites) on a pool of
    host threads: the streams must be those of DE_HOST_THREADS=1, byte for byte (de_program_stream_hash covers every host-side stream and
    the auxiliary program of constant subtrees), and evaluate to the same bits."""
    from dynamicexpressions_jl_amd.node import GraphNode
    n = 3000  # 11 ranges of the pool (>= 256 trees per thread)
    if kind == "parametric":
        ops = de.synth.BENCH_OPERATORS
        trees = de.synth.random_population(n, seed=0xDE05, node_type=de.ParametricNode, nparams=8)
        kw = dict(n_features=5, n_params=8)
    elif kind == "graph":
        ops = de.synth.BENCH_OPERATORS
        base = de.synth.random_population(n, seed=77, node_type=GraphNode)
        trees = []
        for i, t in enumerate(base):  # a shared subtree in every other tree: (t) op (t) with ONE object on both sides
            trees.append(GraphNode(1 + i % 4, t, t) if i % 2 else t)
        kw = dict(n_features=5)
    else:
        ops = de.OperatorEnum(binary_operators=("+", "-", "/", "*", "max"), unary_operators=("cos", "exp", "sin", "safe_log", "abs", "tanh"))
        rng = de.synth.Xoshiro256ss(5)
        trees = [de.synth.gen_random_tree_fixed_size(1 + i % 30, ops, 5, rng, np.float32) for i in range(n)]
        kw = dict(n_features=5)
    X = de.synth.random_X(5, 700, seed=3)
    rng = np.random.default_rng(1)
    params = rng.standard_normal((8, 4)).astype(np.float32) if kind == "parametric" else None
    classes = rng.integers(1, 5, size=700).astype(np.int32) if kind == "parametric" else None
    res = {}
    for threads in ("1", None, "5"):
        if threads is None:
            monkeypatch.delenv("DE_HOST_THREADS", raising=False)
        else:
            monkeypatch.setenv("DE_HOST_THREADS", threads)
        pop = api.Population(trees, ops, np.float32, **kw)
        pop.verify()
        out, ok = pop.eval(X, params, classes) if kind == "parametric" else pop.eval(X)
        res[threads] = (pop.stream_hash(), np.asarray(out).copy(), np.asarray(ok).copy())
        pop.close()
    h1, o1, k1 = res["1"]
    assert h1 != 0
    for threads in (None, "5"):
        h, o, k = res[threads]
        assert h == h1, f"DE_HOST_THREADS={threads}: the streams differ from the serial build's"
        assert np.array_equal(k, k1) and np.array_equal(o[k1].view(np.uint32), o1[k1].view(np.uint32))


def test_recycled_program_buffers_leave_no_trace(api):
    """A context parks the device streams and the host vectors of destroyed programs for the next creation (a search loop creates and
    destroys a program per generation: de_program_destroy of 10^4 trees was 1.9 ms of munmap / hipFree).  A program built from parked
    buffers must be the program a fresh context builds: same host streams (de_program_stream_hash), same rows, flags and gradients —
    whatever was destroyed before it (larger, smaller, another element type, a parametric or a gradient-using program)."""
    ops = de.synth.BENCH_OPERATORS
    X = de.synth.random_X(5, 900, seed=4)
    X64 = X.astype(np.float64)
    y = np.cos(X[0]).astype(np.float32)

    def run(ctx, trees, dtype, grad):
        pop = api.Population(trees, ops, dtype, n_features=5, ctx=ctx)
        Xd = X64 if dtype == np.float64 else X
        out, ok = pop.eval(Xd)
        res = [pop.stream_hash(), np.asarray(out).copy(), np.asarray(ok).copy()]
        if grad:
            lo, dl, okg = pop.eval_loss_grad(Xd, y.astype(dtype))
            res += [np.asarray(lo).copy(), np.concatenate([np.asarray(d).ravel() for d in dl]), np.asarray(okg).copy()]
        pop.close()
        return res

    jobs = [(de.synth.random_population(2500, seed=21), np.float32, True), (de.synth.random_population(40, seed=22), np.float64, False),
            (de.synth.random_population(600, seed=23), np.float32, True), (de.synth.random_population(2500, seed=24), np.float32, False),
            (de.synth.random_population(1, seed=25), np.float32, True)]
    shared = api.Context(0)
    for trees, dtype, grad in jobs:
        got = run(shared, trees, dtype, grad)
        fresh = api.Context(0)
        want = run(fresh, trees, dtype, grad)
        fresh.close()
        assert got[0] == want[0], "host streams differ from a fresh context's"
        ui = np.uint32 if dtype == np.float32 else np.uint64
        assert np.array_equal(got[2], want[2])
        assert np.array_equal(got[1][want[2]].view(ui), want[1][want[2]].view(ui))
        if grad:
            assert np.array_equal(got[5], want[5])
            assert np.array_equal(got[3][want[5]].view(ui), want[3][want[5]].view(ui))
            assert np.array_equal(got[4].view(ui), want[4].view(ui))
    shared.close()


def test_two_host_threads_create_programs_at_once(api):
    """One context per calling thread (SURVEY §8b threading): two host threads create, evaluate and destroy populations at the same time —
    one of them owns the pool of host threads, the other runs its per-tree passes inline in the same partition.  Every program must be the
    one a quiet process builds (stream hash) and evaluate to the same bits."""
    import threading
    ops = de.synth.BENCH_OPERATORS
    X = de.synth.random_X(5, 600, seed=9)
    pops = [de.synth.random_population(1500 + 100 * k, seed=40 + k) for k in range(4)]
    quiet = []
    ctx0 = api.Context(0)
    for trees in pops:
        p = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx0)
        out, ok = p.eval(X)
        quiet.append((p.stream_hash(), np.asarray(out).copy(), np.asarray(ok).copy()))
        p.close()
    ctx0.close()
    errors = []

    def worker(order):
        try:
            ctx = api.Context(0)
            for rep in range(3):
                for k in order:
                    p = api.Population(pops[k], ops, np.float32, n_features=5, ctx=ctx)
                    out, ok = p.eval(X)
                    h, o, q = quiet[k]
                    if p.stream_hash() != h or not np.array_equal(np.asarray(ok), q) or \
                            not np.array_equal(np.asarray(out)[q].view(np.uint32), o[q].view(np.uint32)):
                        errors.append((k, rep))
                    p.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=worker, args=(o,)) for o in ([0, 1, 2, 3], [3, 2, 1, 0], [1, 3, 0, 2])]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_generations_do_not_leak(api):
    """300 generations (create, evaluate, loss gradient every third, destroy) of populations of changing size on ONE context: the device
    memory in use and the process's resident set settle (recycled buffers are bounded: 256 MB of device streams, 512 MB of host vectors
    per context) instead of growing with the generations."""
    import resource
    import torch
    ops = de.synth.BENCH_OPERATORS
    X = de.synth.random_X(5, 256, seed=4)
    y = np.cos(X[1]).astype(np.float32)
    pool = de.synth.random_population(6000, seed=91)
    ctx = api.Context(0)
    rng = np.random.default_rng(3)

    def generation(g):
        n = int(rng.integers(50, 3000))
        start = int(rng.integers(0, len(pool) - n))
        pop = api.Population(pool[start:start + n], ops, np.float32, n_features=5, ctx=ctx)
        pop.eval(X)
        if g % 3 == 0:
            pop.eval_loss_grad(X, y)
        pop.close()

    def used():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return total - free, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024

    for g in range(60):  # warm: every buffer class has been allocated, parked and re-used
        generation(g)
    dev0, rss0 = used()
    for g in range(60, 300):
        generation(g)
    dev1, rss1 = used()
    ctx.close()
    assert dev1 - dev0 < 96 << 20, f"device memory grew by {(dev1 - dev0) >> 20} MiB over 240 generations"
    assert rss1 - rss0 < 192 << 20, f"resident set grew by {(rss1 - rss0) >> 20} MiB over 240 generations"
