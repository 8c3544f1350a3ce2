"""The constant-optimisation inner loop: set_consts + fused loss & gradient, 10^4 trees x 10^3 rows."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api
ops = de.synth.BENCH_OPERATORS
ctx = api.Context(0)
lib = api.library()
for nt, N in ((10000, 1000), (1000, 10000)):
    trees = de.synth.random_population(nt, seed=0xDE02)
    pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn((N, 5), generator=g, device="cuda").t()
    y = torch.randn(N, generator=g, device="cuda")
    consts = np.concatenate([de.flatten(t, ops, np.float32)[1] for t in trees]).astype(np.float32)
    t_set, t_lg, k_lg = [], [], []
    for it in range(6):
        consts = consts * np.float32(1.0001)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pop.set_constants(consts)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss, dl, ok = pop.eval_loss_grad(X, y)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        t_set.append(t1 - t0); t_lg.append(t2 - t1); k_lg.append(ctx.last_kernel_ms())
    print(f"n_trees {nt} N {N}: set_consts {1e3 * np.median(t_set[2:]):.2f} ms, loss_grad wall {1e3 * np.median(t_lg[2:]):.2f} ms (kernel {np.median(k_lg[2:]):.3f} ms)")
    pop.close()

# the C calls alone (what a Julia caller pays)
import ctypes as C
nt, N = 10000, 1000
trees = de.synth.random_population(nt, seed=0xDE02)
pop = api.Population(trees, ops, np.float32, n_features=5, ctx=ctx)
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn((N, 5), generator=g, device="cuda").t(); y = torch.randn(N, generator=g, device="cuda")
ng = pop._n_grad_all(1); tot = int(ng.sum())
loss = torch.empty(nt, device="cuda"); dl = torch.empty(tot, device="cuda"); ok = torch.empty(nt, device="cuda", dtype=torch.uint8)
consts = np.concatenate([de.flatten(t, ops, np.float32)[1] for t in trees]).astype(np.float32)
ts, tl = [], []
for it in range(8):
    consts = consts * np.float32(1.0001)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.de_program_set_consts(pop._h, consts.ctypes.data)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rc |= lib.de_eval_loss_grad(ctx._h, pop._h, X.data_ptr(), N, 5, None, 1, y.data_ptr(), None, 0, loss.data_ptr(), dl.data_ptr(), None, ok.data_ptr())
    torch.cuda.synchronize(); t2 = time.perf_counter()
    assert rc == 0
    ts.append(t1 - t0); tl.append(t2 - t1)
print(f"C ABI only, {nt} trees x {N}: set_consts {1e3 * np.median(ts[2:]):.2f} ms, loss_grad {1e3 * np.median(tl[2:]):.2f} ms (kernel {ctx.last_kernel_ms():.3f} ms)")
