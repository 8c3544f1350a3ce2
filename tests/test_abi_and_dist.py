"""CPU tests: the C-ABI library loads without a GPU and exports every declared symbol; the
Python opcode table equals the C table; multi-process sharding/gather works over gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import api, dist as dedist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = api.library()
    header = open(os.path.join(ROOT, "include", "de_hip.h")).read()
    declared = set(re.findall(r"\b(de_[A-Za-z0-9_]+)\s*\(", header))
    declared -= {"de_status_t", "de_dtype_t"}
    assert declared, "no declarations found"
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libde_hip.so does not export {sym}"
    assert set(api.EXPORTS) <= declared
    assert lib.de_abi_version() == 3 and lib.de_opcode_table_version() == 1


def test_python_opcode_table_matches_c_table():
    lib = api.library()
    for name, degree, code in de.operators.all_opcodes():
        assert lib.de_opcode_by_name(name.encode(), degree) == code, (name, degree)
        assert lib.de_opcode_degree(code) == degree
    assert lib.de_opcode_by_name(b"my_custom_op", 2) == -1
    assert lib.de_opcode_by_name(b"cos", 2) == -1
    assert lib.de_opcode_name(20) == b"cos"


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(api.DeviceError):
        api.Context(0)
    ops = de.synth.BENCH_OPERATORS
    with pytest.raises(api.DeviceError):
        api.eval_tree_array(de.Node(feature=1), np.zeros((1, 4), np.float32), ops)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "dynamicexpressions.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".jl", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert "libde_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_shard_indices_partition():
    for n, w in ((10, 1), (10, 3), (1000, 8), (5, 8), (0, 4)):
        parts = [dedist.shard_indices(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert [len(p) for p in parts] == [dedist.shard_size(n, r, w) for r in range(w)]
        assert list(dedist.unshard_order(n, w)) == sum(parts, [])


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import dynamicexpressions_jl_amd as de
from dynamicexpressions_jl_amd import dist as dedist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n_trees = 11
trees = de.synth.random_population(n_trees, seed=5)
mine = dedist.shard_indices(n_trees, rank, world)
assert [id(t) for t in dedist.scatter_population(trees, rank, world)] == [id(trees[i]) for i in mine]
# stand-in for the device flags of this shard: tree i is "complete" iff i % 3 != 0
local_ok = torch.tensor([1 if i % 3 else 0 for i in mine], dtype=torch.uint8)
allf = dedist.gather_flags(local_ok, n_trees, rank, world)
assert allf.tolist() == [1 if i % 3 else 0 for i in range(n_trees)], allf.tolist()
# max-over-ranks timing reduction used by bench.py
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == world
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_world_size_2_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script), ROOT],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
