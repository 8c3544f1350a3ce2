#!/usr/bin/env python3
"""Where does the output store's cost come from (VALU replay V3 -> V4)?  The same replay with every row at the SAME address (row stride 0:
each workgroup rewrites its own 1 KB piece, 320 MB in all instead of 40 GB — the stores are issued, the data does not go to HBM)."""
import ctypes as C, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import numpy as np, torch
rl = C.CDLL(os.path.join(HERE, "libvalu_replay.so"))
rl.valu_replay_run.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
res = json.load(open(os.path.join(ROOT, "gpurun_out", "valu_replay", "counts_div.json")))
c_div = np.array(res["counts"], dtype=np.uint32)
cd = torch.from_numpy(c_div.view(np.int32)).cuda()
blocks, trees_div, total_div, scale = res["blocks"], res["trees_div"], res["total_div"], res["scale"]
stride = (blocks * 1024 + 4095) // 4096 * 4096
buf = torch.empty(trees_div * stride + 4096, device="cuda", dtype=torch.uint8)
ms = C.c_float(0)
for lds in (7168, 8192):
    for name, v, st in (("V3 (no store)", 3, stride), ("V4, rows 320 MB apart (40 GB written)", 4, stride), ("V4, all rows at one address (320 MB written)", 4, 0)):
        best = min((rl.valu_replay_run(v, cd.data_ptr(), 1, trees_div, total_div, buf.data_ptr(), st, blocks, lds, C.byref(ms)), ms.value)[1] for _ in range(3))
        print(f"LDS {lds}: {name:50s} {best * scale:7.3f} ms", flush=True)
