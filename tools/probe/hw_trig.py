#!/usr/bin/env python3
"""Accuracy of v_cos_f32 / v_sin_f32 (gfx950, argument in revolutions) against float64 (run through gpurun)."""
import ctypes, json, os, subprocess, sys
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libhw_trig.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "hw_trig.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "hw_trig.hip")], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa
lib = ctypes.CDLL(so)
lib.hw_trig.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(1)
res = {}
for name, lo, hi in (("[-0.5, 0.5] rev", -0.5, 0.5), ("[-8, 8] rev", -8, 8), ("[-256, 256] rev", -256, 256)):
    x = rng.uniform(lo, hi, 2_000_000).astype(np.float32)
    c = np.zeros_like(x); s = np.zeros_like(x)
    assert lib.hw_trig(x.ctypes.data, c.ctypes.data, s.ctypes.data, x.size) == 0
    rc = np.cos(2 * np.pi * x.astype(np.float64)); rs = np.sin(2 * np.pi * x.astype(np.float64))
    ec, es = np.abs(c - rc), np.abs(s - rs)
    relc = ec / np.maximum(np.abs(rc), 1e-30)
    res[name] = dict(cos_max_abs=float(ec.max()), sin_max_abs=float(es.max()), cos_rel_p999=float(np.quantile(relc, 0.999)),
                     cos_max_rel_where_gt_0p1=float(relc[np.abs(rc) > 0.1].max()))
    print(name, res[name])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/hw_trig.json", "w"), indent=1)
