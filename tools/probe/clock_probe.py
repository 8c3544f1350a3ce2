#!/usr/bin/env python3
"""Sustained VALU issue clock of the MI355X under an all-VALU load (run through gpurun): builds tools/probe/clock_probe.hip
if needed and prints the effective engine clock for packed-FMA, packed-FMA + v_exp_f32 and plain-FMA loops."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libclock_probe.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "clock_probe.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "clock_probe.hip")], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa: E402  (one HIP runtime per process: torch's)
lib = ctypes.CDLL(so)
simds, res = 256 * 4, {}
for var, name, per_iter in ((0, "v_pk_fma_f32", 32), (1, "v_pk_fma_f32 + v_exp_f32 (1:8, exp counted as 4 slots)", 32 + 8 * 4), (2, "v_fma_f32", 32)):
    for wps in (2, 8):  # waves per SIMD
        blocks, iters = 256 * wps, 20000
        ms = ctypes.c_float(0)
        assert lib.clock_probe(var, blocks, iters, ctypes.byref(ms)) == 0
        slots = blocks * 4 * iters * per_iter
        ghz = slots * 4 / simds / (ms.value * 1e-3) / 1e9
        res[f"{name}, {wps} waves/SIMD"] = dict(ms=ms.value, effective_ghz=ghz)
        print(f"{name:60s} {wps} waves/SIMD  {ms.value:8.2f} ms  -> {ghz:.3f} GHz effective VALU issue clock")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/clock_probe.json", "w"), indent=1)
