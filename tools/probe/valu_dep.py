#!/usr/bin/env python3
"""Dependent vs independent VALU chains at several occupancies on MI355X (run through gpurun) -> gpurun_out/valu_dep.json"""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libvalu_dep.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "valu_dep.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "valu_dep.hip")], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa: E402,F401
lib = ctypes.CDLL(so)
ms = ctypes.c_float(0)
for _ in range(3):
    lib.valu_dep(1, 16, 2048, 8000, ctypes.byref(ms))
res = {}
base = None
for wps in (8, 5, 2, 1):
    for pk in (0, 1):
        for ch in (16, 4, 2, 1):
            blocks, iters = 256 * wps, 8000
            assert lib.valu_dep(pk, ch, blocks, iters, ctypes.byref(ms)) == 0
            per_inst = ms.value * 1e-3 / (wps * iters * 32)  # seconds per instruction slot of one SIMD
            if base is None:
                base = per_inst / 2.0  # v_fma_f32, 16 chains, 8 waves = 2 cycles
            cyc = per_inst / base
            res[f"{'v_pk_fma_f32' if pk else 'v_fma_f32'} chains={ch} waves/SIMD={wps}"] = cyc
            print(f"{'v_pk_fma_f32' if pk else 'v_fma_f32':13s} {ch:2d} independent chain(s) per wave, {wps} waves/SIMD: {cyc:5.2f} cycles per instruction")
json.dump(res, open("gpurun_out/valu_dep.json", "w"), indent=1)
