#!/usr/bin/env python3
"""Scalar-ALU issue rate per SIMD on MI355X, alone and beside vector work (gpurun) -> gpurun_out/issue_probe.json (see issue_probe.hip)."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libissue_probe.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "issue_probe.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "issue_probe.hip")], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa: E402,F401
lib = ctypes.CDLL(so)
names = {0: "s_mov_b32 literal", 1: "v_pk_add_f32 (8 chains)", 2: "s_mov_b32 / v_pk_add_f32 alternating in one wave", 3: "even waves s_mov_b32, odd waves v_pk_add_f32",
         4: "s_mov_b64", 5: "s_add_u32 + s_addc_u32"}
CLK = 2.1e9  # sustained engine clock under load (profiles/r2_clock_probe.json); a lone wave runs at 2.4
res = {}
ms = ctypes.c_float(0)
iters = 4000
for wps, threads, blocks in ((1, 64, 256), (1, 256, 256), (2, 256, 512), (4, 256, 1024), (8, 256, 2048)):
    for mode in range(6):
        if threads == 64 and mode == 3:
            continue
        assert lib.issue_probe_run(mode, iters, blocks, threads, ctypes.byref(ms)) == 0
        # instructions issued per SIMD: waves on the SIMD x iters x 64 (threads == 64: ONE wave per CU, on one of its SIMDs)
        cyc = ms.value * 1e-3 * CLK / (wps * iters * 64)
        label = f"{names[mode]} | {'one wave per CU' if threads == 64 else str(wps) + ' wave(s) per SIMD, every SIMD'}"
        res[label] = dict(ms=ms.value, cycles_per_instruction_per_simd=cyc)
        print(f"{label:90s} {ms.value:8.3f} ms  {cyc:6.2f} cycles per instruction and SIMD (at 2.1 GHz)", flush=True)
for k in (1, 2, 4):  # k workgroups of 512 threads per CU: k SALU waves + k VALU waves on every SIMD
    assert lib.issue_probe_run(6, iters, 256 * k, 512, ctypes.byref(ms)) == 0
    cyc = ms.value * 1e-3 * CLK / (k * iters * 64)
    label = f"{k} s_mov_b32 wave(s) + {k} v_pk_add_f32 wave(s) per SIMD"
    res[label] = dict(ms=ms.value, cycles_per_instruction_of_each_kind_per_simd=cyc)
    print(f"{label:90s} {ms.value:8.3f} ms  {cyc:6.2f} cycles per instruction of EACH kind and SIMD", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/issue_probe.json", "w"), indent=1)
