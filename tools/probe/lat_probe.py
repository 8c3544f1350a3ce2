#!/usr/bin/env python3
"""Unloaded latencies behind one dispatch of the threaded interpreter on MI355X (gpurun) -> gpurun_out/lat_probe.json.
See lat_probe.hip for the five modes; one wave on an idle chip, and the same with 8 waves per SIMD on every CU (blocks = 256 * 32)."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "liblat_probe.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "lat_probe.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-shared", "-fPIC", "-o", so, os.path.join(here, "lat_probe.hip")], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa: E402,F401
lib = ctypes.CDLL(so)
names = {0: "s_load_dwordx4, scalar-cache hit", 1: "s_load_dwordx4, scalar-cache miss (L2)", 2: "ds_read_b128", 3: "s_setpc_b64 (I-cache hit)",
         4: "s_setpc_b64 + s_load_dwordx4 waited for by the next block"}
per_iter = {0: 8, 1: 8, 2: 8, 3: 32, 4: 32}
res = {}
for blocks in (1, 256 * 32):
    for mode in range(5):
        iters = 2000 if mode < 3 else 500
        r = (ctypes.c_uint64 * 3)()
        ms = ctypes.c_float(0)
        rc = lib.lat_probe_run(mode, iters, blocks, r, ctypes.byref(ms))
        assert rc == 0, rc
        n = iters * per_iter[mode]
        ticks, real = r[0], r[1]
        ns_real = real * 10.0 / n  # s_memrealtime: 100 MHz
        res[f"{names[mode]} | blocks={blocks}"] = dict(memtime_ticks_per_op=ticks / n, ns_per_op=ns_real, memtime_ticks_per_ns=ticks / max(real * 10.0, 1), kernel_ms=ms.value)
        print(f"blocks={blocks:5d} {names[mode]:60s} {ns_real:8.1f} ns/op  ({ns_real * 2.1:6.0f} cycles at 2.1 GHz; s_memtime {ticks / n:8.1f} ticks/op)", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/lat_probe.json", "w"), indent=1)
