#!/usr/bin/env python3
"""Issue cost of gfx950 VALU instructions on MI355X (run through gpurun) -> gpurun_out/valu_rate.json.
Cycles are quoted against the measured v_fma_f32 loop taken as 2 cycles per wave64 instruction (32 FP32 FMA lanes per
SIMD and cycle = the 157 TFLOP/s vector FP32 peak of the chip); the implied engine clock is printed next to it."""
import ctypes, json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libvalu_rate.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "valu_rate.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "valu_rate.hip")], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa: E402,F401
lib = ctypes.CDLL(so)
NAMES = ["v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_mul_f32", "v_mov_b32", "v_max_f32", "v_max3_f32", "v_cndmask_b32",
         "v_lshl_add_u32", "v_xor_b32", "v_cmp_lt_f32", "v_exp_f32", "v_rcp_f32", "v_rndne_f32", "v_ldexp_f32", "v_cvt_i32_f32", "v_med3_f32",
         "v_add_u32", "v_pk_add_f32", "v_fma_f64", "v_bfi_b32", "v_and_b32 (literal)", "v_mul_f32 (sgpr operand)",
         "v_pk_fma_f32 (sgpr operand)", "v_fmamk_f32 (literal)", "v_mul_f32 (inline const)", "v_fma_f32 (sgpr operand)", "v_sub_f32",
         "v_mul_f32_e64 (neg/abs modifiers)", "v_cndmask_b32_e64 (sgpr mask)", "ds_read_b128", "v_min_f32", "v_fmac_f32", "v_lshlrev_b32",
         "v_or_b32", "v_sqrt_f32", "v_log_f32", "v_mul_f64", "v_pk_mov_b32"]
simds, blocks, iters = 1024, 2048, 8000
res, base = {}, None
ms = ctypes.c_float(0)
for _ in range(3):  # clocks up
    lib.valu_rate(1, blocks, iters, ctypes.byref(ms))
for v, name in enumerate(NAMES):
    ms = ctypes.c_float(0)
    rc = lib.valu_rate(v, blocks, iters, ctypes.byref(ms))
    assert rc == 0, (name, rc)
    if base is None:
        base = ms.value
    insts_per_simd = blocks * 4 * iters * 32 / simds
    cyc = 2.0 * ms.value / base
    res[name] = dict(ms=ms.value, cycles_per_wave64_inst=cyc)
    print(f"{name:28s} {ms.value:8.2f} ms   {cyc:5.2f} cycles   (clock if v_fma_f32 = 2 cycles: {insts_per_simd * 2 / (base * 1e-3) / 1e9:.2f} GHz)")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/valu_rate.json", "w"), indent=1)
