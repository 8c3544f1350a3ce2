#!/usr/bin/env python3
"""VALU issue slots of every handler of the threaded eval kernel, from the gfx950 ISA.

    python tools/valu_slots.py [--dtype f32] [--obj <device ELF>] [-o profiles/valu_slots.json]

The eval kernel is VALU-issue bound, so the binding ceiling of a launch is
    sum over the dispatched handlers of the SIMD cycles their VALU instructions occupy
per wavefront.  This tool disassembles the device code object build.sh links for de_kernels.hip
(csrc/_obj/irp_de_kernels/k.out), splits it per handler function and reports, per handler id of
csrc/de_bind.h, the VALU slots on the SHORTEST path from the function entry to its return
(`s_setpc_b64` — since the direct-threaded dispatch, the tail call to the next handler): handlers keep their rare cases (division outside [2^-40, 2^40], |x| > 1e5 for
cos/sin, the extremum select) behind wave-uniform branches, and the shortest path is the one a
wavefront of ordinary data takes.  Instructions are priced in SIMD cycles by the classes measured on
MI355X with tools/probe/valu_rate.py (see `weight` below): FP32 fma/add/mul, moves and bit logic run at
2 cycles per wave64 instruction, packed FP32 / min-max / compares / shifts / anything with an SGPR operand
at ~3.7, transcendentals at ~7.4.  bench.py multiplies the table with the dispatch histogram of the
population (`roofline.valu`); no GPU is needed to produce it.
"""
import argparse
import heapq
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM", "/opt/rocm/lib/llvm/bin")
# Issue cost in SIMD cycles per wave64 instruction, MEASURED on MI355X (tools/probe/valu_rate.py ->
# profiles/r2_valu_rate.json, v_fma_f32 = 2 cycles = the 32 FP32-FMA lanes per SIMD and cycle of the chip's vector peak):
#   FULL  2.0  v_fma/fmac/fmamk/fmaak/add/sub/mul_f32, v_mov_b32, v_and/or/xor_b32, v_add/sub_u32 — with VGPR, literal or
#              inline-constant operands and any modifiers;
#   HALF  3.7  every v_pk_*_f32 (two elements: NO throughput gain over two full-rate instructions), v_max/min/max3/min3/
#              med3, v_cmp_*, v_cndmask, shifts, v_lshl_add, v_bfi, v_ldexp, v_rndne, v_cvt, all Float64 arithmetic —
#              and ANY instruction that reads an SGPR operand (a full-rate one becomes half rate: 3.6);
#   TRANS 7.35 v_exp/rcp/log/sqrt/rsq/sin/cos_f32.
FULL = ("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32")
TRANS = ("v_rcp_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32",
         "v_rcp_iflag_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")
C_FULL, C_HALF, C_TRANS = 2.0, 3.7, 7.35
SGPR = re.compile(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc|exec|m0)(?![\w])")


def weight(mn: str, ops: str = "") -> float:
    """SIMD cycles the instruction occupies the VALU for (0 for scalar / memory instructions)."""
    if not mn.startswith("v_"):
        return 0.0
    if any(mn.startswith(t) for t in TRANS):
        return C_TRANS
    if any(mn.startswith(t) for t in FULL) and not SGPR.search(ops):
        return C_FULL
    return C_HALF


def functions(obj):
    """{demangled name: [(addr, mnemonic, operands, branch target or None)]}"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", obj], check=True,
                         capture_output=True, text=True).stdout
    out, cur = {}, None
    head = re.compile(r"^([0-9a-f]+) <(.*)>:$")
    ins = re.compile(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):")
    for line in txt.splitlines():
        m = head.match(line)
        if m:
            cur = out.setdefault(m.group(2), [])
            continue
        m = ins.match(line)
        if m and cur is not None:
            mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
            tgt = None
            if mn.startswith("s_cbranch") or mn == "s_branch":
                off = int(ops.split()[0])
                if off >= 32768:
                    off -= 65536
                tgt = addr + 4 + 4 * off
            cur.append((addr, mn, ops, tgt))
    return out


def shortest_slots(code):
    """(VALU cycles, VALU instructions, all instructions) on the cheapest entry -> return path."""
    idx = {a: i for i, (a, _, _, _) in enumerate(code)}
    dist = {0: (0, 0, 0)}
    heap = [(0, 0, 0, 0)]
    while heap:
        d, nv, ni, i = heapq.heappop(heap)
        if dist.get(i, (1 << 60,))[0] < d:
            continue
        _, mn, ops, tgt = code[i]
        if mn.startswith("s_setpc") or mn == "s_endpgm":
            # a jump through a pc-relative address (s_getpc_b64 + s_add_u32 into the same pair a few instructions up) is the
            # tail call into a handler's out-of-line slow twin (h_un_fast -> h_chain<b_un>, h_tree_end -> h_tree_end_slow): not
            # the path ordinary data takes
            pair = ops.split()[0].rstrip(",") if ops else ""
            if any(code[j][1] == "s_getpc_b64" and code[j][2].split()[0].rstrip(",") == pair for j in range(max(0, i - 8), i)):
                continue
            return round(d, 2), nv, ni
        w = weight(mn, ops)
        nxt = []
        if mn != "s_branch" and i + 1 < len(code):
            nxt.append(i + 1)
        if tgt is not None and tgt in idx:
            nxt.append(idx[tgt])
        for j in nxt:
            c = (d + w, nv + (1 if w else 0), ni + 1)
            if j not in dist or c[0] < dist[j][0]:
                dist[j] = c
                heapq.heappush(heap, (c[0], c[1], c[2], j))
    return None


def handler_names(ty: str, turbo: bool = False):
    """handler id -> template instantiation, restating the id layout of csrc/de_bind.h (checked against
    TOPX_COUNT through de_lower_tape_stage's ids by tests/test_lowering.py::test_valu_slot_table_layout)."""
    b = lambda v: "true" if v else "false"  # noqa: E731
    tb = b(turbo)                       # cos / exp / sin handlers
    tbk = lambda k: b(turbo and k >= 4)  # noqa: E731  binary handlers: only the divisions have a turbo instantiation
    names = {}
    names[0] = f"h_load_row<{ty}>"
    names[1] = f"h_load_const<{ty}>"
    names[2] = f"h_push<{ty}>"
    names[3] = f"h_check_row<{ty}>"
    names[4] = f"h_check_acc<{ty}>"
    BIN_BASE = 5
    for k in range(6):
        for v in range(4):
            names[BIN_BASE + 4 * k + v] = f"h_bin<{ty}, {k}, {v}, {tbk(k)}>"
    UN_BASE = BIN_BASE + 24
    for k in range(3):
        for v in range(4):
            names[UN_BASE + 4 * k + v] = f"h_un<{ty}, {k}, {v}, {tb}>"
    GEN = UN_BASE + 12
    names[GEN + 0] = f"h_gen<{ty}, 0, false>"
    names[GEN + 1] = f"h_gen<{ty}, 1, false>"
    names[GEN + 2] = f"h_gen<{ty}, 2, false>"
    names[GEN + 3] = f"h_param<{ty}, {tb}>"
    names[GEN + 4] = f"h_tern<{ty}>"
    names[GEN + 5] = f"h_gen<{ty}, 2, true>"
    names[GEN + 6] = f"h_gen<{ty}, 0, true>"
    BOP_COUNT = GEN + 7
    LOADROW = BOP_COUNT
    for p in (0, 1):
        for c in (0, 1):
            names[LOADROW + 2 * p + c] = f"h_loadrow_f<{ty}, {b(p)}, {b(c)}>" if (p or c) else f"h_load_row<{ty}>"
    LCP = LOADROW + 4
    names[LCP] = f"h_loadconst_push<{ty}>"
    UNROW = LCP + 1
    for k in range(3):
        for o in (0, 1):
            for p in (0, 1):
                for c in (0, 1):
                    names[UNROW + ((k * 2 + o) * 2 + p) * 2 + c] = f"h_unrow_f<{ty}, {k}, {b(o)}, {b(p)}, {b(c)}, {tb}>"
    BINROWC = UNROW + 24
    for k in range(6):
        for o in (0, 1):
            names[BINROWC + k * 2 + o] = f"h_binrowc<{ty}, {k}, {b(o)}, {tbk(k)}>"
    BIN2 = BINROWC + 12
    for k in range(6):
        for cst in (0, 1):
            for o in (0, 1):
                for p in (0, 1):
                    names[BIN2 + ((k * 2 + cst) * 2 + o) * 2 + p] = f"h_bin2<{ty}, {k}, {b(cst)}, {b(o)}, {b(p)}, {tbk(k)}>"
    TOP_COUNT = BIN2 + 48
    for k in range(3, 13):
        for s in (0, 1):
            names[TOP_COUNT + (k - 3) * 2 + s] = f"h_un2<{ty}, {k}, {s}>"
    XB = TOP_COUNT + 20
    for i, (k, s) in enumerate(((6, 0), (6, 1), (7, 0), (7, 1))):
        names[XB + i] = f"h_maxmin<{ty}, {k}, {s}>"
    return names, dict(BOP_COUNT=BOP_COUNT, TOP_COUNT=TOP_COUNT, TOPX_COUNT=XB + 4)


def table(obj, ty="float", turbo=False):
    fns = functions(obj)
    by_short, fast = {}, {}
    for full, code in fns.items():
        # direct-threaded handlers are h_chain<T, &body>: index them by the body's name with the historical h_ prefix
        m = re.search(r"h_chain<\w+, &de::BState<\w+> de::b_(\w+<[^(]*>)\(", full)
        if m:
            by_short["h_" + m.group(1)] = code
            continue
        m = re.search(r"de::(h_param<[^(]*>)\(", full)
        if m:
            by_short[m.group(1)] = code
            continue
        m = re.search(r"de::h_un_fast<(\d), (\d), (true|false)>\(", full)  # Float32 hot unary handlers: the fast-path-only forms are what the table points at
        if m:
            fast[f"h_un<float, {m.group(1)}, {m.group(2)}, {m.group(3)}>"] = code
            continue
        m = re.search(r"de::h_div_fast<(\d), (\d)>\(", full)  # ... and of the exact Float32 divisions
        if m:
            fast[f"h_bin<float, {m.group(1)}, {m.group(2)}, false>"] = code
            continue
        m = re.search(r"de::h_unrow_fast<(\d), (true|false), (true|false), (true|false), (true|false)>\(", full)  # ... and their fused forms
        if m:
            fast["h_unrow_f<float, %s, %s, %s, %s, %s>" % m.groups()] = code
            continue
        m = re.search(r"de::h_divrowc_fast<(\d), (true|false)>\(", full)
        if m:
            fast["h_binrowc<float, %s, %s, false>" % m.groups()] = code
            continue
        m = re.search(r"de::h_div2_fast<(\d), (true|false), (true|false), (true|false)>\(", full)
        if m:
            fast["h_bin2<float, %s, %s, %s, %s, false>" % m.groups()] = code
    if ty == "float":
        by_short.update(fast)
    names, counts = handler_names(ty, turbo)
    slots = {}
    for hid, nm in names.items():
        code = by_short.get(nm)
        if code is None:
            continue
        r = shortest_slots(code)
        if r is None:
            continue
        slots[hid] = dict(name=nm, valu_cycles=r[0], valu_insts=r[1], insts=r[2])
    return slots, counts


def static_mix(obj):
    """Static VALU instruction mix of a code object: (full-rate, half-rate, transcendental) counts over every function.  Used to
    price EXECUTED instruction counts from the hardware counters (bench.py `roofline.valu_measured`): the counters tell how many
    VALU instructions ran, not whether they were packed / SGPR-operand forms."""
    n = [0, 0, 0]
    for _, code in functions(obj).items():
        for _, mn, ops, _ in code:
            w = weight(mn, ops)
            if w == C_FULL:
                n[0] += 1
            elif w == C_HALF:
                n[1] += 1
            elif w == C_TRANS:
                n[2] += 1
    return n


def write_static_mix(out):
    """profiles/valu_static_mix.json: per code-object module of the build (csrc/_obj/irp_*/k.out)."""
    import glob
    res = {}
    for k in sorted(glob.glob(os.path.join(ROOT, "dynamicexpressions.jl_amd", "csrc", "_obj", "irp_*", "k.out"))):
        mod = os.path.basename(os.path.dirname(k))[4:]
        full, half, trans = static_mix(k)
        res[mod] = dict(full_rate=full, half_rate=half, transcendental=trans, half_share_of_non_trans=half / max(full + half, 1))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_profiles import kernel_source_hash
    doc = dict(kernel_source_hash=kernel_source_hash(), cycles=dict(full=C_FULL, half=C_HALF, trans=C_TRANS),
               source="tools/valu_slots.py --static-mix: llvm-objdump of every device code object of the build, instruction classes of "
                      "profiles/r2_valu_rate.json", modules=res)
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(f"{len(res)} modules -> {out}")


def main():
    if "--static-mix" in sys.argv:
        return write_static_mix(os.path.join(ROOT, "profiles", "valu_static_mix.json"))
    ap = argparse.ArgumentParser()
    ap.add_argument("--obj", default=os.path.join(ROOT, "dynamicexpressions.jl_amd", "csrc", "_obj", "irp_de_kernels", "k.out"))
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("-o", "--out", default=os.path.join(ROOT, "profiles", "valu_slots.json"))
    a = ap.parse_args()
    slots, counts = table(a.obj, "float" if a.dtype == "f32" else "double")
    turbo = table(a.obj, "float", True)[0] if a.dtype == "f32" else {}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_profiles import kernel_source_hash
    doc = dict(kernel_source_hash=kernel_source_hash(),  # bench.py refuses the table when the library was built from other sources
               source=f"tools/valu_slots.py over {os.path.relpath(a.obj, ROOT)} (llvm-objdump of the gfx950 code object)",
               rule="VALU cycles (wave64 instruction occupancy of a SIMD) on the shortest entry->return path; measured classes: "
                    f"full rate {C_FULL}, half rate / packed / SGPR operand {C_HALF}, transcendental {C_TRANS} (profiles/r2_valu_rate.json)",
               per_tree_overhead_cycles=24.0,  # state zeroing, output address, ballot compare around the chain (~10 instructions)
               layout=counts, handlers={str(k): v for k, v in sorted(slots.items())},
               handlers_turbo={str(k): v for k, v in sorted(turbo.items())})  # the same ids in a DE_OPT_TURBO program
    with open(a.out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(f"{len(slots)} handlers -> {a.out}")
    for k in sorted(slots)[:0]:
        print(k, slots[k])


if __name__ == "__main__":
    sys.exit(main())
