#!/usr/bin/env python3
"""VALU replay of the threaded eval kernel — generator.  No GPU needed.

    python tools/probe/valu_replay/gen.py          ->  tools/probe/valu_replay/replay_gen.hip  (+ handlers.json)

The eval kernel is said to sit at the vector ALU's practical ceiling.  This probe REPLAYS the real thing: for every
handler of the shipped code object (csrc/_obj/irp_de_kernels/k.out, the disassembly tools/valu_slots.py prices) it takes the
instructions on the handler's shortest entry -> return path VERBATIM — same mnemonics, same registers, same modifiers, same order, same
`s_nop` hazards — and keeps only what a variant is about:

    V0  the VALU instructions (+ s_nop)                                       "the pipe alone"
    V1  V0 + the handlers' own scalar ALU instructions (literal moves, the pointer bump ...)
    V2  V1 + the LDS operand reads / spill writes (+ the waits for them)
    V3  V2 + the bookkeeping of V4 without its store (three scalar instructions and a branch per dispatch: what V4 adds besides the store)
    V4  V3 + one 1 KB output store per tree (non-temporal, a row stride apart, as h_tree_end stores)

and never a record load, a jump, or a dispatch.  Each handler's body sits in a loop whose trip count comes from a table in memory
(the dispatch histogram of a population, made by run.py from de_program_dump): one binary replays any population.  A wave executes
the histogram of the WHOLE population once per outer iteration — what the real launch spreads over the ~16 tree chunks of a sample
tile — so `blocks x iters` passes correspond to that many sample tiles of the real launch.  Values are garbage (registers start as
lane * 16, scalars as 0); the vector ALU's timing does not depend on them, branches are gone, and LDS addresses stay inside the
allocation (out-of-range LDS accesses are dropped by the hardware anyway)."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import valu_slots as VS  # noqa: E402

OBJ = os.path.join(ROOT, "dynamicexpressions.jl_amd", "csrc", "_obj", "irp_de_kernels", "k.out")
DROP_SALU = ("s_load", "s_setpc", "s_getpc", "s_swappc", "s_branch", "s_cbranch", "s_waitcnt", "s_endpgm", "s_barrier", "s_sendmsg",
             "s_sleep", "s_setprio", "s_call", "s_trap", "s_icache", "s_dcache", "s_buffer", "s_store", "s_atc", "s_memtime", "s_memreal")
CTR, CREDIT = 96, 97          # scalar registers of the replay's own loops (checked: no handler names them)
OUTER, TOTAL, NTREES = 98, 99, 95
ROWP, ROWB = 90, 92           # s[90:91]: the output row the next store of V3 goes to; s[92:93]: row 0 of the current pass


def shortest_path(code):
    """instructions on the cheapest entry -> return path (the path tools/valu_slots.py prices), the return itself excluded"""
    import heapq
    idx = {a: i for i, (a, _, _, _) in enumerate(code)}
    dist, parent = {0: 0.0}, {0: None}
    heap = [(0.0, 0)]
    while heap:
        d, i = heapq.heappop(heap)
        if dist.get(i, 1e30) < d:
            continue
        _, mn, ops, tgt = code[i]
        if mn.startswith("s_setpc") or mn == "s_endpgm":
            pair = ops.split()[0].rstrip(",") if ops else ""
            if any(code[j][1] == "s_getpc_b64" and code[j][2].split()[0].rstrip(",") == pair for j in range(max(0, i - 8), i)):
                continue  # the tail call into the slow twin: not the path ordinary data takes
            path = []
            k = parent[i]
            while k is not None:
                path.append(code[k])
                k = parent[k]
            return path[::-1]
        w = VS.weight(mn, ops)
        nxt = []
        if mn != "s_branch" and i + 1 < len(code):
            nxt.append(i + 1)
        if tgt is not None and tgt in idx:
            nxt.append(idx[tgt])
        for j in nxt:
            if j not in dist or d + w < dist[j]:
                dist[j] = d + w
                parent[j] = i
                heapq.heappush(heap, (d + w, j))
    return None


def handler_code(obj, ty="float"):
    """handler id -> (name, code) exactly as valu_slots.table maps them"""
    fns = VS.functions(obj)
    by_short, fast, end = {}, {}, None
    for full, code in fns.items():
        m = re.search(r"h_chain<\w+, &de::BState<\w+> de::b_(\w+<[^(]*>)\(", full)
        if m:
            by_short["h_" + m.group(1)] = code
            continue
        m = re.search(r"de::(h_param<[^(]*>)\(", full)
        if m:
            by_short[m.group(1)] = code
            continue
        if re.search(r"de::h_tree_end<float>\(", full):
            end = code
            continue
        for pat, fmt in ((r"de::h_un_fast<(\d), (\d), (true|false)>\(", "h_un<float, %s, %s, %s>"),
                         (r"de::h_div_fast<(\d), (\d)>\(", "h_bin<float, %s, %s, false>"),
                         (r"de::h_unrow_fast<(\d), (true|false), (true|false), (true|false), (true|false)>\(", "h_unrow_f<float, %s, %s, %s, %s, %s>"),
                         (r"de::h_divrowc_fast<(\d), (true|false)>\(", "h_binrowc<float, %s, %s, false>"),
                         (r"de::h_div2_fast<(\d), (true|false), (true|false), (true|false)>\(", "h_bin2<float, %s, %s, %s, %s, false>")):
            m = re.search(pat, full)
            if m:
                fast[fmt % m.groups()] = code
                break
    by_short.update(fast)
    names, _ = VS.handler_names(ty, False)
    return {hid: (nm, by_short[nm]) for hid, nm in names.items() if nm in by_short}, end


def keep(mn, ops, variant):
    """the instruction as the replay issues it, or None"""
    if "exec" in ops and not mn.startswith("v_"):
        return None
    if mn.startswith("v_"):
        if mn.startswith("v_cmpx"):
            mn = "v_cmp" + mn[6:]
        return mn, ops
    if mn == "s_nop":
        return mn, ops
    if mn.startswith("ds_"):
        return (mn, ops) if variant >= 2 else None
    if mn == "s_waitcnt":
        return ("s_waitcnt", "lgkmcnt(0)") if (variant >= 2 and "lgkmcnt" in ops) else None
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return None  # (the output store of V3 is the replay's own: the real one's address comes from the record)
    if mn.startswith("s_"):
        if variant < 1 or mn.startswith(DROP_SALU):
            return None
        return mn, ops
    return None


def regs(ops):
    v, s = set(), set()
    for m in re.finditer(r"(?<![\w.])([vsa])(\d+)(?![\w\[:])", ops):
        (v if m.group(1) in "va" else s).add(int(m.group(2)))
    for m in re.finditer(r"(?<![\w.])([vsa])\[(\d+):(\d+)\]", ops):
        (v if m.group(1) in "va" else s).update(range(int(m.group(2)), int(m.group(3)) + 1))
    return v, s


def emit_body(path, variant):
    lines, vr, sr, n_valu = [], set(), set(), 0
    for _, mn, ops, _ in path:
        k = keep(mn, ops, variant)
        if k is None:
            continue
        mn2, ops2 = k
        ops2 = re.sub(r"\|([^|]+)\|", r"abs(\1)", ops2)  # |v1| -> abs(v1): '|' is a dialect separator of GCC-style asm strings
        v, s = regs(ops2)
        vr |= v
        sr |= s
        n_valu += mn2.startswith("v_")
        lines.append(f"{mn2} {ops2}".strip())
    return lines, vr, sr, n_valu


def main():
    handlers, end_code = handler_code(OBJ)
    ids = sorted(handlers)
    doc = {"handlers": {}, "source": os.path.relpath(OBJ, ROOT)}
    kernels = []
    all_v, all_s = set(), set()
    for variant in range(5):
        body = []
        for n, hid in enumerate(ids):
            nm, code = handlers[hid]
            path = shortest_path(code)
            if path is None:
                continue
            lines, vr, sr, n_valu = emit_body(path, min(variant, 2))
            all_v |= vr
            all_s |= sr
            if variant == 0:
                doc["handlers"][str(hid)] = {"name": nm, "slot": n, "valu_insts": n_valu, "path_insts": len(path)}
            store = []
            if variant >= 3:  # Bresenham: one 1 KB store per TOTAL / NTREES dispatches, rows a stride apart (s[ROWP:ROWP+1] walks the rows)
                store = [f"s_sub_i32 s{CREDIT}, s{CREDIT}, s{NTREES}", f"s_cmp_lt_i32 s{CREDIT}, 0", "s_cbranch_scc0 4f",
                         f"global_store_dwordx4 v14, v[0:3], s[{ROWP}:{ROWP + 1}] nt" if variant == 4 else "s_nop 0", f"s_add_i32 s{CREDIT}, s{CREDIT}, s{TOTAL}",
                         f"s_add_u32 s{ROWP}, s{ROWP}, %4", f"s_addc_u32 s{ROWP + 1}, s{ROWP + 1}, %5", "4:"]
            body += [f"// ---- handler {hid}: {nm}", f"s_load_dword s{CTR}, %0, {4 * n}", "s_waitcnt lgkmcnt(0)", f"s_cmp_eq_u32 s{CTR}, 0", "s_cbranch_scc1 3f", "2:"]
            body += lines + store
            body += [f"s_sub_u32 s{CTR}, s{CTR}, 1", f"s_cmp_lg_u32 s{CTR}, 0", "s_cbranch_scc1 2b", "3:"]
        # the end of a tree (h_tree_end's own vector instructions: state zeroing, poison ballot), NTREES times per pass
        endl, vr, sr, n_end = emit_body(shortest_path(end_code) or [], min(variant, 1))
        all_v |= vr
        all_s |= sr
        if variant == 0:
            doc["tree_end_valu_insts"] = n_end
        body += ["// ---- h_tree_end (per tree)", f"s_mov_b32 s{CTR}, s{NTREES}", "2:"] + endl + [f"s_sub_u32 s{CTR}, s{CTR}, 1", f"s_cmp_lg_u32 s{CTR}, 0", "s_cbranch_scc1 2b"]
        kernels.append(body)
    own = {CTR, CREDIT, OUTER, TOTAL, NTREES, ROWP, ROWP + 1, ROWB, ROWB + 1}
    assert not (all_s & own), sorted(all_s)
    vmax, smax = max(all_v), max(all_s)
    doc.update(vgprs_named=vmax + 1, sgprs_named=smax + 1, slots=len(ids))
    with open(os.path.join(HERE, "handlers.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    vclob = ", ".join(f'"v{i}"' for i in range(vmax + 1))
    sclob = ", ".join(f'"s{i}"' for i in sorted(set(range(smax + 1)) | own))
    out = ['// GENERATED by tools/probe/valu_replay/gen.py from the disassembly of csrc/_obj/irp_de_kernels/k.out — do not edit.',
           '#include <hip/hip_runtime.h>', '#include <cstdint>', '']
    for variant, body in enumerate(kernels):
        # (the inputs are copied into the replay's own registers FIRST: the compiler may have put them anywhere outside the clobber list)
        init = [f"s_mov_b32 s{OUTER}, %1", f"s_mov_b32 s{NTREES}, %6", f"s_mov_b32 s{TOTAL}, %7", f"s_mov_b32 s{CREDIT}, 0",
                f"s_mov_b32 s{ROWB}, %8", f"s_mov_b32 s{ROWB + 1}, %9"]
        init += [f"v_mov_b32 v{i}, %3" for i in range(vmax + 1)] + [f"s_mov_b32 s{i}, 0" for i in range(smax + 1)]
        init += ["1:", f"s_mov_b64 s[{ROWP}:{ROWP + 1}], s[{ROWB}:{ROWB + 1}]"]  # a pass stores one 1 KB piece into every row; the next pass the piece behind it
        tail = [f"s_add_u32 s{ROWB}, s{ROWB}, 1024", f"s_addc_u32 s{ROWB + 1}, s{ROWB + 1}, 0", f"s_sub_u32 s{OUTER}, s{OUTER}, 1", f"s_cmp_lg_u32 s{OUTER}, 0", "s_cbranch_scc1 1b", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
        text = "\n".join(f'        "{ln}\\n"' if not ln.startswith("//") else f"        {ln}" for ln in init + body + tail)
        out += [f"extern \"C\" __global__ void __launch_bounds__(64) valu_replay_v{variant}(const uint32_t *counts, uint32_t iters, uint32_t n_trees, uint32_t total, "
                "char *out, uint64_t row_stride, uint32_t tile_bytes) {",
                "    extern __shared__ char lds[];",
                "    const uint32_t lane16 = threadIdx.x * 16u;",
                "    char *row0 = out + (uint64_t)blockIdx.x * iters * tile_bytes;  // this wave's first 1 KB piece of row 0",
                "    const uint32_t st_lo = (uint32_t)row_stride, st_hi = (uint32_t)(row_stride >> 32);",
                "    const uint32_t r_lo = (uint32_t)(uintptr_t)row0, r_hi = (uint32_t)((uintptr_t)row0 >> 32);",
                "    asm volatile(", text,
                '        : : "s"(counts), "s"(iters), "s"(lds), "v"(lane16), "s"(st_lo), "s"(st_hi), "s"(n_trees), "s"(total), "s"(r_lo), "s"(r_hi)',
                f'        : "memory", "vcc", "scc", {vclob}, {sclob});', "}", ""]
    out += ['extern "C" int valu_replay_run(int variant, const uint32_t *counts, uint32_t iters, uint32_t n_trees, uint32_t total, char *out, uint64_t row_stride,',
            '                               uint32_t blocks, uint32_t lds_bytes, float *ms) {',
            '    hipEvent_t e0, e1;', '    hipEventCreate(&e0); hipEventCreate(&e1);',
            '    void (*k[5])(const uint32_t *, uint32_t, uint32_t, uint32_t, char *, uint64_t, uint32_t) = {valu_replay_v0, valu_replay_v1, valu_replay_v2, valu_replay_v3, valu_replay_v4};',
            '    if (variant < 0 || variant > 4) return 1;',
            '    if (lds_bytes > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(k[variant]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);',
            '    hipEventRecord(e0, 0);',
            '    hipLaunchKernelGGL(k[variant], dim3(blocks), dim3(64), lds_bytes, 0, counts, iters, n_trees, total, out, row_stride, 1024u);',
            '    hipEventRecord(e1, 0);', '    if (hipEventSynchronize(e1) != hipSuccess) return 2;', '    hipEventElapsedTime(ms, e0, e1);',
            '    hipEventDestroy(e0); hipEventDestroy(e1);', '    return hipGetLastError() == hipSuccess ? 0 : 3;', '}',
            '// resident one-wave workgroups per CU for this much dynamic LDS (what the launch will really have)',
            'extern "C" int valu_replay_occupancy(int variant, uint32_t lds_bytes) {',
            '    void (*k[5])(const uint32_t *, uint32_t, uint32_t, uint32_t, char *, uint64_t, uint32_t) = {valu_replay_v0, valu_replay_v1, valu_replay_v2, valu_replay_v3, valu_replay_v4};',
            '    int n = -1;',
            '    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(k[variant]), 64, lds_bytes) != hipSuccess) return -1;',
            '    return n;', '}', '']
    with open(os.path.join(HERE, "replay_gen.hip"), "w") as fh:
        fh.write("\n".join(out))
    print(f"{len(ids)} handlers, VGPRs named v0..v{vmax}, SGPRs s0..s{smax} -> replay_gen.hip")


if __name__ == "__main__":
    main()
