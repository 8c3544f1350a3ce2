#!/usr/bin/env python3
"""asmpatch.py OBJECT.o — let the output store of one tree overlap with the evaluation of the next.

The AMDGPU backend opens every non-kernel function with `s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)` (SIInsertWaitcnts: a
callee cannot know what its caller left in flight).  For the direct-threaded handlers (de_kernels.hip: h_chain<...>, h_param<...>, h_tree_skip<...> — the early-exit walk runs right behind a tree's
output store and reads only LDS and the scalar stream;
de_grad_threaded.hip: gh_chain<...>; de_rev_threaded.hip: rh_chain<...>) the only vector-memory operations that can be in
flight at their entry are the `global_store`s of the PREVIOUS tree's results (h_tree_end / the gradient kernels' epilogues)
— nothing a handler reads — yet the first handler of every tree would wait for their write acknowledgement (~1-2 us), once
per tree and wavefront.  This pass rewrites that entry wait to
`s_waitcnt expcnt(0) lgkmcnt(0)` in those functions only.  Why it is safe:
  * handlers take their inputs in registers and LDS (lgkmcnt is still drained); none consumes a VMEM result it did not
    issue itself, and a handler that does issue loads (h_param) waits for them with counts computed from its own
    instructions — an extra OLDER store in flight can only make such a wait longer (vmcnt retires loads in order), never
    let it pass early;
  * gfx9 VMEM stores read their data registers when they issue, so overwriting those registers afterwards is fine;
  * the kernel itself waits for vmcnt(0) before it ends (s_endpgm drains stores).
  * ONLY handlers that contain no vector-memory instruction at all are relaxed — plus the eval kernel's h_tree_end (and its out-of-line
    twin h_tree_end_slow: ragged stores, the fused loss's partial — every tree of a fused-loss launch ends there), which
    only STORE (checked: no vector load, and a vmcnt(0) wait in front of every return to the kernel; end_handler_is_safe).  A handler with a stack frame (h_param,
    the generic handlers that call cold_op) restores its callee-saved VGPR with a `scratch_load` right before its tail call
    and relies on the NEXT function's entry wait to complete it; a relaxed successor never touches that register (it has
    no scratch instruction to save it with), so the restore lands harmlessly while it runs, and the chain always ends in
    an end handler (h_tree_end, g_end, r_end) — left untouched, full wait — before control returns to the kernel, which
    does use those registers and which assumes, like every caller, that a call has drained the loads issued before it.
Works on the relocatable gfx950 object (one 32-bit instruction word per function); refuses to touch a function whose first
instruction is not exactly that wait."""
import os
import re
import struct
import subprocess
import sys

TARGETS = re.compile(r"^_ZN2de(7h_chainI|7h_paramI|9h_un_fastI|10h_div_fastI|12h_unrow_fastI|14h_divrowc_fastI|11h_div2_fastI|13h_un_end_fastI|14h_div_end_fastI|10h_tree_endI|15h_tree_end_slowI|11h_tree_skipI|11h_chain_endI|\d+[gr]tm_\w+?8[gr]h_chainI)")  # never an end handler (h_tree_end, g_end, r_end): the end of every chain keeps the full wait
VMEM = re.compile(r"^\s*(scratch_|flat_|global_|buffer_|tbuffer_|image_)")
LLVM = os.environ.get("LLVM", "/opt/rocm/lib/llvm/bin")


def vmem_free_functions(path):
    """Names of the functions of the object whose bodies contain no vector-memory instruction."""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", path], check=True, capture_output=True, text=True).stdout
    ok, name, clean = set(), None, True
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            if name and clean:
                ok.add(name)
            name, clean = m.group(1), True
        elif name and VMEM.match(line):
            clean = False
    if name and clean:
        ok.add(name)
    return ok
END = re.compile(r"^_ZN2de(10h_tree_endI|15h_tree_end_slowI|11h_chain_endI|13h_un_end_fastI|14h_div_end_fastI)")  # the eval kernel's end-of-tree handlers: they store, and read nothing back


def end_handler_is_safe(path, name):
    """h_tree_end issues the tree's output store itself, so it is not VMEM-free — but its entry wait only makes it wait for
    the PREVIOUS tree's store.  It may be relaxed too if it never reads vector memory (only stores) and every return to the
    kernel (s_setpc_b64 s[30:31]) is preceded by a wait for vmcnt(0): the guarantee the callers of a chain rely on."""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", path], check=True, capture_output=True, text=True).stdout
    body, on = [], False
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            on = m.group(1) == name
            continue
        if on and "//" in line:
            body.append(line.split("//")[0].strip())
    if not body or any(re.match(r"(scratch_|flat_|buffer_|tbuffer_|image_|global_load|global_atomic)", i) for i in body):
        return False
    rets = [k for k, i in enumerate(body) if i.startswith("s_setpc_b64 s[30:31]")]
    return bool(rets) and all(k > 0 and body[k - 1].startswith("s_waitcnt") and "vmcnt(0)" in body[k - 1] for k in rets)


ENTRY_WAIT = 0xBF8C0000      # s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)
RELAXED_WAIT = 0xBF8CC00F    # s_waitcnt expcnt(0) lgkmcnt(0)    (gfx9 simm16: vmcnt = [15:14]:[3:0] = 63 = no wait)


def main(path):
    """Patches the relocatable object in place (the backend's own assembly does not survive a round trip through the
    assembler: its resource-usage symbols of functions with indirect calls cannot be re-evaluated)."""
    leaf = vmem_free_functions(path)
    blob = bytearray(open(path, "rb").read())
    assert blob[:4] == b"\x7fELF" and blob[4] == 2 and blob[5] == 1, "not a little-endian ELF64 file"
    e_shoff, = struct.unpack_from("<Q", blob, 0x28)
    e_shentsize, e_shnum, e_shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    secs = []
    for i in range(e_shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", blob, e_shoff + i * e_shentsize)
        secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    symtab = next(s for s in secs if s["type"] == 2)  # SHT_SYMTAB
    strtab = secs[symtab["link"]]
    n = skipped = 0
    for k in range(symtab["size"] // 24):
        st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", blob, symtab["off"] + 24 * k)
        if (st_info & 0xF) != 2 or st_shndx == 0 or st_shndx >= len(secs):  # STT_FUNC, defined
            continue
        end = blob.index(b"\0", strtab["off"] + st_name)
        name = blob[strtab["off"] + st_name:end].decode()
        if not TARGETS.match(name):
            continue
        if name not in leaf and not (END.match(name) and end_handler_is_safe(path, name)):
            skipped += 1
            continue
        sec = secs[st_shndx]
        pos = sec["off"] + (st_value - sec["addr"])
        word, = struct.unpack_from("<I", blob, pos)
        if word != ENTRY_WAIT:
            sys.exit(f"asmpatch: {name} does not start with the entry wait (0x{word:08X})")
        struct.pack_into("<I", blob, pos, RELAXED_WAIT)
        n += 1
    if n == 0:
        sys.exit("asmpatch: no handler entry found")
    open(path, "wb").write(bytes(blob))
    print(f"asmpatch: entry vmcnt wait dropped in {n} eval handlers ({skipped} with vector-memory instructions left alone)")
    return dict(asmpatch_relaxed=n, asmpatch_left_alone=skipped)


if __name__ == "__main__":
    stats = main(sys.argv[1])
    if len(sys.argv) > 2:
        import patch_expect
        patch_expect.check(sys.argv[2], stats)
