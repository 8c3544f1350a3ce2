#!/usr/bin/env python3
"""asmopt.py IN.s OUT.s [module] — peephole pass over the backend's assembly of a handler module (build.sh, between llc and the assembler).

The threaded interpreter's cheap handlers are bound by the NUMBER of instructions a wavefront issues per dispatch (a wave issues one
instruction per ~8 cycles whatever its kind; tools/probe/issue_probe.py, tools/exp_dispatch_cost.py), so every instruction of the
dispatch overhead counts.  One of them cannot be removed from the source: a handler saves the address of its successor — it arrives in
the SGPR pair the next record is loaded into — with TWO s_mov_b32, because the calling convention splits a 64-bit `inreg` argument into
32-bit halves and the register allocator copies each half on its own.

  P1  s_mov_b32 s(2k+1), s(2m+1)  ...  s_mov_b32 s(2k), s(2m)      (either order, same basic block, nothing in between touches the
      four registers)                                 ->  s_mov_b64 s[2k:2k+1], s[2m:2m+1]   at the place of the first

  P2  s_mov_b32 sA, sB ; [s_load_dword*/v_*/ds_*/s_waitcnt/s_nop that do not touch sA, sC or write sB... the load of the next record
      overwrites sB] ; s_add_u32 sA, sC, sA          ->  s_add_u32 sA, sC, sB   at the place of the copy
      (the gradient / reverse chains: successor address = handler base + the offset that arrived in the quad the next record is
      loaded into; the backend copies the offset out of the way first).  Nothing that reads or writes SCC may stand between the copy
      and the add — the add's carry feeds the s_addc_u32 behind it, which stays where it is.

The pass also appends the `.set amdgpu.max_num_named_barrier` the printer of this toolchain forgets (without it the assembler cannot
evaluate the kernels' resource symbols: "cannot evaluate equated symbol ... num_named_barrier"); with it the round trip
llc -S -> assembler gives the same text, kernel descriptors and metadata as llc -c (checked once per build by build.sh: DE_ASMOPT_VERIFY=1).
Prints the number of merged pairs; csrc/patch_expect/ pins it per module."""
import re
import sys

MOV = re.compile(r'^\ts_mov_b32 s(\d+), s(\d+)\s*$')
SREG = re.compile(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]')
# a line that ends a basic block or that the pass must not look across
BARRIER = re.compile(r'^\t(s_cbranch|s_branch|s_setpc|s_swappc|s_call|s_endpgm|s_getpc|s_barrier)|^[^\t;]|^\t\.')


def sregs(line):
    """every SGPR number a line mentions (operands of any instruction; ranges expanded)"""
    out = set()
    code = line.split(';', 1)[0]
    for m in SREG.finditer(code):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def merge_pairs(lines):
    merged = 0
    i = 0
    n = len(lines)
    while i < n:
        m = MOV.match(lines[i])
        if not m:
            i += 1
            continue
        d0, s0 = int(m.group(1)), int(m.group(2))
        if (d0 & 1) != (s0 & 1):
            i += 1
            continue
        d1, s1 = d0 ^ 1, s0 ^ 1  # the other half of both pairs
        quad = {d0, s0, d1, s1}
        j = i + 1
        found = -1
        while j < n and j - i <= 12:
            lj = lines[j]
            if BARRIER.match(lj):
                break
            mj = MOV.match(lj)
            if mj and int(mj.group(1)) == d1 and int(mj.group(2)) == s1:
                found = j
                break
            if sregs(lj) & quad:
                break
            j += 1
        if found < 0:
            i += 1
            continue
        dl, sl = min(d0, d1), min(s0, s1)
        lines[i] = f'\ts_mov_b64 s[{dl}:{dl + 1}], s[{sl}:{sl + 1}]\n'
        del lines[found]
        n -= 1
        merged += 1
        i += 1
    return merged


ADD = re.compile(r'^\ts_add_u32 s(\d+), s(\d+), s(\d+)\s*$')
# instructions the add may be hoisted over: they neither read nor write SCC (vector, LDS, scalar loads, waits, plain scalar moves)
SCC_NEUTRAL = re.compile(r'^\t(v_|ds_|s_load_dword|s_waitcnt|s_nop|s_mov_b32|s_mov_b64|global_|buffer_|;)')


def fold_copy_into_add(lines):
    folded = 0
    i = 0
    n = len(lines)
    while i < n:
        m = MOV.match(lines[i])
        if not m:
            i += 1
            continue
        a, b = int(m.group(1)), int(m.group(2))
        j = i + 1
        found = -1
        while j < n and j - i <= 8:
            lj = lines[j]
            if BARRIER.match(lj):
                break
            ma = ADD.match(lj)
            if ma and int(ma.group(1)) == a and a in (int(ma.group(2)), int(ma.group(3))) and int(ma.group(2)) != int(ma.group(3)):
                c = int(ma.group(3)) if int(ma.group(2)) == a else int(ma.group(2))
                # sC must hold the same value at the copy's place: nothing in between may mention it as a destination — checked below
                ok = c not in (a, b) and all(c not in sregs_written(lines[k]) for k in range(i + 1, j))
                if ok:
                    found = j
                    cc = c
                break
            if not SCC_NEUTRAL.match(lj) or a in sregs(lj):
                break
            j += 1
        if found < 0:
            i += 1
            continue
        lines[i] = f'\ts_add_u32 s{a}, s{cc}, s{b}\n'
        del lines[found]
        n -= 1
        folded += 1
        i += 1
    return folded


def sregs_written(line):
    """SGPRs an instruction of the SCC_NEUTRAL set may write: the first operand of s_load / s_mov; any SGPR a vector instruction names"""
    code = line.split(';', 1)[0].strip()
    if code.startswith('v_'):
        # v_readfirstlane / v_readlane, VOP3 compares, carry-out adds, v_div_scale, v_mad_u64_u32 ... DO write SGPRs, in various operand
        # positions: every SGPR a vector instruction mentions counts as written (conservative: the hoist is refused)
        return sregs(line)
    if not (code.startswith('s_load') or code.startswith('s_mov')):
        return set()
    first = code.split(None, 1)[1].split(',')[0] if len(code.split(None, 1)) > 1 else ''
    return sregs('\t' + first)


def main(src, dst, module):
    lines = open(src).read().splitlines(keepends=True)
    merged = merge_pairs(lines)
    folded = fold_copy_into_add(lines)
    text = ''.join(lines)
    if 'amdgpu.max_num_named_barrier' in text and not re.search(r'^\t\.set amdgpu\.max_num_named_barrier,', text, re.M):
        text += '\t.set amdgpu.max_num_named_barrier, 0\n'
    open(dst, 'w').write(text)
    print(f'asmopt: {merged} s_mov_b32 pair(s) merged into s_mov_b64, {folded} copies folded into s_add_u32' + (f' ({module})' if module else ''))
    if module:
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import patch_expect
        patch_expect.check(module, {'asmopt_pairs': merged, 'asmopt_folded': folded})


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else '')
