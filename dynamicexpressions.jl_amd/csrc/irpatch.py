#!/usr/bin/env python3
"""irpatch.py IN.ll OUT.ll — mark the interpreter's indirect handler calls as needing no implicit
kernel inputs.

The threaded eval kernel (de_kernels.hip) dispatches every instruction through an indirect call.
For an indirect call LLVM cannot see the callee, so the AMDGPU backend re-materialises ALL implicit
inputs of the fixed function ABI before every call (dispatch ptr, queue ptr, implicit-arg ptr,
dispatch id, workgroup ids, packed work-item id: 8 scalar + 1 vector instruction per dispatch,
a third of the dispatch overhead), although no handler uses any of them.  The backend skips an
input when the CALL SITE carries the matching "amdgpu-no-*" attribute (SIISelLowering
passSpecialInputs); clang has no source-level spelling for call-site string attributes, so this
script adds them to the optimised device IR between clang's middle end and llc.

Safety: an attribute is added only if EVERY possible callee — every function defined in the module
whose return type is the handler state struct — already carries it (inferred by the AMDGPU
attributor from the function bodies and their transitive callees)."""
import re
import sys

NO_ATTRS = ["amdgpu-no-dispatch-ptr", "amdgpu-no-queue-ptr", "amdgpu-no-implicitarg-ptr", "amdgpu-no-dispatch-id",
            "amdgpu-no-workgroup-id-x", "amdgpu-no-workgroup-id-y", "amdgpu-no-workgroup-id-z",
            "amdgpu-no-workitem-id-x", "amdgpu-no-workitem-id-y", "amdgpu-no-workitem-id-z",
            "amdgpu-no-lds-kernel-id", "amdgpu-no-hostcall-ptr", "amdgpu-no-heap-ptr", "amdgpu-no-default-queue",
            "amdgpu-no-completion-action", "amdgpu-no-multigrid-sync-arg", "amdgpu-no-flat-scratch-init",
            "amdgpu-no-cluster-id-x", "amdgpu-no-cluster-id-y", "amdgpu-no-cluster-id-z"]
HSTATE = r'%"struct\.de::(?:\w+::)?[HG]State(?:\.\d+)?"'


def main(src, dst):
    text = open(src).read()
    groups = {int(m.group(1)): m.group(2) for m in re.finditer(r'^attributes #(\d+) = \{(.*)\}$', text, re.M)}
    # every handler definition and the attributes all of them share
    defs = re.findall(r'^define [^\n]*?' + HSTATE + r' @[^\n(]+\([^\n]*\)[^\n#]*#(\d+)', text, re.M)
    if not defs:
        sys.exit("irpatch: no handler definitions found")
    allowed = [a for a in NO_ATTRS if all(f'"{a}"' in groups[int(g)] for g in defs)]
    # indirect calls returning the handler state: callee operand is a local value (%...), not @global
    call_re = re.compile(r'^(\s*%[\w.]+ = (?:tail |musttail |notail )?call ' + HSTATE + r' %[\w.]+\([^\n]*\)) #(\d+)$', re.M)
    calls = call_re.findall(text)
    if not calls:
        sys.exit("irpatch: no indirect handler call found")
    new_ids = {}
    next_id = max(groups) + 1
    for _, g in calls:
        g = int(g)
        if g not in new_ids:
            extra = " ".join(f'"{a}"' for a in allowed if f'"{a}"' not in groups[g])
            new_ids[g] = (next_id, groups[g].rstrip() + " " + extra + " ")
            next_id += 1
    text = call_re.sub(lambda m: f"{m.group(1)} #{new_ids[int(m.group(2))][0]}", text)
    text = text.rstrip("\n") + "\n" + "".join(f"attributes #{i} = {{{body}}}\n" for i, body in new_ids.values())
    open(dst, "w").write(text)
    print(f"irpatch: {len(calls)} indirect handler call(s), {len(defs)} handlers, {len(allowed)}/{len(NO_ATTRS)} inputs dropped")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
