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
attributor from the function bodies and their transitive callees).

Second job (direct-threaded eval handlers, de_kernels.hip): the instruction-stream pointer every handler receives
and hands to the next one is wave-uniform, but the C calling convention passes pointers in VGPRs; each handler would
then spend two v_readfirstlane to read its record with a scalar load and could not tail-call (the backend refuses a
sibling call through a divergent address).  The IR parameter attribute `inreg` puts the argument in SGPRs; HIP has no
source spelling for it on a device function, so it is added here to the `ptr addrspace(4)` parameter — and to the integer
parameters behind it, the operand words one handler hands to the next — of every handler definition AND of every indirect handler call (caller and callee must agree, and `musttail` requires identical
prototypes).  Only functions returning the handler state struct that have such a parameter are touched."""
import re
import sys

NO_ATTRS = ["amdgpu-no-dispatch-ptr", "amdgpu-no-queue-ptr", "amdgpu-no-implicitarg-ptr", "amdgpu-no-dispatch-id",
            "amdgpu-no-workgroup-id-x", "amdgpu-no-workgroup-id-y", "amdgpu-no-workgroup-id-z",
            "amdgpu-no-workitem-id-x", "amdgpu-no-workitem-id-y", "amdgpu-no-workitem-id-z",
            "amdgpu-no-lds-kernel-id", "amdgpu-no-hostcall-ptr", "amdgpu-no-heap-ptr", "amdgpu-no-default-queue",
            "amdgpu-no-completion-action", "amdgpu-no-multigrid-sync-arg", "amdgpu-no-flat-scratch-init",
            "amdgpu-no-cluster-id-x", "amdgpu-no-cluster-id-y", "amdgpu-no-cluster-id-z"]
HSTATE = r'%"struct\.de::(?:\w+::)?[HG]State(?:\.\d+)?"'


# what a function must not contain (directly) for the implicit input to be dead in it
USES = {
    "amdgpu-no-dispatch-ptr": ["llvm.amdgcn.dispatch.ptr"],
    "amdgpu-no-queue-ptr": ["llvm.amdgcn.queue.ptr", "llvm.trap", "llvm.debugtrap"],
    "amdgpu-no-implicitarg-ptr": ["llvm.amdgcn.implicitarg.ptr"],
    "amdgpu-no-dispatch-id": ["llvm.amdgcn.dispatch.id"],
    "amdgpu-no-workgroup-id-x": ["llvm.amdgcn.workgroup.id.x"], "amdgpu-no-workgroup-id-y": ["llvm.amdgcn.workgroup.id.y"],
    "amdgpu-no-workgroup-id-z": ["llvm.amdgcn.workgroup.id.z"],
    "amdgpu-no-workitem-id-x": ["llvm.amdgcn.workitem.id.x"], "amdgpu-no-workitem-id-y": ["llvm.amdgcn.workitem.id.y"],
    "amdgpu-no-workitem-id-z": ["llvm.amdgcn.workitem.id.z"],
    "amdgpu-no-lds-kernel-id": ["llvm.amdgcn.lds.kernel.id"],
    # the five below live behind the implicit-argument pointer
    "amdgpu-no-hostcall-ptr": ["llvm.amdgcn.implicitarg.ptr"], "amdgpu-no-heap-ptr": ["llvm.amdgcn.implicitarg.ptr"],
    "amdgpu-no-default-queue": ["llvm.amdgcn.implicitarg.ptr"], "amdgpu-no-completion-action": ["llvm.amdgcn.implicitarg.ptr"],
    "amdgpu-no-multigrid-sync-arg": ["llvm.amdgcn.implicitarg.ptr"],
    "amdgpu-no-flat-scratch-init": ["addrspacecast ptr addrspace(5)"],
    "amdgpu-no-cluster-id-x": ["llvm.amdgcn.cluster.id"], "amdgpu-no-cluster-id-y": ["llvm.amdgcn.cluster.id"],
    "amdgpu-no-cluster-id-z": ["llvm.amdgcn.cluster.id"],
}


def main(src, dst):
    text = open(src).read()
    groups = {int(m.group(1)): m.group(2) for m in re.finditer(r'^attributes #(\d+) = \{(.*)\}$', text, re.M)}
    # every function definition: name -> (attribute group, body)
    funcs = {}
    for m in re.finditer(r'^define [^\n]*? @([^\s(]+)\([^\n]*\)[^\n#]*#(\d+)[^\n]*\{\n(.*?)^\}', text, re.M | re.S):
        funcs[m.group(1)] = (int(m.group(2)), m.group(3))
    # every handler definition and the attributes all of them share
    hdefs = re.findall(r'^define [^\n]*?' + HSTATE + r' @([^\s(]+)\([^\n]*\)[^\n#]*#(\d+)', text, re.M)
    if not hdefs:
        sys.exit("irpatch: no handler definitions found")
    defs = [g for _, g in hdefs]

    def dead_in(name, attr, seen):
        """The implicit input behind `attr` is dead in function `name`: it carries the attribute (AMDGPU attributor), or its
        body neither reads the input nor calls — directly — anything in which it is live.  INDIRECT calls inside a handler
        are the tail calls to other handlers (closed world: every callee is checked by the caller of this function)."""
        if name not in funcs:
            return name.startswith("llvm.") and not any(u in name for u in USES[attr])
        g, body = funcs[name]
        if f'"{attr}"' in groups[g]:
            return True
        if name in seen:
            return True
        seen.add(name)
        if any(u in body for u in USES[attr]):
            return False
        return all(dead_in(c, attr, seen) for c in set(re.findall(r'call [^\n]*? @([^\s(]+)\(', body)))
    allowed = [a for a in NO_ATTRS if all(dead_in(n, a, set()) for n, _ in hdefs)]
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
    # ... and to the handler DEFINITIONS that lack them (handlers that end in an indirect tail call: the attributor cannot
    # see their callee, the closed-world check above can), so that the inputs are not kept alive as live-ins either
    def_ids = {}
    for g in sorted(set(int(g) for g in defs)):
        extra = " ".join(f'"{a}"' for a in allowed if f'"{a}"' not in groups[g])
        if extra:
            def_ids[g] = (next_id, groups[g].rstrip() + " " + extra + " ")
            next_id += 1
    if def_ids:
        def_re = re.compile(r'^(define [^\n]*?' + HSTATE + r' @[^\n(]+\([^\n]*\)[^\n#]*)#(\d+)', re.M)
        text = def_re.sub(lambda m: f"{m.group(1)}#{def_ids.get(int(m.group(2)), (int(m.group(2)),))[0]}", text)
    # stream pointer in SGPRs: definitions and indirect calls of handlers with a constant-address-space pointer parameter
    n_inreg = 0

    def add_inreg(m):
        """The stream pointer and EVERY parameter after it (the operand words the predecessor hands over) are wave-uniform."""
        nonlocal n_inreg
        line = m.group(0)
        k = line.find("ptr addrspace(4)")
        if k < 0 or "ptr addrspace(4) inreg" in line:
            return line
        end = line.rfind(")")
        head, params, tail = line[:k], line[k:end], line[end:]
        out = []
        for prm in params.split(", "):
            mt = re.match(r"(ptr addrspace\(4\)|i32|i64)( |$)", prm)
            if mt:
                prm = mt.group(1) + " inreg" + prm[len(mt.group(1)):]
                n_inreg += 1
            out.append(prm)
        return head + ", ".join(out) + tail
    text = re.sub(r'^define [^\n]*?' + HSTATE + r' @[^\n(]+\([^\n]*$', add_inreg, text, flags=re.M)
    # (indirect calls, and direct calls of one handler by another: h_tree_end -> h_tree_end_slow)
    text = re.sub(r'^\s*%[\w.]+ = (?:tail |musttail |notail )?call ' + HSTATE + r' [%@][^\s(]+\([^\n]*$', add_inreg, text, flags=re.M)
    text = text.rstrip("\n") + "\n" + "".join(f"attributes #{i} = {{{body}}}\n" for i, body in list(new_ids.values()) + list(def_ids.values()))
    open(dst, "w").write(text)
    print(f"irpatch: {len(calls)} indirect handler call(s), {len(defs)} handlers, {len(allowed)}/{len(NO_ATTRS)} inputs dropped, "
          f"{n_inreg} stream-pointer parameter(s) moved to SGPRs")
    return dict(irpatch_indirect_calls=len(calls), irpatch_handlers=len(defs), irpatch_inputs_dropped=len(allowed), irpatch_inreg=n_inreg)


if __name__ == "__main__":
    stats = main(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 3:  # build.sh: compare with / record the counts this module is known to produce
        import patch_expect
        patch_expect.check(sys.argv[3], stats)
