"""CPU test of csrc/irpatch.py (the build-time IR pass that marks the interpreter's indirect handler calls as
needing no implicit kernel inputs): attributes are added to the indirect calls only, and only those that
EVERY possible callee already carries."""
import importlib.util
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SPEC = importlib.util.spec_from_file_location(
    "irpatch", os.path.join(HERE, "..", "dynamicexpressions.jl_amd", "csrc", "irpatch.py"))
irpatch = importlib.util.module_from_spec(SPEC)
SPEC.loader.exec_module(irpatch)

IR = '''
%"struct.de::HState" = type { <4 x float>, <2 x float>, [8 x i8] }
declare ptr addrspace(4) @llvm.amdgcn.queue.ptr()
declare ptr addrspace(4) @llvm.amdgcn.implicitarg.ptr()
define internal %"struct.de::HState" @h_a(<4 x float> %0, i32 %1, ptr addrspace(4) noundef %2) #0 {
  ret %"struct.de::HState" undef
}
define internal %"struct.de::HState" @h_b(<4 x float> %0, i32 %1, ptr addrspace(4) noundef %2) #1 {
  %q = call ptr addrspace(4) @llvm.amdgcn.queue.ptr()
  %n = load ptr, ptr addrspace(4) %2
  %r = musttail call %"struct.de::HState" %n(<4 x float> %0, i32 %1, ptr addrspace(4) noundef %2) #3
  ret %"struct.de::HState" %r
}
define internal %"struct.de::HState" @h_c(<4 x float> %0, i32 %1, ptr addrspace(4) noundef %2) #1 {
  call void @other(i32 %1)
  ret %"struct.de::HState" undef
}
define void @other(i32 %x) #2 {
  %p = call ptr addrspace(4) @llvm.amdgcn.implicitarg.ptr()
  ret void
}
define protected amdgpu_kernel void @kern(ptr %fn, ptr addrspace(4) %code) #2 {
  %r = tail call %"struct.de::HState" %fn(<4 x float> zeroinitializer, i32 0, ptr addrspace(4) %code) #3
  %d = tail call %"struct.de::HState" @h_a(<4 x float> zeroinitializer, i32 0, ptr addrspace(4) %code) #3
  ret void
}
attributes #0 = { nounwind "amdgpu-no-dispatch-ptr" "amdgpu-no-queue-ptr" "amdgpu-no-workitem-id-x" }
attributes #1 = { nounwind }
attributes #2 = { nounwind }
attributes #3 = { convergent nounwind }
'''


def test_only_indirect_handler_calls_get_the_attributes_every_possible_callee_allows(tmp_path):
    src, dst = tmp_path / "k.ll", tmp_path / "k2.ll"
    src.write_text(IR)
    irpatch.main(str(src), str(dst))
    out = dst.read_text()
    ind = re.search(r'%r = tail call [^\n]* %fn\([^\n]*\) #(\d+)', out)
    direct = re.search(r'%d = tail call [^\n]* @h_a\([^\n]*\) #(\d+)', out)
    assert ind and direct
    assert direct.group(1) == "3"  # the direct call keeps its group
    new_group = re.search(r'^attributes #%s = \{(.*)\}$' % ind.group(1), out, re.M).group(1)
    assert ind.group(1) != "3" and "convergent" in new_group
    # inputs no handler reads — itself or through a direct callee — are dropped, whether the attributor said so (h_a) or
    # the closed-world body check did (h_b ends in an indirect tail call: the attributor gives up on it)
    assert '"amdgpu-no-dispatch-ptr"' in new_group and '"amdgpu-no-workitem-id-x"' in new_group
    assert '"amdgpu-no-queue-ptr"' not in new_group       # h_b reads the queue pointer
    assert '"amdgpu-no-implicitarg-ptr"' not in new_group  # h_c calls @other, which reads the implicit-argument pointer
    assert '"amdgpu-no-heap-ptr"' not in new_group         # ... and everything that lives behind it
    # the handler definitions get the same attributes (the inputs are not live-ins either)
    hb = re.search(r'^define [^\n]* @h_b\([^\n]*\) #(\d+)', out, re.M)
    assert '"amdgpu-no-dispatch-ptr"' in re.search(r'^attributes #%s = \{(.*)\}$' % hb.group(1), out, re.M).group(1)
    # the stream pointer of every handler definition and of every handler call is `inreg` — the direct call of one handler by
    # another too (h_tree_end -> h_tree_end_slow: caller and callee must agree, musttail demands identical prototypes)
    assert out.count("ptr addrspace(4) inreg") == 6  # 3 definitions + the musttail call + the kernel's indirect call + the direct call
    assert "ptr addrspace(4) inreg" in direct.group(0)


def test_refuses_a_module_without_handlers_or_without_indirect_calls(tmp_path):
    src, dst = tmp_path / "k.ll", tmp_path / "k2.ll"
    src.write_text("define void @f() #0 {\n  ret void\n}\nattributes #0 = { nounwind }\n")
    with pytest.raises(SystemExit):
        irpatch.main(str(src), str(dst))
    no_indirect = IR.replace('  %r = tail call %"struct.de::HState" %fn(<4 x float> zeroinitializer, i32 0, ptr addrspace(4) %code) #3\n', '')
    no_indirect = no_indirect.replace('  %r = musttail call %"struct.de::HState" %n(<4 x float> %0, i32 %1, ptr addrspace(4) noundef %2) #3\n', '')
    assert no_indirect.count("%r") == 1
    src.write_text(no_indirect)
    with pytest.raises(SystemExit):
        irpatch.main(str(src), str(dst))


def test_asmpatch_relaxes_only_the_entry_wait_of_eval_handlers():
    """csrc/asmpatch.py rewrites ONE instruction word — the function-entry `s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)` — of
    the direct-threaded eval handlers that contain no vector-memory instruction; checked on the shipped code object: those
    start with the relaxed wait, every other function (handlers with a stack frame, h_tree_end, cold_op, flag_incomplete, OCML
    helpers ...) keeps the full one."""
    import subprocess
    obj = os.path.join(HERE, "..", "dynamicexpressions.jl_amd", "csrc", "_obj", "irp_de_kernels", "k.out")
    if not os.path.exists(obj):
        pytest.skip("device code object not built here (csrc/_obj is a build artefact)")
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", obj], check=True, capture_output=True, text=True).stdout
    first, vmem = {}, {}
    name = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            name = m.group(1)
            vmem[name] = False
            continue
        if name and "//" in line:
            ins = line.split("//")[0].strip()
            first.setdefault(name, ins)
            if re.match(r"(scratch_|flat_|global_|buffer_)", ins):
                vmem[name] = True
    handlers = {n: i for n, i in first.items() if re.match(r"_ZN2de(7h_chainI|7h_paramI|9h_un_fastI|10h_div_fastI|12h_unrow_fastI|14h_divrowc_fastI|11h_div2_fastI|11h_tree_skipI)", n)}  # (h_un_fast / h_div_fast: fast-path-only Float32 cos / exp / sin and exact divisions; h_tree_skip: the early-exit walk, LDS + scalar loads only)
    others = {n: i for n, i in first.items() if n not in handlers and n.startswith("_ZN2de") and "kernel" not in n and "fill_handlers" not in n}
    assert len(handlers) > 300 and others
    relaxed = {n for n, i in handlers.items() if i == "s_waitcnt expcnt(0) lgkmcnt(0)"}
    assert len(relaxed) > 250
    assert all(not vmem[n] for n in relaxed)                 # only handlers without any vector-memory instruction
    assert all(vmem[n] for n in handlers if n not in relaxed)  # ... and all of those
    assert all(i == "s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)" for n, i in handlers.items() if n not in relaxed)
    # ... except h_tree_end and its out-of-line twin (ragged stores, the fused loss's partials), which only store: relaxed too
    # (and the end-fused last-instruction handlers h_chain_end<body>, relaxed where the same two conditions hold)
    ends = {n: i for n, i in others.items() if ("10h_tree_endI" in n or "15h_tree_end_slowI" in n or "11h_chain_endI" in n or "13h_un_end_fastI" in n or "14h_div_end_fastI" in n) and i == "s_waitcnt expcnt(0) lgkmcnt(0)"}
    assert sum("10h_tree_endI" in n for n in ends) == 2 and sum("11h_chain_endI" in n for n in ends) >= 20
    assert all(i == "s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)" for n, i in others.items() if n not in ends), others
    assert sum("15h_tree_end_slowI" in n for n in others) == 2
    # every return of h_tree_end to the kernel drains vector memory first (what the callers of a chain rely on)
    body, on = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            on = m.group(1) if m.group(1) in ends else None
            continue
        if on and "//" in line:
            body.setdefault(on, []).append(line.split("//")[0].strip())
    for n, ins in body.items():
        rets = [k for k, i in enumerate(ins) if i.startswith("s_setpc_b64 s[30:31]")]
        assert rets and all("vmcnt(0)" in ins[k - 1] for k in rets), n
        assert not any(re.match(r"(global_load|scratch_|flat_|buffer_)", i) for i in ins), n
