"""csrc/asmopt.py — the assembly-level peephole stage of build.sh — on hand-written snippets: what it merges, what it must leave alone."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("asmopt", os.path.join(ROOT, "dynamicexpressions.jl_amd", "csrc", "asmopt.py"))
asmopt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(asmopt)


def run(text, fn):
    lines = [l + "\n" for l in text.strip("\n").split("\n")]
    n = fn(lines)
    return n, "".join(lines)


def test_pair_becomes_one_s_mov_b64_at_the_place_of_the_first():
    n, out = run("\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 s17, s7\n\tv_add_u32_e32 v15, s4, v14\n\ts_mov_b32 s16, s6\n\ts_load_dwordx4 s[4:7], s[0:1], 0x0\n",
                 asmopt.merge_pairs)
    assert n == 1
    assert out.split("\n")[1] == "\ts_mov_b64 s[16:17], s[6:7]" and "s_mov_b32" not in out
    assert out.index("s_mov_b64") < out.index("v_add_u32") < out.index("s_load_dwordx4")


@pytest.mark.parametrize("between", [
    "\ts_load_dwordx4 s[4:7], s[0:1], 0x0",      # writes the source of the second half
    "\tv_pk_add_f32 v[0:1], v[0:1], s[16:17]",   # reads the destination
    "\ts_cbranch_scc1 .LBB0_2",                  # end of the basic block
    ".LBB0_1:",                                  # a label
])
def test_pair_is_left_alone_when_something_in_between_touches_it(between):
    n, out = run("\ts_mov_b32 s17, s7\n" + between + "\n\ts_mov_b32 s16, s6\n", asmopt.merge_pairs)
    assert n == 0 and out.count("s_mov_b32") == 2


def test_pair_needs_aligned_pairs_of_the_same_parity():
    assert run("\ts_mov_b32 s8, s5\n\ts_mov_b32 s9, s4\n", asmopt.merge_pairs)[0] == 0   # (la, nx) swapped: not a 64-bit copy
    assert run("\ts_mov_b32 s17, s7\n\ts_mov_b32 s18, s8\n", asmopt.merge_pairs)[0] == 0  # two different pairs


def test_copy_is_folded_into_the_add_over_the_record_load():
    n, out = run("\ts_mov_b32 s8, s4\n\ts_load_dwordx4 s[4:7], s[0:1], 0x0\n\tv_fmac_f32_e32 v6, 0, v0\n\ts_add_u32 s8, s2, s8\n\ts_addc_u32 s9, s3, 0\n",
                 asmopt.fold_copy_into_add)
    assert n == 1
    assert out.split("\n")[0] == "\ts_add_u32 s8, s2, s4" and out.count("s_add_u32") == 1 and "s_mov_b32" not in out
    assert out.index("s_add_u32") < out.index("s_load_dwordx4") < out.index("s_addc_u32")  # the carry still reaches its s_addc_u32


@pytest.mark.parametrize("between", [
    "\ts_add_i32 s0, s0, 16",                    # writes SCC between the hoisted add and the s_addc that consumes its carry
    "\ts_mov_b32 s2, s9",                        # rewrites the other addend
    "\tv_add_u32_e32 v0, s8, v4",                # reads the copy
])
def test_copy_is_not_folded_across_scc_writers_or_users(between):
    n, out = run("\ts_mov_b32 s8, s4\n" + between + "\n\ts_add_u32 s8, s2, s8\n\ts_addc_u32 s9, s3, 0\n", asmopt.fold_copy_into_add)
    assert n == 0 and "s_mov_b32 s8, s4" in out


def test_missing_resource_symbol_is_appended_once(tmp_path):
    src, dst = tmp_path / "a.s", tmp_path / "b.s"
    src.write_text("\t.set k.num_named_barrier, max(0, amdgpu.max_num_named_barrier)\n\t.set amdgpu.max_num_vgpr, 62\n")
    asmopt.main(str(src), str(dst), "")
    out = dst.read_text()
    assert out.count(".set amdgpu.max_num_named_barrier, 0") == 1
    asmopt.main(str(dst), str(src), "")
    assert src.read_text().count(".set amdgpu.max_num_named_barrier, 0") == 1
