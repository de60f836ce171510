"""Static guard for the frame kernel's register allocation (no GPU needed: hipcc cross-compiles).

The frame time follows the DYNAMIC spill count: one reload inside a depth-3 loop cost 46 % in
round 2, and harmless-looking source changes move spills into the march / walk
loops.  This test compiles the kernels to gfx950 assembly with the product flags and checks, for the
default instantiation of render_frame_kernel (7 waves/SIMD, row-major tables), where the spills
sit and that the launch resources are the ones the occupancy argument of DESIGN.md rests on."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
# the bench instantiations: accelerated, 7 waves/SIMD, table layout 5 (row-major tables of the 256^3
# grid, edge compiled in), arithmetic contracts gfx950-default (library and bench default, ArithOf index 3),
# gfx950-strict (2) and cpu (0) -- and layout 2, the same with the edge read at run time (other cubic
# power-of-two grids)
KEYS = ["render_frame_kernelILb1ELi7ELb0ELi5ELi3E", "render_frame_kernelILb1ELi7ELb0ELi5ELi2E",
        "render_frame_kernelILb1ELi7ELb0ELi5ELi0E", "render_frame_kernelILb1ELi7ELb0ELi2ELi3E",
        "render_frame_kernelILb1ELi7ELb0ELi2ELi2E", "render_frame_kernelILb1ELi7ELb0ELi2ELi0E"]
KEY = KEYS[0]
# spilled VGPRs: round 5 found the frame time to follow the allocation (70 -> 30 spilled VGPRs = -9 %);
# the bench instantiation must not drift back (tools/spill_screen.py finds the site when it does)
SPILL_BUDGET = {KEYS[0]: 36}


@pytest.fixture(scope="module")
def frame_kernel_asm(tmp_path_factory):
    from raymarchcl_amd import _native

    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "k.s"
    flags = [f for f in _native.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", os.path.join(_native.CSRC, "rm_kernels.hip"),
                                       "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


@pytest.mark.parametrize("key", KEYS)
def test_spills_stay_out_of_the_inner_loops(frame_kernel_asm, key):
    import isa_spills

    res = isa_spills.analyse(frame_kernel_asm.split("\n"), key)
    assert res is not None, "frame kernel instantiation not found"
    n_ins, hist, lanes, per_loop = res
    deep = {d: v for d, v in hist.items() if d >= 3}
    assert not deep, f"scratch traffic inside depth >= 3 loops: {deep}"
    at2 = hist.get(2, {"load": 0, "store": 0})
    assert at2["load"] + at2["store"] <= 4, f"scratch traffic inside depth-2 loops: {at2}"
    assert 7000 < n_ins < 12000  # the kernel the profiles describe, not a different shape


@pytest.mark.parametrize("key", KEYS)
def test_launch_resources_of_the_default_kernel(frame_kernel_asm, key):
    # kernel descriptor metadata of the bench instantiations
    m = None
    for blk in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: +\d+", frame_kernel_asm, re.S):
        if key in blk.group(0):
            m = blk.group(0)
    assert m, "metadata block not found"
    get = lambda k: int(re.search(r"\." + k + r": +(\d+)", m).group(1))
    assert get("vgpr_count") <= 72          # 7 wavefronts per SIMD
    assert get("agpr_count") == 0
    assert get("group_segment_fixed_size") <= 5851  # 160 KB / 28 wavefronts per CU
    assert get("private_segment_fixed_size") <= 128
    assert get("vgpr_spill_count") <= SPILL_BUDGET.get(key, 48)
    assert get("sgpr_spill_count") == 0     # no v_writelane / v_readlane spill carriers (round 2: 61)
    assert get("wavefront_size") == 64


def test_no_spill_reload_in_a_block_entered_with_exec_zero(frame_kernel_asm):
    """The code-generation fault DESIGN_HISTORY.md 4c pins down (round 3, with rocgdb): the register allocator puts
    the reload of a spilled VGPR into the exit block of a loop it lowered as divergent -- a block entered
    through `s_cbranch_execz`, with NO lane enabled, in front of the `s_or_b64 exec` that brings the lanes
    back.  The reload reaches nobody; the lanes go on with whatever the loop used the register for (seen:
    the per-lane LDS slot address -> owners post into nowhere -> helpers fetch through a pointer made of
    leftovers -> memory aperture violation; earlier in the round the same shape as wrong pixels).  No
    kernel of the library may contain that shape (tools/isa_exec_lint.py; every kernel, not only the
    bench instantiations)."""
    from raymarchcl_amd import isa_exec_lint

    lines = frame_kernel_asm.split("\n")
    fatal, kernels = [], 0
    for name, lo, hi in isa_exec_lint.kernels(lines):
        kernels += 1
        for off, reload, restore, dead in isa_exec_lint.lint(lines, lo, hi):
            if dead:
                fatal.append((name, off, reload))
        # the same for ANY vector instruction (the allocator's live-range copies land there too)
        fatal += [(name, off, ins) for off, ins in isa_exec_lint.dead_vector_instructions(lines, lo, hi)]
    assert kernels >= 30            # all instantiations of rm_kernels.hip were looked at
    assert not fatal, fatal


def test_no_kernel_of_the_library_spills_an_sgpr(frame_kernel_asm):
    """Round 2 blamed spilled SGPRs (spill lanes of a VGPR also handed to a vector value) for the fault that
    round 3 traced to misplaced spill code; never confirmed, but since the in-kernel loop over pass groups is
    gone no kernel of the library spills an SGPR, and it stays that way."""
    n = 0
    for blk in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", frame_kernel_asm, re.S):
        n += 1
        assert int(re.search(r"sgpr_spill_count:\s+(\d+)", blk.group(2)).group(1)) == 0, blk.group(1)
    assert n >= 30


def test_the_lint_recognises_the_shape_and_nothing_else():
    """tools/isa_exec_lint.py on hand-written assembly: the fatal shape (reload in a loop-exit block that is
    only entered through s_cbranch_execz, in front of the exec restore), the same with the exit reached by
    falling out of an s_cbranch_execnz loop, and two harmless neighbours (a reload after the restore; a
    reload at the end of an `if` region under the region's own mask)."""
    from raymarchcl_amd import isa_exec_lint

    def kernel(body):
        return ("_Zk:\n" + body + "\n.Lfunc_end0:\n").split("\n")

    fatal = """
.LBB0_1:
	v_add_f32_e32 v1, v1, v2
	s_andn2_b64 exec, exec, s[4:5]
	s_cbranch_execz .LBB0_2
	s_branch .LBB0_1
.LBB0_2:
	scratch_load_dword v1, off, off offset:4 ; 4-byte Folded Reload
.LBB0_3:
	s_or_b64 exec, exec, s[6:7]
	s_endpgm"""
    fallthrough = """
.LBB0_1:
	v_add_f32_e32 v1, v1, v2
	s_andn2_b64 exec, exec, s[4:5]
	s_cbranch_execnz .LBB0_1
.LBB0_2:
	scratch_load_dword v1, off, off offset:4 ; 4-byte Folded Reload
	s_or_b64 exec, exec, s[6:7]
	s_endpgm"""
    after_restore = """
.LBB0_1:
	s_andn2_b64 exec, exec, s[4:5]
	s_cbranch_execz .LBB0_2
	s_branch .LBB0_1
.LBB0_2:
	s_or_b64 exec, exec, s[6:7]
	scratch_load_dword v1, off, off offset:4 ; 4-byte Folded Reload
	s_endpgm"""
    end_of_if = """
	s_and_saveexec_b64 s[6:7], vcc
	s_cbranch_execz .LBB0_2
	v_mov_b32_e32 v1, v3
	scratch_load_dword v1, off, off offset:4 ; 4-byte Folded Reload
.LBB0_2:
	s_or_b64 exec, exec, s[6:7]
	s_endpgm"""
    copy = fatal.replace("scratch_load_dword v1, off, off offset:4 ; 4-byte Folded Reload", "v_mov_b32_e32 v1, v9")
    for body, want in ((fatal, 1), (fallthrough, 1), (after_restore, 0), (end_of_if, 0)):
        lines = kernel(body)
        (name, lo, hi), = list(isa_exec_lint.kernels(lines))
        assert sum(1 for x in isa_exec_lint.lint(lines, lo, hi) if x[3]) == want, body
        assert len(isa_exec_lint.dead_vector_instructions(lines, lo, hi)) == want, body
    lines = kernel(copy)  # an allocator copy instead of a reload: not a reload, but just as dead
    (name, lo, hi), = list(isa_exec_lint.kernels(lines))
    assert not isa_exec_lint.lint(lines, lo, hi) and len(isa_exec_lint.dead_vector_instructions(lines, lo, hi)) == 1
