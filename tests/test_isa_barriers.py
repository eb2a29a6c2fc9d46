"""tools/isa_barrier_check.py: no kernel may reach an s_barrier with an LDS store still in flight (hipcc's
wait-count pass dropped the wait in front of an in-loop barrier of the tilebook builder's sort — one corrupted rulebook list in 400
builds at 2 M voxels).  CPU-only: the kernels are cross-compiled to gfx950 ISA and their control-flow graphs analysed."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("isa_barrier_check", os.path.join(ROOT, "tools", "isa_barrier_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_analysis_sees_a_store_that_reaches_a_barrier_over_a_back_edge():
    """The shape of the miscompiled sort: the barrier is the loop header, the conditional ds_write pair sits in a block of its own
    further down, and nothing waits for it on the way back.  With the wait in front of the barrier the report is gone."""
    tool = _tool()
    bad = """_Z4sortv:
	s_waitcnt lgkmcnt(0)
	s_barrier
.LBB0_1:
	s_barrier
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_3
	ds_read_b32 v4, v2
	ds_read_b32 v5, v3
	s_waitcnt lgkmcnt(0)
	v_cmp_gt_u32_e64 s[2:3], v4, v5
	s_and_saveexec_b64 s[4:5], s[2:3]
	s_cbranch_execz .LBB0_3
	ds_write_b32 v2, v5
	ds_write_b32 v3, v4
.LBB0_3:
	s_or_b64 exec, exec, s[0:1]
	s_lshr_b32 s6, s6, 1
	s_cmp_lt_u32 s6, 1
	s_cbranch_scc0 .LBB0_1
	s_waitcnt lgkmcnt(0)
	s_barrier
	s_endpgm""".splitlines()
    rep = tool.check_kernel(bad[0], bad)
    assert list(rep) == [4] and "ds_write_b32" in bad[rep[4]]            # the in-loop barrier, not the two waited ones
    good = list(bad)
    good.insert(4, "\ts_waitcnt lgkmcnt(0)")
    assert tool.check_kernel(good[0], good) == {}
    raw = [l.replace("s_waitcnt lgkmcnt(0)", "s_waitcnt 0xc07f") for l in good]     # (a raw immediate: lgkmcnt = bits 11:8 = 0)
    assert tool.check_kernel(raw[0], raw) == {}


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_library_kernels_reach_no_barrier_with_a_pending_lds_store():
    """The sources whose barriers the compiler places on its own (the tilebook builder first of all); the conv / weight-gradient
    files take minutes to compile and are checked by `python tools/isa_barrier_check.py` (all files: zero reports at this commit,
    and exactly one — tilebook_build — at the revision before the fix)."""
    tool = _tool()
    for name in ("tilebook", "rulebook", "voxelize_idx", "core", "glue", "loss"):
        assert tool.check_file(os.path.join(ROOT, "doda_amd", "csrc", name + ".hip")) == 0, name


def test_the_analysis_tracks_lds_dma_by_vmcnt():
    """`buffer_load ... lds` writes LDS under vmcnt: lgkmcnt(0) in front of the barrier does not publish it, vmcnt(0) does."""
    tool = _tool()
    k = """_Z3dmav:
	s_mov_b32 m0, s4
	buffer_load_dwordx4 v1, s[8:11], 0 offen lds
	s_waitcnt lgkmcnt(0)
	s_barrier
	s_waitcnt vmcnt(0)
	s_barrier
	s_endpgm""".splitlines()
    assert list(tool.check_kernel(k[0], k, "dma")) == [4]
    assert tool.check_kernel(k[0], k, "store") == {}


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_wgrad_dma16_publishes_its_staged_buffers_behind_vmcnt0():
    """The LDS-DMA weight-gradient kernel (spconv_wdma.hip) is double-buffered: of its eleven barriers exactly the eight of the two
    inlined flush() sites run with the next item's DMA in flight (they exchange through the buffer that DMA does not target); the three
    that publish a staged buffer -- prologue and the two loop tops -- are behind s_waitcnt vmcnt(0)."""
    tool = _tool()
    assert tool.check_file(os.path.join(ROOT, "doda_amd", "csrc", "spconv_wdma.hip"), verbose=False) == 0
    assert tool.check_file.last_dma_barriers == 8


def test_every_source_barrier_is_doda_sync():
    """No kernel source calls __syncthreads() itself: common.hpp's doda_sync() (explicit s_waitcnt lgkmcnt(0) + barrier) is the only
    spelling, so the compiler's wait-count pass is never what an LDS exchange depends on.  (Hand-written `s_barrier` in inline asm
    carries its own wait on the same asm statement.)"""
    import glob
    import re
    bad = []
    for f in sorted(glob.glob(os.path.join(ROOT, "doda_amd", "csrc", "*.h*"))):
        for n, line in enumerate(open(f), 1):
            code = line.split("//")[0]
            if "__syncthreads" in code and not (f.endswith("common.hpp") and "doda_sync" not in code):
                bad.append("%s:%d" % (os.path.basename(f), n))
            if re.search(r"s_barrier", code) and "s_waitcnt" not in code and "asm" in code:
                bad.append("%s:%d (bare s_barrier in inline asm)" % (os.path.basename(f), n))
    assert not bad, bad


def _ring_tool():
    spec = importlib.util.spec_from_file_location("isa_ring_check", os.path.join(ROOT, "tools", "isa_ring_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_ring_analysis_sees_a_compiler_load_between_ring_loads():
    """conv_fast's unit ring: inline-asm loads with hand-counted vmcnt waits.  A compiler-visible load sunk between two of them (round
    6: the epilogue's operands requested in front of the loop) makes the count wrong; in front of the first / behind the last it is fine."""
    tool = _ring_tool()
    ring = "\tbuffer_load_dwordx4 v[22:25], v26, s[16:19], s4 offen"
    ok = ["\tbuffer_load_dwordx2 v[46:47], v4, s[4:7], 0 offen", ring, "\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]", ring,
          "\ts_waitcnt vmcnt(0)", "\tglobal_load_dwordx4 v[14:17], v[6:7], off", "\tbuffer_store_dwordx2 v[4:5], v10, s[12:15], 0 offen"]
    assert tool.check_kernel(ok) == []
    bad = [ring, "\tbuffer_load_dwordx2 v[46:47], v4, s[4:7], 0 offen", ring, "\tglobal_load_dwordx4 v[14:17], v[6:7], off", ring]
    assert [k for k, _ in tool.check_kernel(bad)] == [1, 3]


def test_conv_fast_rings_hold_only_ring_loads():
    """Every conv_fast instantiation of the BUILT library object (264 kernels): zero foreign vector-memory instructions inside a ring."""
    tool = _ring_tool()
    obj = os.path.join(ROOT, "doda_amd", "csrc", "_obj", "spconv_gather.o")
    if not (os.path.exists(obj) and os.path.exists(tool.OBJDUMP)):
        pytest.skip("no built object / llvm-objdump")
    n, report = tool.check_object(obj)
    assert n >= 100, n
    assert report == {}, {k: v[:3] for k, v in report.items()}
