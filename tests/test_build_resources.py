"""What the hand-scheduled kernels assume about their own register allocation, checked on the BUILT library's code-object metadata (CPU test).

ADVICE r4 (medium): `gemm_nt_256w4` is written as the only wave on its SIMD; with fewer than 512 registers a low-register wave of a kernel on another
stream can be placed beside it (round 4: sporadic NaNs).  The kernel claims the whole file through an asm clobber of v255 — a compiler or flag change
that drops it must fail here, not in a training run."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import kernel_resources as kr  # noqa: E402

SO = os.path.join(os.path.dirname(__file__), "..", "visper-lm_amd", "libvisper_hip.so")


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(SO):
        pytest.skip("library not built")
    k = kr.kernels(SO)
    assert len(k) > 50, "could not read the code objects' metadata"
    return k


def test_one_wave_per_simd_kernels_own_the_whole_register_file(kernels):
    w4 = {n: d for n, d in kernels.items() if "gemm_nt_256w4" in n}
    assert len(w4) >= 4
    for n, d in w4.items():
        assert d["vgpr_count"] == 512 and d["agpr_count"] == 256, (n, d)
        assert d.get("vgpr_spill_count", 0) == 0 and d.get("private_segment_fixed_size", 0) == 0, (n, d)
    # the one-wave-per-SIMD attention backward kernels (round 5) are written the same way.  (A few prologue values that cross the fenced loops may
    # be spilled OUTSIDE them; the ISA audit below checks that nothing spills between the barriers.)
    for n, d in kernels.items():
        if "attn_bwd_dkdv64w" in n or "attn_bwd_dq64w" in n:
            assert d["vgpr_count"] == 512 and d["agpr_count"] == 256, (n, d)
            assert d.get("vgpr_spill_count", 0) <= 16, (n, d)


def test_register_resident_row_kernels_do_not_go_to_scratch(kernels):
    """only attention.hip / gemm.hip are built with -amdgpu-spill-vgpr-to-agpr=0 now; the cross-entropy kernel that keeps its row in
    registers (128 data VGPRs under launch_bounds(512)) and the 4-wide AdamW must not touch scratch (a spill may use a free AGPR)."""
    hit = 0
    for n, d in kernels.items():
        if "ce_fwd_bwd_reg_kernel" in n or "adamw4_kernel" in n:
            hit += 1
            assert d.get("private_segment_fixed_size", 0) == 0, (n, d)
    assert hit >= 2, hit


def test_dma_ring_kernels_do_not_spill(kernels):
    """a scratch reload inside a kernel that keeps an LDS-DMA ring in flight is followed by s_waitcnt vmcnt(0): it drains the ring every iteration"""
    for n, d in kernels.items():
        if any(t in n for t in ("attn_bwd_dkdv128", "attn_bwd_dq128", "attn_fwd128m")):
            assert d.get("vgpr_spill_count", 0) == 0 and d.get("private_segment_fixed_size", 0) == 0, (n, d)


def test_asm_owned_registers_are_never_touched_by_the_compiler():
    """csrc/attention_bwd64.h names v64..v255 and every AGPR literally inside asm statements; tools/audit_asm_owned.py compiles attention.hip to ISA
    and checks that no compiler-generated instruction uses them while they are live, that nothing spills inside the streams and that no scalar
    load sits inside the counted-lgkmcnt regions (needs hipcc: ~1 min)."""
    import shutil
    import audit_asm_owned as au
    if not (shutil.which("hipcc") or os.path.exists(au.HIPCC)):
        pytest.skip("needs hipcc")
    for dbg in (False, True):                      # the sealed product build AND the -DVP_DEBUG build
        n, problems = au.audit(au.isa(debug=dbg))
        assert n >= 12, (dbg, n)
        assert not problems, (dbg, problems[:10])


def test_w4_epilogue_accumulator_reads_never_overwrite_pending_store_data():
    """Round 6, the root cause of round 4's "NaNs beside other kernels": an `asm volatile("v_accvgpr_read_b32 ...")` of the one-wave-per-SIMD GEMM's
    epilogue (opaque to the compiler's hazard recogniser) was allocated the VGPR holding dword 0 of the buffer_store issued just before; when the CU's
    memory pipeline is shared with another kernel's waves the store fetches its data late and writes the next block's raw accumulator
    (tools/nan_pattern_r06.py, profiles/r06_nan_root_cause.txt).  gemm.hip's W4_KEEP2 keeps the previous block's store-data registers live across the
    next block's accumulator reads; tools/w4_store_data_audit.py checks the built ISA of every instantiation: no asm accumulator read writes a data
    register of the last two 16-byte stores (0 with the fix, 647-1247 per instantiation without; needs hipcc: ~1 min)."""
    import shutil
    import w4_store_data_audit as au
    if not (shutil.which("hipcc") or os.path.exists(au.HIPCC)):
        pytest.skip("needs hipcc")
    res = au.audit(au.isa())
    assert len(res) >= 4, list(res)
    assert sum(r[0] for r in res.values()) >= 3000 and sum(r[1] for r in res.values()) >= 400          # the audit saw the epilogues
    for name, (_, _, probs) in res.items():
        assert not probs, (name, probs[:5])
