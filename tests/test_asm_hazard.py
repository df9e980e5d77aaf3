"""The static hazard check that guards the inline-asm LDS reads (scripts/check_asm_hazard.py):
it must flag a copy of an in-flight destination, follow branches, honour partial waits, and pass the ISA of the
kernels that were built into the shipped library."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_asm_hazard as H  # noqa: E402

HEAD = "\t.text\nkern:\n"


def _scan(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text(HEAD + body + "\ts_endpgm\n.Lfunc_end0:\n")
    return [l for _, _, l in H.scan(str(p))]


def test_copy_of_in_flight_tile_is_flagged(tmp_path):
    bad = _scan(tmp_path, "\tds_read_b128 v[20:23], v79 offset:0\n\tv_mfma_f32_32x32x2_f32 a[0:15], v2, v0, a[0:15]\n"
                          "\tv_accvgpr_write_b32 a117, v23\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v1, v20\n")
    assert bad == ["v_accvgpr_write_b32 a117, v23"]


def test_partial_wait_retires_the_oldest_reads_only(tmp_path):
    body = ("\tds_read_b128 v[4:7], v1 offset:0\n\tds_read_b128 a[8:11], v1 offset:16\n\ts_waitcnt lgkmcnt(1)\n"
            "\tv_add_f32_e32 v2, v4, v5\n\tv_accvgpr_read_b32 v3, a9\n")
    assert _scan(tmp_path, body) == ["v_accvgpr_read_b32 v3, a9"]


def test_overwrite_of_a_pending_destination_is_flagged_and_reissue_is_not(tmp_path):
    body = ("\tds_read_b128 v[4:7], v1 offset:0\n\tds_read_b128 v[4:7], v1 offset:64\n\tv_mov_b32_e32 v6, 0\n"
            "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_mov_b32_e32 v6, 0\n")
    assert _scan(tmp_path, body) == ["v_mov_b32_e32 v6, 0"]


def test_state_follows_branches_not_layout(tmp_path):
    # the read is waited for on the fall-through path only; the taken path reaches the use with the read in flight
    body = ("\tds_read_b128 v[4:7], v1 offset:0\n\ts_cbranch_scc1 .LBB0_2\n\ts_waitcnt lgkmcnt(0)\n\ts_branch .LBB0_3\n"
            ".LBB0_2:\n\tv_mov_b32_e32 v9, v5\n.LBB0_3:\n\tv_mov_b32_e32 v8, v4\n")
    assert sorted(_scan(tmp_path, body)) == ["v_mov_b32_e32 v8, v4", "v_mov_b32_e32 v9, v5"]
    # a loop whose back edge carries a pending read into the header
    body = (".LBB0_1:\n\tv_mov_b32_e32 v8, v4\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b128 v[4:7], v1 offset:0\n"
            "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt lgkmcnt(0)\n")
    assert _scan(tmp_path, body) == ["v_mov_b32_e32 v8, v4"]


def test_shipped_kernels_use_no_scratch_and_spill_nothing():
    """Every kernel of every translation unit: private segment 0 bytes, 0 spilled VGPRs, 0 spilled SGPRs (the
    metadata the compiler writes next to the ISA that went into the shipped library)."""
    import check_no_scratch as S
    isa = sorted(glob.glob(os.path.join(ROOT, "dm_nerf_amd", "csrc", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not isa:
        pytest.skip("no kernel ISA in the tree (the library was not built here)")
    total = 0
    for f in isa:
        ks = S.kernels(f)
        total += len(ks)
        assert S.violations(f) == [], f
    assert total >= 46


def test_shipped_kernels_are_hazard_free():
    isa = sorted(glob.glob(os.path.join(ROOT, "dm_nerf_amd", "csrc", "build", "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not isa:
        pytest.skip("no kernel ISA in the tree (the library was not built here)")
    for f in isa:
        assert H.scan(f) == [], f
