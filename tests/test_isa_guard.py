"""Compiler-drift guard: read the gfx950 code objects of the built ``libgs_hip.so`` (no GPU needed).

SURVEY.md §7 ("hard parts"): bit-exact parity with ``compute/naive/src/lib.rs:63-79`` rests on every multiply and add
being rounded separately, so the strict kernels must not contain a single fused multiply-add -- whatever a future
compiler or a flag change would like to contract.  The occupancy the production kernels are tuned for is forced with
``amdgpu_waves_per_eu``, which turns a register-allocation regression into silent spills: the shipped entries must
stay spill-free.  The float mode of the kernel descriptor is the GPU's ``DenormalsFlusher``
(``compute/shared/src/lib.rs:123-207``): mode 1 (flush results, keep inputs) on strict, 3 (keep) on fused.

Red when the strict translation units are built with ``-ffp-contract=fast`` (checked by hand when this test was
written: 3 000+ ``v_fmac_f32`` / ``v_fma_f32`` appear in ``gs_step_tb_dx_k_strict<4, 4>``).
"""
from __future__ import annotations

import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import codeobj  # noqa: E402


@pytest.fixture(scope="module")
def kernels(built):
    ks = codeobj.kernels()
    assert ks, "no gfx950 code object in libgs_hip.so"
    return {k.name: k for k in ks}


def strict(kernels):
    return [k for k in kernels.values() if "_strict" in k.name]


def fused(kernels):
    return [k for k in kernels.values() if "_fused" in k.name]


def test_every_translation_unit_is_there(kernels):
    names = set(kernels)
    # one kernel of each translation unit: strict, strict parameter-specialised, fused, utilities
    for probe in ("gs_step_stream_k_strict<2>", "gs_step_tb_dx_k_strict<4, 4>", "gs_step_stream_k_fused<2>",
                  "gs_fill_rect_k"):
        assert probe in names, (probe, sorted(names))
    assert len(strict(kernels)) >= 60 and len(fused(kernels)) >= 20


def test_strict_kernels_hold_no_fused_multiply_add(kernels):
    """The bit-exact contract: ``acc + w*(s-c)``, ``Du*acc - uvv``, ``u + du*dt`` ... each op rounded once."""
    bad = {}
    for k in strict(kernels):
        assert len(k.insts) > 100, (k.name, len(k.insts))          # the disassembly really is this kernel's
        hits = k.matching(codeobj.FLOAT_FMA)
        if hits:
            bad[k.name] = hits[:3]
    assert not bad, bad


def test_the_fma_pattern_sees_fmas(kernels):
    """The pattern above is not vacuous: the fused flavour is made of exactly those instructions."""
    for k in fused(kernels):
        assert k.count(codeobj.FLOAT_FMA) > 0, k.name
    for text in ("v_fmac_f32_e32 v1, v2, v3", "v_fma_f32 v1, v2, v3, v4", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]",
                 "v_fmamk_f32 v1, v2, 0x3dcccccd, v3", "v_fmaak_f32 v1, v2, v3, 0x3dcccccd", "v_mad_f32 v1, v2, v3, v4",
                 "v_fma_mix_f32 v1, v2, v3, v4", "v_mac_f32_e32 v1, v2, v3", "v_pk_fma_f16 v1, v2, v3, v4",
                 "v_fmac_f32_dpp v1, v2, v3 wave_shl:1"):
        assert re.match(codeobj.FLOAT_FMA, text), text
    for text in ("v_mad_u64_u32 v[0:1], s[2:3], v2, v3, v[4:5]", "v_mad_u32_u24 v1, v2, v3, v4", "v_mul_f32_e32 v1, v2, v3",
                 "v_add_f32_e32 v1, v2, v3", "v_sub_f32_e64 v1, v2, v3 div:2", "v_mad_i32_i24 v1, v2, v3, v4"):
        assert not re.match(codeobj.FLOAT_FMA, text), text


def test_float_mode_is_the_denormals_flusher(kernels):
    """x86 MXCSR.FTZ without DAZ = FP_DENORM single 1 (flush results, keep inputs); the fused flavour keeps both."""
    for k in strict(kernels):
        assert k.denorm_mode_32 == 1, (k.name, k.denorm_mode_32)
        assert k.round_mode_32 == 0, k.name                       # round to nearest even
    for k in fused(kernels):
        assert k.denorm_mode_32 == 3, (k.name, k.denorm_mode_32)
        assert k.round_mode_32 == 0, k.name


# The entries the library launches by default: name -> (max VGPRs, LDS bytes as DESIGN states them)
SHIPPED = {
    "gs_step_tb_dx_k_strict<4, 4>": (128, 16640),      # gs_run at 16384^2: 4 waves per SIMD, halo board 4 x 4160 B
    "gs_step_tb_dx_k_strict<4, 16>": (128, 66688),     # one-round grids: 16 waves in step (+ the progress board)
    "gs_step_tb_dx_k_strict<3, 4>": (128, 12480),      # remainder passes of 3 / 2 steps
    "gs_step_tb_dx_k_strict<2, 4>": (128, 8320),
    "gs_step_tb_k_strict<1, 3, 2, 4>": (128, 0),       # ... and of 1 step
    "gs_step_stream_k_strict<2>": (128, 0),            # gs_step: the HBM-bound single-step kernel
    "gs_run_tile_k_strict<4, 3>": (128, None),         # small grids
    "gs_run_resident_k_strict<3, 1>": (128, None),
}


@pytest.mark.parametrize("name", sorted(SHIPPED))
def test_shipped_kernels_are_spill_free(kernels, name):
    k = kernels[name]
    max_vgpr, lds = SHIPPED[name]
    assert k.vgpr <= max_vgpr and k.agpr == 0, (name, k.vgpr, k.agpr)
    assert k.vgpr_spill == 0 and k.sgpr_spill == 0, (name, k.vgpr_spill, k.sgpr_spill)
    assert k.scratch == 0 and not k.dynamic_stack, (name, k.scratch)
    assert k.count(r"^scratch_") == 0, name
    # SGPR spills go to VGPR lanes: none may be left
    assert k.count(r"^v_(readlane|writelane)_b32") == 0, (name, k.count(r"^v_(readlane|writelane)_b32"))
    if lds is not None:
        assert k.lds == lds, (name, k.lds)


def test_window_kernel_has_no_scratch_and_no_spill_traffic_in_its_step_loops(kernels):
    """gs_run_window_k (the reference's default 1080 x 1920 in long calls): eight kinds of window, one branch each, in one
    persistent kernel.  Values that only the exchange between super-steps needs may be parked in VGPR lanes around the
    step loops (SGPR spills: v_writelane before, v_readlane after), but nothing may go to scratch and no step loop -- the
    innermost loops around a step: rows published to LDS, the wave's wait for the waves above and below it, its priority
    set by how it stands to them -- may contain a lane read or write (round 5 had 528 SGPR spills, 2 VGPR spills, 8 B of
    scratch and up to 25 v_readlane per step in the corner windows) or a workgroup barrier (one per step until round 6:
    the LDS read burst of 16 waves in lock-step and the lone last wave in front of it cost 8 % of a step)."""
    for name in ("gs_run_window_k_strict<5, 7>", "gs_run_window_k_strict<5, 3>", "gs_run_window_k_strict<5, 0>", "gs_run_window_k_fused<5, 0>"):
        k = kernels[name]
        assert k.vgpr <= 128 and k.agpr == 0 and k.vgpr_spill == 0 and k.scratch == 0 and not k.dynamic_stack, (name, k.vgpr, k.vgpr_spill, k.scratch)
        assert k.count(r"^scratch_") == 0, name
        lane_ops = r"^v_(readlane|writelane)_b32"
        # a step loop: rows published (ds_write2), the wave's priority set behind its wait for its neighbours, hundreds of VALU
        # instructions, and none of the exchange's stores to the exchange planes
        step_like = [l for l in k.loops() if sum(t.startswith("s_setprio") for t in l) >= 3 and sum(t.startswith("ds_write2") for t in l) >= 4
                     and sum(t.startswith("v_") for t in l) >= 300 and not any(t.startswith("buffer_store") for t in l)]
        assert len(step_like) >= 6, (name, len(step_like))             # every kind of window has one
        # innermost: no other step-like loop is a proper part of it
        inner = [l for l in step_like if not any(len(m) < len(l) and " ".join(m) in " ".join(l) for m in step_like)]
        assert len(inner) >= 3, (name, len(inner))
        for l in inner:
            assert sum(bool(re.match(lane_ops, t)) for t in l) == 0, (name, len(l))
            assert not any(t.startswith("s_barrier") for t in l), (name, len(l))
            assert not any(t.startswith(("flat_", "global_", "scratch_")) for t in l), (name, len(l))
        # ... and what surrounds a step loop inside a super-step (executed once per K steps) reads back a few lanes at most
        for l in step_like:
            assert sum(bool(re.match(lane_ops, t)) for t in l) <= 6, (name, len(l))


def test_no_kernel_touches_scratch(kernels):
    """Scratch is a register-allocation regression, never a design choice -- in either flavour (round 5 shipped
    gs_step_tb_k_fused<4, 0, 2, 16> with one spilled register: that 16-wave variant is gone)."""
    bad = {k.name: (k.scratch, k.vgpr_spill) for k in kernels.values() if k.scratch or k.vgpr_spill or k.count(r"^scratch_")}
    assert not bad, bad
    assert "gs_step_tb_k_fused<4, 0, 2, 16>" not in kernels and "gs_step_tb_k_fused<4, 0, 1, 16>" in kernels


def test_production_march_keeps_four_waves_per_simd(kernels):
    """512 VGPRs per SIMD lane / 4 waves = 128: one register more and the forced occupancy spills instead."""
    for name, k in kernels.items():
        if name.startswith(("gs_step_tb_dx_k_strict", "gs_step_tb_ds_k_strict")):
            assert k.vgpr <= 128 and k.vgpr_spill == 0 and k.sgpr_spill == 0 and k.scratch == 0, (name, k.vgpr)
