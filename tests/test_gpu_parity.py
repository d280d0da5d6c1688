"""Parity tests proper: the HIP path, called through the C ABI (libgs_hip.so), against the CPU
oracle on the same inputs -- BIT EXACT (memcmp) for the strict flavour, which is a stronger
statement than north_star's tolerance of 1e-5 relative.  The fused flavour is held to
bit-exactness wherever no sub-normal intermediate occurs and to |diff| <= 1e-37 elsewhere.

All tests need a real MI355X:  python -m pytest tests -m gpu
"""
import glob
import os

import numpy as np
import pytest

import oracle
from grayscott_amd import GsError, HipArgs, Parameters, Simulation, capi
from tests.helpers import (STRESS_SHAPES, assert_bits_equal, gpu_run, oracle_params,
                           species_from_arrays, stress_fields)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-5  # north_star: "results within 1e-5 rel of the naive CPU reference"


def args(**kw):
    kw.setdefault("devices", [0])
    return HipArgs(**kw)


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    assert capi.device_count() >= 1, "no MI355X visible"


# ---- small and ragged shapes, random data ---------------------------------------------------
@pytest.mark.parametrize("kernel", [capi.GS_KERNEL_SIMPLE, capi.GS_KERNEL_STREAM, capi.GS_KERNEL_LDS])
@pytest.mark.parametrize("shape", STRESS_SHAPES + [(5, 256), (9, 257), (33, 255), (6, 1024), (40, 1030)])
def test_stress_shapes_bit_exact(shape, kernel):
    for seed in (0, 1, 2):
        u0, v0 = stress_fields(shape, seed)
        for steps in (1, 7):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(kernel=kernel))
            assert_bits_equal(got_u, ref_u, f"U {shape} seed {seed} steps {steps} {info[0]}")
            assert_bits_equal(got_v, ref_v, f"V {shape} seed {seed} steps {steps} {info[0]}")


@pytest.mark.parametrize("rpb", [1, 2, 3, 8, 64])
def test_rows_per_block_does_not_change_results(rpb):
    u0, v0 = stress_fields((77, 600), 3)
    ref_u, ref_v = oracle.run(u0, v0, 5, ftz=True)
    got_u, got_v, _ = gpu_run(u0, v0, 5, args=args(kernel=capi.GS_KERNEL_STREAM, rows_per_block=rpb))
    assert_bits_equal(got_u, ref_u, f"U rpb {rpb}")
    assert_bits_equal(got_v, ref_v, f"V rpb {rpb}")


# ---- committed golden vectors (no oracle needed at run time) -------------------------------
def test_golden_vectors():
    for path in sorted(glob.glob(os.path.join(GOLDEN, "stress_*.npz"))):
        g = np.load(path)
        for steps in (1, 20):
            got_u, got_v, _ = gpu_run(g["u0"], g["v0"], steps)
            assert_bits_equal(got_u, g[f"u_{steps}"], f"{os.path.basename(path)} U {steps}")
            assert_bits_equal(got_v, g[f"v_{steps}"], f"{os.path.basename(path)} V {steps}")
    g = np.load(os.path.join(GOLDEN, "species_new_64x128.npz"))
    sim = Simulation.new(Parameters(), args())
    species = sim.make_species([64, 128])
    done = 0
    for steps in (1, 10, 100, 1000):
        sim.perform_steps(species, steps - done)
        done = steps
        in_u, in_v, _, _ = species.in_out()
        assert_bits_equal(in_u.make_scalar_view(sim.context), g[f"u_{steps}"], f"species_new U {steps}")
        assert_bits_equal(in_v.make_scalar_view(sim.context), g[f"v_{steps}"], f"species_new V {steps}")


@pytest.mark.parametrize("kw", [dict(kernel=capi.GS_KERNEL_SIMPLE), dict(kernel=capi.GS_KERNEL_STREAM), dict(kernel=capi.GS_KERNEL_LDS),
                                dict(kernel=capi.GS_KERNEL_TB), dict(kernel=capi.GS_KERNEL_TB, cols_per_lane=1, rows_per_block=4),
                                dict(kernel=capi.GS_KERNEL_TB, cols_per_lane=2, fuse_steps=3), dict(kernel=capi.GS_KERNEL_TB, cols_per_lane=4, fuse_steps=2),
                                dict(kernel=capi.GS_KERNEL_TILE, tile_shape=1), dict(kernel=capi.GS_KERNEL_TILE, tile_shape=2),
                                dict(kernel=capi.GS_KERNEL_TILE, tile_shape=3, fuse_steps=5), dict(devices=[0, 0, 0])],
                         ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_golden_vectors_every_kernel(kw):
    """The committed fixtures (inputs and expected outputs, no oracle at run time) through every kernel
    family and a slab chain, not only through what kernel = auto picks for their sizes."""
    for path in sorted(glob.glob(os.path.join(GOLDEN, "stress_*.npz"))):
        g = np.load(path)
        if len(kw.get("devices", [0])) > g["u0"].shape[0]:
            continue                                            # fewer rows than slabs
        for steps in (1, 20):
            got_u, got_v, _ = gpu_run(g["u0"], g["v0"], steps, args=args(**kw))
            assert_bits_equal(got_u, g[f"u_{steps}"], f"{os.path.basename(path)} U {steps} {kw}")
            assert_bits_equal(got_v, g[f"v_{steps}"], f"{os.path.basename(path)} V {steps} {kw}")
    g = np.load(os.path.join(GOLDEN, "species_new_64x128.npz"))
    sim = Simulation.new(Parameters(), args(**kw))
    species = sim.make_species([64, 128])
    done = 0
    for steps in (1, 10, 100, 1000):
        sim.perform_steps(species, steps - done)
        done = steps
        in_u, in_v, _, _ = species.in_out()
        assert_bits_equal(in_u.make_scalar_view(sim.context), g[f"u_{steps}"], f"species_new U {steps} {kw}")
        assert_bits_equal(in_v.make_scalar_view(sim.context), g[f"v_{steps}"], f"species_new V {steps} {kw}")
    g = np.load(os.path.join(GOLDEN, "zero_halo_64x128.npz"))
    for steps in (1, 20):
        got_u, got_v, _ = gpu_run(g["stress_u0"], g["stress_v0"], steps, args=args(boundary=capi.GS_BOUNDARY_ZERO_HALO, **kw))
        assert_bits_equal(got_u, g[f"stress_u_{steps}"], f"zero halo stress U {steps} {kw}")
        assert_bits_equal(got_v, g[f"stress_v_{steps}"], f"zero halo stress V {steps} {kw}")


def test_golden_vectors_zero_halo_and_stencils():
    """The committed fixtures of the widened rows, without the oracle at run time."""
    from grayscott_amd.simulation import STENCILS

    g = np.load(os.path.join(GOLDEN, "zero_halo_64x128.npz"))
    sim = Simulation.new(Parameters(), args(boundary=capi.GS_BOUNDARY_ZERO_HALO))
    species = sim.make_species([64, 128])
    done = 0
    for steps in (1, 10, 100):
        sim.perform_steps(species, steps - done)
        done = steps
        in_u, in_v, _, _ = species.in_out()
        assert_bits_equal(in_u.make_scalar_view(sim.context), g[f"u_{steps}"], f"zero halo U {steps}")
        assert_bits_equal(in_v.make_scalar_view(sim.context), g[f"v_{steps}"], f"zero halo V {steps}")
    for steps in (1, 20):
        got_u, got_v, _ = gpu_run(g["stress_u0"], g["stress_v0"], steps, args=args(boundary=capi.GS_BOUNDARY_ZERO_HALO))
        assert_bits_equal(got_u, g[f"stress_u_{steps}"], f"zero halo stress U {steps}")
        assert_bits_equal(got_v, g[f"stress_v_{steps}"], f"zero halo stress V {steps}")
    g = np.load(os.path.join(GOLDEN, "stencils_37x60.npz"))
    for name in STENCILS:
        p = Parameters.with_stencil(name, time_step=0.25 if name == "pretty" else 1.0)
        got_u, got_v, _ = gpu_run(g["u0"], g["v0"], 12, params=p)
        assert_bits_equal(got_u, g[f"u_{name}"], f"stencil {name} U")
        assert_bits_equal(got_v, g[f"v_{name}"], f"stencil {name} V")


def test_ftz_rule_matches_cpu_flush_to_zero():
    """Strict flavour = MXCSR.FTZ semantics (flush results, keep inputs): V's sub-normal front."""
    g = np.load(os.path.join(GOLDEN, "ftz_front_64x128.npz"))
    u0, v0 = oracle.init_species(64, 128)
    for steps in (30, 40):
        got_u, got_v, _ = gpu_run(u0, v0, steps)
        assert_bits_equal(got_v, g[f"v_{steps}"], f"FTZ V {steps}")
        assert_bits_equal(got_u, g[f"u_{steps}"], f"FTZ U {steps}")
        # and it is NOT the keep-denormals answer
        assert np.count_nonzero(got_v.view(np.uint32) != g[f"v_{steps}_noftz"].view(np.uint32)) > 100


# ---- BASELINE config 1: 1080 x 1920, Species::new, default feed/kill, 1000 steps -----------
@pytest.mark.parametrize("kernel,label", [(capi.GS_KERNEL_AUTO, "window-r5/strict.op"), (capi.GS_KERNEL_TB, "tb-k")])
def test_config1_1080x1920_1000_steps(kernel, label):
    """As the library runs it by default (one persistent launch of the window kernel for a call of >= 32 steps on a grid of
    one round of windows) and with the marching kernel, which ran it until round 4 and still runs short calls."""
    rows, cols, steps = 1080, 1920, 1000
    sim = Simulation.new(Parameters(), args(kernel=kernel))
    species = sim.make_species([rows, cols])
    assert species.shape() == (rows, cols)
    sim.perform_steps(species, steps)
    assert sim.context.info()[0].startswith(label), sim.context.info()
    in_u, in_v, _, _ = species.in_out()
    got_u, got_v = in_u.make_scalar_view(sim.context), in_v.make_scalar_view(sim.context)
    u0, v0 = oracle.init_species(rows, cols)
    ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
    # the stated tolerance first (field-relative), then the stronger bit-exact claim
    assert np.max(np.abs(got_u - ref_u)) <= REL_TOL * np.max(np.abs(ref_u))
    assert np.max(np.abs(got_v - ref_v)) <= REL_TOL * np.max(np.abs(ref_v))
    assert_bits_equal(got_u, ref_u, "config 1 U")
    assert_bits_equal(got_v, ref_v, "config 1 V")
    # SURVEY section 8c self-consistency sums
    assert abs(got_u.sum(dtype=np.float64) - 2072039.383495) < 1e-5
    assert abs(got_v.sum(dtype=np.float64) - 354.492044) < 1e-5


# ---- fused flavour -----------------------------------------------------------------------------
def test_fused_flavour():
    # no sub-normals anywhere (few steps, fields O(1)): identical bits
    u0, v0 = stress_fields((250, 130), 1)
    ref_u, ref_v = oracle.run(u0, v0, 20, ftz=True)
    got_u, got_v, info = gpu_run(u0, v0, 20, args=args(math=capi.GS_MATH_FUSED))
    assert "fused" in info[0]
    assert_bits_equal(got_u, ref_u, "fused U")
    assert_bits_equal(got_v, ref_v, "fused V")
    # sub-normal front present: U still identical, V within 1e-37 absolute
    u0, v0 = oracle.init_species(64, 128)
    ref_u, ref_v = oracle.run(u0, v0, 100, ftz=True)
    got_u, got_v, _ = gpu_run(u0, v0, 100, args=args(math=capi.GS_MATH_FUSED))
    assert_bits_equal(got_u, ref_u, "fused U with sub-normal V front")
    assert np.max(np.abs(got_v.astype(np.float64) - ref_v.astype(np.float64))) <= 1e-37
    assert np.max(np.abs(got_v - ref_v)) <= REL_TOL * np.max(np.abs(ref_v))


def test_fused_refuses_non_power_of_two_weights():
    pk = ((1 / 6, 4 / 6, 1 / 6), (4 / 6, 0.0, 4 / 6), (1 / 6, 4 / 6, 1 / 6))
    with pytest.raises(GsError) as e:
        Simulation.new(Parameters(weights=pk), args(math=capi.GS_MATH_FUSED))
    assert e.value.code == capi.GS_ERR_UNSUPPORTED


# ---- temporal blocking: K fused steps per launch are bit-identical to K single steps ---------
@pytest.mark.parametrize("cpl", [4, 2, 1])
@pytest.mark.parametrize("fuse", [1, 2, 3, 4])
def test_temporal_blocking_bit_exact(fuse, cpl):
    """K fused steps, wide and narrow lane layouts; the widths straddle the strip widths of every
    (K, columns per lane) pair: (64 - 2 * ceil(K / cpl)) * cpl output columns per wave."""
    for shape in [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (17, 33), (64, 128), (250, 130), (9, 247), (9, 248),
                  (9, 249), (30, 252), (30, 253), (41, 500), (12, 1030), (9, 55), (9, 56), (9, 57), (7, 61),
                  (7, 62), (7, 63), (11, 119), (11, 120), (11, 121), (11, 124), (11, 125), (5, 113)]:
        u0, v0 = stress_fields(shape, 2)
        for steps in (1, 4, 7):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(kernel=capi.GS_KERNEL_TB, fuse_steps=fuse,
                                                                  rows_per_block=5, cols_per_lane=cpl))
            assert info[0].startswith("tb-k"), info
            assert ("c%d/" % cpl in info[0]) == (cpl != 4), info
            assert_bits_equal(got_u, ref_u, f"TB{fuse} U {shape} steps {steps}")
            assert_bits_equal(got_v, ref_v, f"TB{fuse} V {shape} steps {steps}")


@pytest.mark.parametrize("fuse,math", [(2, 0), (4, 0), (4, 1)])
def test_temporal_blocking_species_new_1000_steps(fuse, math):
    g = np.load(os.path.join(GOLDEN, "species_new_64x128.npz"))
    sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_TB, fuse_steps=fuse, math=math))
    species = sim.make_species([64, 128])
    sim.perform_steps(species, 1000)
    in_u, in_v, _, _ = species.in_out()
    got_u, got_v = in_u.make_scalar_view(sim.context), in_v.make_scalar_view(sim.context)
    if math == 0:
        assert_bits_equal(got_u, g["u_1000"], "TB U 1000")
        assert_bits_equal(got_v, g["v_1000"], "TB V 1000")
    else:
        assert np.max(np.abs(got_u - g["u_1000"])) <= REL_TOL * np.max(np.abs(g["u_1000"]))
        assert np.max(np.abs(got_v - g["v_1000"])) <= REL_TOL * np.max(np.abs(g["v_1000"]))


def test_temporal_blocking_large_grid_vs_stream():
    rows, cols, steps = 2048, 4096, 13  # 13 = 3 launches of 4 + 1 of 1
    rng = np.random.default_rng(5)
    u0 = rng.random((rows, cols), dtype=np.float32)
    v0 = (rng.random((rows, cols), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    ref = gpu_run(u0, v0, steps, args=args(kernel=capi.GS_KERNEL_STREAM))
    for fuse in (2, 3, 4):
        got = gpu_run(u0, v0, steps, args=args(kernel=capi.GS_KERNEL_TB, fuse_steps=fuse))
        assert_bits_equal(got[0], ref[0], f"TB{fuse} vs stream U")
        assert_bits_equal(got[1], ref[1], f"TB{fuse} vs stream V")


@pytest.mark.parametrize("case", [
    dict(rows=1100, cols=1920, cpl=1, rpu=20),                       # 1 column per lane, 1925 units + edge halves
    dict(rows=1024, cols=2048, cpl=2, rpu=10),                       # 2 columns per lane, short units
    dict(rows=1500, cols=1500, cpl=2, rpu=12, boundary=capi.GS_BOUNDARY_ZERO_HALO),
    dict(rows=1300, cols=1700, cpl=2, rpu=16, math=capi.GS_MATH_FUSED),   # (no in-step form: it needed 129 registers and spilled)
    dict(rows=1300, cols=1700, cpl=1, rpu=33, math=capi.GS_MATH_FUSED),   # the fused flavour's in-step form
    dict(rows=1300, cols=1700, cpl=1, rpu=33, general=True),         # non-default parameters: the general kernels
    dict(rows=2000, cols=130, cpl=2, rpu=2),                         # two strips, both edge strips, 2-row units in halves
])
def test_in_step_workgroups_bit_exact(case):
    """The form of the marching kernel that launches of one round run on a single slab: 16-wave workgroups
    whose waves keep step through an LDS progress board and s_setprio (tb-k4c?f).  Same arithmetic, same
    unit decomposition -- the results must be the oracle's bit for bit, edge units, ragged right end,
    remainder passes and both boundary rules included."""
    rows, cols = case["rows"], case["cols"]
    u0, v0 = stress_fields((rows, cols), 21)
    p = Parameters(feed_rate=0.03, kill_rate=0.06, time_step=0.5, diffusion_rate_u=0.2) if case.get("general") else Parameters()
    kw = dict(kernel=capi.GS_KERNEL_TB, fuse_steps=4, rows_per_block=case["rpu"], cols_per_lane=case["cpl"],
              boundary=case.get("boundary", capi.GS_BOUNDARY_CLIPPED), math=case.get("math", capi.GS_MATH_STRICT))
    steps = 11                                                        # a 3-step pass, then two in-step passes
    ref_u, ref_v = oracle.run(u0, v0, steps, oracle_params(p), ftz=True,
                              boundary=oracle.ZERO_HALO if kw["boundary"] == capi.GS_BOUNDARY_ZERO_HALO else oracle.CLIPPED)
    got_u, got_v, info = gpu_run(u0, v0, steps, params=p, args=args(**kw))
    in_step = not (kw["math"] == capi.GS_MATH_FUSED and case["cpl"] == 2)
    assert info[0].startswith("tb-k4c%d%s/" % (case["cpl"], "f" if in_step else "")), info
    if kw["math"] == capi.GS_MATH_STRICT:
        assert_bits_equal(got_u, ref_u, f"U {case}")
        assert_bits_equal(got_v, ref_v, f"V {case}")
    else:
        # stress fields hold no sub-normal intermediates in 11 steps: the fused taps agree bit for bit too
        assert np.max(np.abs(got_u.astype(np.float64) - ref_u)) <= 1e-37 and np.max(np.abs(got_v.astype(np.float64) - ref_v)) <= 1e-37


# ---- in-place row bands of a single slab (cross-pass overlap schedule) --------------------------
@pytest.mark.parametrize("split", [2, 3, 5])
def test_single_slab_row_bands_bit_exact(split):
    for shape, steps in (((200, 300), 23), ((64, 128), 50), ((97, 1030), 9)):
        u0, v0 = stress_fields(shape, 13)
        ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
        got_u, got_v, info = gpu_run(u0, v0, steps, args=args(split=split, rows_per_block=6))
        assert_bits_equal(got_u, ref_u, f"U split {split} {shape}")
        assert_bits_equal(got_v, ref_v, f"V split {split} {shape}")
    # interleaved with single steps, parameter changes and asynchronous downloads
    from grayscott_amd import pinned_empty
    u0, v0 = stress_fields((160, 520), 3)
    sim = Simulation.new(Parameters(), args(split=split))
    sp = species_from_arrays(sim, u0, v0)
    img = pinned_empty((160, 520))
    sim.perform_steps(sp, 10)
    sp.write_result_view_after(img)
    sim.perform_steps(sp, 17)
    sim.perform_step(sp)
    sim.perform_steps(sp, 8)
    sim.context.download_wait()
    assert_bits_equal(img, oracle.run(u0, v0, 10)[1], "image behind band passes")
    in_u, in_v, _, _ = sp.in_out()
    ref = oracle.run(u0, v0, 36)
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref[0], "U bands + single steps")
    assert_bits_equal(in_v.make_scalar_view(sim.context), ref[1], "V bands + single steps")


@pytest.mark.parametrize("split", [0, 2])
def test_row_bands_on_a_large_grid(split):
    """8192 x 8192 with the default schedule and with two row bands (plus the on-line tuning
    passes of a 300-step run); compare with the single-step kernel run step by step."""
    rows = cols = 8192
    rng = np.random.default_rng(21)
    u0 = rng.random((rows, cols), dtype=np.float32)
    v0 = (rng.random((rows, cols), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    steps = 14 if split == 0 else 300
    ref = gpu_run(u0, v0, steps, args=args(kernel=capi.GS_KERNEL_STREAM))
    got = gpu_run(u0, v0, steps, args=args(split=split))
    assert got[2][0].startswith("tb-k"), got[2]
    assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32))
    assert np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))


# ---- non-default parameters (CLI overrides -k -f -t, ui/src/lib.rs:51-63) and weights ------
# ---- GS_KERNEL_TILE: up to 8 steps per launch on LDS-resident windows (kernel = auto on mid-size grids) ---
@pytest.mark.parametrize("tile_shape,fuse", [(1, 0), (2, 0), (3, 0), (1, 5), (2, 7), (3, 1)])
@pytest.mark.parametrize("boundary", [capi.GS_BOUNDARY_CLIPPED, capi.GS_BOUNDARY_ZERO_HALO])
def test_tile_kernel_bit_exact(boundary, tile_shape, fuse):
    """gs_run_tile_k (GS_KERNEL_TILE), every window shape: every shape class of grid -- single cells and
    lines, one partial window, multiples of a window's output, ragged right and bottom windows, several
    windows each way -- and step counts that are a short launch, full launches and both."""
    for shape in [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (17, 33), (32, 64), (33, 65), (31, 63), (64, 128), (250, 130),
                  (65, 129), (96, 200), (40, 1000), (1000, 40), (1, 5000), (5000, 1), (129, 257)]:
        u0, v0 = stress_fields(shape, 4)
        for steps in (1, 3, 8, 9, 21):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True, boundary=boundary)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(kernel=capi.GS_KERNEL_TILE, boundary=boundary,
                                                                  tile_shape=tile_shape, fuse_steps=fuse))
            kmax = fuse or (4 if tile_shape == 2 else 8)
            assert info[0] == ("tile32x64", "tile16x64", "tile64x64")[tile_shape - 1] + "/strict.op", info
            assert info[1] == (steps + kmax - 1) // kmax, info
            assert_bits_equal(got_u, ref_u, f"tile U {shape} steps {steps}")
            assert_bits_equal(got_v, ref_v, f"tile V {shape} steps {steps}")


def test_tile_kernel_variants_and_auto_choice():
    """General weights / dt != 1 (no specialised variant), the fused flavour, Species::new over many
    launches with mixed entry points; what kernel = auto picks, and that a slab chain falls back to
    temporal blocking."""
    shape = (150, 333)
    u0, v0 = stress_fields(shape, 6)
    for params in (Parameters.with_stencil("patrakarttunen"), Parameters(time_step=0.5), Parameters(feed_rate=0.03, kill_rate=0.06)):
        ref_u, ref_v = oracle.run(u0, v0, 19, params=oracle_params(params), ftz=True)
        got_u, got_v, info = gpu_run(u0, v0, 19, params=params, args=args(kernel=capi.GS_KERNEL_TILE))
        assert info[0].startswith("tile") and "/strict" in info[0], info
        assert info[0].endswith(".op") == (params.weights == Parameters().weights and params.time_step == 1.0), info
        assert_bits_equal(got_u, ref_u, f"tile U {params}")
        assert_bits_equal(got_v, ref_v, f"tile V {params}")
    ref_u, ref_v = oracle.run(u0, v0, 19, ftz=False)
    got_u, got_v, info = gpu_run(u0, v0, 19, args=args(math=capi.GS_MATH_FUSED, kernel=capi.GS_KERNEL_TILE))
    assert info[0].startswith("tile") and info[0].endswith("/fused"), info
    assert np.max(np.abs(got_u - ref_u)) <= 1e-37 and np.max(np.abs(got_v - ref_v)) <= 1e-37
    sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_TILE))
    species = sim.make_species([128, 256])
    sim.perform_steps(species, 41)
    assert sim.context.info()[0].startswith("tile") and sim.context.info()[0].endswith("/strict.op")
    for _ in range(3):
        sim.perform_step(species)
    sim.perform_steps(species, 16)
    u, v = oracle.run(*oracle.init_species(128, 256), 60)
    in_u, in_v, _, _ = species.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), u, "tile + single steps U")
    assert_bits_equal(in_v.make_scalar_view(sim.context), v, "tile + single steps V")
    # kernel = auto: the window kernel between the resident kernel's 4096 cells and 1.5 M cells when nothing is
    # pinned (window and steps per launch from a cost model, gs_tuner.cpp: pick_tile_config); temporal
    # blocking for slab chains, pinned schedules and everything larger
    ref_u, ref_v = oracle.run(u0, v0, 29, ftz=True)
    for kw, want in ((dict(), "tile"), (dict(devices=[0, 0], kernel=capi.GS_KERNEL_TILE), "tb-k"),
                     (dict(kernel=capi.GS_KERNEL_TB), "tb-k"), (dict(fuse_steps=4), "tb-k"), (dict(rows_per_block=8), "tb-k"),
                     (dict(devices=[0, 0]), "tb-k")):
        got_u, got_v, info = gpu_run(u0, v0, 29, args=args(**kw))
        assert info[0].startswith(want), (kw, info)
        assert_bits_equal(got_u, ref_u, f"auto choice U {kw}")
        assert_bits_equal(got_v, ref_v, f"auto choice V {kw}")
    for shape2, want in (((300, 400), "tile32x64/"), ((600, 900), "tile64x64/"), ((1024, 1024), "tile64x64/"), ((3000, 40), "tile"),
                         ((12, 3000), "tile"), ((1300, 1300), "tb-k")):
        a0, b0 = stress_fields(shape2, 9)
        ref_u, ref_v = oracle.run(a0, b0, 21, ftz=True)
        got_u, got_v, info = gpu_run(a0, b0, 21, args=args())
        assert info[0].startswith(want), (shape2, info)
        assert_bits_equal(got_u, ref_u, f"auto choice U {shape2}")
        assert_bits_equal(got_v, ref_v, f"auto choice V {shape2}")


# ---- small grids: the whole run in one launch, LDS-resident ----------------------------------------
@pytest.mark.parametrize("boundary", [capi.GS_BOUNDARY_CLIPPED, capi.GS_BOUNDARY_ZERO_HALO])
def test_resident_kernel_small_grids(boundary):
    """Grids of at most 1536 cells run gs_run as ONE launch (gs_run_resident_k; above, the LDS-window kernel
    is faster): every shape class, odd and even step counts (the result lands in the other slot), long
    runs, both boundary rules, default and general parameters (the specialised and the general variant)."""
    for shape in [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (8, 16), (16, 32), (17, 33), (32, 48), (1, 1536), (1536, 1), (5, 307)]:
        u0, v0 = stress_fields(shape, 16)
        for steps in (1, 2, 9, 256):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True, boundary=boundary)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(boundary=boundary))
            assert info[0] == "resident-lds/strict.op" and info[1] == 1, info      # one launch
            assert_bits_equal(got_u, ref_u, f"resident U {shape} steps {steps}")
            assert_bits_equal(got_v, ref_v, f"resident V {shape} steps {steps}")
        params = Parameters(feed_rate=0.03, kill_rate=0.06, time_step=0.5)
        ref_u, ref_v = oracle.run(u0, v0, 9, params=oracle_params(params), ftz=True, boundary=boundary)
        got_u, got_v, info = gpu_run(u0, v0, 9, params=params, args=args(boundary=boundary))
        assert info[0] == "resident-lds/strict", info
        assert_bits_equal(got_u, ref_u, f"resident U {shape}, general parameters")
        assert_bits_equal(got_v, ref_v, f"resident V {shape}, general parameters")
    # just above the limit the ordinary kernels take over
    u0, v0 = stress_fields((32, 49), 16)
    assert not gpu_run(u0, v0, 8, args=args(boundary=boundary))[2][0].startswith("resident")


def test_resident_kernel_species_new_and_mixed_entry_points():
    """Species::new on a 32 x 48 grid through 1000 steps in uneven calls, single steps in between
    (gs_step uses the stream kernel), parameters changed on the way, fused flavour within tolerance."""
    sim = Simulation.new(Parameters(), args())
    species = sim.make_species([32, 48])
    u, v = oracle.init_species(32, 48)
    p = Parameters()
    for steps in (1, 7, 256, 333, 403):
        sim.perform_steps(species, steps)
        sim.perform_step(species)
        u, v = oracle.run(u, v, steps + 1, oracle_params(p), ftz=True)
        iu, iv, _, _ = species.in_out()
        assert_bits_equal(iu.make_scalar_view(sim.context), u, f"U after +{steps}+1")
        assert_bits_equal(iv.make_scalar_view(sim.context), v, f"V after +{steps}+1")
        p = Parameters(feed_rate=0.03, kill_rate=0.06, time_step=0.5) if steps == 7 else p
        sim.context.set_params(p)
    u0, v0 = stress_fields((30, 50), 17)
    ref_u, ref_v = oracle.run(u0, v0, 50, ftz=True)
    got_u, got_v, info = gpu_run(u0, v0, 50, args=args(math=capi.GS_MATH_FUSED))
    assert info[0] == "resident-lds/fused"
    assert np.max(np.abs(got_u - ref_u)) <= REL_TOL * np.max(np.abs(ref_u))
    assert np.max(np.abs(got_v - ref_v)) <= REL_TOL * np.max(np.abs(ref_v))


def test_unusual_call_sequences():
    """Zero steps, a fill_slice between runs, an overlapped download on the LDS-resident path, a very
    long run on a tiny grid (the uniform state is a fixed point, so the answer is known)."""
    from grayscott_amd import pinned_empty

    for shape in ((24, 40), (96, 200), (1100, 1000)):    # resident path / temporally blocked path, small and large
        sim = Simulation.new(Parameters(), args())
        species = sim.make_species(list(shape))
        sim.perform_steps(species, 0)                    # no-op, result stays in the input slot
        u, v = oracle.init_species(*shape)
        assert_bits_equal(species.make_result_view(), v, f"V after 0 steps {shape}")
        sim.perform_steps(species, 13)
        u, v = oracle.run(u, v, 13)
        in_u, in_v, _, _ = species.in_out()
        in_v.fill_slice(sim.context, [range(2, 5), range(3, 9)], 0.75)          # edit the state, carry on
        v[2:5, 3:9] = 0.75
        img = pinned_empty(shape)
        sim.perform_steps(species, 22)
        species.write_result_view_after(img)             # overlapped with ...
        sim.perform_steps(species, 5)                    # ... the next steps
        sim.context.download_wait()
        u, v = oracle.run(u, v, 22)
        assert_bits_equal(np.array(img), v, f"async image {shape}")
        u, v = oracle.run(u, v, 5)
        assert_bits_equal(species.make_result_view(), v, f"V after the overlapped download {shape}")
        sim.context.close()
    sim = Simulation.new(Parameters(), args())
    ones = np.ones((8, 16), np.float32)
    species = species_from_arrays(sim, ones, np.zeros((8, 16), np.float32))
    sim.perform_steps(species, 3_000_001)                # one launch, three million barriers
    in_u, in_v, _, _ = species.in_out()
    assert (in_u.make_scalar_view(sim.context) == 1).all() and (in_v.make_scalar_view(sim.context) == 0).all()


@pytest.mark.parametrize("kw", [dict(devices=[0, 0]), dict(devices=[0, 0, 0]), dict(split=2), dict()])
def test_fill_slice_between_asynchronous_runs(kw):
    """A fill_slice right behind an ASYNCHRONOUS run on a slab chain / on row bands: the last pass's
    boundary kernels and ghost pushes are still in flight on the side streams when the fill arrives; it
    must wait for them (rows 2..5 lie in slab 0's boundary rows, 46..50 straddle a seam of 2 slabs /
    bands, 63..66 one of 3 slabs)."""
    shape = (96, 200)
    sim = Simulation.new(Parameters(), args(**kw))
    species = sim.make_species(list(shape))
    u, v = oracle.init_species(*shape)
    for rows_, cols_, value, steps in ((range(2, 5), range(3, 9), 0.75, 13), (range(46, 50), range(0, 200), 0.5, 16),
                                       (range(63, 66), range(100, 180), 0.25, 9)):
        sim.prepare_steps(species, steps)                # enqueue only
        u, v = oracle.run(u, v, steps)
        in_u, in_v, _, _ = species.in_out()
        in_v.fill_slice(sim.context, [rows_, cols_], value)
        v[rows_.start:rows_.stop, cols_.start:cols_.stop] = value
        in_u.fill_slice(sim.context, [rows_, cols_], 1.0 - value)
        u[rows_.start:rows_.stop, cols_.start:cols_.stop] = np.float32(1.0 - value)
    sim.perform_steps(species, 21)
    u, v = oracle.run(u, v, 21)
    in_u, in_v, _, _ = species.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), u, f"U after edits between runs {kw}")
    assert_bits_equal(in_v.make_scalar_view(sim.context), v, f"V after edits between runs {kw}")
    sim.context.close()


def test_short_alternating_runs_finish_tuning_without_restarting():
    """A driver loop with 32 steps per image on TWO grids in turn: each shape's tuning advances in its
    own state (it used to restart whenever the other shape ran), short calls never wait for their
    timing windows, and both shapes end up tuned; the bits stay those of the oracle throughout."""
    sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_TB))
    shapes = [(96, 200), (64, 700)]
    state = {s: stress_fields(s, 23) for s in shapes}
    species = {s: species_from_arrays(sim, *state[s]) for s in shapes}
    total = {s: 0 for s in shapes}
    tuned = {}
    for it in range(400):
        for s in shapes:
            sim.prepare_steps(species[s], 32)
            total[s] += 32
            if sim.context.get_tuned(*s)[0] > 0:
                tuned.setdefault(s, it)
        sim.context.download_wait()                     # a driver waits for its image, not for the steps ...
        sim.context.sync()                              # ... (here: for everything, to keep the queue short)
        if len(tuned) == len(shapes):
            break
    assert len(tuned) == len(shapes), f"not tuned after 400 short calls each: {tuned}"
    for s in shapes:
        ref_u, ref_v = oracle.run(state[s][0], state[s][1], total[s], ftz=True)
        iu, iv, _, _ = species[s].in_out()
        assert_bits_equal(iu.make_scalar_view(sim.context), ref_u, f"U {s}")
        assert_bits_equal(iv.make_scalar_view(sim.context), ref_v, f"V {s}")
    sim.context.close()


def test_no_tune_option_and_handed_in_configuration():
    """no_tune: gs_run never times candidates; a configuration handed in with set_tuned is what runs
    (single slab and slab chain, incl. fewer steps per pass than the default 4)."""
    shape = (200, 300)
    u0, v0 = stress_fields(shape, 3)
    ref_u, ref_v = oracle.run(u0, v0, 47, ftz=True)
    for kw in (dict(), dict(devices=[0, 0, 0])):
        sim = Simulation.new(Parameters(), args(no_tune=1, kernel=capi.GS_KERNEL_TB, **kw))
        sp = species_from_arrays(sim, u0, v0)
        sim.perform_steps(sp, 30)
        assert "@" not in sim.context.info()[0]
        slab_rows = shape[0] // len(kw.get("devices", [0]))
        assert sim.context.get_tuned(slab_rows, shape[1]) == (0, 0, 0, 0)
        sim.context.set_tuned(slab_rows, shape[1], 16, 3, 1)
        # (the EFFECTIVE form is reported: 1 column per lane has no sharing variant, whatever was handed in -- ADVICE round 5)
        assert sim.context.get_tuned(slab_rows, shape[1]) == (16, 3, 1, 2)
        sim.perform_steps(sp, 17)
        label = sim.context.info()[0]
        assert label.startswith("tb-k3c1/") and "@16x" in label, label
        iu, iv, _, _ = sp.in_out()
        assert_bits_equal(iu.make_scalar_view(sim.context), ref_u, f"U {kw}")
        assert_bits_equal(iv.make_scalar_view(sim.context), ref_v, f"V {kw}")
        with pytest.raises(GsError):
            sim.context.set_tuned(slab_rows, shape[1], 16, 5, 1)
        sim.context.close()


# ---- one context, several grids ------------------------------------------------------------------
def test_alternating_shapes_on_one_context_keep_their_tuning():
    """The on-line choice is remembered per shape: going back to a grid does not tune again, and the
    results stay bit-exact throughout."""
    sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_TB))
    shapes = [(96, 200), (64, 700)]
    state = {s: stress_fields(s, 15) for s in shapes}
    species = {s: species_from_arrays(sim, *state[s]) for s in shapes}
    labels = {}
    for round_ in range(3):
        for s in shapes:
            steps = 2400 if round_ == 0 else 40
            sim.perform_steps(species[s], steps)
            state[s] = oracle.run(state[s][0], state[s][1], steps, ftz=True)
            iu, iv, _, _ = species[s].in_out()
            assert_bits_equal(iu.make_scalar_view(sim.context), state[s][0], f"U {s} round {round_}")
            assert_bits_equal(iv.make_scalar_view(sim.context), state[s][1], f"V {s} round {round_}")
            label = sim.context.info()[0]
            assert "@" in label, label                      # a tuned configuration is active
            assert labels.setdefault(s, label) == label     # ... and it is the one chosen the first time
    assert labels[shapes[0]] is not None


# ---- extreme aspect ratios ---------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 100003), (3, 70001), (70001, 3), (100003, 1), (2, 32768), (16385, 17)])
def test_extreme_aspect_ratios(shape):
    """One very long axis: many strips and one row unit, or many units and one (mostly masked)
    strip; every kernel family and both lane-layout extremes against the oracle."""
    u0, v0 = stress_fields(shape, 13)
    ref_u, ref_v = oracle.run(u0, v0, 6, ftz=True)
    for kw in (dict(), dict(kernel=capi.GS_KERNEL_STREAM), dict(kernel=capi.GS_KERNEL_LDS), dict(cols_per_lane=4),
               dict(cols_per_lane=1, fuse_steps=3), dict(devices=[0, 0]) if shape[0] >= 2 else dict(split=2)):
        got_u, got_v, info = gpu_run(u0, v0, 6, args=args(**kw))
        assert_bits_equal(got_u, ref_u, f"U {shape} {info[0]} {kw}")
        assert_bits_equal(got_v, ref_v, f"V {shape} {info[0]} {kw}")


# ---- the other boundary rule and the named stencils -------------------------
@pytest.mark.parametrize("kw", [dict(kernel=capi.GS_KERNEL_SIMPLE), dict(kernel=capi.GS_KERNEL_STREAM),
                                dict(kernel=capi.GS_KERNEL_LDS),
                                dict(kernel=capi.GS_KERNEL_TB, fuse_steps=1, cols_per_lane=4),
                                dict(kernel=capi.GS_KERNEL_TB, fuse_steps=2, cols_per_lane=1),
                                dict(kernel=capi.GS_KERNEL_TB, fuse_steps=3, cols_per_lane=2),
                                dict(kernel=capi.GS_KERNEL_TB, fuse_steps=4, cols_per_lane=4),
                                dict(kernel=capi.GS_KERNEL_TB, fuse_steps=4, cols_per_lane=2),
                                dict(kernel=capi.GS_KERNEL_TB, fuse_steps=4, cols_per_lane=1),
                                dict(devices=[0, 0, 0]), dict(split=2, rows_per_block=6), dict(math=capi.GS_MATH_FUSED)])
def test_zero_halo_boundary_rule(kw):
    """GS_BOUNDARY_ZERO_HALO (the rule of the reference's Vulkan and SIMD backends) against its
    checker, every kernel and schedule, bit for bit (fused flavour: tolerance)."""
    for shape in STRESS_SHAPES + [(9, 257), (40, 1030), (30, 120), (30, 121), (12, 56), (12, 57)]:
        if len(kw.get("devices", [0])) > shape[0]:
            continue
        u0, v0 = stress_fields(shape, 12)
        for steps in (1, 9):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True, boundary=oracle.ZERO_HALO)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(rows_per_block=kw.get("rows_per_block", 5),
                                                                  boundary=capi.GS_BOUNDARY_ZERO_HALO,
                                                                  **{k: v for k, v in kw.items() if k != "rows_per_block"}))
            if kw.get("math"):
                assert np.max(np.abs(got_u - ref_u)) <= REL_TOL * np.max(np.abs(ref_u)), (shape, steps)
                assert np.max(np.abs(got_v - ref_v)) <= REL_TOL * np.max(np.abs(ref_v)), (shape, steps)
            else:
                assert_bits_equal(got_u, ref_u, f"zero halo U {info[0]} {shape} steps {steps}")
                assert_bits_equal(got_v, ref_v, f"zero halo V {info[0]} {shape} steps {steps}")


def test_zero_halo_species_new_1080x1920():
    """Config-1 shape under the zero-halo rule, 200 steps, and the known answer on the border:
    U = 1, V = 0 next to an edge loses Du per step at first (0.9 after one step; corner 0.825)."""
    rows, cols = 1080, 1920
    sim = Simulation.new(Parameters(), args(boundary=capi.GS_BOUNDARY_ZERO_HALO))
    species = sim.make_species([rows, cols])
    sim.perform_step(species)
    u1 = species.in_out()[0].make_scalar_view(sim.context)
    assert u1[0, 100] == np.float32(1.0) + np.float32(0.1) * np.float32(-1.0) and abs(float(u1[0, 0]) - 0.825) < 1e-6
    assert u1[5, 5] == 1.0
    sim.perform_steps(species, 199)
    u0, v0 = oracle.init_species(rows, cols)
    ref_u, ref_v = oracle.run(u0, v0, 200, ftz=True, boundary=oracle.ZERO_HALO)
    in_u, in_v, _, _ = species.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref_u, "zero halo U 1080x1920")
    assert_bits_equal(in_v.make_scalar_view(sim.context), ref_v, "zero halo V 1080x1920")


@pytest.mark.parametrize("name", ["oono-puri", "5points", "patrakarttunen", "pretty"])
def test_named_stencils(name):
    """The reference's cargo-feature stencils (data/Cargo.toml:28-58) as run-time weights."""
    p = Parameters.with_stencil(name, time_step=0.25 if name == "pretty" else 1.0)   # "pretty" needs a small dt
    for shape, seed in (((37, 300), 4), ((64, 128), 5)):
        u0, v0 = stress_fields(shape, seed)
        ref_u, ref_v = oracle.run(u0, v0, 12, oracle_params(p), ftz=True)
        for kw in (dict(), dict(kernel=capi.GS_KERNEL_STREAM), dict(cols_per_lane=4, fuse_steps=4)):
            got_u, got_v, info = gpu_run(u0, v0, 12, params=p, args=args(**kw))
            assert np.isfinite(ref_u).all()
            assert_bits_equal(got_u, ref_u, f"{name} U {info[0]}")
            assert_bits_equal(got_v, ref_v, f"{name} V {info[0]}")


# ---- hipGraph replay of pass batches (gs_options.use_graph) -----------------------------------
def test_graph_replay_bit_exact_and_invalidated_by_every_input():
    """Batches of 16 passes replayed through a hipGraph: same bits; the captured launches carry
    plane addresses, shape, tuning and parameters, so changing any of them must rebuild the graph
    (two species alternating on one context, parameters changed between runs, pinned unit height)."""
    g = np.load(os.path.join(GOLDEN, "species_new_64x128.npz"))
    sim = Simulation.new(Parameters(), args(use_graph=1))
    species = sim.make_species([64, 128])
    sim.perform_steps(species, 1000)          # tuning passes first, then graph batches, then a remainder
    in_u, in_v, _, _ = species.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), g["u_1000"], "graph U 1000")
    assert_bits_equal(in_v.make_scalar_view(sim.context), g["v_1000"], "graph V 1000")

    u0, v0 = stress_fields((41, 500), 8)
    u1, v1 = stress_fields((41, 500), 9)
    p2 = Parameters(feed_rate=0.03, kill_rate=0.06, time_step=0.5)
    sim = Simulation.new(Parameters(), args(use_graph=1, rows_per_block=8))
    a, b = species_from_arrays(sim, u0, v0), species_from_arrays(sim, u1, v1)
    done_a = done_b = 0
    ref_a, ref_b = (u0, v0), (u1, v1)
    for params, steps in ((Parameters(), 150), (p2, 131), (Parameters(), 64)):
        sim.context.set_params(params)
        for sp, which in ((a, "a"), (b, "b"), (a, "a")):
            sim.perform_steps(sp, steps)
            if which == "a":
                ref_a = oracle.run(ref_a[0], ref_a[1], steps, oracle_params(params), ftz=True)
                ref = ref_a
            else:
                ref_b = oracle.run(ref_b[0], ref_b[1], steps, oracle_params(params), ftz=True)
                ref = ref_b
            iu, iv, _, _ = sp.in_out()
            assert_bits_equal(iu.make_scalar_view(sim.context), ref[0], f"graph U {which} {steps} {params}")
            assert_bits_equal(iv.make_scalar_view(sim.context), ref[1], f"graph V {which} {steps} {params}")


# ---- parameter-specialised variants of the temporal-blocking kernel ---------------------------
def _tiny_fields(shape, seed):
    """The left 60 % of the columns hold values of both signs around the flush threshold (2^-130 ..
    2^-118 times an ordinary value), the rest ordinary values: V stays tiny inside the region for
    many steps, so differences, halved differences, tap sums and reaction terms are sub-normal
    there step after step (U leaves the range in one step: + F * (1 - u))."""
    rng = np.random.default_rng(seed)
    u, v = stress_fields(shape, seed)
    scale = np.float32(2.0) ** rng.integers(-130, -118, size=shape).astype(np.float32)
    sign = np.where(rng.random(shape) < 0.5, np.float32(-1), np.float32(1))
    tiny = np.zeros(shape, bool)
    tiny[:, : (shape[1] * 3) // 5] = True
    u = np.where(tiny, u * scale * sign, u).astype(np.float32)
    v = np.where(tiny, v * scale * sign, v).astype(np.float32)
    return u, v


@pytest.mark.parametrize("general", [0, 1])
@pytest.mark.parametrize("fuse,cpl", [(1, 4), (2, 4), (3, 4), (4, 4), (4, 2), (3, 1), (4, 1), (2, 2)])
def test_specialised_variants_bit_exact(fuse, cpl, general):
    """GsStepArgs::fast: side weights == 0.5 (v_sub_f32 div:2 instead of sub, mul) and dt == 1
    (no multiply) are specialisations of the strict kernel that must not change one bit, with
    ordinary data and with data around the flush-to-zero threshold (the output modifier flushes
    to +0 where the multiply flushes to -0)."""
    quarter_corners = ((0.125, 0.5, 0.375), (0.5, 0.0, 0.5), (0.75, 0.5, 1.0))   # sides 0.5, odd corners
    cases = [
        (Parameters(), True),                                                       # both
        (Parameters(time_step=0.5), True),                                          # sides only
        (Parameters(weights=quarter_corners, diffusion_rate_u=0.05), True),         # both, other corners
        (Parameters(weights=((0.25, 0.5, 0.25), (0.5, 0, 0.25), (0.25, 0.5, 0.25))), False),  # dt == 1 alone: no variant
        (Parameters(weights=((0.25, 0.5, 0.25), (0.5, 0, 0.25), (0.25, 0.5, 0.25)), time_step=0.75), False),
    ]
    for p, special in cases:
        for shape, maker, seed in [((41, 500), stress_fields, 5), ((64, 300), _tiny_fields, 6), ((9, 1030), _tiny_fields, 7)]:
            u0, v0 = maker(shape, seed)
            for steps in (1, 4, 9):
                ref_u, ref_v = oracle.run(u0, v0, steps, oracle_params(p), ftz=True)
                got_u, got_v, info = gpu_run(u0, v0, steps, params=p,
                                             args=args(kernel=capi.GS_KERNEL_TB, fuse_steps=fuse, rows_per_block=12,
                                                       general_kernels=general, cols_per_lane=cpl))
                assert (".op" in info[0]) == (special and not general), (info, p)
                assert_bits_equal(got_u, ref_u, f"U {info[0]} {shape} steps {steps} {p}")
                assert_bits_equal(got_v, ref_v, f"V {info[0]} {shape} steps {steps} {p}")


@pytest.mark.parametrize("fuse,cpl", [(4, 2), (2, 2), (4, 1), (4, 4)])
def test_shared_difference_with_signed_zeros_and_equal_neighbours(fuse, cpl):
    """Two cells side by side in a lane share one difference (cells_interior): `x[c+1] - x[c]` is formed once and
    subtracted in the second cell's fold where the reference adds `x[c] - x[c+1]`.  The two agree up to the sign of an
    exact or flushed zero, which the accumulator must not see: fields drawn from a handful of values -- both zeros,
    sub-normals and the smallest normals of both signs, a few ordinary numbers -- so that neighbours are often equal,
    differ only in the sign of zero, or differ by less than the flush threshold."""
    values = np.array([0.0, -0.0, 2.0 ** -127, -(2.0 ** -127), 2.0 ** -126, -(2.0 ** -126), 3 * 2.0 ** -126, 2.0 ** -125,
                       0.25, 0.5, 0.5, 1.0, 1.0, 1.0], np.float32)
    for shape, seed in [((40, 520), 11), ((64, 300), 12)]:
        rng = np.random.default_rng(seed)
        u0 = values[rng.integers(0, len(values), size=shape)]
        v0 = values[rng.integers(0, len(values), size=shape)]
        # runs of equal neighbours along the rows, as a smooth field has them
        u0[:, 1::2] = np.where(rng.random(u0[:, 1::2].shape) < 0.5, u0[:, 0:-1:2][:, : u0[:, 1::2].shape[1]], u0[:, 1::2])
        v0[:, 1::2] = np.where(rng.random(v0[:, 1::2].shape) < 0.5, v0[:, 0:-1:2][:, : v0[:, 1::2].shape[1]], v0[:, 1::2])
        for steps in (1, 4, 8):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(kernel=capi.GS_KERNEL_TB, fuse_steps=fuse, rows_per_block=12,
                                                                  cols_per_lane=cpl))
            assert_bits_equal(got_u, ref_u, f"U {info[0]} {shape} steps {steps}")
            assert_bits_equal(got_v, ref_v, f"V {info[0]} {shape} steps {steps}")


def test_specialised_variant_sees_subnormals():
    """The sub-normal test data is real: the keep-denormals answer differs from the FTZ one."""
    u0, v0 = _tiny_fields((64, 300), 6)
    for steps in (1, 4, 9):
        _, a_v = oracle.run(u0, v0, steps, ftz=True)
        _, b_v = oracle.run(u0, v0, steps, ftz=False)
        assert np.count_nonzero(a_v.view(np.uint32) != b_v.view(np.uint32)) > 1000


@pytest.mark.parametrize("kernel", [capi.GS_KERNEL_SIMPLE, capi.GS_KERNEL_STREAM, capi.GS_KERNEL_LDS])
def test_non_default_parameters(kernel):
    pk = ((1 / 6, 4 / 6, 1 / 6), (4 / 6, 0.0, 4 / 6), (1 / 6, 4 / 6, 1 / 6))
    for p in (Parameters(feed_rate=0.03, kill_rate=0.06, time_step=0.5),
              Parameters(weights=pk, time_step=0.75),
              Parameters(weights=((0, 1, 0), (1, 0, 1), (0, 1, 0)), diffusion_rate_u=0.2)):
        u0, v0 = stress_fields((37, 300), 4)
        ref_u, ref_v = oracle.run(u0, v0, 9, oracle_params(p), ftz=True)
        got_u, got_v, _ = gpu_run(u0, v0, 9, params=p, args=args(kernel=kernel))
        assert_bits_equal(got_u, ref_u, f"U {p}")
        assert_bits_equal(got_v, ref_v, f"V {p}")


# ---- the Species / Simulate contract -----------------------------------------------------------
def test_species_new_and_flip_contract():
    sim = Simulation.new(Parameters(), args())
    species = sim.make_species([64, 128])
    in_u, in_v, out_u, out_v = species.in_out()
    u0, v0 = oracle.init_species(64, 128)
    assert_bits_equal(in_u.make_scalar_view(sim.context), u0, "Species::new U")
    assert_bits_equal(in_v.make_scalar_view(sim.context), v0, "Species::new V")
    assert in_u.raw_shape() == (64 + 8, 128)  # one slab: rows + 2 x 4 ghost rows, pitch 128
    # perform_step + flip == perform_steps(1); results land in the input slot either way
    sim.perform_step(species)
    a = species.make_result_view()
    sim2 = Simulation.new(Parameters(), args())
    sp2 = sim2.make_species([64, 128])
    sim2.perform_steps(sp2, 1)
    b = sp2.make_result_view()
    ref_u, ref_v = oracle.run(u0, v0, 1)
    assert_bits_equal(a, ref_v, "perform_step result view")
    assert_bits_equal(b, ref_v, "perform_steps(1) result view")
    # steps = 0 is a no-op; even and odd counts both leave the result in the input slot
    sim2.perform_steps(sp2, 0)
    assert_bits_equal(sp2.make_result_view(), ref_v, "perform_steps(0)")
    sim2.perform_steps(sp2, 2)
    ref3 = oracle.run(u0, v0, 3)
    assert_bits_equal(sp2.make_result_view(), ref3[1], "perform_steps(1)+(2)")
    target = np.empty((64, 128), np.float32)
    sp2.write_result_view(target)
    assert_bits_equal(target, ref3[1], "write_result_view")
    with pytest.raises(AssertionError):  # validate_write panics on shape mismatch (mod.rs:291-295)
        sp2.write_result_view(np.empty((64, 127), np.float32))


def test_set_params_between_runs_and_mixed_entry_points():
    """gs_ctx_set_params mid-run, and gs_step (single-step kernel) interleaved with gs_run (fused)."""
    u0, v0 = stress_fields((90, 400), 9)
    p1, p2 = Parameters(), Parameters(feed_rate=0.03, kill_rate=0.06, time_step=0.5)
    sim = Simulation.new(p1, args())
    sp = species_from_arrays(sim, u0, v0)
    sim.perform_steps(sp, 6)
    sim.perform_step(sp)
    sim.context.set_params(p2)
    sim.perform_steps(sp, 9)
    sim.perform_step(sp)
    sim.perform_steps(sp, 2)
    ref = oracle.run(u0, v0, 7, oracle_params(p1))
    ref = oracle.run(ref[0], ref[1], 12, oracle_params(p2))
    in_u, in_v, _, _ = sp.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref[0], "U after set_params")
    assert_bits_equal(in_v.make_scalar_view(sim.context), ref[1], "V after set_params")
    with pytest.raises(GsError):  # fused math refuses non power-of-two weights at set_params too
        s2 = Simulation.new(p1, args(math=capi.GS_MATH_FUSED))
        s2.context.set_params(Parameters(weights=((1 / 6, 4 / 6, 1 / 6), (4 / 6, 0.0, 4 / 6), (1 / 6, 4 / 6, 1 / 6))))


def test_error_behaviour():
    sim = Simulation.new(Parameters(), args())
    ctx = sim.context
    from grayscott_amd import HipConcentration

    a = HipConcentration(ctx, (8, 8))
    b = HipConcentration(ctx, (8, 9))
    lib = ctx._lib
    assert lib.gs_step(ctx.handle, a.handle, a.handle, a.handle, a.handle) == capi.GS_ERR_INVALID
    assert lib.gs_step(ctx.handle, a.handle, b.handle, a.handle, b.handle) == capi.GS_ERR_INVALID
    with pytest.raises(GsError):
        a.fill_slice(ctx, [range(0, 9), range(0, 8)], 1.0)  # out of range, as ndarray slicing panics
    with pytest.raises(GsError):
        Simulation.new(Parameters(), HipArgs(devices=[99]))
    with pytest.raises(GsError):                              # 2 rows cannot be split over 3 slabs
        Simulation.new(Parameters(), args(devices=[0, 0, 0])).make_species([2, 8])
    a.fill_slice(ctx, [range(2, 2), range(0, 8)], 5.0)  # empty range is fine
    assert float(a.make_scalar_view(ctx).sum()) == 0.0


@pytest.mark.parametrize("shape", [(0, 0), (0, 7), (5, 0)])
def test_empty_grids_are_legal_no_ops(shape):
    """ndarray holds zero-sized arrays, so Species::new, perform_steps and the result view all work
    on an empty grid in the reference (and do nothing); same here."""
    sim = Simulation.new(Parameters(), args())
    species = sim.make_species(list(shape))
    sim.perform_steps(species, 9)
    sim.perform_step(species)
    view = species.make_result_view()
    assert view.shape == shape and view.dtype == np.float32
    u, v = oracle.run(*oracle.init_species(*shape), 10)
    assert u.shape == shape and v.shape == shape


# ---- in-process row slabs with ghost rows (the multi-GPU data path on one GPU) --------------
@pytest.mark.parametrize("nslabs", [2, 3, 5])
def test_row_slabs_bit_identical_to_single(nslabs):
    for shape, steps in (((64, 128), 50), ((37, 300), 12), ((nslabs, 9), 4), ((250, 1030), 6)):
        u0, v0 = stress_fields(shape, 7)
        ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
        got_u, got_v, _ = gpu_run(u0, v0, steps, args=args(devices=[0] * nslabs))
        assert_bits_equal(got_u, ref_u, f"U {nslabs} slabs {shape}")
        assert_bits_equal(got_v, ref_v, f"V {nslabs} slabs {shape}")


def test_row_slabs_species_new_and_stepwise():
    sim = Simulation.new(Parameters(), args(devices=[0, 0, 0, 0]))
    species = sim.make_species([128, 256])
    assert species.raw_shape() == (128 + 4 * 8, 256)
    for _ in range(30):
        sim.perform_step(species)
    sim.perform_steps(species, 31)
    u0, v0 = oracle.init_species(128, 256)
    ref_u, ref_v = oracle.run(u0, v0, 61)
    in_u, in_v, _, _ = species.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref_u, "4 slabs U")
    assert_bits_equal(in_v.make_scalar_view(sim.context), ref_v, "4 slabs V")
    with pytest.raises(GsError):
        sim.make_species([3, 16])  # fewer rows than slabs


def test_slab_chain_never_refreshes_between_runs():
    """with a handed-in configuration of 3 steps per pass the leading short pass used to be
    sized with steps % 4, the run ended on a remainder and EVERY following gs_run started with a blocking,
    collective ghost refresh.  Now the short pass is sized with the steps per pass in force and every pass
    exchanges ghost-depth rows: the only refreshes are the two after the upload."""
    shape = (240, 500)
    u0, v0 = stress_fields(shape, 5)
    sim = Simulation.new(Parameters(), args(devices=[0, 0, 0], no_tune=1))
    sim.context.set_tuned(shape[0] // 3, shape[1], 16, 3, 2)
    sp = species_from_arrays(sim, u0, v0)
    base = sim.context.stats()
    total = 0
    for n in (10, 7, 9, 1, 12, 5):        # remainders of every kind, a single step in between
        sim.perform_steps(sp, n)
        total += n
    sim.perform_step(sp)
    sim.perform_steps(sp, 8)
    total += 9
    st = sim.context.stats()
    assert st["ghost_refreshes"] - base["ghost_refreshes"] == 2, (base, st)
    assert st["steps"] - base["steps"] == total
    assert sim.context.info()[0].startswith("tb-k3c2/")
    ref_u, ref_v = oracle.run(u0, v0, total, ftz=True)
    iu, iv, _, _ = sp.in_out()
    assert_bits_equal(iu.make_scalar_view(sim.context), ref_u, "U")
    assert_bits_equal(iv.make_scalar_view(sim.context), ref_v, "V")
    sim.context.close()


def test_pass_timing_reports_halo_and_interior_times():
    """gs_ctx_set_pass_timing / gs_ctx_stats on a slab chain: what bench.py prints per rank for N > 1."""
    sim = Simulation.new(Parameters(), args(devices=[0, 0]))
    sp = sim.make_species([4096, 2048])
    sim.perform_steps(sp, 40)
    sim.context.set_pass_timing(6)
    sim.perform_steps(sp, 40)          # 10 passes, the first 6 timed
    st = sim.context.stats()
    assert st["timed_passes"] == 6 and st["interior_ms"] > 0.0 and st["halo_ms"] > 0.0, st
    assert st["halo_exposed_ms"] >= 0.0 and st["halo_exposed_ms"] < st["interior_ms"] + st["halo_ms"] + 1.0
    sim.context.set_pass_timing(0)
    sim.perform_steps(sp, 8)
    assert sim.context.stats()["timed_passes"] == 0
    u0, v0 = oracle.init_species(4096, 2048)
    # the timed passes are ordinary passes: the state is the oracle's (crop around the seed and the seam)
    ref_u, ref_v = oracle.run(u0[1500:2300], v0[1500:2300], 88, ftz=True)
    iu, iv, _, _ = sp.in_out()
    got_u, got_v = iu.make_scalar_view(sim.context), iv.make_scalar_view(sim.context)
    assert_bits_equal(got_u[1500 + 88:2300 - 88], ref_u[88:-88], "U around the seam")
    assert_bits_equal(got_v[1500 + 88:2300 - 88], ref_v[88:-88], "V around the seam")
    sim.context.close()


def test_two_contexts_launch_the_large_lds_kernels():
    """The kernels that need more than 64 KB of dynamic LDS (64 x 64 windows: what AUTO picks for 600 x 900;
    the resident kernel on a 1 x 1536 grid) from two contexts and two slab-chain entries of one process: the
    opt-in is remembered per (device, function) -- tests/test_capi_cpu.py pins the key -- and a second
    context (on this 1-GPU box: the same device) still launches."""
    for shape, steps in (((600, 900), 24), ((1, 1536), 9)):
        u0, v0 = stress_fields(shape, 9)
        ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
        sims = [Simulation.new(Parameters(), args()) for _ in range(2)]
        if capi.device_count() >= 2:
            sims.append(Simulation.new(Parameters(), args(devices=[1])))
        for k, sim in enumerate(sims):
            sp = species_from_arrays(sim, u0, v0)
            sim.perform_steps(sp, steps)
            label = sim.context.info()[0]
            assert label.startswith("tile64x64/") or label.startswith("resident-lds/"), label
            iu, iv, _, _ = sp.in_out()
            assert_bits_equal(iu.make_scalar_view(sim.context), ref_u, f"U context {k} {shape}")
            assert_bits_equal(iv.make_scalar_view(sim.context), ref_v, f"V context {k} {shape}")
        for sim in sims:
            sim.context.close()


# ---- large grids: size-independent properties + GPU-vs-GPU equivalence chain ----------------
def test_large_grid_properties_4096():
    """4096 x 4096 (BASELINE config 2): (a) stream kernel == simple kernel bit for bit,
    (b) the far field is an exact fixed point (U=1, V=0), (c) a band of rows around the seed
    equals the oracle run on a cropped domain whose edges the signal has not reached."""
    rows = cols = 4096
    steps = 24
    out = {}
    for kernel in (capi.GS_KERNEL_SIMPLE, capi.GS_KERNEL_STREAM):
        sim = Simulation.new(Parameters(), args(kernel=kernel))
        sp = sim.make_species([rows, cols])
        sim.perform_steps(sp, steps)
        in_u, in_v, _, _ = sp.in_out()
        out[kernel] = (in_u.make_scalar_view(sim.context), in_v.make_scalar_view(sim.context))
        sim.context.close()
    su, sv = out[capi.GS_KERNEL_SIMPLE]
    tu, tv = out[capi.GS_KERNEL_STREAM]
    assert_bits_equal(tu, su, "stream vs simple U 4096^2")
    assert_bits_equal(tv, sv, "stream vs simple V 4096^2")
    (r0, r1), (c0, c1) = oracle.seed_ranges(rows, cols)
    m = steps + 2
    far = np.ones((rows, cols), bool)
    far[r0 - m:r1 + m, c0 - m:c1 + m] = False
    assert (tu[far] == 1.0).all() and (tv[far] == 0.0).all()
    # crop: the influence cone of the seed after `steps` steps stays inside [r0-m, r1+m) x [c0-m, c1+m);
    # outside it the state is the fixed point, so a crop with a 2m margin evolves identically
    R0, R1, C0, C1 = r0 - 2 * m, r1 + 2 * m, c0 - 2 * m, c1 + 2 * m
    u0, v0 = oracle.init_species(rows, cols)
    cu, cv = oracle.run(u0[R0:R1, C0:C1], v0[R0:R1, C0:C1], steps)
    inner = (slice(m, R1 - R0 - m), slice(m, C1 - C0 - m))
    assert_bits_equal(tu[R0:R1, C0:C1][inner], cu[inner], "crop U")
    assert_bits_equal(tv[R0:R1, C0:C1][inner], cv[inner], "crop V")


def test_headline_grid_16384_equivalence_chain():
    """16384 x 16384 (BASELINE config 3 shape): random-ish data everywhere, 3 steps,
    production kernel vs cross-check kernel vs 2 in-process slabs: identical bits.
    (The oracle cannot run this size in test time; the chain oracle == simple kernel ==
    stream kernel is closed on the smaller sizes above.)"""
    rows = cols = 16384
    rng = np.random.default_rng(11)
    base_u = rng.random((256, cols), dtype=np.float32)
    base_v = (rng.random((256, cols), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    u0 = np.tile(base_u, (rows // 256, 1))
    v0 = np.tile(base_v, (rows // 256, 1))
    u0[::7] = u0[::7][:, ::-1]  # break the vertical periodicity
    results = []
    for kw in (dict(kernel=capi.GS_KERNEL_SIMPLE), dict(kernel=capi.GS_KERNEL_STREAM),
               dict(kernel=capi.GS_KERNEL_STREAM, devices=[0, 0])):
        sim = Simulation.new(Parameters(), args(**kw))
        sp = species_from_arrays(sim, u0, v0)
        sim.perform_steps(sp, 3)
        in_u, in_v, _, _ = sp.in_out()
        results.append((in_u.make_scalar_view(sim.context), in_v.make_scalar_view(sim.context)))
        for c in sp.u._pair + sp.v._pair:
            c.destroy()
        sim.context.close()
    for k in (1, 2):
        for f, name in ((0, "U"), (1, "V")):
            same = np.array_equal(results[k][f].view(np.uint32), results[0][f].view(np.uint32))
            assert same, f"{name} variant {k} differs from the cross-check kernel"
    # oracle on the top-left corner block (exact for cells whose 3-step cone stays inside it)
    n = 160
    cu, cv = oracle.run(u0[:n, :n], v0[:n, :n], 3)
    assert_bits_equal(results[1][0][:n - 3, :n - 3], cu[:n - 3, :n - 3], "corner U")
    assert_bits_equal(results[1][1][:n - 3, :n - 3], cv[:n - 3, :n - 3], "corner V")
    # and the bottom-right corner (clipped-window rule on the far edges)
    cu, cv = oracle.run(u0[-n:, -n:], v0[-n:, -n:], 3)
    assert_bits_equal(results[1][0][-n + 3:, -n + 3:], cu[3:, 3:], "far corner U")
    assert_bits_equal(results[1][1][-n + 3:, -n + 3:], cv[3:, 3:], "far corner V")


@pytest.mark.parametrize("kw", [dict(kernel=capi.GS_KERNEL_SIMPLE), dict(kernel=capi.GS_KERNEL_STREAM), dict(),
                                dict(kernel=capi.GS_KERNEL_LDS), dict(devices=[0, 0]), dict(math=capi.GS_MATH_FUSED)])
def test_non_finite_values_spread_like_the_reference(kw):
    """NaN / Inf cells: the same cells are non-finite as in the oracle after every step count and
    all finite cells are bit-identical (NaN payloads are not compared: x86 and the GPU quiet NaNs
    differently).  Checks that skipped centre taps, blends and sacrificial lanes neither hide nor
    invent non-finite values."""
    u0, v0 = stress_fields((40, 300), 17)
    u0[5, 7] = np.nan
    v0[20, 100] = np.inf
    u0[39, 299] = -np.inf
    v0[0, 0] = np.nan
    for steps in (1, 3, 6):
        ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
        got_u, got_v, _ = gpu_run(u0, v0, steps, args=args(**kw))
        for got, ref, name in ((got_u, ref_u, "U"), (got_v, ref_v, "V")):
            bad_ref, bad_got = ~np.isfinite(ref), ~np.isfinite(got)
            assert np.array_equal(np.isnan(ref), np.isnan(got)), f"{name} NaN positions, {steps} steps, {kw}"
            assert np.array_equal(bad_ref, bad_got), f"{name} non-finite positions"
            fin = ~bad_ref
            assert np.array_equal(got[fin].view(np.uint32), ref[fin].view(np.uint32)), f"{name} finite cells"
            inf = np.isinf(ref)
            assert np.array_equal(got[inf], ref[inf]), f"{name} infinities (sign)"


def test_huge_grid_32768_index_width():
    """32768 x 32768 (2^30 cells, 4 GiB per plane: byte offsets exceed 32 bits): the seed region
    evolves exactly like the oracle on a crop, the far field stays the exact fixed point, and
    the last row / column (largest offsets) follow the clipped-window rule."""
    rows = cols = 32768
    steps = 9
    sim = Simulation.new(Parameters(), args())
    sp = sim.make_species([rows, cols])
    sim.perform_steps(sp, steps)
    in_u, in_v, _, _ = sp.in_out()
    v = in_v.make_scalar_view(sim.context)
    (r0, r1), (c0, c1) = oracle.seed_ranges(rows, cols)
    m = steps + 2
    R0, R1, C0, C1 = r0 - 2 * m, r1 + 2 * m, c0 - 2 * m, c1 + 2 * m
    u0, v0 = oracle.init_species(rows, cols)
    cu, cv = oracle.run(u0[R0:R1, C0:C1], v0[R0:R1, C0:C1], steps)
    del u0, v0
    inner = (slice(m, R1 - R0 - m), slice(m, C1 - C0 - m))
    assert_bits_equal(v[R0:R1, C0:C1][inner], cv[inner], "crop V at 32768^2")
    assert not v[:R0].any() and not v[R1:].any() and not v[:, :C0].any() and not v[:, C1:].any()
    del v
    u = in_u.make_scalar_view(sim.context)
    assert_bits_equal(u[R0:R1, C0:C1][inner], cu[inner], "crop U at 32768^2")
    assert (u[-1] == 1.0).all() and (u[:, -1] == 1.0).all() and (u[0] == 1.0).all()


def test_xcd_aware_unit_order_bit_exact():
    """Large launches renumber their workgroups so that the ones an XCD is dealt are neighbours in the grid
    (GsStepArgs::xcd_m; groups of 8 x 16 workgroups).  The group size is read from the environment once per process,
    so a child process runs the marching kernel and the single-step kernel with groups of 8 x 2 and 8 x 3 -- small
    enough to permute the units of a 300 x 2000 grid -- against the oracle, every kind of unit included."""
    import subprocess
    import sys

    code = r"""
import numpy as np, oracle
from grayscott_amd import HipArgs, capi
from tests.helpers import assert_bits_equal, gpu_run, stress_fields
u0, v0 = stress_fields((300, 2000), 21)
for kernel, kw in ((capi.GS_KERNEL_TB, dict(fuse_steps=4, rows_per_block=8, cols_per_lane=2)),
                   (capi.GS_KERNEL_TB, dict(fuse_steps=3, rows_per_block=12, cols_per_lane=1)),
                   (capi.GS_KERNEL_STREAM, dict(rows_per_block=16))):
    for steps in (1, 4, 9):
        ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
        got_u, got_v, info = gpu_run(u0, v0, steps, args=HipArgs(devices=[0], kernel=kernel, no_tune=1, **kw))
        assert_bits_equal(got_u, ref_u, f"U {info[0]} steps {steps}")
        assert_bits_equal(got_v, ref_v, f"V {info[0]} steps {steps}")
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for m in ("2", "3"):
        env = dict(os.environ, GS_HIP_XCD_M=m, GS_HIP_XCD_M_STREAM=m, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (m, r.stdout[-2000:], r.stderr[-4000:])
