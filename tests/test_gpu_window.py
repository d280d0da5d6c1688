"""gs_run_window_k (GS_KERNEL_WINDOW): the whole gs_run as ONE persistent launch on grids of at most one window per
compute unit -- workgroups keep their window in registers and trade their k-cell aprons through exchange planes with
flags, inside the launch.  Bit for bit against the oracle: every shape class (single cells and lines, one partial
window, several windows each way with ragged right / bottom ends, exactly one owned region), both boundary rules,
both window heights, every k, step counts that are a short super-step, whole ones and both; general parameters
(no specialised variant); the fused flavour; NaN / Inf spreading; Species::new through uneven calls with single
steps in between; BASELINE config 1 (1080 x 1920 x 1000 steps) end to end; a launch that gives up; a soak against the
marching kernel under chaotic dynamics (every exchanged word matters).  A measured alternative: kernel = auto never
picks it (profiles/r04_window_kernel.md).
Spec: compute/naive/src/lib.rs:42-83 (arithmetic, clipped window), compute/shared/src/cpu.rs:30-42 (step; flip);
zero-halo rule: compute/gpu/naive/src/pipeline.rs:105-113."""
import numpy as np
import pytest

import oracle
from grayscott_amd import HipArgs, Parameters, Simulation, capi
from tests.helpers import assert_bits_equal, gpu_run, oracle_params, stress_fields

pytestmark = pytest.mark.gpu


def args(**kw):
    return HipArgs(devices=[0], **kw)


SHAPES = [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (17, 33), (72, 120), (73, 121), (71, 119), (88, 120), (96, 300), (250, 130),
          (145, 241), (40, 1000), (1000, 40), (1, 3000), (3000, 1), (300, 500)]


@pytest.mark.parametrize("window_rows,k", [(0, 0), (80, 2), (80, 4), (80, 6), (80, 8), (96, 4), (96, 8)])
@pytest.mark.parametrize("boundary", [capi.GS_BOUNDARY_CLIPPED, capi.GS_BOUNDARY_ZERO_HALO])
def test_window_kernel_bit_exact(boundary, window_rows, k):
    for shape in SHAPES:
        u0, v0 = stress_fields(shape, 4)
        for steps in (1, 3, 8, 9, 22):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True, boundary=boundary)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(kernel=capi.GS_KERNEL_WINDOW, boundary=boundary,
                                                                  rows_per_block=window_rows, fuse_steps=k))
            assert info[0] == f"window-r{(window_rows or 80) // 16}/strict.op" and info[1] == 1, info    # ONE launch
            assert_bits_equal(got_u, ref_u, f"window U {shape} steps {steps} k {k}")
            assert_bits_equal(got_v, ref_v, f"window V {shape} steps {steps} k {k}")


def test_window_kernel_variants():
    shape = (150, 333)
    u0, v0 = stress_fields(shape, 6)
    for params in (Parameters.with_stencil("patrakarttunen"), Parameters(time_step=0.5), Parameters(feed_rate=0.03, kill_rate=0.06)):
        for boundary in (capi.GS_BOUNDARY_CLIPPED, capi.GS_BOUNDARY_ZERO_HALO):
            ref_u, ref_v = oracle.run(u0, v0, 19, params=oracle_params(params), ftz=True, boundary=boundary)
            got_u, got_v, info = gpu_run(u0, v0, 19, params=params, args=args(kernel=capi.GS_KERNEL_WINDOW, boundary=boundary))
            assert info[0].startswith("window-r5/strict"), info
            assert info[0].endswith(".op") == (params.weights == Parameters().weights and params.time_step == 1.0), info
            assert_bits_equal(got_u, ref_u, f"window U {params}")
            assert_bits_equal(got_v, ref_v, f"window V {params}")
    ref_u, ref_v = oracle.run(u0, v0, 19, ftz=False)
    got_u, got_v, info = gpu_run(u0, v0, 19, args=args(math=capi.GS_MATH_FUSED, kernel=capi.GS_KERNEL_WINDOW))
    assert info[0] == "window-r5/fused", info
    assert np.max(np.abs(got_u - ref_u)) <= 1e-37 and np.max(np.abs(got_v - ref_v)) <= 1e-37
    # the general path of every edge window (no cheap kinds): same bits
    import os
    ref_u, ref_v = oracle.run(u0, v0, 19, ftz=True)
    got_u, got_v, _ = gpu_run(u0, v0, 19, args=args(kernel=capi.GS_KERNEL_WINDOW, general_kernels=1))
    assert_bits_equal(got_u, ref_u, "window U, general kernels")
    assert_bits_equal(got_v, ref_v, "window V, general kernels")


def test_window_kernel_spreads_nan_and_inf_like_the_reference():
    shape = (200, 300)
    u0, v0 = stress_fields(shape, 8)
    for (r, c, val) in ((0, 0, np.nan), (0, 299, np.inf), (199, 0, -np.inf), (100, 150, np.nan), (71, 119, np.inf), (72, 120, np.nan),
                        (0, 150, np.nan), (100, 0, np.inf), (100, 299, np.nan), (199, 150, -np.inf)):
        a, b = u0.copy(), v0.copy()
        a[r, c] = val
        b[(r + 37) % shape[0], (c + 91) % shape[1]] = val
        for boundary in (capi.GS_BOUNDARY_CLIPPED, capi.GS_BOUNDARY_ZERO_HALO):
            ref_u, ref_v = oracle.run(a, b, 7, ftz=True, boundary=boundary)
            got_u, got_v, _ = gpu_run(a, b, 7, args=args(kernel=capi.GS_KERNEL_WINDOW, boundary=boundary))
            # NaN payloads are not part of the contract: compare the masks, and the bits of everything finite
            for got, ref, name in ((got_u, ref_u, "U"), (got_v, ref_v, "V")):
                assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{name}: NaN mask differs ({r},{c},{val})"
                fin = ~np.isnan(ref)
                assert np.array_equal(got[fin].view(np.uint32), ref[fin].view(np.uint32)), f"{name} differs ({r},{c},{val})"


def test_window_kernel_species_new_uneven_calls_and_single_steps():
    sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_WINDOW))
    species = sim.make_species([250, 400])
    u, v = oracle.init_species(250, 400)
    total = 0
    for steps in (1, 7, 256, 333, 403):
        sim.perform_steps(species, steps)
        assert sim.context.info()[0] == "window-r5/strict.op"
        sim.perform_step(species)          # gs_step: the stream kernel, then back
        total += steps + 1
    u, v = oracle.run(u, v, total, ftz=True)
    in_u, in_v, _, _ = species.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), u, "window + single steps U")
    assert_bits_equal(in_v.make_scalar_view(sim.context), v, "window + single steps V")
    st = sim.context.stats()
    assert st["steps"] == total and st["launches"] == 10


def test_config1_1080x1920_1000_steps_through_the_window_kernel():
    """BASELINE config 1 as written, end to end: Species::new([1080, 1920]), default feed / kill, 1000 steps."""
    rows, cols, steps = 1080, 1920, 1000
    sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_WINDOW))
    species = sim.make_species([rows, cols])
    sim.perform_steps(species, steps)
    assert sim.context.info() == ("window-r5/strict.op", 1)
    u, v = oracle.run(*oracle.init_species(rows, cols), steps, ftz=True)
    in_u, in_v, _, _ = species.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), u, "config 1 U")
    assert_bits_equal(in_v.make_scalar_view(sim.context), v, "config 1 V")
    # stress fields on the same grid, both rules, odd step count
    u0, v0 = stress_fields((rows, cols), 12)
    for boundary in (capi.GS_BOUNDARY_CLIPPED, capi.GS_BOUNDARY_ZERO_HALO):
        ref_u, ref_v = oracle.run(u0, v0, 37, ftz=True, boundary=boundary)
        got_u, got_v, info = gpu_run(u0, v0, 37, args=args(kernel=capi.GS_KERNEL_WINDOW, boundary=boundary))
        assert_bits_equal(got_u, ref_u, "1080x1920 stress U")
        assert_bits_equal(got_v, ref_v, "1080x1920 stress V")


def test_window_kernel_refuses_grids_of_more_than_one_window_per_cu():
    from grayscott_amd import GsError

    sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_WINDOW))
    species = sim.make_species([4096, 4096])
    with pytest.raises(GsError) as e:
        sim.perform_steps(species, 4)
    assert e.value.code == capi.GS_ERR_UNSUPPORTED
    # a slab chain falls back to the temporally blocked kernel
    u0, v0 = stress_fields((200, 300), 3)
    ref_u, ref_v = oracle.run(u0, v0, 13, ftz=True)
    got_u, got_v, info = gpu_run(u0, v0, 13, args=HipArgs(devices=[0, 0], kernel=capi.GS_KERNEL_WINDOW))
    assert info[0].startswith("tb-k"), info
    assert_bits_equal(got_u, ref_u, "chain U")
    assert_bits_equal(got_v, ref_v, "chain V")


def test_a_launch_that_gives_up_is_reported_and_destroys_nothing(monkeypatch):
    """GS_HIP_WINDOW_PATIENCE = 1 poll: on a grid of many windows some workgroup's neighbour is late at some exchange,
    the launch gives up, gs_sync says so, the input planes are intact and the context falls back to the marching
    kernel -- which then gives the right answer from the same planes."""
    from grayscott_amd import GsError

    monkeypatch.setenv("GS_HIP_WINDOW_PATIENCE", "1")
    rows, cols = 1080, 1920
    u0, v0 = stress_fields((rows, cols), 5)
    sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_WINDOW))
    from tests.helpers import species_from_arrays
    sp = species_from_arrays(sim, u0, v0)
    gave_up = False
    try:
        sim.perform_steps(sp, 400)
    except GsError as e:
        gave_up = True
        assert "gave up" in str(e)
    if not gave_up:
        pytest.skip("every poll of 100 exchanges matched at once on this box")
    in_u, in_v, _, _ = sp.in_out()
    # the host mirror points its handles back at the input planes, which the launch never wrote
    assert_bits_equal(in_u.make_scalar_view(sim.context), u0, "input U after a launch that gave up")
    sim.perform_steps(sp, 40)
    assert sim.context.info()[0].startswith("tb-k")
    ref_u, ref_v = oracle.run(u0, v0, 40, ftz=True)
    in_u, in_v, _, _ = sp.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref_u, "U after the fallback")
    assert_bits_equal(in_v.make_scalar_view(sim.context), ref_v, "V after the fallback")


def test_window_kernel_soak_against_the_marching_kernel():
    """1080 x 1920, a pattern-forming start, 3000 steps in uneven calls, three contexts in a row: ~750 exchanges of 240
    workgroups each -- one stale or misplaced exchange word anywhere changes bits that the dynamics then spread --
    against the default schedule (the marching kernel), every word of U and V."""
    import torch

    import bench

    rows, cols = 1080, 1920
    u0, v0 = bench.developed_start(rows, cols)
    ref = Simulation.new(Parameters(), args())
    sr = bench.upload_species(ref, u0, v0)
    calls = (997, 1003, 1000)
    for n in calls:
        ref.perform_steps(sr, n)
    assert ref.context.info()[0].startswith("tb-k")
    for attempt in range(3):
        sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_WINDOW))
        sp = bench.upload_species(sim, u0, v0)
        for n in calls:
            sim.perform_steps(sp, n)
        assert sim.context.info()[0] == "window-r5/strict.op"
        torch.cuda.synchronize()
        for name, a, b in (("U", sp.in_out()[0], sr.in_out()[0]), ("V", sp.in_out()[1], sr.in_out()[1])):
            (_, _, x), = a.torch_views()
            (_, _, y), = b.torch_views()
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"{name} differs after {sum(calls)} steps (context {attempt})"
        sim.context.close()
    (_, _, v), = sr.in_out()[1].torch_views()
    assert float(v.max()) > 0.3
    ref.context.close()
