"""gs_run_window_k (GS_KERNEL_WINDOW): the whole gs_run as ONE persistent launch on grids of at most one window per
compute unit -- workgroups keep their window in registers and trade their k-cell aprons through exchange planes with
flags, inside the launch.  Bit for bit against the oracle: every shape class (single cells and lines, one partial
window, several windows each way with ragged right / bottom ends, exactly one owned region), both boundary rules,
both window heights, every k, step counts that are a short super-step, whole ones and both; general parameters
(no specialised variant); the fused flavour; NaN / Inf spreading; Species::new through uneven calls with single
steps in between; BASELINE config 1 (1080 x 1920 x 1000 steps) end to end; a launch that gives up; a soak against the
marching kernel under chaotic dynamics (every exchanged word matters); what kernel = auto picks
(profiles/r04_window_kernel.md).
Spec: compute/naive/src/lib.rs:42-83 (arithmetic, clipped window), compute/shared/src/cpu.rs:30-42 (step; flip);
zero-halo rule: compute/gpu/naive/src/pipeline.rs:105-113."""
import numpy as np
import pytest

import oracle
from grayscott_amd import GsError, HipArgs, Parameters, Simulation, capi
from tests.helpers import assert_bits_equal, gpu_run, oracle_params, stress_fields

pytestmark = pytest.mark.gpu


def args(**kw):
    return HipArgs(devices=[0], **kw)


SHAPES = [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (17, 33), (72, 120), (73, 121), (71, 119), (88, 120), (96, 300), (250, 130),
          (145, 241), (40, 1000), (1000, 40), (1, 3000), (3000, 1), (300, 500)]


@pytest.mark.parametrize("window_rows,k", [(0, 0), (80, 2), (80, 4), (80, 6), (80, 8)])
@pytest.mark.parametrize("boundary", [capi.GS_BOUNDARY_CLIPPED, capi.GS_BOUNDARY_ZERO_HALO])
def test_window_kernel_bit_exact(boundary, window_rows, k):
    for shape in SHAPES:
        u0, v0 = stress_fields(shape, 4)
        for steps in (1, 3, 8, 9, 22):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True, boundary=boundary)
            # share_taps 0: the default form, full difference sharing inside a wave's band in the windows inside the grid
            # (".op.ds", round 6); 2: without (".op", what ran until round 5) -- both against the oracle
            for share, suffix in ((0, ".op.ds"), (2, ".op")) if k in (0, 4) else ((0, ".op.ds"),):
                got_u, got_v, info = gpu_run(u0, v0, steps, args=args(kernel=capi.GS_KERNEL_WINDOW, boundary=boundary,
                                                                      rows_per_block=window_rows, fuse_steps=k, share_taps=share))
                assert info[0] == f"window-r{(window_rows or 80) // 16}/strict{suffix}" and info[1] == 1, info    # ONE launch
                assert_bits_equal(got_u, ref_u, f"window U {shape} steps {steps} k {k} {suffix}")
                assert_bits_equal(got_v, ref_v, f"window V {shape} steps {steps} k {k} {suffix}")


def test_window_kernel_variants():
    shape = (150, 333)
    u0, v0 = stress_fields(shape, 6)
    for params in (Parameters.with_stencil("patrakarttunen"), Parameters(time_step=0.5), Parameters(feed_rate=0.03, kill_rate=0.06)):
        for boundary in (capi.GS_BOUNDARY_CLIPPED, capi.GS_BOUNDARY_ZERO_HALO):
            ref_u, ref_v = oracle.run(u0, v0, 19, params=oracle_params(params), ftz=True, boundary=boundary)
            got_u, got_v, info = gpu_run(u0, v0, 19, params=params, args=args(kernel=capi.GS_KERNEL_WINDOW, boundary=boundary))
            assert info[0].startswith("window-r5/strict"), info
            assert info[0].endswith(".op.ds") == (params.weights == Parameters().weights and params.time_step == 1.0), info
            assert ".op" not in info[0] or info[0].endswith(".op.ds"), info
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
        assert sim.context.info()[0] == "window-r5/strict.op.ds"
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
    assert sim.context.info() == ("window-r5/strict.op.ds", 1)
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


def test_a_launch_that_gives_up_is_run_again_by_the_marching_kernel(monkeypatch):
    """GS_HIP_WINDOW_PATIENCE = 1 poll: on a grid of many windows some workgroup's neighbour is late at some exchange and
    the launch gives up -- as it would if its workgroups were not all resident.  Nothing of it is kept: the launches
    enqueued since the last synchronisation took no step (sticky abort word, input planes only read), gs_sync runs them
    again with the marching kernel and puts the results where gs_run said they would be (also when the step count's
    parity would put them elsewhere), and the context stays with the marching kernel.  The caller sees right answers."""
    from tests.helpers import species_from_arrays

    monkeypatch.setenv("GS_HIP_WINDOW_PATIENCE", "1")
    rows, cols = 1080, 1920
    u0, v0 = stress_fields((rows, cols), 5)
    sim = Simulation.new(Parameters(), args())           # kernel = auto: the window kernel for calls of >= 32 steps
    sp = species_from_arrays(sim, u0, v0)
    sim.prepare_steps(sp, 400)                           # two calls in flight, the second with an odd step count
    sim.prepare_steps(sp, 77)
    sim.context.sync()
    label = sim.context.info()[0]
    if label.startswith("window"):
        pytest.skip("every poll of 120 exchanges matched at once on this box")
    assert label.startswith("tb-k"), label
    ref_u, ref_v = oracle.run(u0, v0, 477, ftz=True)
    in_u, in_v, _, _ = sp.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref_u, "U after the fallback")
    assert_bits_equal(in_v.make_scalar_view(sim.context), ref_v, "V after the fallback")
    sim.perform_steps(sp, 100)                           # the context stays with the marching kernel
    assert sim.context.info()[0].startswith("tb-k")
    ref_u, ref_v = oracle.run(ref_u, ref_v, 100, ftz=True)
    in_u, in_v, _, _ = sp.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref_u, "U, 100 steps later")


def test_only_the_launch_that_gave_up_and_the_later_ones_are_run_again(monkeypatch):
    """Two short window launches queued without a wait, the SECOND with a patience of one poll.  Its workgroups poll once
    at their only exchange: those whose neighbours have arrived run on to the end and STORE their windows into the
    launch's output planes -- the first launch's INPUT planes -- before the others give up.  The abort
    word holds the number of the launch that gave up: the first launch's result stands, the second is run again by the
    marching kernel from ITS input planes, which nothing has written.  Right answers, one fallback, and the counters
    count every step once."""
    from tests.helpers import species_from_arrays

    rows, cols = 1080, 1920
    u0, v0 = stress_fields((rows, cols), 6)
    ref = {}
    fallbacks = 0
    for first, second in ((64, 8), (70, 8), (9, 8), (64, 12), (33, 8), (128, 8)):
        sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_WINDOW))
        sp = species_from_arrays(sim, u0, v0)
        monkeypatch.delenv("GS_HIP_WINDOW_PATIENCE", raising=False)
        sim.prepare_steps(sp, first)
        monkeypatch.setenv("GS_HIP_WINDOW_PATIENCE", "1")
        sim.prepare_steps(sp, second)
        monkeypatch.delenv("GS_HIP_WINDOW_PATIENCE", raising=False)
        sim.context.sync()
        st = sim.context.stats()
        fallbacks += st["window_fallbacks"]
        assert st["steps"] == first + second, st
        total = first + second
        if total not in ref:
            ref[total] = oracle.run(u0, v0, total, ftz=True)
        in_u, in_v, _, _ = sp.in_out()
        assert_bits_equal(in_u.make_scalar_view(sim.context), ref[total][0], f"U after {first} + {second} steps, {st}")
        assert_bits_equal(in_v.make_scalar_view(sim.context), ref[total][1], f"V after {first} + {second} steps, {st}")
        if st["window_fallbacks"]:
            assert sim.context.info()[0].startswith("tb-k"), sim.context.info()
        sim.context.close()
    if fallbacks == 0:
        pytest.skip("every single poll matched at once on this box: no launch gave up")


def test_what_kernel_auto_picks_around_the_window_kernel():
    """kernel = auto: the window kernel for calls of >= 32 steps (the reference's steps per image) on single-slab grids from 0.8 M cells up to one window per
    compute unit when nothing is pinned; the marching (or, below 1.5 M cells, the LDS-window) kernel for short calls, pinned schedules, slab chains, other grids."""
    u0, v0 = stress_fields((1080, 1920), 3)
    ref = {n: oracle.run(u0, v0, n, ftz=True) for n in (64, 32, 28)}
    for kw, steps, want in ((dict(), 64, "window-r5/"), (dict(), 32, "window-r5/"), (dict(), 28, "tb-k"), (dict(fuse_steps=4), 64, "tb-k"),
                            (dict(rows_per_block=10), 64, "tb-k"), (dict(devices=[0, 0]), 64, "tb-k"),
                            (dict(boundary=capi.GS_BOUNDARY_ZERO_HALO), 64, "window-r5/")):
        got_u, got_v, info = gpu_run(u0, v0, steps, args=HipArgs(**{"devices": [0], **kw}))
        assert info[0].startswith(want), (kw, steps, info)
        if "boundary" not in kw:
            assert_bits_equal(got_u, ref[steps][0], f"auto U {kw} {steps}")
            assert_bits_equal(got_v, ref[steps][1], f"auto V {kw} {steps}")
    # ... and from 0.8 M cells on (round 6: the window kernel without its barriers is ahead of the LDS-window kernel there),
    # with as many steps per exchange as leave one window per compute unit (8 here)
    a0, b0 = stress_fields((720, 1280), 5)
    for steps, want in ((64, "window-r5/"), (32, "window-r5/"), (28, "tile")):
        got_u, got_v, info = gpu_run(a0, b0, steps, args=args())
        assert info[0].startswith(want), (steps, info)
        ref_u, ref_v = oracle.run(a0, b0, steps, ftz=True)
        assert_bits_equal(got_u, ref_u, f"auto U 720 x 1280, {steps}")
        assert_bits_equal(got_v, ref_v, f"auto V 720 x 1280, {steps}")
    a0, b0 = stress_fields((512, 1024), 4)                # 0.5 M cells: the LDS-window kernel
    assert gpu_run(a0, b0, 64, args=args())[2][0].startswith("tile")
    a0, b0 = stress_fields((2048, 2048), 4)               # 4.2 M cells: more than one window per CU
    assert gpu_run(a0, b0, 64, args=args())[2][0].startswith("tb-k")
    a0, b0 = stress_fields((1200, 2000), 4)               # 2.4 M cells: more than one round of 80-row windows
    assert gpu_run(a0, b0, 70, args=args())[2][0].startswith("tb-k")
    with pytest.raises(GsError):                          # (round 4's 96-row windows, which covered it, are gone: they spilled)
        gpu_run(a0, b0, 70, args=args(kernel=capi.GS_KERNEL_WINDOW))


def test_window_kernel_soak_against_the_marching_kernel():
    """1080 x 1920, a pattern-forming start, 3000 steps in uneven calls, three contexts in a row: ~750 exchanges of 240
    workgroups each -- one stale or misplaced exchange word anywhere changes bits that the dynamics then spread --
    against the default schedule (the marching kernel), every word of U and V."""
    import torch

    import bench

    rows, cols = 1080, 1920
    u0, v0 = bench.developed_start(rows, cols)
    ref = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_TB))
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
        assert sim.context.info()[0] == "window-r5/strict.op.ds"
        torch.cuda.synchronize()
        for name, a, b in (("U", sp.in_out()[0], sr.in_out()[0]), ("V", sp.in_out()[1], sr.in_out()[1])):
            (_, _, x), = a.torch_views()
            (_, _, y), = b.torch_views()
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"{name} differs after {sum(calls)} steps (context {attempt})"
        sim.context.close()
    (_, _, v), = sr.in_out()[1].torch_views()
    assert float(v.max()) > 0.3
    ref.context.close()


def test_window_launches_whose_workgroups_start_far_apart():
    """Uneven load (MI355X_MICROARCH.md: test every hand-off under it): right before every window launch another stream
    starts kernels that hold compute units for tens to hundreds of microseconds, so the launch's workgroups become
    resident at very different times -- the early ones poll their aprons hundreds of times (the bounded-wait path, the
    check of the abort word every 64 polls), waves of one window start their steps far apart -- for 120 launches of
    uneven lengths, four, two and eight steps per exchange and the default size with its 12-wave edge windows.  Every word
    against the marching kernel; no launch may have given up (the hogs are far shorter than the patience)."""
    import torch

    import bench

    hog_stream = torch.cuda.Stream()
    hog = torch.zeros(1 << 26, device="cuda")
    for (rows, cols), k in (((1080, 1920), 0), ((1024, 2048), 0), ((900, 1600), 0), ((1080, 1920), 2)):
        u0, v0 = bench.developed_start(rows, cols)
        ref = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_TB))
        sr = bench.upload_species(ref, u0, v0)
        sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_WINDOW, fuse_steps=k))
        sp = bench.upload_species(sim, u0, v0)
        rng = np.random.default_rng(rows + k)
        total = 0
        for call in range(30):
            n = int(rng.integers(1, 90))
            with torch.cuda.stream(hog_stream):
                for _ in range(int(rng.integers(1, 4))):      # elementwise kernels over 256 MB: ~0.1 ms each, three per round,
                    hog.mul_(1.0001).add_(0.5).sin_()          # their workgroups draining off the CUs while the launch comes on
            sim.prepare_steps(sp, n)
            ref.prepare_steps(sr, n)
            total += n
            if call % 10 == 9:
                sim.context.sync()
        sim.context.sync()
        ref.context.sync()
        torch.cuda.synchronize()
        st = sim.context.stats()
        assert st["window_fallbacks"] == 0 and sim.context.info()[0].startswith("window-r5/"), (st, sim.context.info())
        for name, a, b in (("U", sp.in_out()[0], sr.in_out()[0]), ("V", sp.in_out()[1], sr.in_out()[1])):
            (_, _, x), = a.torch_views()
            (_, _, y), = b.torch_views()
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"{name} differs after {total} steps at {rows} x {cols}, k = {k}"
        sim.context.close()
        ref.context.close()


def test_images_enqueued_behind_window_launches_never_wait_and_are_right(monkeypatch):
    """The reference's driver loop (simulate/src/main.rs:99-115) on the window kernel: prepare_steps, then the image of
    the newest state enqueued BEHIND the launch (gs_field_download_async no longer waits for a window launch in flight),
    validated when it is waited for.  Every image equals the oracle's V after as many steps -- also the images of launches
    that gave up (patience of one poll from the third call on: those launches and the later ones are run again by the
    marching kernel, each image fetched again right after its own launch's replay, before the next one overwrites the
    planes)."""
    import time

    from grayscott_amd.simulation import pinned_empty
    from tests.helpers import species_from_arrays

    rows, cols, n, calls = 1080, 1920, 64, 6
    u0, v0 = stress_fields((rows, cols), 11)
    refs, ru, rv = [], u0, v0
    for _ in range(calls):
        ru, rv = oracle.run(ru, rv, n, ftz=True)
        refs.append(rv)
    for give_up_from in (None, 2):
        monkeypatch.delenv("GS_HIP_WINDOW_PATIENCE", raising=False)
        sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_WINDOW))
        sp = species_from_arrays(sim, u0, v0)
        images = [pinned_empty((rows, cols)) for _ in range(calls)]
        enqueue_s = []
        for i in range(calls):
            if give_up_from is not None and i == give_up_from:
                monkeypatch.setenv("GS_HIP_WINDOW_PATIENCE", "1")
            sim.prepare_steps(sp, n)
            t0 = time.perf_counter()
            sp.write_result_view_after(images[i])
            enqueue_s.append(time.perf_counter() - t0)
            sim.context.download_wait(in_flight=1)   # two images in flight: the one before this is complete on return
            if i:
                assert_bits_equal(images[i - 1], refs[i - 1], f"image {i - 1} when handed over (give up from call {give_up_from})")
        sim.context.download_wait()
        st = sim.context.stats()
        for i in range(calls):
            assert_bits_equal(images[i], refs[i], f"image {i} (give up from call {give_up_from}: {st})")
        sim.context.sync()
        in_u, in_v, _, _ = sp.in_out()
        assert_bits_equal(in_v.make_scalar_view(sim.context), refs[-1], "V at the end")
        if give_up_from is None:
            assert st["window_fallbacks"] == 0 and sim.context.info()[0].startswith("window"), (st, sim.context.info())
            # enqueueing an image costs microseconds, not the 270 us a 64-step launch takes (it used to wait for it)
            assert sorted(enqueue_s)[len(enqueue_s) // 2] < 150e-6, enqueue_s
        sim.context.close()
