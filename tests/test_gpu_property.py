"""Randomised parity: hypothesis draws shapes, parameters, step counts and scheduling options
(kernel incl. the LDS-window kernel and its window shapes, fused steps, unit height, row bands, in-process
slabs, or nothing pinned at all: kernel = auto's own choice); every combination must be
bit-identical to the oracle.  Scheduling options never change results -- that is the property."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
from grayscott_amd import Parameters, capi
from tests.helpers import assert_bits_equal, gpu_run, oracle_params
from tests.test_gpu_parity import args

pytestmark = pytest.mark.gpu

POW2 = [0.0, 0.125, 0.25, 0.5, 1.0]


@st.composite
def cases(draw):
    rows = draw(st.integers(1, 140))
    cols = draw(st.one_of(st.integers(1, 40), st.integers(240, 270), st.integers(480, 530), st.integers(990, 1040)))
    steps = draw(st.integers(1, 13))
    seed = draw(st.integers(0, 2 ** 16))
    kernel = draw(st.sampled_from([capi.GS_KERNEL_AUTO, capi.GS_KERNEL_AUTO, capi.GS_KERNEL_STREAM, capi.GS_KERNEL_TB, capi.GS_KERNEL_SIMPLE,
                                   capi.GS_KERNEL_LDS, capi.GS_KERNEL_TILE, capi.GS_KERNEL_WINDOW]))
    fuse = draw(st.integers(0, 8 if kernel == capi.GS_KERNEL_TILE else 4))
    tile_shape = draw(st.integers(0, 3))
    rpb = draw(st.sampled_from([0, 1, 2, 3, 5, 8, 16, 33]))
    if kernel == capi.GS_KERNEL_WINDOW:           # steps per exchange (even) and full window rows
        fuse = draw(st.sampled_from([0, 2, 4, 6, 8]))
        rpb = draw(st.sampled_from([0, 80]))
    split = draw(st.integers(0, 4))
    slabs = draw(st.integers(1, 4))
    cpl = draw(st.sampled_from([0, 1, 2, 4]))
    general = draw(st.integers(0, 1))
    graph = draw(st.integers(0, 1))
    boundary = draw(st.integers(0, 1))
    w = [[draw(st.sampled_from(POW2)) for _ in range(3)] for _ in range(3)]
    if draw(st.booleans()):                       # the default side weights: specialised kernels
        w[0][1] = w[1][0] = w[1][2] = w[2][1] = 0.5
    p = Parameters(weights=tuple(tuple(r) for r in w),
                   diffusion_rate_u=draw(st.sampled_from([0.1, 0.2, 0.05])),
                   diffusion_rate_v=draw(st.sampled_from([0.05, 0.1])),
                   feed_rate=draw(st.sampled_from([0.014, 0.03, 0.0])),
                   kill_rate=draw(st.sampled_from([0.054, 0.06])),
                   time_step=draw(st.sampled_from([1.0, 0.5, 0.75])))
    tiny = draw(st.booleans())  # sprinkle values near the flush-to-zero threshold
    if kernel == capi.GS_KERNEL_AUTO and draw(st.booleans()):
        # nothing pinned, one slab: what kernel = auto picks by grid size (resident / window / marching kernel)
        fuse = rpb = split = cpl = graph = 0
        slabs = 1
    return rows, cols, steps, seed, kernel, fuse, rpb, split, slabs, p, tiny, cpl, general, graph, boundary, tile_shape


@settings(max_examples=int(os.environ.get("GS_PROPERTY_EXAMPLES", "80")), deadline=None, suppress_health_check=list(HealthCheck))
@given(cases())
def test_any_schedule_matches_the_oracle(built, case):
    rows, cols, steps, seed, kernel, fuse, rpb, split, slabs, p, tiny, cpl, general, graph, boundary, tile_shape = case
    steps = steps * 9 if graph else steps   # long enough for at least one batch of 16 passes
    slabs = min(slabs, rows)
    rng = np.random.default_rng(seed)
    u0 = rng.random((rows, cols), dtype=np.float32)
    v0 = (rng.random((rows, cols), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    if tiny:
        mask = rng.random((rows, cols)) < 0.3
        v0[mask] = (v0[mask] * np.float32(1e-37)).astype(np.float32)
        u0[rng.random((rows, cols)) < 0.05] = np.float32(3e-38)
    if kernel in (capi.GS_KERNEL_STREAM, capi.GS_KERNEL_SIMPLE, capi.GS_KERNEL_LDS):
        fuse = 0
    ref_u, ref_v = oracle.run(u0, v0, steps, oracle_params(p), ftz=True, boundary=boundary)
    got_u, got_v, info = gpu_run(u0, v0, steps, params=p,
                                 args=args(kernel=kernel, fuse_steps=fuse, rows_per_block=rpb, split=split,
                                           devices=[0] * slabs, cols_per_lane=cpl, general_kernels=general, use_graph=graph, boundary=boundary,
                                           tile_shape=tile_shape))
    what = (f"{rows}x{cols} steps={steps} kernel={info[0]} fuse={fuse} rpb={rpb} split={split} slabs={slabs} tile_shape={tile_shape} "
            f"cpl={cpl} general={general} graph={graph} boundary={boundary} {p}")
    assert_bits_equal(got_u, ref_u, "U " + what)
    assert_bits_equal(got_v, ref_v, "V " + what)


def _window_plan_exists(rows, cols, boundary, window_rows, k):
    """Is the grid one round of windows on this device (the planner's own answer, gs_debug_window_plan)?"""
    import ctypes

    import torch

    lib = capi.load()
    f = lib.gs_debug_window_plan
    f.restype = ctypes.c_int32
    f.argtypes = [ctypes.c_uint64, ctypes.c_uint64] + [ctypes.c_int32] * 5 + [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    return f(rows, cols, cus, boundary, 1, window_rows, k, None, 0, None, None) > 0


@st.composite
def larger_cases(draw):
    """Grids from 40 k to 2.7 M cells: the window kernel's three windows, its hand-over to the marching kernel at
    1.5 M cells, single- and multi-round launches of the marching kernel (halved edge units, tapered tails)."""
    rows = draw(st.integers(200, 1300))
    cols = draw(st.integers(200, 2100))
    steps = draw(st.integers(1, 12))
    seed = draw(st.integers(0, 2 ** 16))
    kernel = draw(st.sampled_from([capi.GS_KERNEL_AUTO, capi.GS_KERNEL_AUTO, capi.GS_KERNEL_TB, capi.GS_KERNEL_TILE, capi.GS_KERNEL_WINDOW]))
    pinned = kernel != capi.GS_KERNEL_AUTO or draw(st.booleans())
    fuse = draw(st.integers(0, 8 if kernel == capi.GS_KERNEL_TILE else 4)) if pinned else 0
    rpb = draw(st.sampled_from([0, 0, 3, 8, 10, 16, 21, 39, 64])) if pinned else 0
    if kernel == capi.GS_KERNEL_WINDOW:           # (grids that are not one round of windows: the test falls back to auto)
        fuse = draw(st.sampled_from([0, 4, 8]))
        rpb = draw(st.sampled_from([0, 80]))
    cpl = draw(st.sampled_from([0, 1, 2, 4])) if pinned else 0
    slabs = draw(st.sampled_from([1, 1, 2, 3])) if pinned else 1
    tile_shape = draw(st.integers(0, 3))
    boundary = draw(st.integers(0, 1))
    default_params = draw(st.booleans())
    return rows, cols, steps, seed, kernel, fuse, rpb, cpl, slabs, tile_shape, boundary, default_params


@settings(max_examples=int(os.environ.get("GS_PROPERTY_EXAMPLES_LARGER", "12")), deadline=None, suppress_health_check=list(HealthCheck))
@given(larger_cases())
def test_any_schedule_matches_the_oracle_larger_grids(built, case):
    rows, cols, steps, seed, kernel, fuse, rpb, cpl, slabs, tile_shape, boundary, default_params = case
    rng = np.random.default_rng(seed)
    u0 = rng.random((rows, cols), dtype=np.float32)
    v0 = (rng.random((rows, cols), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    p = Parameters() if default_params else Parameters(feed_rate=0.03, kill_rate=0.06, time_step=0.5)
    ref_u, ref_v = oracle.run(u0, v0, steps, oracle_params(p), ftz=True, boundary=boundary)
    if kernel == capi.GS_KERNEL_WINDOW and slabs == 1 and not _window_plan_exists(rows, cols, boundary, rpb, fuse):
        kernel, fuse, rpb = capi.GS_KERNEL_AUTO, 0, 0
    got_u, got_v, info = gpu_run(u0, v0, steps, params=p,
                                 args=args(kernel=kernel, fuse_steps=fuse, rows_per_block=rpb, devices=[0] * slabs,
                                           cols_per_lane=cpl, boundary=boundary, tile_shape=tile_shape))
    what = (f"{rows}x{cols} steps={steps} kernel={info[0]} fuse={fuse} rpb={rpb} slabs={slabs} cpl={cpl} "
            f"tile_shape={tile_shape} boundary={boundary} default_params={default_params}")
    assert_bits_equal(got_u, ref_u, "U " + what)
    assert_bits_equal(got_v, ref_v, "V " + what)
