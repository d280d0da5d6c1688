"""The kernel that is TIMED, at the sizes it is timed at, under the parity run.

bench.py times the library's default schedule (kernel = AUTO: the temporally blocked kernel, its
parameter-specialised `.op` variant, unit height / steps per pass / columns per lane chosen on
line) on 16384 x 16384 (BASELINE config 3) and the profiles also quote 4096 x 4096 (config 2).
These tests run exactly that schedule at exactly those sizes, long enough to contain tuning passes,
full passes and a remainder pass, and compare every word of U and V with the single-step stream
kernel -- which tests/test_gpu_parity.py ties to the oracle bit for bit on every shape the oracle
can reach -- and, where the oracle can follow, with the oracle itself.
Spec: compute/naive/src/lib.rs:42-83 (arithmetic), compute/shared/src/cpu.rs:30-42 (step; flip).
"""
import numpy as np
import pytest

import oracle
from grayscott_amd import HipArgs, Parameters, Simulation, capi
from tests.helpers import assert_bits_equal, species_from_arrays

pytestmark = pytest.mark.gpu


def _destroy(sim, sp):
    for c in sp.u._pair + sp.v._pair:
        c.destroy()
    sim.context.close()


def _run(u0, v0, steps, call=None, pretune=0, **kw):
    """upload -> perform_steps in calls of `call` steps -> download; returns (U, V, kernel label).
    `pretune` steps on a throw-away Species of the same shape first let the context finish its on-line
    tuning when `steps` alone would be too few passes for it."""
    sim = Simulation.new(Parameters(), HipArgs(devices=[0], **kw))
    if pretune:
        scratch = sim.make_species(list(u0.shape))
        sim.perform_steps(scratch, pretune)
        for c in scratch.u._pair + scratch.v._pair:
            c.destroy()
    sp = species_from_arrays(sim, u0, v0)
    done = 0
    while done < steps:
        n = min(steps - done, call or steps)
        sim.perform_steps(sp, n)
        done += n
    in_u, in_v, _, _ = sp.in_out()
    out = in_u.make_scalar_view(sim.context), in_v.make_scalar_view(sim.context), sim.context.info()[0]
    _destroy(sim, sp)
    return out


def _tiled_random(rows, cols, seed):
    """Random data everywhere without a rows x cols call to the generator: a 256-row block tiled
    down the grid, every 7th row mirrored to break the vertical period."""
    rng = np.random.default_rng(seed)
    base_u = rng.random((256, cols), dtype=np.float32)
    base_v = rng.random((256, cols), dtype=np.float32) * np.float32(0.5)
    u0 = np.tile(base_u, (rows // 256, 1))
    v0 = np.tile(base_v, (rows // 256, 1))
    u0[::7] = u0[::7][:, ::-1]
    return u0, v0


def _seeded(rows, cols):
    """tools/soak.py's pattern-forming start: U = 1, V = 0, one 12 x 12 seed per 40 000 cells, 1 % noise."""
    rng = np.random.default_rng(2024)
    u0 = np.ones((rows, cols), np.float32)
    v0 = np.zeros((rows, cols), np.float32)
    for _ in range(max(4, rows * cols // 40000)):
        r, c = int(rng.integers(0, rows - 12)), int(rng.integers(0, cols - 12))
        u0[r:r + 12, c:c + 12] = 0.5
        v0[r:r + 12, c:c + 12] = 0.25
    u0 += rng.random(u0.shape, dtype=np.float32) * np.float32(0.01)
    v0 += rng.random(v0.shape, dtype=np.float32) * np.float32(0.01)
    return u0, v0


def _assert_production(label):
    # the default schedule, tuned: "tb-k<K>[c<cpl>]/strict.op[.ds|.dx]@<rows>x<bands>" (.ds / .dx: full difference sharing
    # within a lane / across lanes too)
    assert label.startswith("tb-k") and any(f"/strict.op{x}@" in label for x in ("", ".ds", ".dx")), label


def test_default_schedule_16384_403_steps_vs_stream_kernel():
    """BASELINE config 3's grid, data everywhere, 403 steps = tuning passes + full passes + a 3-step
    remainder: the timed kernel against the stream kernel, every word."""
    rows = cols = 16384
    u0, v0 = _tiled_random(rows, cols, 11)
    pu, pv, label = _run(u0, v0, 403)
    _assert_production(label)
    su, sv, slabel = _run(u0, v0, 403, kernel=capi.GS_KERNEL_STREAM)
    assert slabel.startswith("stream")
    assert np.array_equal(pu.view(np.uint32), su.view(np.uint32)), "U: default schedule != stream kernel at 16384^2"
    assert np.array_equal(pv.view(np.uint32), sv.view(np.uint32)), "V: default schedule != stream kernel at 16384^2"
    assert np.isfinite(pu).all() and np.isfinite(pv).all()


def test_default_schedule_16384_developed_pattern_2000_steps():
    """A shortened tools/soak.py as a test: pattern-forming start, 2000 steps in calls of 997 (uneven
    call lengths: remainders and re-entry), default schedule vs stream kernel at 16384 x 16384.  A
    stale read, a missed dependency between passes or a mis-indexed unit anywhere changes bits that
    the chaotic dynamics then spread."""
    rows = cols = 16384
    u0, v0 = _seeded(rows, cols)
    pu, pv, label = _run(u0, v0, 2000, call=997)
    _assert_production(label)
    su, sv, _ = _run(u0, v0, 2000, call=997, kernel=capi.GS_KERNEL_STREAM)
    assert np.array_equal(pu.view(np.uint32), su.view(np.uint32)), "U differs after 2000 steps"
    assert np.array_equal(pv.view(np.uint32), sv.view(np.uint32)), "V differs after 2000 steps"
    assert np.isfinite(pv).all() and float(pv.max()) > 0.3          # the pattern is alive


def test_default_schedule_4096_403_steps_vs_stream_and_oracle():
    """BASELINE config 2's grid.  (a) random data, 403 steps on a context that has finished tuning on
    this shape: default schedule == stream kernel, every word; (b) Species::new start, 403 steps: the region the seed has influenced equals the ORACLE run
    on a crop whose edges the signal cannot have reached, and the far field is the exact fixed point."""
    rows = cols = 4096
    steps = 403
    u0, v0 = _tiled_random(rows, cols, 5)
    pu, pv, label = _run(u0, v0, steps, pretune=4000)
    _assert_production(label)
    su, sv, _ = _run(u0, v0, steps, kernel=capi.GS_KERNEL_STREAM)
    assert np.array_equal(pu.view(np.uint32), su.view(np.uint32)), "U: default schedule != stream kernel at 4096^2"
    assert np.array_equal(pv.view(np.uint32), sv.view(np.uint32)), "V: default schedule != stream kernel at 4096^2"
    del pu, pv, su, sv
    u0, v0 = oracle.init_species(rows, cols)
    gu, gv, label = _run(u0, v0, steps, pretune=4000)
    _assert_production(label)
    (r0, r1), (c0, c1) = oracle.seed_ranges(rows, cols)
    m = steps + 2
    far = np.ones((rows, cols), bool)
    far[r0 - m:r1 + m, c0 - m:c1 + m] = False
    assert (gu[far] == 1.0).all() and (gv[far] == 0.0).all()
    R0, R1, C0, C1 = r0 - 2 * m, r1 + 2 * m, c0 - 2 * m, c1 + 2 * m
    assert R0 >= 0 and C0 >= 0 and R1 <= rows and C1 <= cols
    cu, cv = oracle.run(u0[R0:R1, C0:C1], v0[R0:R1, C0:C1], steps, ftz=True)
    inner = (slice(m, R1 - R0 - m), slice(m, C1 - C0 - m))
    assert_bits_equal(gu[R0:R1, C0:C1][inner], cu[inner], "crop U, 403 steps, default schedule")
    assert_bits_equal(gv[R0:R1, C0:C1][inner], cv[inner], "crop V, 403 steps, default schedule")


def test_config3_as_written_10000_steps_vs_stream_kernel():
    """BASELINE config 3 as written: 16384 x 16384 f32, Species::new, default feed/kill, double-buffered U/V in
    HBM, 10 000 steps -- the library's default schedule in uneven calls (tuning passes, remainders, re-entry)
    against the single-step stream kernel, every word of U and V, compared on the device.
    Spec: compute/naive/src/lib.rs:42-83, compute/shared/src/cpu.rs:30-42."""
    import torch

    rows = cols = 16384
    calls = (4001, 2999, 1777, 1223)
    assert sum(calls) == 10000
    prod = Simulation.new(Parameters(), HipArgs(devices=[0]))
    sp = prod.make_species([rows, cols])
    for n in calls:
        prod.perform_steps(sp, n)
    _assert_production(prod.context.info()[0])
    ref = Simulation.new(Parameters(), HipArgs(devices=[0], kernel=capi.GS_KERNEL_STREAM))
    sr = ref.make_species([rows, cols])
    ref.perform_steps(sr, sum(calls))
    assert ref.context.info()[0].startswith("stream")
    torch.cuda.synchronize()
    for name, a, b in (("U", sp.in_out()[0], sr.in_out()[0]), ("V", sp.in_out()[1], sr.in_out()[1])):
        (_, _, x), = a.torch_views()
        (_, _, y), = b.torch_views()
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"{name} differs after 10000 steps at 16384^2"
        assert bool(torch.isfinite(x[::53]).all())
    (_, _, v), = sp.in_out()[1].torch_views()
    assert float(v.max()) > 0.3 and float((v > 0.1).sum()) > 1e4          # the seed has grown into a pattern
    _destroy(prod, sp)
    _destroy(ref, sr)
