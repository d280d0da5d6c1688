"""Full difference sharing in the temporally blocked kernel (gs_options.share_taps; cells_vshare and cells_xshare in
grayscott_amd/csrc/gs_march.h): the S / SE / SW taps of a row are carried to the next row, where they are --
negated -- its N / NW / NE taps; in the second form (share_taps = 3, ".op.dx") the differences that cross a lane
boundary are also computed once, by the lane on the right, and read by the lane on the left as DPP operands.  The
reference forms every tap afresh (compute/naive/src/lib.rs:63-71); both shared forms must give the same BITS: ordinary data, data around the flush-to-zero threshold, signed zeros and equal neighbours
(where a tap is an exact or a flushed zero of either sign), non-finite values, and the stencils / rates for which the
variant must NOT run.  Everything against the CPU oracle, through the C ABI.
"""
import numpy as np
import pytest

import oracle
from grayscott_amd import HipArgs, Parameters, Simulation, capi
from tests.helpers import assert_bits_equal, gpu_run, oracle_params, species_from_arrays, stress_fields

pytestmark = pytest.mark.gpu


def args(**kw):
    kw.setdefault("devices", [0])
    return HipArgs(**kw)


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    assert capi.device_count() >= 1, "no MI355X visible"


def _tiny_fields(shape, seed):
    """Most of the grid within a few binades of the smallest normal f32, both signs (products and differences flush)."""
    rng = np.random.default_rng(seed)
    u, v = stress_fields(shape, seed)
    scale = np.float32(2.0) ** rng.integers(-130, -118, size=shape).astype(np.float32)
    sign = np.where(rng.random(shape) < 0.5, np.float32(-1), np.float32(1))
    tiny = np.zeros(shape, bool)
    tiny[:, : (shape[1] * 3) // 5] = True
    return (np.where(tiny, u * scale * sign, u).astype(np.float32),
            np.where(tiny, v * scale * sign, v).astype(np.float32))


def _few_values_fields(shape, seed, values):
    rng = np.random.default_rng(seed)
    values = np.asarray(values, np.float32)
    u0 = values[rng.integers(0, len(values), size=shape)]
    v0 = values[rng.integers(0, len(values), size=shape)]
    # runs of equal neighbours along rows and columns, as a smooth field has them
    u0[1::2, :] = np.where(rng.random(u0[1::2, :].shape) < 0.4, u0[0:-1:2, :][: u0[1::2, :].shape[0]], u0[1::2, :])
    v0[:, 1::2] = np.where(rng.random(v0[:, 1::2].shape) < 0.4, v0[:, 0:-1:2][:, : v0[:, 1::2].shape[1]], v0[:, 1::2])
    return u0, v0


# (gs_options.share_taps, what the launch label ends in): within a lane / across lanes too
MODES = [(1, ".op.ds"), (3, ".op.dx")]
mode_cases = pytest.mark.parametrize("mode", MODES, ids=["ds", "dx"])


def shared(mode):
    return dict(kernel=capi.GS_KERNEL_TB, cols_per_lane=2, share_taps=mode[0])


@mode_cases
@pytest.mark.parametrize("fuse", [4, 3, 2])
@pytest.mark.parametrize("rpb", [1, 5, 12, 40])
def test_shared_taps_bit_exact(fuse, rpb, mode):
    """Interior units of every height (1 row: nothing but ramp-up ticks and one stored row; 40: several trips of the
    six-tick loop and its remainders), 2 to 4 fused steps, strips and chunks on every side of the interior ones."""
    for shape, maker, seed in [((61, 700), stress_fields, 21), ((64, 500), _tiny_fields, 22), ((130, 380), stress_fields, 23)]:
        u0, v0 = maker(shape, seed)
        for steps in (1, fuse, 2 * fuse + 1, 23):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(fuse_steps=fuse, rows_per_block=rpb, **shared(mode)))
            if steps >= fuse:
                assert info[0].startswith(f"tb-k{fuse}c2/strict{mode[1]}"), info
            assert_bits_equal(got_u, ref_u, f"U {info[0]} {shape} steps {steps} rpb {rpb}")
            assert_bits_equal(got_v, ref_v, f"V {info[0]} {shape} steps {steps} rpb {rpb}")


@mode_cases
def test_shared_taps_signed_zeros_and_equal_neighbours(mode):
    """Taps that are exact zeros of either sign, differences that flush: the carried tap has the other sign of zero than
    the tap the reference forms, and the accumulator must not be able to tell.  Also with feed = kill = 0, where the sign
    of a zero accumulator reaches the output (du = Du * acc - 0 + 0 * (1 - u))."""
    values = [0.0, -0.0, 0.0, -0.0, 2.0 ** -127, -(2.0 ** -127), 2.0 ** -126, -(2.0 ** -126), 3 * 2.0 ** -126, 2.0 ** -125,
              0.25, 0.5, 1.0, 1.0]
    for p in (Parameters(), Parameters(feed_rate=0.0, kill_rate=0.0)):
        for shape, seed in [((40, 520), 31), ((64, 300), 32)]:
            for vals in (values, [0.0, -0.0]):
                u0, v0 = _few_values_fields(shape, seed, vals)
                for steps in (1, 4, 9):
                    ref_u, ref_v = oracle.run(u0, v0, steps, oracle_params(p), ftz=True)
                    got_u, got_v, info = gpu_run(u0, v0, steps, params=p, args=args(fuse_steps=4, rows_per_block=6, **shared(mode)))
                    assert_bits_equal(got_u, ref_u, f"U {info[0]} {shape} steps {steps} {p}")
                    assert_bits_equal(got_v, ref_v, f"V {info[0]} {shape} steps {steps} {p}")


@mode_cases
def test_a_positive_zero_among_negative_zeros(mode):
    """The case a folded `0 - x` would get wrong: a cell +0 whose eight neighbours are -0 (every tap the reference forms
    is -0, its accumulator stays +0; the carried taps are +0), feed = 0 so that nothing hides the accumulator's sign."""
    shape = (40, 400)
    u0 = np.full(shape, -0.0, np.float32)
    v0 = np.full(shape, -0.0, np.float32)
    u0[3::5, 2::3] = 0.0
    v0[2::4, 1::5] = 0.0
    p = Parameters(feed_rate=0.0, kill_rate=0.0)
    for steps in (1, 2, 4, 8):
        ref_u, ref_v = oracle.run(u0, v0, steps, oracle_params(p), ftz=True)
        got_u, got_v, info = gpu_run(u0, v0, steps, params=p, args=args(fuse_steps=4, rows_per_block=8, **shared(mode)))
        assert mode[1] in info[0] or steps < 4, info
        assert_bits_equal(got_u, ref_u, f"U {info[0]} steps {steps}")
        assert_bits_equal(got_v, ref_v, f"V {info[0]} steps {steps}")


@mode_cases
def test_shared_taps_spread_non_finite_values_like_the_reference(mode):
    u0, v0 = stress_fields((48, 520), 41)
    u0[20, 250] = np.nan
    v0[30, 130] = np.inf
    u0[10, 400] = -np.inf
    for steps in (1, 4, 6):
        ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True)
        got_u, got_v, info = gpu_run(u0, v0, steps, args=args(fuse_steps=4, rows_per_block=8, **shared(mode)))
        # NaN payloads are not part of the contract: NaNs must sit in the same cells
        assert np.array_equal(np.isnan(got_u), np.isnan(ref_u)) and np.array_equal(np.isnan(got_v), np.isnan(ref_v)), info
        fin = ~np.isnan(ref_u) & ~np.isnan(ref_v)
        assert_bits_equal(np.where(fin, got_u, 0).astype(np.float32), np.where(fin, ref_u, 0).astype(np.float32), f"U {steps}")
        assert_bits_equal(np.where(fin, got_v, 0).astype(np.float32), np.where(fin, ref_v, 0).astype(np.float32), f"V {steps}")


def test_variant_only_where_its_conditions_hold():
    """Side weights 0.5, dt == 1 and pairwise equal diagonal weights, or the plain specialised / general kernel runs:
    same bits either way, the label says which."""
    sym = ((0.125, 0.5, 0.375), (0.5, 0.0, 0.5), (0.375, 0.5, 0.125))      # w00 == w22, w02 == w20
    asym = ((0.125, 0.5, 0.375), (0.5, 0.0, 0.5), (0.75, 0.5, 1.0))
    cases = [(Parameters(), 1, ".op.ds"), (Parameters(), 3, ".op.dx"), (Parameters(), 2, ".op"),
             (Parameters(weights=sym), 1, ".op.ds"), (Parameters(weights=sym), 3, ".op.dx"),
             (Parameters(weights=asym), 1, ".op"), (Parameters(weights=asym), 3, ".op"), (Parameters(time_step=0.5), 3, ".op"),
             (Parameters(weights=((1, 1, 1), (1, 0, 1), (1, 1, 1))), 3, "strict")]
    u0, v0 = stress_fields((50, 600), 51)
    for p, share, suffix in cases:
        ref_u, ref_v = oracle.run(u0, v0, 9, oracle_params(p), ftz=True)
        got_u, got_v, info = gpu_run(u0, v0, 9, params=p, args=args(kernel=capi.GS_KERNEL_TB, cols_per_lane=2, fuse_steps=4,
                                                                    rows_per_block=10, share_taps=share))
        assert info[0].split("@")[0].endswith(suffix), (info, p, share)
        assert_bits_equal(got_u, ref_u, f"U {info[0]} {p}")
        assert_bits_equal(got_v, ref_v, f"V {info[0]} {p}")
    # general_kernels = 1 switches every specialisation off
    got_u, got_v, info = gpu_run(u0, v0, 9, args=args(kernel=capi.GS_KERNEL_TB, cols_per_lane=2, fuse_steps=4, rows_per_block=10,
                                                      share_taps=1, general_kernels=1))
    assert info[0].split("@")[0].endswith("strict"), info


def test_species_new_and_a_developing_pattern_at_a_few_megacells():
    """A grid of many interior units per launch, library defaults but for the pinned sharing: Species::new for 600 steps
    (uneven calls) and a developing pattern for 400, shared against unshared (same kernels otherwise), a crop of each
    against the oracle."""
    rows, cols = 1536, 2300
    rng = np.random.default_rng(7)
    u0 = np.ones((rows, cols), np.float32)
    v0 = np.zeros((rows, cols), np.float32)
    for _ in range(120):
        r, c = int(rng.integers(0, rows - 12)), int(rng.integers(0, cols - 12))
        u0[r:r + 12, c:c + 12] = 0.5
        v0[r:r + 12, c:c + 12] = 0.25
    u0 += rng.random(u0.shape, dtype=np.float32) * np.float32(0.01)
    v0 += rng.random(v0.shape, dtype=np.float32) * np.float32(0.01)
    outs = []
    for share in (1, 3, 2):
        # (a launch of one round of wave slots: the in-step form, 16-wave workgroups with 132 / 65 KB of halo boards)
        sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_TB, cols_per_lane=2, share_taps=share, no_tune=1))
        sp_new = sim.make_species([rows, cols])
        sp_dev = species_from_arrays(sim, u0, v0)
        for n in (197, 3, 400):
            sim.perform_steps(sp_new, n)
        sim.perform_steps(sp_dev, 400)
        label = sim.context.info()[0]
        assert label.split("@")[0] == "tb-k4c2f/strict" + {1: ".op.ds", 3: ".op.dx", 2: ".op"}[share], label
        outs.append([x.make_scalar_view(sim.context) for x in sp_new.in_out()[:2] + sp_dev.in_out()[:2]])
        sim.context.close()
    for a, b, c, what in zip(outs[0], outs[1], outs[2], ("new U", "new V", "pattern U", "pattern V")):
        assert_bits_equal(a, c, what + " shared vs unshared")
        assert_bits_equal(b, c, what + " shared across lanes vs unshared")
    # the oracle on the small pattern start, 40 steps, through the same configurations
    ref_u, ref_v = oracle.run(u0[:300, :600].copy(), v0[:300, :600].copy(), 40, ftz=True)
    for share, suffix in MODES:
        got_u, got_v, info = gpu_run(u0[:300, :600].copy(), v0[:300, :600].copy(), 40,
                                     args=args(kernel=capi.GS_KERNEL_TB, cols_per_lane=2, share_taps=share, no_tune=1))
        assert suffix in info[0], info
        assert_bits_equal(got_u, ref_u, "pattern crop U " + suffix)
        assert_bits_equal(got_v, ref_v, "pattern crop V " + suffix)


def test_the_tuner_decides_on_sharing_and_results_do_not_depend_on_it():
    """share_taps = 0: gs_run's on-line tuner times the configuration it chose (sharing across lanes) without sharing
    (phase E) on passes of the run itself and keeps the faster -- sharing within a lane only is never its choice; whatever it keeps, and while it is still trying, the planes equal
    those of a run with sharing pinned off."""
    rows, cols = 1700, 2300
    outs, tuned = [], None
    for share in (0, 2):
        sim = Simulation.new(Parameters(), args(kernel=capi.GS_KERNEL_TB, share_taps=share))
        sp = sim.make_species([rows, cols])
        for n in (50, 1200, 1200, 1200, 353):
            sim.perform_steps(sp, n)
        if share == 0:
            tuned = sim.context.get_tuned(rows, cols)
        outs.append([x.make_scalar_view(sim.context) for x in sp.in_out()[:2]])
        sim.context.close()
    assert tuned[0] > 0 and tuned[3] in (2, 3), tuned            # a choice was made, sharing included
    assert_bits_equal(outs[0][0], outs[1][0], f"U, tuner's choice {tuned} vs sharing off")
    assert_bits_equal(outs[0][1], outs[1][1], f"V, tuner's choice {tuned} vs sharing off")


@mode_cases
@pytest.mark.parametrize("kw", [dict(boundary=capi.GS_BOUNDARY_ZERO_HALO), dict(devices=[0, 0, 0]), dict(use_graph=1), dict(split=2),
                                dict(devices=[0, 0], boundary=capi.GS_BOUNDARY_ZERO_HALO)])
def test_shared_taps_under_the_other_schedules(kw, mode):
    """The sharing variant inside everything else gs_run can do with the marching kernel: the zero-halo rule (edge units
    change, interior ones share), slab chains (interior units next to a seam read the neighbour's ghost rows, the
    boundary bands are 4-row units), hipGraph replay, row bands."""
    boundary = kw.get("boundary", capi.GS_BOUNDARY_CLIPPED)
    for shape, seed in [((150, 700), 61), ((97, 1250), 62)]:
        u0, v0 = stress_fields(shape, seed)
        for steps in (9, 70):
            ref_u, ref_v = oracle.run(u0, v0, steps, ftz=True, boundary=boundary)
            a = dict(rows_per_block=9, **shared(mode))
            a.update(kw)
            got_u, got_v, info = gpu_run(u0, v0, steps, args=args(**a))
            assert mode[1] in info[0], info
            assert_bits_equal(got_u, ref_u, f"U {info[0]} {shape} steps {steps} {kw}")
            assert_bits_equal(got_v, ref_v, f"V {info[0]} {shape} steps {steps} {kw}")
