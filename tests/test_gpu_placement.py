"""Placement by measurement (gs_fields_place): the four planes of a Species are given the best of 4 + n candidate
allocations as single-step probes time them, and keep their contents.  Results cannot depend on it: Species::new through
the placed planes against the oracle (data/src/concentration/mod.rs:36-59, compute/naive/src/lib.rs:42-83), and a Species
placed in the middle of a run."""
import ctypes

import numpy as np
import pytest

import oracle
from grayscott_amd import GsError, HipArgs, HipConcentration, Parameters, Simulation, capi
from tests.helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    assert capi.device_count() >= 1, "no MI355X visible"


@pytest.mark.parametrize("shape,candidates", [((300, 700), 3), ((64, 128), 1), ((1, 1), 2), ((1500, 2100), 4), ((200, 300), 40), ((96, 256), 124)])
def test_species_new_on_placed_planes_matches_the_oracle(shape, candidates):
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    sp = sim.make_species(shape, place_candidates=candidates)
    first, best = sp.placement
    assert first > 0 and 0 < best <= first * 1.0001, sp.placement
    sim.perform_steps(sp, 37)
    u0, v0 = oracle.init_species(*shape)
    ref_u, ref_v = oracle.run(u0, v0, 37, ftz=True)
    in_u, in_v, _, _ = sp.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref_u, f"U {shape}")
    assert_bits_equal(in_v.make_scalar_view(sim.context), ref_v, f"V {shape}")
    sim.context.close()


def test_a_species_placed_in_the_middle_of_a_run_keeps_its_planes():
    """Species.place on planes that hold a state (both slots, ghost rows and all): the run goes on as if nothing had
    happened."""
    shape = (210, 640)
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    sp = sim.make_species(shape)
    sim.perform_steps(sp, 21)
    first, best = sp.place(5)
    assert first > 0 and 0 < best <= first * 1.0001
    sim.perform_steps(sp, 30)
    u0, v0 = oracle.init_species(*shape)
    ref_u, ref_v = oracle.run(u0, v0, 51, ftz=True)
    in_u, in_v, _, _ = sp.in_out()
    assert_bits_equal(in_u.make_scalar_view(sim.context), ref_u, "U")
    assert_bits_equal(in_v.make_scalar_view(sim.context), ref_v, "V")
    sim.context.close()


def test_placed_planes_keep_their_contents_and_the_call_checks_its_arguments():
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    ctx = sim.context
    planes = [HipConcentration.ones(ctx, (90, 333)) for _ in range(4)]
    planes[2].fill_slice(ctx, [range(3, 40), range(100, 222)], 0.25)
    before = [p.make_scalar_view(ctx).copy() for p in planes]
    arr = (ctypes.c_void_p * 4)(*[p.handle for p in planes])
    capi.check(ctx._lib.gs_fields_place(ctx.handle, arr, 2, None, None))
    for p, b in zip(planes, before):
        assert np.array_equal(p.make_scalar_view(ctx), b)          # whichever blocks they have now
    for bad in (0, 125):
        with pytest.raises(GsError):
            capi.check(ctx._lib.gs_fields_place(ctx.handle, arr, bad, None, None))
    dup = (ctypes.c_void_p * 4)(planes[0].handle, planes[1].handle, planes[2].handle, planes[0].handle)
    with pytest.raises(GsError):
        capi.check(ctx._lib.gs_fields_place(ctx.handle, dup, 2, None, None))
    other = HipConcentration(ctx, (90, 334))
    mixed = (ctypes.c_void_p * 4)(planes[0].handle, planes[1].handle, planes[2].handle, other.handle)
    with pytest.raises(GsError):
        capi.check(ctx._lib.gs_fields_place(ctx.handle, mixed, 2, None, None))
    chain = Simulation.new(Parameters(), HipArgs(devices=[0, 0]))
    cp = [HipConcentration(chain.context, (90, 333)) for _ in range(4)]
    with pytest.raises(GsError):
        capi.check(chain.context._lib.gs_fields_place(chain.context.handle, (ctypes.c_void_p * 4)(*[p.handle for p in cp]), 2, None, None))
    chain.context.close()
    ctx.close()
    assert np.float32(0) == 0
