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


def test_make_species_places_large_species_by_default_and_small_ones_never():
    """The library's default (all three hosts): a Species of >= 2^26 cells on a context with one slab is placed by
    measurement with at most 12 extra blocks; smaller ones and slab chains are left where hipMalloc put them."""
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    assert sim.context.args.place_candidates == 12
    small = sim.make_species((1080, 1920))
    assert small.placement is None and sim.context.place_stats() == (0, 0)
    big = sim.make_species((8192, 8192))                       # 2^26 cells: planes of 256 MiB
    first, best = big.placement
    probes, drawn = sim.context.place_stats()
    assert first > 0 and 0 < best <= first * 1.0001 and probes >= 6 and 0 <= drawn <= 12, (big.placement, probes, drawn)   # (256 MiB planes: no deep stage)
    # Species::new survives the move: U = 1, V = 0 but for the seed rectangle (data/src/concentration/mod.rs:36-59)
    in_u, in_v, _, _ = big.in_out()
    v = in_v.make_scalar_view(sim.context)
    assert float(v.sum(dtype=np.float64)) == (8192 // 16) * (8192 // 16) and v[8192 * 7 // 16 - 4, 8192 * 7 // 16] == 1.0
    off = Simulation.new(Parameters(), HipArgs(devices=[0], place_candidates=0))
    assert off.make_species((8192, 8192)).placement is None
    off.context.close()
    chain = Simulation.new(Parameters(), HipArgs(devices=[0, 0]))
    assert chain.make_species((8192, 8192)).placement is None   # (several local slabs: not placed, not an error)
    chain.context.close()
    sim.context.close()


def test_placement_at_the_headline_size_separates_u_from_v():
    """16384^2: after placement each slot's (U, V) pair is a cross-group pair -- the probe pass over it takes 0.72-0.79 ms
    per GiB pair where two blocks of one group take 0.86-0.96 (profiles/r06_placement.md) -- unless the box handed out one
    group only among 4 + 48 blocks, in which case all 48 were drawn.  The planes' contents move with them."""
    sim = Simulation.new(Parameters(), HipArgs(devices=[0], kernel=capi.GS_KERNEL_STREAM))
    sp = sim.make_species((16384, 16384))
    first, best = sp.placement
    probes, drawn = sim.context.place_stats()
    assert 0 < best <= first * 1.0001 and drawn <= 48
    assert best < 0.83 or drawn == 48, (first, best, probes, drawn)      # (beyond 12 draws: the deep stage, one probe per block)
    sim.perform_steps(sp, 3)
    # against an unplaced Species of the same context: same bits
    ref = sim.make_species((16384, 16384), place_candidates=0)
    sim.perform_steps(ref, 3)
    import torch
    for a, b in zip(sp.in_out()[:2], ref.in_out()[:2]):
        for (_, _, x), (_, _, y) in zip(a.torch_views(), b.torch_views()):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    sim.context.close()


@pytest.mark.parametrize("force", ["1,0,3,2", "1,2,3,0", "4,0,1,2", "0,1,3,2", "5,4,1,0", "3,2,1,0", "0,1,2,3", "2,3,0,4"])
def test_every_shape_of_move_keeps_every_plane(force, monkeypatch):
    """The planes' moves are device copies one at a time, a plane moving when nobody holds its chosen block: chains resolve in
    order, planes that wait for each other (swaps, a cycle through all four) go through a spare block.  GS_HIP_PLACE_FORCE
    dictates the arrangement (0-3: the planes' own blocks, 4 and up: drawn ones); four planes with four different contents,
    ghost rows included, must come out as they went in, and a simulation on them must go on bit for bit."""
    monkeypatch.setenv("GS_HIP_PLACE_FORCE", force)
    shape = (130, 257)
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    ctx = sim.context
    u0, v0 = np.random.default_rng(7).random(shape, dtype=np.float32), np.random.default_rng(8).random(shape, dtype=np.float32) * np.float32(0.5)
    from tests.helpers import species_from_arrays
    sp = species_from_arrays(sim, u0, v0)
    sim.perform_steps(sp, 6)                     # both slots hold a state, ghost rows are in use
    before = [p.make_scalar_view(ctx).copy() for p in sp.in_out()]
    first, best = sp.place(2)
    assert first > 0 and best > 0
    after = [p.make_scalar_view(ctx) for p in sp.in_out()]
    for i, (a, b) in enumerate(zip(after, before)):
        assert a.tobytes() == b.tobytes(), f"plane {i} changed under GS_HIP_PLACE_FORCE={force}"
    sim.perform_steps(sp, 9)
    ref_u, ref_v = oracle.run(u0, v0, 15, ftz=True)
    in_u, in_v, _, _ = sp.in_out()
    assert_bits_equal(in_u.make_scalar_view(ctx), ref_u, f"U after moves {force}")
    assert_bits_equal(in_v.make_scalar_view(ctx), ref_v, f"V after moves {force}")
    ctx.close()


def test_deep_stage_goes_on_with_one_probe_per_block(monkeypatch):
    """Beyond `candidates` draws the search goes on (planes of >= 512 MiB, more than half of the device's memory free) up to
    4 x `candidates` blocks, timing each new block against ONE block of the best arrangement -- and against the rest of it
    only when it lies in another region.  GS_HIP_PLACE_ALL=1 keeps it from stopping early, so every draw happens here."""
    monkeypatch.setenv("GS_HIP_PLACE_ALL", "1")
    shape = (16384, 8192)                                    # planes of 512 MiB
    sim = Simulation.new(Parameters(), HipArgs(devices=[0], kernel=capi.GS_KERNEL_STREAM))
    sp = sim.make_species(shape, place_candidates=2)
    first, best = sp.placement
    probes, drawn = sim.context.place_stats()
    assert drawn == 8 and 0 < best <= first * 1.0001, (sp.placement, probes, drawn)
    # 6 pairs, two blocks timed against all held (4 + 5), six blocks with 1 probe -- or 4 when of another region
    assert 6 + 9 + 6 <= probes <= 6 + 9 + 6 * 4, probes
    monkeypatch.delenv("GS_HIP_PLACE_ALL")
    monkeypatch.setenv("GS_HIP_PLACE_DEEP", "0")
    sim.perform_steps(sp, 2)
    ref = sim.make_species(shape, place_candidates=0)
    sim.perform_steps(ref, 2)
    import torch
    for a, b in zip(sp.in_out()[:2], ref.in_out()[:2]):
        for (_, _, x), (_, _, y) in zip(a.torch_views(), b.torch_views()):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    sim.context.close()
