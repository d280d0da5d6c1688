"""Colour mapping (SURVEY section 8f row 4 remainder): data-to-pics' per-pixel work behind gs_field_colormap."""
import os

import numpy as np
import pytest

from oracle import colormap_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def palette():
    return np.load(os.path.join(GOLDEN, "inferno_256.npy"))


def test_inferno_table_and_index_rule():
    p = palette()
    assert p.shape == (256, 3) and p.dtype == np.uint8
    # first / last entries of d3-scale-chromatic's inferno string, which colorous ports
    assert p[0].tolist() == [0x00, 0x00, 0x04] and p[1].tolist() == [0x01, 0x00, 0x05]
    assert p[-2].tolist() == [0xfa, 0xfd, 0xa1] and p[-1].tolist() == [0xfc, 0xff, 0xa4]
    v = np.array([[0.0, 0.25, 0.5, 0.4999, 0.75, -0.1, np.nan, np.inf, 1.0 / 512, 1.0 / 513]], np.float32)
    got = colormap_ref.colormap(v, p)
    # t = 2 v: 0 -> entry 0, 0.5 -> entry 128, 1.0 (and beyond) -> the last entry, negatives / NaN -> entry 0
    want = [0, 128, 255, 255, 255, 0, 0, 255, 1, 0]
    assert [int(np.flatnonzero((p == px).all(axis=1))[0]) for px in got[0]] == want


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_gpu_colormap_matches_the_restatement(devices):
    from grayscott_amd import HipArgs, Parameters, Simulation
    from tests.helpers import species_from_arrays

    rng = np.random.default_rng(9)
    shape = (301, 517)
    v = (rng.random(shape, dtype=np.float32) * np.float32(0.7) - np.float32(0.05)).astype(np.float32)
    v[3, 4] = np.nan
    v[5, 6] = np.inf
    v[7, 8] = -np.inf
    v[9, 10] = np.float32(0.5)
    sim = Simulation.new(Parameters(), HipArgs(devices=devices))
    sp = species_from_arrays(sim, np.ones(shape, np.float32), v)
    p = palette()
    got = sp.access_result(lambda plane, ctx: plane.colormap(ctx, p))
    assert got.shape == shape + (3,) and got.dtype == np.uint8
    assert np.array_equal(got, colormap_ref.colormap(v, p))
    # another palette size and scale: the rule is generic
    small = p[::37]
    assert np.array_equal(sp.v.in_out()[0].colormap(sim.context, small, scale=1.5), colormap_ref.colormap(v, small, scale=1.5))
    # a developed run paints something: Species::new after a few steps has V in (0, 1]
    sim2 = Simulation.new(Parameters(), HipArgs(devices=[0]))
    s2 = sim2.make_species([64, 128])
    sim2.perform_steps(s2, 10)
    img = s2.access_result(lambda plane, ctx: plane.colormap(ctx, p))
    assert np.array_equal(img, colormap_ref.colormap(s2.make_result_view(), p)) and len(np.unique(img.reshape(-1, 3), axis=0)) > 4
