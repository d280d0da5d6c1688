"""Known-answer tests that pin the CPU oracle (CPU only, no GPU).

The reference holds no numerical tests for this path ("parity unpinned", SURVEY.md section
8c), so the oracle is pinned by answers derived BY HAND from the cited reference source:
  compute/naive/src/lib.rs:42-83   (step)        data/src/concentration/mod.rs:36-59 (init)
  data/src/parameters.rs:72-83,116-122 (constants)
and by bit-for-bit agreement between two independently written restatements
(oracle/gs_oracle.c and oracle/numpy_ref.py), and against the committed golden vectors.
"""
import glob
import os

import numpy as np
import pytest

import oracle
from oracle import numpy_ref

f32 = np.float32
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---- KAT 1: Species::new --------------------------------------------------------------
@pytest.mark.parametrize("shape,rows,cols", [
    ((1080, 1920), (468, 536), (840, 960)),
    ((64, 128), (24, 28), (56, 64)),
    ((16384, 16384), (7164, 8188), (7168, 8192)),
    ((4, 4), (0, 0), (1, 2)),   # saturating_sub: 4*7/16-4 -> 0, 4*8/16-4 -> 0 (empty row range)
    ((16, 16), (3, 4), (7, 8)),
])
def test_kat1_seed_ranges(shape, rows, cols):
    assert oracle.seed_ranges(*shape) == (rows, cols)


def test_kat1_species_new_fields():
    u, v = oracle.init_species(64, 128)
    eu = np.ones((64, 128), f32)
    ev = np.zeros((64, 128), f32)
    eu[24:28, 56:64] = 0
    ev[24:28, 56:64] = 1
    assert (u == eu).all() and (v == ev).all()
    nu, nv = numpy_ref.init_species(64, 128)
    assert (u == nu).all() and (v == nv).all()


# ---- KAT 2: uniform U=1, V=0 is a fixed point of the naive rule, borders included --------
@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (9, 1), (5, 7), (32, 48)])
def test_kat2_uniform_fixed_point(shape):
    u = np.ones(shape, f32)
    v = np.zeros(shape, f32)
    ou, ov = oracle.step(u, v)
    assert (bits(ou) == bits(u)).all() and (bits(ov) == bits(v)).all()


# ---- KAT 3: one step from the initial condition -------------------------------------------
def test_kat3_first_step_values():
    rows, cols = 1080, 1920
    u0, v0 = oracle.init_species(rows, cols)
    u, v = oracle.step(u0, v0)
    F, K, DU, DV = f32(0.014), f32(0.054), f32(0.1), f32(0.05)
    one, zero = f32(1), f32(0)
    # deep inside the seed: u=0, v=1, all neighbours equal -> acc = 0
    #   du = (DU*0 - 0) + F*(1-0) = F ; dv = (DV*0 + 0) - (F+K)*1
    assert u[500, 900] == (zero + F * one) and v[500, 900] == one + ((zero + zero) - (F + K) * one)
    assert abs(float(v[500, 900]) - 0.932) < 1e-6 and abs(float(u[500, 900]) - 0.014) < 1e-9
    # outside cell just above the seed (row 467), its three lower neighbours in the seed:
    #   acc_u = .25*(0-1)+.5*(0-1)+.25*(0-1) = -1 ; acc_v = +1 ; uvv = 0
    assert u[467, 900] == one + ((DU * f32(-1) - zero) + F * zero)       # 0.9
    assert v[467, 900] == zero + ((DV * one + zero) - (F + K) * zero)    # 0.05
    # inside cell on the seed's top row (468): three upper neighbours outside
    #   acc_u = +1, acc_v = -1, u=0, v=1, uvv = 0
    assert u[468, 900] == zero + ((DU * one - zero) + F * one)           # 0.114
    assert v[468, 900] == one + ((DV * f32(-1) + zero) - (F + K) * one)  # 0.882
    # outside corner cell (536, 959): seed neighbours (535,958) w=.25 and (535,959) w=.5
    acc = (zero + f32(.25) * f32(-1)) + f32(.5) * f32(-1)
    assert u[536, 959] == one + ((DU * acc - zero) + F * zero)           # 0.925
    assert v[536, 959] == zero + ((DV * (-acc) + zero) - (F + K) * zero) # 0.0375
    assert abs(float(u[536, 959]) - 0.925) < 1e-7 and abs(float(v[536, 959]) - 0.0375) < 1e-8


# ---- KAT 4: the top-row / left-column weight anchoring of the clipped window ---------------
def test_kat4_top_row_quirk():
    u = np.ones((3, 3), f32)
    u[1, 1] = 0
    v = np.zeros((3, 3), f32)
    ou, _ = oracle.step(u, v)
    # cell (0,1): window rows {0,1}; weight row 0 on row 0, weight row 1 = [.5, 0, .5] on row 1
    # -> the neighbour below, u[1][1], gets weight 0 (an aligned stencil would give .5)
    assert ou[0, 1] == f32(1.0)
    # cell (1,0): window cols {0,1}; weight column 0 on col 0, column 1 on col 1 -> w[1][1] = 0
    assert ou[1, 0] == f32(1.0)
    # the aligned sides do see the hole: cell (2,1) has (1,1) at window index (0,1) -> w = .5
    assert ou[2, 1] == f32(1) + (f32(0.1) * (f32(.5) * f32(-1)))          # 0.95
    assert ou[1, 2] == f32(1) + (f32(0.1) * (f32(.5) * f32(-1)))
    # corner (0,0): window {0,1}x{0,1}; (1,1) sits at window index (1,1) -> weight 0
    assert ou[0, 0] == f32(1.0)
    # corner (2,2): (1,1) at window index (0,0) -> weight .25
    assert ou[2, 2] == f32(1) + (f32(0.1) * (f32(.25) * f32(-1)))         # 0.975


def test_kat4_single_row_and_column():
    # R = 1: window is one row, weights row 0 = [.25, .5, .25] anchored at the window start
    u = np.array([[1, 0, 1, 1]], f32)
    v = np.zeros_like(u)
    ou, _ = oracle.step(u, v)
    # cell (0,0): window cols {0,1} -> w[0][0]*(u0-u0) + w[0][1]*(u1-u0) = .5*(0-1)
    assert ou[0, 0] == f32(1) + f32(0.1) * (f32(.5) * f32(-1))
    # cell (0,2): window cols {1,2,3} -> w[0][0]*(0-1) = -.25
    assert ou[0, 2] == f32(1) + f32(0.1) * (f32(.25) * f32(-1))
    # cell (0,1): u=0: .25*(1-0) + .25*(1-0) = .5 ; du = .1*.5 + F*(1-0)
    assert ou[0, 1] == f32(0) + ((f32(0.1) * f32(.5) - f32(0)) + f32(0.014) * f32(1))


# ---- two independent restatements agree bit for bit ----------------------------------------
@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (17, 33), (64, 128), (250, 130)])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_c_oracle_equals_numpy_restatement(shape, seed):
    rng = np.random.default_rng(seed)
    u = rng.random(shape, dtype=f32)
    v = (rng.random(shape, dtype=f32) * f32(0.5)).astype(f32)
    a, b = (u, v), (u, v)
    for _ in range(4):
        a = oracle.step(*a, ftz=False)
        b = numpy_ref.step(*b)
        assert (bits(a[0]) == bits(b[0])).all() and (bits(a[1]) == bits(b[1])).all()


def test_c_oracle_equals_numpy_restatement_under_ftz():
    """The sub-normal front: both restatements under MXCSR.FTZ, 40 steps from Species::new."""
    u0, v0 = oracle.init_species(64, 128)
    a = oracle.run(u0, v0, 40, ftz=True, nthreads=1)
    was = oracle.set_ftz(True)
    try:
        b = numpy_ref.run(u0, v0, 40)
    finally:
        oracle.set_ftz(was)
    assert (bits(a[0]) == bits(b[0])).all() and (bits(a[1]) == bits(b[1])).all()
    c = oracle.run(u0, v0, 40, ftz=False)
    assert np.count_nonzero(bits(a[1]) != bits(c[1])) > 500  # the flush rule does matter here


def test_non_default_weights_and_rates():
    """Patra-Karttunen weights (data/src/parameters.rs:99-104) and a non-unit time step."""
    p = oracle.default_params()
    w = np.array([[1 / 6, 4 / 6, 1 / 6], [4 / 6, 0, 4 / 6], [1 / 6, 4 / 6, 1 / 6]], f32)
    p.set_weights(w)
    p.dt, p.feed, p.kill = 0.5, 0.03, 0.06
    q = numpy_ref.default_params()
    q.update(w=w, dt=f32(0.5), feed=f32(0.03), kill=f32(0.06))
    rng = np.random.default_rng(5)
    u = rng.random((19, 23), dtype=f32)
    v = (rng.random((19, 23), dtype=f32) * f32(0.5)).astype(f32)
    a = oracle.run(u, v, 6, p, ftz=False)
    b = numpy_ref.run(u, v, 6, q)
    assert (bits(a[0]) == bits(b[0])).all() and (bits(a[1]) == bits(b[1])).all()


def test_threading_and_row_partition_are_bit_exact():
    u0, v0 = oracle.init_species(96, 80)
    a = oracle.run(u0, v0, 25, nthreads=1)
    b = oracle.run(u0, v0, 25, nthreads=4)
    assert (bits(a[0]) == bits(b[0])).all() and (bits(a[1]) == bits(b[1])).all()
    ou, ov = np.empty_like(u0), np.empty_like(v0)
    p = oracle.default_params()
    oracle.step_rows(u0, v0, ou, ov, p, 0, 37)
    oracle.step_rows(u0, v0, ou, ov, p, 37, 96)
    su, sv = oracle.step(u0, v0)
    assert (bits(ou) == bits(su)).all() and (bits(ov) == bits(sv)).all()


# ---- self-consistency with the survey's scratch restatement ----------
def test_survey_checksums_1080x1920():
    u0, v0 = oracle.init_species(1080, 1920)
    u, v = oracle.run(u0, v0, 1)
    assert abs(u.sum(dtype=np.float64) - 2065554.239995) < 1e-5
    assert abs(v.sum(dtype=np.float64) - 7605.119844) < 1e-5
    u, v = oracle.run(u0, v0, 100)
    assert abs(u.sum(dtype=np.float64) - 2068985.006178) < 1e-5
    assert abs(v.sum(dtype=np.float64) - 379.712084) < 1e-5


# ---- committed golden vectors ---------------------------------------------------------------
def test_golden_species_new():
    g = np.load(os.path.join(GOLDEN, "species_new_64x128.npz"))
    u0, v0 = oracle.init_species(64, 128)
    for steps in (1, 10, 100, 1000):
        u, v = oracle.run(u0, v0, steps)
        assert (bits(u) == bits(g[f"u_{steps}"])).all(), steps
        assert (bits(v) == bits(g[f"v_{steps}"])).all(), steps


def test_golden_stress_and_ftz():
    files = sorted(glob.glob(os.path.join(GOLDEN, "stress_*.npz")))
    assert len(files) == 8
    for path in files:
        g = np.load(path)
        for steps in (1, 20):
            u, v = oracle.run(g["u0"], g["v0"], steps)
            assert (bits(u) == bits(g[f"u_{steps}"])).all(), (path, steps)
            assert (bits(v) == bits(g[f"v_{steps}"])).all(), (path, steps)
    g = np.load(os.path.join(GOLDEN, "ftz_front_64x128.npz"))
    u0, v0 = oracle.init_species(64, 128)
    for steps in (30, 40):
        u, v = oracle.run(u0, v0, steps, ftz=True)
        assert (bits(v) == bits(g[f"v_{steps}"])).all() and (bits(u) == bits(g[f"u_{steps}"])).all()
        u, v = oracle.run(u0, v0, steps, ftz=False)
        assert (bits(v) == bits(g[f"v_{steps}_noftz"])).all()


# ---- the reference's other boundary rule (zero halo), checker of GS_BOUNDARY_ZERO_HALO --------
def test_zero_halo_known_answers():
    """Uniform U=1, V=0 (KAT 2): a fixed point of the clipped-window rule, but
    under the zero-halo rule a border cell loses Du * (sum of the weights that fall outside) per
    step: 1.0 for an edge cell, .25+.5+.25+.5+.25 = 1.75 for a corner."""
    u = np.ones((5, 6), np.float32)
    v = np.zeros((5, 6), np.float32)
    ou, ov = oracle.step(u, v, boundary=oracle.ZERO_HALO)
    f = np.float32
    edge = f(1.0) + (f(0.1) * (f(0.0) + f(0.25) * f(-1) + f(0.5) * f(-1) + f(0.25) * f(-1)))
    assert ou[2, 3] == f(1.0) and ou[0, 2] == edge and ou[2, 0] == edge and ou[4, 3] == edge and ou[1, 5] == edge
    assert abs(float(ou[0, 0]) - 0.825) < 1e-6 and ou[0, 0] == ou[4, 5] == ou[0, 5] == ou[4, 0]
    assert (ov == 0).all()
    cu, cv = oracle.step(u, v)                          # the clipped rule keeps it fixed
    assert (cu == 1).all() and (cv == 0).all()


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (17, 33), (64, 128)])
def test_zero_halo_c_checker_equals_numpy_restatement(shape):
    rng = np.random.default_rng(11)
    u = rng.random(shape, dtype=np.float32)
    v = (rng.random(shape, dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    for _ in range(5):
        cu, cv = oracle.step(u, v, ftz=False, boundary=oracle.ZERO_HALO)
        nu, nv = numpy_ref.step_zero_halo(u, v)
        assert cu.tobytes() == nu.tobytes() and cv.tobytes() == nv.tobytes()
        u, v = cu, cv


def test_boundary_rules_agree_away_from_the_border():
    """The two rules differ on the border only; the difference travels one cell per step."""
    u0, v0 = oracle.init_species(64, 96)
    steps = 7
    a = oracle.run(u0, v0, steps)
    b = oracle.run(u0, v0, steps, boundary=oracle.ZERO_HALO)
    k = steps
    assert a[0][k:-k, k:-k].tobytes() == b[0][k:-k, k:-k].tobytes()
    assert a[1][k:-k, k:-k].tobytes() == b[1][k:-k, k:-k].tobytes()
    assert a[0].tobytes() != b[0].tobytes()


def test_zero_halo_checker_against_the_parallel_backend_port():
    """Same boundary rule as the reference's block/parallel family, different association (FMA,
    three accumulator chains): agreement to rounding on the WHOLE grid, border included."""
    from oracle import cpu_parallel

    w = cpu_parallel.simd_width()
    rows, cols = 8 * w, 72
    sim = cpu_parallel.ParallelSimulation(rows, cols, num_threads=2)
    sim.perform_steps(60)
    u0, v0 = oracle.init_species(rows, cols)
    ref_u, ref_v = oracle.run(u0, v0, 60, boundary=oracle.ZERO_HALO)
    assert np.max(np.abs(sim.read(0) - ref_u)) <= 1e-5 * np.max(np.abs(ref_u))
    assert np.max(np.abs(sim.read(1) - ref_v)) <= 1e-5 * max(np.max(np.abs(ref_v)), 1e-3)
    # ... which the clipped rule does not give (the border differs at the 1e-1 level)
    clip_u, _ = oracle.run(u0, v0, 60)
    assert np.max(np.abs(sim.read(0) - clip_u)) > 1e-2
    sim.close()


def test_golden_zero_halo_and_stencils():
    """Fixtures of the widened rows (the other boundary rule, the named stencils)."""
    from grayscott_amd.simulation import STENCILS

    g = np.load(os.path.join(GOLDEN, "zero_halo_64x128.npz"))
    u0, v0 = oracle.init_species(64, 128)
    for steps in (1, 10, 100):
        u, v = oracle.run(u0, v0, steps, ftz=True, boundary=oracle.ZERO_HALO)
        assert u.tobytes() == g[f"u_{steps}"].tobytes() and v.tobytes() == g[f"v_{steps}"].tobytes()
    for steps in (1, 20):
        u, v = oracle.run(g["stress_u0"], g["stress_v0"], steps, ftz=True, boundary=oracle.ZERO_HALO)
        assert u.tobytes() == g[f"stress_u_{steps}"].tobytes() and v.tobytes() == g[f"stress_v_{steps}"].tobytes()
    g = np.load(os.path.join(GOLDEN, "stencils_37x60.npz"))
    for name, w in STENCILS.items():
        q = oracle.default_params()
        q.set_weights(w)
        q.dt = 0.25 if name == "pretty" else 1.0
        u, v = oracle.run(g["u0"], g["v0"], 12, q, ftz=True)
        assert u.tobytes() == g[f"u_{name}"].tobytes() and v.tobytes() == g[f"v_{name}"].tobytes()
        assert np.isfinite(u).all() and np.isfinite(v).all()

