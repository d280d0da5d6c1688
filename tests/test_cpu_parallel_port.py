"""The CPU timing baseline (oracle/gs_cpu_parallel.c: the reference's parallel(block(autovec))
backend restated) -- CPU only.

It is not the parity target (zero-halo boundary rule, FMA association), but it is a second,
differently formulated statement of the same interior arithmetic taken from the reference
(compute/autovec/src/lib.rs:63-115: sum(w * elem) with the corrected centre weight -sum(w),
three accumulator chains), so agreement with the naive oracle away from the border
cross-checks both restatements."""
import numpy as np
import pytest

import oracle
from oracle import cpu_parallel

f32 = np.float32


def test_species_new_layout_round_trip():
    w = cpu_parallel.simd_width()
    rows, cols = 8 * w, 40
    sim = cpu_parallel.ParallelSimulation(rows, cols, num_threads=2)
    u0, v0 = oracle.init_species(rows, cols)
    assert (sim.read(0) == u0).all() and (sim.read(1) == v0).all()
    with pytest.raises(ValueError):  # rows must be a multiple of the SIMD width (simd/mod.rs:83-87)
        cpu_parallel.ParallelSimulation(rows + 1, cols)


def test_zero_halo_boundary_rule_by_hand():
    """Uniform U=1, V=0: under the zero-halo rule of the block/parallel family the border is NOT
    a fixed point (boundary-rule summary): an edge cell loses Du * (sum of the
    weights that fall outside) per step."""
    w = cpu_parallel.simd_width()
    rows, cols = 4 * w, 24
    sim = cpu_parallel.ParallelSimulation(rows, cols, num_threads=2)
    # overwrite the seed: read-modify is not exposed, so use a shape whose seed is empty instead
    sim.close()
    rows, cols = w, 8  # rows*7/16-4 saturates to 0 and rows*8/16-4 too -> empty seed for w <= 8
    assert oracle.seed_ranges(rows, cols)[0] == (0, 0)
    sim = cpu_parallel.ParallelSimulation(rows, cols, num_threads=1)
    sim.perform_steps(1)
    u = sim.read(0)
    one = f32(1)
    # an edge (non-corner) cell has 3 outside neighbours with weights .25+.5+.25 = 1
    assert abs(float(u[0, 3]) - (1 - 0.1 * 1.0)) < 1e-6
    # a corner cell has 5 outside neighbours: .25+.5+.25+.5+.25 = 1.75
    assert abs(float(u[0, 0]) - (1 - 0.1 * 1.75)) < 1e-6
    if rows > 2:
        assert u[rows // 2, cols // 2] == one  # interior of a uniform field is a fixed point
    # naive keeps the whole uniform field fixed (KAT 2) -- the two rules really differ
    nu, _ = oracle.step(np.ones((rows, cols), f32), np.zeros((rows, cols), f32))
    assert (nu == 1).all()


@pytest.mark.parametrize("steps", [1, 10])
def test_interior_agrees_with_naive_oracle(steps):
    w = cpu_parallel.simd_width()
    rows, cols = 16 * w, 160
    sim = cpu_parallel.ParallelSimulation(rows, cols, num_threads=3, l1_block_size=4096, l2_block_size=65536,
                                          seq_block_size=32768)  # small blocks: exercise every split
    sim.perform_steps(steps)
    u0, v0 = oracle.init_species(rows, cols)
    ru, rv = oracle.run(u0, v0, steps)
    m = steps + 1  # the two boundary rules differ on the border; the difference travels 1 cell/step
    du = np.abs(sim.read(0)[m:-m, m:-m] - ru[m:-m, m:-m]).max()
    dv = np.abs(sim.read(1)[m:-m, m:-m] - rv[m:-m, m:-m]).max()
    assert du <= 5e-7 and dv <= 5e-7, (du, dv)
    if steps == 1:  # one step, exact-weight products: only association/FMA differences, <= 1 ulp
        assert du <= 1.2e-7 and dv <= 1.2e-7


def test_thread_count_and_blocking_do_not_change_results():
    w = cpu_parallel.simd_width()
    rows, cols = 8 * w, 96
    a = cpu_parallel.ParallelSimulation(rows, cols, num_threads=1)
    b = cpu_parallel.ParallelSimulation(rows, cols, num_threads=4, l1_block_size=2048, l2_block_size=16384,
                                        seq_block_size=8192)
    a.perform_steps(7)
    b.perform_steps(7)
    assert a.read(0).tobytes() == b.read(0).tobytes() and a.read(1).tobytes() == b.read(1).tobytes()
