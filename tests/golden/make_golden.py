"""Generates tests/golden/*.npz with the C oracle (oracle/gs_oracle.c), AFTER it has passed
the known-answer tests (tests/test_oracle_kat.py).  The reference has no fixtures for this
path and cannot run here (Rust, no toolchain), so these vectors pin the oracle against
itself across compilers/machines and give the GPU tests data that does not need the oracle
at run time.  Re-run:  python tests/golden/make_golden.py

Contents (float32 bit patterns; FTZ on, as under the reference's DenormalsFlusher):
  species_new_64x128.npz   Species::new([64,128]) after 1, 10, 100, 1000 steps (default params)
  stress_*.npz             random U~[0,1), V~[0,0.5) (numpy default_rng(seed)) after 1 and 20 steps
  ftz_front_64x128.npz     Species::new([64,128]) after 30 and 40 steps, with FTZ and without: V's
                           diffusion front is in the sub-normal range there (hundreds of cells
                           differ between the two), which pins the flush-to-zero rule
  zero_halo_64x128.npz     the other boundary rule (GS_BOUNDARY_ZERO_HALO): Species::new([64,128])
                           after 1, 10, 100 steps and a 17x33 stress field after 1 and 20 steps
  stencils_37x60.npz       a stress field after 12 steps with each of the reference's named stencils
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

STRESS = [((1, 1), 0), ((1, 7), 1), ((7, 1), 2), ((2, 2), 0), ((3, 5), 1), ((17, 33), 2),
          ((64, 128), 0), ((250, 130), 1)]


def stress_fields(shape, seed):
    rng = np.random.default_rng(seed)
    u = rng.random(shape, dtype=np.float32)
    v = (rng.random(shape, dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    return u, v


def main():
    p = oracle.default_params()
    u0, v0 = oracle.init_species(64, 128)
    out = {}
    for steps in (1, 10, 100, 1000):
        u, v = oracle.run(u0, v0, steps, p, ftz=True)
        out[f"u_{steps}"], out[f"v_{steps}"] = u, v
    np.savez_compressed(os.path.join(HERE, "species_new_64x128.npz"), **out)

    for shape, seed in STRESS:
        u0, v0 = stress_fields(shape, seed)
        out = {"u0": u0, "v0": v0}
        for steps in (1, 20):
            u, v = oracle.run(u0, v0, steps, p, ftz=True)
            out[f"u_{steps}"], out[f"v_{steps}"] = u, v
        np.savez_compressed(os.path.join(HERE, f"stress_{shape[0]}x{shape[1]}_seed{seed}.npz"), **out)

    u0, v0 = oracle.init_species(64, 128)
    out, ndiff = {}, 0
    for steps in (30, 40):
        u, v = oracle.run(u0, v0, steps, p, ftz=True)
        un, vn = oracle.run(u0, v0, steps, p, ftz=False)
        out[f"u_{steps}"], out[f"v_{steps}"] = u, v
        out[f"u_{steps}_noftz"], out[f"v_{steps}_noftz"] = un, vn
        ndiff += int(np.count_nonzero(v.view(np.uint32) != vn.view(np.uint32)))
    np.savez_compressed(os.path.join(HERE, "ftz_front_64x128.npz"), **out)
    print("cells differing between FTZ and no-FTZ runs:", ndiff)
    assert ndiff > 1000

    u0, v0 = oracle.init_species(64, 128)
    out = {}
    for steps in (1, 10, 100):
        u, v = oracle.run(u0, v0, steps, p, ftz=True, boundary=oracle.ZERO_HALO)
        out[f"u_{steps}"], out[f"v_{steps}"] = u, v
    su, sv = stress_fields((17, 33), 2)
    out["stress_u0"], out["stress_v0"] = su, sv
    for steps in (1, 20):
        u, v = oracle.run(su, sv, steps, p, ftz=True, boundary=oracle.ZERO_HALO)
        out[f"stress_u_{steps}"], out[f"stress_v_{steps}"] = u, v
    np.savez_compressed(os.path.join(HERE, "zero_halo_64x128.npz"), **out)

    from grayscott_amd.simulation import STENCILS   # the weight tables only; no GPU involved
    su, sv = stress_fields((37, 60), 4)
    out = {"u0": su, "v0": sv}
    for name, w in STENCILS.items():
        q = oracle.default_params()
        q.set_weights(w)
        q.dt = 0.25 if name == "pretty" else 1.0
        u, v = oracle.run(su, sv, 12, q, ftz=True)
        out[f"u_{name}"], out[f"v_{name}"] = u, v
    np.savez_compressed(os.path.join(HERE, "stencils_37x60.npz"), **out)

if __name__ == "__main__":
    main()
