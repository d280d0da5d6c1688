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

if __name__ == "__main__":
    main()
