#!/usr/bin/env python3
"""Wide-window kernel (gs_run_band_k, tile_shape 4..6 = 48 / 64 / 80 rows x 128 columns): parity against the oracle
under the zero-halo rule on a few grids, then its rate next to kernel = auto on grids of 1-4 M cells.

    python tools/band_probe.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402
from tests.helpers import gpu_run, stress_fields  # noqa: E402


def main():
    ok = True
    for shape in ((200, 300), (97, 131), (333, 517), (1, 40), (40, 1), (130, 257)):
        u0, v0 = stress_fields(shape, 3)
        for ts in (4, 5, 6):
            for k, steps in ((4, 9), (3, 7), (8, 17), (1, 2)):
                ref = oracle.run(u0, v0, steps, ftz=True, boundary=oracle.ZERO_HALO)
                got = gpu_run(u0, v0, steps, args=HipArgs(devices=[0], kernel=capi.GS_KERNEL_TILE, tile_shape=ts, fuse_steps=k,
                                                          boundary=capi.GS_BOUNDARY_ZERO_HALO))
                same = got[0].tobytes() == ref[0].tobytes() and got[1].tobytes() == ref[1].tobytes()
                ok = ok and same
                if not same:
                    bad = np.argwhere(got[0].view(np.uint32) != ref[0].view(np.uint32))
                    print(f"MISMATCH {shape} tile_shape {ts} K {k}: {len(bad)} cells, first {bad[:3].tolist()}  {got[2][0]}")
    print("parity (zero-halo rule):", "ok" if ok else "FAILED", flush=True)
    for rows, cols in ((1080, 1920), (1024, 2048), (768, 1536), (2048, 2048), (1440, 2560), (512, 1024)):
        cells = rows * cols
        line = [f"{rows}x{cols}"]
        for label, kw in (("auto", {}), ("band48 K4", dict(kernel=capi.GS_KERNEL_TILE, tile_shape=4, fuse_steps=4)),
                          ("band64 K4", dict(kernel=capi.GS_KERNEL_TILE, tile_shape=5, fuse_steps=4)),
                          ("band80 K4", dict(kernel=capi.GS_KERNEL_TILE, tile_shape=6, fuse_steps=4)),
                          ("band80 K6", dict(kernel=capi.GS_KERNEL_TILE, tile_shape=6, fuse_steps=6)),
                          ("band80 K8", dict(kernel=capi.GS_KERNEL_TILE, tile_shape=6, fuse_steps=8))):
            sim = Simulation.new(Parameters(), HipArgs(devices=[0], boundary=capi.GS_BOUNDARY_ZERO_HALO, **kw))
            sp = sim.make_species([rows, cols])
            for _ in range(3):
                sim.perform_steps(sp, 2000)
            times = []
            for _ in range(5):
                sim.context.timer_start()
                sim.prepare_steps(sp, 2000)
                times.append(sim.context.timer_stop())
            ms = sorted(times)[2]
            line.append(f"{label} {cells * 2000 / ms / 1e3:.0f} k ({sim.context.info()[0]})")
            sim.context.close()
        print(" | ".join(line), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
