#!/usr/bin/env python3
"""Mid-size grids: the LDS-tile kernel (every tile shape x steps per launch) against the temporally
blocked kernel, the reference's "compute" workload with 256 steps per call (criterion style: warm-up,
then the median of 15 timed calls, each call synchronised).  Prints Mcells x steps / s."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402


def rate(shape, steps=256, **kw):
    sim = Simulation.new(Parameters(), HipArgs(devices=[0], **kw))
    sp = sim.make_species(shape)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        sim.perform_steps(sp, steps)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        sim.perform_steps(sp, steps)
        ts.append(time.perf_counter() - t0)
    label = sim.context.info()[0]
    sim.context.close()
    return shape[0] * shape[1] * steps / statistics.median(ts) / 1e6, label


def main():
    sizes = [(64, 128), (128, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 4096)]
    if len(sys.argv) > 1:                       # e.g. 96x192 384x768
        sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    ks = (4, 6, 8)
    print("| rows x cols | tb (tuned) | auto | " + " | ".join(f"tile{n} K={k}" for n in ("32x64", "16x64", "64x64") for k in ks) + " |")
    print("|---|---|---|" + "---|" * 9)
    for shape in sizes:
        row = [f"{shape[0]} x {shape[1]}"]
        r, label = rate(shape, kernel=capi.GS_KERNEL_TB)
        row.append(f"{r:.0f} ({label})")
        r, label = rate(shape)
        row.append(f"{r:.0f} ({label})")
        for ts in (1, 2, 3):
            for k in ks:
                if ts == 2 and k == 8:
                    row.append("-")      # 2K must stay below the window's 16 rows
                    continue
                r, _ = rate(shape, kernel=capi.GS_KERNEL_TILE, tile_shape=ts, fuse_steps=k)
                row.append(f"{r:.0f}")
        print("| " + " | ".join(row) + " |", flush=True)


if __name__ == "__main__":
    main()
