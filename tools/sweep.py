#!/usr/bin/env python3
"""In-process A/B timing of kernel variants (interleaved rounds, median + min; guide rule 24).

    python tools/sweep.py [--rows 16384 --cols 16384 --steps 48 --rounds 5] VARIANT...

A VARIANT is comma-separated key=value pairs over HipArgs fields, e.g.
    kernel=2,rows_per_block=32   kernel=3,fuse_steps=2,math=1
Prints Mcells*steps/s and algorithmic GB/s (16 B per cell-step) per variant, from HIP events.
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    sims = []
    for spec in a.variants:
        kw = {"devices": [0]}
        for kv in spec.split(","):
            k, v = kv.split("=")
            if k == "slabs":
                kw["devices"] = [0] * int(v)
            else:
                kw[k] = int(v)
        sim = Simulation.new(Parameters(), HipArgs(**kw))
        sp = sim.make_species([a.rows, a.cols])
        sim.perform_steps(sp, a.steps)  # warm-up
        sim.context.sync()
        sims.append((spec, sim, sp, []))
    for _ in range(a.rounds):
        for spec, sim, sp, times in sims:
            sim.context.timer_start()
            sim.perform_steps(sp, a.steps)
            times.append(sim.context.timer_stop())
    cells = a.rows * a.cols
    print(f"grid {a.rows}x{a.cols}, {a.steps} steps per timing, {a.rounds} rounds")
    for spec, sim, sp, times in sims:
        med, best = statistics.median(times), min(times)
        name, _ = sim.context.info()
        print(f"{spec:45s} {name:18s} median {cells*a.steps/med/1e3:10.0f} Mcs/s {16*cells*a.steps/med/1e6:8.0f} GB/s"
              f" | best {cells*a.steps/best/1e3:10.0f} Mcs/s  ms/step {med/a.steps:.4f}", flush=True)


if __name__ == "__main__":
    main()
