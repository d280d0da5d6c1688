#!/usr/bin/env python3
"""Shader clock, socket power and energy per cell-step while ONE kernel choice runs for a few seconds on a grid:

    python tools/archive/clock_under_kernel.py ROWS COLS SECONDS VARIANT [VARIANT ...]     (VARIANT = key=value,key=value over HipArgs)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402


def main():
    rows, cols, seconds = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    for spec in sys.argv[4:]:
        kw = {"devices": [0]}
        for kv in spec.split(","):
            if kv:
                k, v = kv.split("=")
                kw[k] = int(v)
        sim = Simulation.new(Parameters(), HipArgs(**kw))
        sp = sim.make_species([rows, cols])
        for _ in range(6):
            sim.perform_steps(sp, 2000)
        t0 = time.perf_counter()
        sim.perform_steps(sp, 4000)
        n = int(seconds * 4000 / (time.perf_counter() - t0)) // 8 * 8
        box = {}

        def work():
            t0 = time.perf_counter()
            sim.context.timer_start()
            sim.prepare_steps(sp, n)
            box["ms"] = sim.context.timer_stop()
            sim.context.sync()
            return (time.perf_counter() - t0,)

        s = bench.sample_clock_and_power(work, 0, float(rows * cols) * n) or {}
        print(json.dumps({"variant": spec, "kernel": sim.context.info()[0], "steps": n,
                          "Mcells_steps_per_s": rows * cols * n / (box["ms"] * 1e-3) / 1e6, "us_per_step": box["ms"] * 1e3 / n,
                          "sclk_MHz": s.get("sclk_MHz"), "power_W": s.get("power_W"), "energy_W": s.get("energy_W"),
                          "pJ_per_cell_step": s.get("energy_pJ_per_cell_step")}), flush=True)
        sim.context.close()


if __name__ == "__main__":
    main()
