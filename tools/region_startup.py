#!/usr/bin/env python3
"""What does a timed region pay for starting on an idle chip?  T(n) for regions of n passes (n = 1 ... 12) after a
sync, each the median of 9; a straight line a + b n gives the per-pass time b and the start-up cost a.

    python tools/region_startup.py [rows cols] [idle_ms]
"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 2 else 16384
    cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    idle_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    sp = sim.make_species([rows, cols])
    ctx = sim.context
    for _ in range(8):
        if ctx.get_tuned(rows, cols)[0] > 0:
            break
        sim.perform_steps(sp, 400)
    k = ctx.get_tuned(rows, cols)[1] or 4
    sim.perform_steps(sp, 400)
    print(f"grid {rows}x{cols}, {k} steps per pass, idle before each region {idle_ms} ms (+ the sync)")
    xs, ys = [], []
    for n in (1, 2, 3, 4, 5, 6, 8, 12, 25):
        t = []
        for _ in range(9):
            ctx.sync()
            if idle_ms:
                time.sleep(idle_ms / 1e3)
            ctx.timer_start()
            sim.prepare_steps(sp, n * k)
            t.append(ctx.timer_stop())
        m = statistics.median(t)
        xs.append(n)
        ys.append(m)
        print(f"  {n:3d} passes: median {m:8.4f} ms  ({m / n:.4f} per pass; min {min(t):.4f} max {max(t):.4f})", flush=True)
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    a = my - b * mx
    print(f"fit: T(n) = {a:.4f} + {b:.4f} n ms  -> a region of 5 passes runs at {5 * b / (a + 5 * b):.3f} of the steady rate")


if __name__ == "__main__":
    main()
