#!/usr/bin/env python3
"""How often does the on-line tuner settle on the same layout?  N fresh contexts on one grid, each tuned inside one
long call and timed afterwards.

    python tools/archive/tuner_repeat.py ROWS COLS [contexts=8] [steps=4000] [calls=1]

Prints per context the layout the tuner kept after `calls` calls of `steps` steps (short calls never wait for the
tuner's windows: a driver loop of 34-step calls is tuned over its first few hundred calls) and the rate of three
more calls (median; of 4000 steps when the calls are short).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
    calls = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    timed = max(steps, 4000)
    picks = {}
    for i in range(n):
        sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
        sp = sim.make_species([rows, cols])
        for _ in range(calls):
            sim.perform_steps(sp, steps)
            sim.context.sync()
        rates = []
        for _ in range(3):
            sim.context.timer_start()
            sim.perform_steps(sp, timed)
            rates.append(rows * cols * timed / sim.context.timer_stop() / 1e3)
        name, _ = sim.context.info()
        picks.setdefault(name, []).append(sorted(rates)[1])
        print(f"context {i}: {name:28s} {sorted(rates)[1]:10.0f} Mcells x steps / s", flush=True)
        sim.context.close()
    for name, r in sorted(picks.items()):
        print(f"{name:28s} kept {len(r)} of {n} times, {min(r):.0f} - {max(r):.0f}")


if __name__ == "__main__":
    main()
