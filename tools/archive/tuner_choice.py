#!/usr/bin/env python3
"""What gs_run's on-line tuner chooses (unit height, steps per pass, columns per lane, form of difference sharing) on
Species::new and on a developed pattern, several fresh contexts each, and the rate each choice then sustains.

    GS_HIP_TRACE_TUNER=1 python tools/archive/tuner_choice.py [--rows 16384 --cols 16384] [--contexts 3] [--seconds 3]
One JSON line per context: input, tuned configuration, kernel label, Mcells x steps/s over `--seconds` (HIP events)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--contexts", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--data", default="new,developed")
    ap.add_argument("--place", type=int, default=0, help="Species::new with placement by measurement over this many extra blocks")
    a = ap.parse_args()
    rows, cols = a.rows, a.cols
    start = bench.developed_start(rows, cols) if "developed" in a.data else None
    for data in a.data.split(","):
        for i in range(a.contexts):
            sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
            ctx = sim.context
            sp = sim.make_species([rows, cols], place_candidates=a.place) if data == "new" else bench.upload_species(sim, *start)
            tuned = (0, 0, 0, 0)
            for _ in range(10):
                sim.perform_steps(sp, 400)
                tuned = ctx.get_tuned(rows, cols)
                if tuned[0] > 0:
                    break
            steps, t0 = 0, time.time()
            ctx.timer_start()
            while time.time() - t0 < a.seconds:
                sim.prepare_steps(sp, 200)
                ctx.sync()
                steps += 200
            ms = ctx.timer_stop()
            print(json.dumps({"data": data, "context": i, "tuned": tuned, "placement_ms": getattr(sp, "placement", None), "kernel": ctx.info()[0],
                              "Mcells_steps_per_s": rows * cols * steps / (ms * 1e-3) / 1e6}), flush=True)
            del sp
            ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
