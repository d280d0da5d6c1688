#!/usr/bin/env python3
"""A plain run for profilers: one context, Species::new, `--calls` calls of perform_steps(`--steps`) after a tuning run.

    rocprofv3 --kernel-trace --stats -- python3 tools/run_steps.py --rows 1080 --cols 1920 --steps 1000 --calls 5
HipArgs come from the environment (GS_HIP_KERNEL, GS_HIP_SHARE_TAPS, ...).  Prints one JSON line: kernel label, rate of
the timed calls (HIP events), launches per call."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--tune-steps", type=int, default=4000)
    ap.add_argument("--developed", action="store_true",
                    help="a developed spot pattern instead of Species::new (bench.py's co-headline input: developed_start + 4000 steps)")
    a = ap.parse_args()
    sim = Simulation.new(Parameters(), HipArgs())
    if a.developed:
        from benchkit.legs import developed_start, upload_species
        sp = upload_species(sim, *developed_start(a.rows, a.cols))
        sim.perform_steps(sp, max(a.tune_steps, 4000))
    else:
        sp = sim.make_species([a.rows, a.cols])
        sim.perform_steps(sp, a.tune_steps)
    ctx = sim.context
    l0 = ctx.stats()["launches"]
    ctx.timer_start()
    for _ in range(a.calls):
        sim.prepare_steps(sp, a.steps)
    ms = ctx.timer_stop()
    ctx.sync()
    st = ctx.stats()
    print(json.dumps({"kernel": ctx.info()[0], "input": "developed" if a.developed else "Species::new", "placement": sp.placement, "rows": a.rows, "cols": a.cols, "steps_per_call": a.steps, "calls": a.calls,
                      "Mcells_steps_per_s": a.rows * a.cols * a.steps * a.calls / (ms * 1e-3) / 1e6,
                      "launches_per_call": (st["launches"] - l0) / a.calls, "ms_per_call": ms / a.calls,
                      "window_fallbacks": st["window_fallbacks"],
                      # the layout that ran: chosen by gs_run's tuner, or pinned through GS_HIP_ROWS_PER_BLOCK / _FUSE_STEPS / _COLS_PER_LANE
                      "tuned": ctx.get_tuned(a.rows, a.cols) if ctx.get_tuned(a.rows, a.cols)[0] else
                      (ctx.args.rows_per_block, ctx.args.fuse_steps, ctx.args.cols_per_lane, ctx.args.share_taps),
                      "pinned": bool(ctx.args.rows_per_block)}), flush=True)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
