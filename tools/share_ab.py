#!/usr/bin/env python3
"""The three forms of the marching kernel's arithmetic -- no difference sharing (".op"), sharing within a lane
(".op.ds"), sharing across lanes too (".op.dx") -- on THE SAME PLANES: one context, one Species per input, the form
switched between timing windows with gs_ctx_set_tuned.  (Separate contexts draw separate planes, and at 1.1-1.2 M
Mcells x steps/s the kernel's rate depends on where its four planes landed in HBM -- profiles/r05_cross_lane.md -- so
only windows on the same planes compare kernels.)  `--contexts` fresh contexts show the spread between placements;
`--place N` draws Species::new's planes by measurement (gs_fields_place) out of 4 + N blocks.

    python tools/share_ab.py [--rows 16384 --cols 16384] [--seconds 3] [--rounds 2] [--contexts 2] [--place 0]
One JSON line per window: input, context, form, Mcells x steps/s, shader clock, socket power, pJ per cell-step.
Throughput unit: compute/shared/src/benchmark.rs:55-60."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402

FORMS = {"op": 2, "ds": 1, "dx": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--place", type=int, default=0)
    ap.add_argument("--rows-per-unit", type=int, default=122, help="0 = the on-line tuner's schedule")
    ap.add_argument("--data", default="new,developed")
    ap.add_argument("--forms", default="ds,dx")
    ap.add_argument("--fresh-steps", type=int, default=0,
                    help="Species::new only: every window is this many steps of a FRESH Species after 100 warm-up steps (small grids "
                         "develop their pattern within one long window); median of --rounds windows per form")
    a = ap.parse_args()
    rows, cols = a.rows, a.cols
    cells = rows * cols
    start = bench.developed_start(rows, cols) if "developed" in a.data else None
    for c in range(a.contexts):
        sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
        ctx = sim.context
        if a.rows_per_unit > 0:
            ctx.set_tuned(rows, cols, a.rows_per_unit, 4, 2, 1)     # no on-line tuning: the schedule is pinned
        species = {}
        if "new" in a.data:
            species["new"] = sim.make_species([rows, cols], place_candidates=a.place)
        if "developed" in a.data:
            species["developed"] = bench.upload_species(sim, *start)
            sim.perform_steps(species["developed"], 2000)
        rpu, k, cpl = a.rows_per_unit, 4, 2
        if a.fresh_steps > 0:
            import statistics
            if rpu == 0:
                scratch = sim.make_species([rows, cols])
                for _ in range(12):
                    sim.perform_steps(scratch, 400)
                    rpu, k, cpl, _ = ctx.get_tuned(rows, cols)
                    if rpu > 0:
                        break
                del scratch
            rates = {f: [] for f in a.forms.split(",")}
            for rnd in range(a.rounds):
                for form in rates:
                    ctx.set_tuned(rows, cols, rpu, k, cpl, FORMS[form])
                    sp = sim.make_species([rows, cols], place_candidates=a.place)
                    sim.perform_steps(sp, 100)
                    ctx.timer_start()
                    sim.prepare_steps(sp, a.fresh_steps)
                    ms = ctx.timer_stop()
                    rates[form].append(cells * a.fresh_steps / (ms * 1e-3) / 1e6)
                    label = ctx.info()[0]
                    del sp
            for form, r in rates.items():
                print(json.dumps({"data": "new, fresh Species per window", "context": c, "form": form, "steps": a.fresh_steps, "windows": len(r),
                                  "schedule": [rpu, k, cpl], "Mcells_steps_per_s_median": statistics.median(r), "min": min(r), "max": max(r)}), flush=True)
            ctx.close()
            continue
        if a.rows_per_unit == 0:                                    # ... or what the tuner chooses on the first input
            first = next(iter(species.values()))
            for _ in range(12):
                sim.perform_steps(first, 400)
                rpu, k, cpl, _ = ctx.get_tuned(rows, cols)
                if rpu > 0:
                    break
        for rnd in range(a.rounds):
            for data, sp in species.items():
                for form in a.forms.split(","):
                    ctx.set_tuned(rows, cols, rpu, k, cpl, FORMS[form])
                    sim.perform_steps(sp, 200)
                    done = [0, 0.0]

                    def work():
                        t0 = time.time()
                        ctx.timer_start()
                        while time.time() - t0 < a.seconds:
                            sim.prepare_steps(sp, 200)
                            ctx.sync()
                            done[0] += 200
                        done[1] = ctx.timer_stop() * 1e-3
                        return (done[1],)

                    smp = bench.sample_clock_and_power(work, 0, 0.0) or {}
                    rate = cells * done[0] / done[1] / 1e6
                    watts = smp.get("energy_W") or smp.get("power_W")
                    print(json.dumps({"data": data, "context": c, "round": rnd, "form": form, "kernel": ctx.info()[0],
                                      "placement_ms": getattr(sp, "placement", None),
                                      "Mcells_steps_per_s": rate, "sclk_MHz": smp.get("sclk_MHz"), "power_W": smp.get("power_W"),
                                      "energy_W": smp.get("energy_W"),
                                      "pJ_per_cell_step": watts / (rate * 1e6) * 1e12 if watts else None}), flush=True)
        species.clear()
        ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
