#!/usr/bin/env python3
"""Placement by measurement (gs_fields_place, round-6 form: pair probes, the library's default for large Species) against
planes as hipMalloc hands them out, per context of one process: what the HBM-bound single-step kernel and the marching
kernel of gs_run read on each, how many blocks were drawn and how many probes it took.  GS_HIP_TRACE_TUNER=1 prints every
probe.  (Which physical cause the probes see: tools/ubench/hbm_kinds.hip, profiles/r06_placement.md.)

    python tools/placement_probe.py [--rows 16384 --cols 16384 --contexts 3 --candidates 12 --hold 1]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402


def rate(sim, sp, steps, cells, repeats=3):
    """Median Mcells x steps / s of `steps` steps (HIP events on the library's stream)."""
    out = []
    for _ in range(repeats):
        sim.context.timer_start()
        sim.prepare_steps(sp, steps)
        out.append(cells * steps / (sim.context.timer_stop() * 1e-3) / 1e6)
    sim.context.sync()
    return sorted(out)[len(out) // 2]


def destroy(sp):
    for conc in sp.u._pair + sp.v._pair:
        conc.destroy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--contexts", type=int, default=3)
    ap.add_argument("--candidates", type=int, default=12)
    ap.add_argument("--hold", type=int, default=1, help="1 = a Species of another context is allocated first (as in bench.py)")
    ap.add_argument("--march", type=int, default=1, help="1 = also the marching kernel of gs_run on both sets of planes")
    a = ap.parse_args()
    cells = a.rows * a.cols
    frac = lambda r: r * 1e6 * 16 / 8e12     # noqa: E731  (Mcells x steps / s of a single-step kernel -> fraction of 8 TB/s)
    hold = None
    if a.hold:
        other = Simulation.new(Parameters(), HipArgs(devices=[0]))
        hold = (other, other.make_species([a.rows, a.cols], place_candidates=0))
    print(f"grid {a.rows} x {a.cols}; per context: planes as hipMalloc hands them out | placed by measurement (at most {a.candidates} extra blocks)")
    for c in range(a.contexts):
        sim_s = Simulation.new(Parameters(), HipArgs(devices=[0], kernel=capi.GS_KERNEL_STREAM))
        sim_m = Simulation.new(Parameters(), HipArgs(devices=[0])) if a.march else None
        row = {}
        for how, cand in (("unplaced", 0), ("placed", a.candidates)):
            t0 = time.perf_counter()
            sp = sim_s.make_species([a.rows, a.cols], place_candidates=cand)
            t_make = time.perf_counter() - t0
            sim_s.perform_steps(sp, 50)
            row[how] = {"single_step": rate(sim_s, sp, 100, cells), "placement": sp.placement, "make_s": t_make,
                        "stats": sim_s.context.place_stats()}
            if sim_m is not None:
                # the SAME planes under the marching kernel: a Species of the other context cannot share them, so the
                # marching context places (or not) a Species of its own right after -- it draws from the same allocator state
                destroy(sp)
                spm = sim_m.make_species([a.rows, a.cols], place_candidates=cand)
                sim_m.perform_steps(spm, 2400)               # on-line tuning done
                row[how]["march"] = rate(sim_m, spm, 1000, cells)
                row[how]["march_placement"] = spm.placement
                destroy(spm)
            else:
                destroy(sp)
        u, p = row["unplaced"], row["placed"]
        probes, drawn = p["stats"]
        print(f"context {c}: single step {u['single_step']:,.0f} k = {frac(u['single_step']):.3f} of 8 TB/s | "
              f"{p['single_step']:,.0f} k = {frac(p['single_step']):.3f}  (probe pass {p['placement'][0]:.3f} -> {p['placement'][1]:.3f} ms, "
              f"{drawn} blocks drawn and {probes} probes so far on this context, make_species {p['make_s']:.2f} s against {u['make_s']:.2f} s)"
              + (f"; marching {u['march']:,.0f} k | {p['march']:,.0f} k (probe pass {p['march_placement'][0]:.3f} -> {p['march_placement'][1]:.3f} ms)"
                 if sim_m is not None else ""), flush=True)
        sim_s.context.close()
        if sim_m is not None:
            sim_m.context.close()
    del hold
    return 0


if __name__ == "__main__":
    sys.exit(main())
