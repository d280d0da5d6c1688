#!/usr/bin/env python3
"""Where four planes land decides the HBM-bound single-step kernel's rate (profiles/r04_sweeps.md, section 8).  Is the
level a property of the PROCESS (every set of blocks reads alike) or of the ALLOCATION (some sets of blocks read better)?
Several contexts in one process, each drawing 4 + 12 blocks of 1 GiB and timing 48 four-subsets of them
(gs_fields_place with GS_HIP_TRACE_TUNER=1 prints every probe); then the distribution per context.

    GS_HIP_TRACE_TUNER=1 python tools/placement_probe.py [--rows 16384 --cols 16384 --contexts 3 --candidates 12]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GS_HIP_TRACE_TUNER", "1")
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--contexts", type=int, default=3)
    ap.add_argument("--candidates", type=int, default=12)
    ap.add_argument("--generations", type=int, default=1, help="species placed (and freed) per context before the one that is kept")
    ap.add_argument("--hold", type=int, default=0, help="1 = a Species of another context is allocated first (as in bench.py)")
    a = ap.parse_args()
    cells = a.rows * a.cols
    hold = None
    if a.hold:   # as in bench.py: another context's Species (4 planes) is allocated before the placed one
        other = Simulation.new(Parameters(), HipArgs(devices=[0]))
        hold = (other, other.make_species([a.rows, a.cols]))
    for c in range(a.contexts):
        sim = Simulation.new(Parameters(), HipArgs(devices=[0], kernel=capi.GS_KERNEL_STREAM))
        for g in range(a.generations - 1):    # throw-away generations: place, step a little, free
            tmp = sim.make_species([a.rows, a.cols], place_candidates=a.candidates)
            print(f"context {c} generation {g}: first four {tmp.placement[0]:.4f} ms, chosen {tmp.placement[1]:.4f} ms", flush=True)
            sim.perform_steps(tmp, 50)
            for conc in tmp.u._pair + tmp.v._pair:
                conc.destroy()
            del tmp
        sp = sim.make_species([a.rows, a.cols], place_candidates=a.candidates)
        first, best = sp.placement
        sim.perform_steps(sp, 50)
        sim.context.timer_start()
        sim.prepare_steps(sp, 200)
        ms = sim.context.timer_stop() / 200
        f = lambda t: 16 * cells / (t * 1e-3) / 8e12     # noqa: E731
        print(f"context {c}: first four blocks {first:.4f} ms = {f(first):.3f} of 8 TB/s, chosen {best:.4f} ms = {f(best):.3f}, "
              f"200 steps on the chosen planes {ms:.4f} ms = {f(ms):.3f}", flush=True)
        keep = sp                                            # keep this context's planes allocated: the next draws elsewhere
        del keep
        sim.context.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
