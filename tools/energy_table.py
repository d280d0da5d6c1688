#!/usr/bin/env python3
"""What a joule buys: rate, sustained clock, socket power and energy per cell-step of the production kernel's
flavours, on the reference's benchmark input (Species::new: > 99 % of the cells at the fixed point) and on a
developed spot pattern -- one process, the flavours interleaved, every window >= `--seconds` of back-to-back
launches with rocm-smi sampled beside it (clock, power, the accumulated-energy counter).

    python tools/energy_table.py [--rows 16384 --cols 16384] [--seconds 4] [--rounds 2] [--data new,developed]
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \
        -- python3 tools/energy_table.py --profile --data new        (short: a few launches per flavour)

Flavours (all on the same pinned schedule: 4 steps per pass, 2 columns per lane, `--rows-per-unit` rows):
  strict.op.dx  the kernel for the default parameters with full difference sharing, across lanes too (share_taps = 3)
  strict.op.ds  ... with full difference sharing within a lane (share_taps = 1)
  strict.op   the kernel for the default parameters (side taps `v_sub_f32 ... div:2`, no `* dt`), share_taps = 2
  strict      the same build without the parameter specialisations (general_kernels = 1): sub, mul, add taps
  fused       GS_MATH_FUSED: sub + fma taps, denormals kept (not bit-exact below 1e-37)
Throughput unit: compute/shared/src/benchmark.rs:55-60.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402

FLAVOURS = (("strict.op.dx", {"share_taps": 3}), ("strict.op.ds", {"share_taps": 1}), ("strict.op", {"share_taps": 2}), ("strict", {"general_kernels": 1}),
            ("fused", {"math": capi.GS_MATH_FUSED}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--rows-per-unit", type=int, default=122)
    ap.add_argument("--data", default="new,developed")
    ap.add_argument("--flavours", default="strict.op.dx,strict.op.ds,strict.op,strict,fused", help="comma- or plus-separated")
    ap.add_argument("--profile", action="store_true", help="under rocprofv3: 40 steps per flavour, no sampling")
    ap.add_argument("--heights", default="", help="comma-separated unit heights: the FIRST flavour of --flavours at each of them instead of the "
                                                  "flavours side by side (rows recomputed at unit seams against launch rounds and tail)")
    a = ap.parse_args()
    rows, cols = a.rows, a.cols
    cells = rows * cols
    datas = a.data.split(",")
    sims = {}
    wanted = a.flavours.replace("+", ",").split(",")
    if a.heights:
        name, kw = next((n, k) for n, k in FLAVOURS if n == wanted[0])
        for h in (int(x) for x in a.heights.split(",")):
            sims[f"{name}@{h}"] = [Simulation.new(Parameters(), HipArgs(devices=[0], rows_per_block=h, fuse_steps=4, cols_per_lane=2, **kw)), {}]
    for name, kw in FLAVOURS:
        if name not in wanted or a.heights:
            continue
        sim = Simulation.new(Parameters(), HipArgs(devices=[0], rows_per_block=a.rows_per_unit, fuse_steps=4,
                                                   cols_per_lane=2, **kw))
        sims[name] = [sim, {}]
    start = bench.developed_start(rows, cols) if "developed" in datas else None
    place = None                     # the library default: Species of >= 2^26 cells are placed by measurement
    for name, (sim, species) in sims.items():
        if "new" in datas:
            # (planes placed by measurement: unplaced, the flavours would be compared on different draws of blocks)
            species["new"] = sim.make_species([rows, cols], place_candidates=place)
        if "developed" in datas:
            species["developed"] = bench.upload_species(sim, *start, place)
            sim.perform_steps(species["developed"], 4000)
    del start
    if a.profile:
        for name, (sim, species) in sims.items():
            for d, sp in species.items():
                sim.perform_steps(sp, 40)
                print(json.dumps({"flavour": name, "data": d, "kernel": sim.context.info()[0], "steps": 40}), flush=True)
        return 0
    results = []
    for rnd in range(a.rounds):
        for d in datas:
            for name, (sim, species) in sims.items():
                sp, ctx = species[d], sim.context
                sim.perform_steps(sp, 400)                                     # warm: clocks, caches
                t0 = time.perf_counter()
                sim.perform_steps(sp, 400)
                rate = 400 / (time.perf_counter() - t0)                        # steps per second
                n = max(400, int(a.seconds * rate) // 4 * 4)
                box = {}

                def work():
                    ctx.sync()
                    t0 = time.perf_counter()
                    ctx.timer_start()
                    sim.prepare_steps(sp, n)
                    box["ms"] = ctx.timer_stop()
                    ctx.sync()
                    return (time.perf_counter() - t0,)

                s = bench.sample_clock_and_power(work, 0, float(cells) * n) or {}
                row = {"round": rnd, "data": d, "flavour": name, "kernel": ctx.info()[0], "steps": n,
                       "Mcells_steps_per_s": cells * n / (box["ms"] * 1e-3) / 1e6, "launch_ms": box["ms"] / (n / 4),
                       "sclk_MHz": s.get("sclk_MHz"), "power_W": s.get("power_W"), "power_cap_W": s.get("power_cap_W"),
                       "energy_W": s.get("energy_W"), "pJ_per_cell_step": s.get("energy_pJ_per_cell_step"),
                       "samples": s.get("samples")}
                if row["pJ_per_cell_step"] is None and row["power_W"]:
                    row["pJ_per_cell_step_from_power_samples"] = row["power_W"] / (row["Mcells_steps_per_s"] * 1e6) * 1e12
                results.append(row)
                print(json.dumps(row), flush=True)
    print("\n| data | flavour | kernel | Mcells×steps/s | sclk MHz | power W (samples) | power W (energy counter) | pJ per cell-step |")
    print("|---|---|---|---|---|---|---|---|")
    for d in datas:
        for name in sims:
            rs = [r for r in results if r["data"] == d and r["flavour"] == name]

            def med(key):
                v = sorted(r[key] for r in rs if r.get(key) is not None)
                return v[len(v) // 2] if v else None

            pj = med("pJ_per_cell_step") or med("pJ_per_cell_step_from_power_samples")
            fmt = lambda x, f: ("%" + f) % x if x is not None else "–"     # noqa: E731
            print(f"| {d} | {name} | `{rs[0]['kernel']}` | {fmt(med('Mcells_steps_per_s'), '.0f')} | {fmt(med('sclk_MHz'), '.0f')} | "
                  f"{fmt(med('power_W'), '.0f')} | {fmt(med('energy_W'), '.0f')} | {fmt(pj, '.0f')} |")
    return 0


if __name__ == "__main__":
    sys.exit(main())
