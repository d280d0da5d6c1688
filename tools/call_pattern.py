#!/usr/bin/env python3
"""The reference's real call pattern at its default size: `simulate` calls perform_steps(32) per image and copies the V
plane out after each (simulate/src/main.rs:52, 113-115; grid: ui/src/lib.rs:32-37).  Which kernel should `kernel = auto`
run for calls of that length?  Per call length: Mcells x steps / s of (a) perform_steps alone, back to back, (b)
perform_steps + a blocking V download per call (the reference's default build), (c) prepare_steps + an overlapped
download (its `async-gpu` build), for the marching kernel (GS_KERNEL_TB), the persistent window kernel
(GS_KERNEL_WINDOW) and whatever `auto` picks.

    python tools/call_pattern.py [--rows 1080 --cols 1920 --calls 200]
"""
import argparse
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402
from grayscott_amd.simulation import pinned_empty  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--lengths", default="8,16,32,64,128,1000")
    ap.add_argument("--kinds", default="", help="auto,marching,window (default: all three)")
    a = ap.parse_args()
    cells = a.rows * a.cols
    kinds = (("auto", {}), ("marching", {"kernel": capi.GS_KERNEL_TB}), ("window", {"kernel": capi.GS_KERNEL_WINDOW}))
    only = [k for k in kinds if not a.kinds or k[0] in a.kinds.split(",")]

    # ONE context alive at a time, as in the reference's `simulate`: a process holds few hardware queues, and streams of
    # several contexts that land on one of them wait for each other (three contexts side by side: the same kernel 364 k in
    # one context and 419 k in another, profiles/r06_logs/call_pattern_pack_kernel.log).
    def fresh(kw):
        sim = Simulation.new(Parameters(), HipArgs(devices=[0], **kw))
        sp = sim.make_species([a.rows, a.cols])
        sim.perform_steps(sp, 4000)                       # on-line tuning done
        return sim, sp

    image = np.empty((a.rows, a.cols), np.float32)
    pinned = [pinned_empty((a.rows, a.cols)) for _ in range(3)]
    # the floor a download per call sets: images back to back through the overlapped path, no steps in between
    sim0, sp0 = fresh({"kernel": capi.GS_KERNEL_TB})
    per_image = per_image_2 = float("inf")
    for _ in range(4):                                   # (the best of four: an idle chip's first copies run at idle clocks)
        sim0.perform_steps(sp0, 400)
        t0 = time.perf_counter()
        for i in range(100):
            sp0.write_result_view_after(pinned[i & 1])
            sim0.context.download_wait()
        per_image = min(per_image, (time.perf_counter() - t0) / 100)
        t0 = time.perf_counter()
        for i in range(100):
            sp0.write_result_view_after(pinned[i % 3])
            sim0.context.download_wait(in_flight=1)
        sim0.context.download_wait()
        per_image_2 = min(per_image_2, (time.perf_counter() - t0) / 100)
    sim0.context.close()
    print(f"an image alone (staging copy + {cells * 4 / 1e6:.1f} MB over PCIe into pinned memory + wait), back to back, one at a time: {per_image * 1e6:.0f} us = "
          f"{cells * 4 / per_image / 1e9:.1f} GB/s; two in flight: {per_image_2 * 1e6:.0f} us = {cells * 4 / per_image_2 / 1e9:.1f} GB/s: a call of n steps "
          f"with a download each cannot beat {cells / per_image_2 / 1e6:,.0f} x n Mcells x steps / s (n = 32: {32 * cells / per_image_2 / 1e6:,.0f})")
    print(f"grid {a.rows} x {a.cols}, {a.calls} calls per figure, median of 3; Mcells x steps / s; one context at a time")
    print("| kernel (label) | steps per call | steps only | + blocking V download | + overlapped V download |")
    print("|---|---|---|---|---|")
    for name, kw in only:
        sim, sp = fresh(kw)
        ctx = sim.context
        for n in (int(x) for x in a.lengths.split(",")):
            calls = max(20, min(a.calls, 40000 // n))

            def steps_only():
                for _ in range(calls):
                    sim.perform_steps(sp, n)

            def blocking():
                for _ in range(calls):
                    sim.perform_steps(sp, n)
                    sp.write_result_view(image)

            def overlapped():
                # the driver loop of grayscott_amd/simulate.py: two images in flight
                for i in range(calls):
                    sim.prepare_steps(sp, n)
                    sp.write_result_view_after(pinned[i % 3])
                    if i:
                        ctx.download_wait(in_flight=1)
                ctx.download_wait()
                ctx.sync()

            row = []
            for fn in (steps_only, blocking, overlapped):
                fn()
                rates = []
                for _ in range(3):
                    ctx.sync()
                    t0 = time.perf_counter()
                    fn()
                    ctx.sync()
                    rates.append(cells * n * calls / (time.perf_counter() - t0) / 1e6)
                row.append(statistics.median(rates))
            print(f"| {name} (`{ctx.info()[0]}`) | {n} | {row[0]:,.0f} | {row[1]:,.0f} | {row[2]:,.0f} |", flush=True)
        ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
