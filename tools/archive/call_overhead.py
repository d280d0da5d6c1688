#!/usr/bin/env python3
"""Host-side cost of the reference's benchmark iteration on small grids: `perform_steps(1)` = gs_run + gs_sync
(compute/shared/src/benchmark.rs:77-83 calls it once per criterion iteration, starting at 8 x 16 cells).

    python tools/archive/call_overhead.py

Per grid: microseconds per call of (a) gs_run + gs_sync, (b) gs_run alone, enqueued 200 deep and synchronised once,
(c) gs_sync on an idle context, (d) the ctypes call overhead itself (gs_abi_version).
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402


def per_call(fn, n=2000):
    for _ in range(200):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    lib = capi.load()
    print("| grid | steps | run + sync | run alone (200 deep) | sync, idle | ctypes call |")
    print("|---|---|---|---|---|---|")
    for rows, cols in ((8, 16), (32, 64), (64, 128), (256, 512), (1024, 2048)):
        sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
        sp = sim.make_species([rows, cols])
        ctx = sim.context
        sim.perform_steps(sp, 64)
        in_u, in_v, out_u, out_v = sp.in_out()
        h = (ctx.handle, in_u.handle, in_v.handle, out_u.handle, out_v.handle)
        slot = ctypes.c_int32(0)
        for steps in (2,):      # an even count keeps the result in the input slot: no handle swap in the loop
            def run():
                lib.gs_run(*h, steps, ctypes.byref(slot))

            def run_sync():
                lib.gs_run(*h, steps, ctypes.byref(slot))
                lib.gs_sync(h[0])

            a = per_call(run_sync)
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                for _ in range(200):
                    run()
                lib.gs_sync(h[0])
            b = (time.perf_counter() - t0) / (200 * reps) * 1e6
            c = per_call(lambda: lib.gs_sync(h[0]))
            d = per_call(lib.gs_abi_version)
            print(f"| {rows} x {cols} | {steps} | {a:.1f} | {b:.1f} | {c:.2f} | {d:.2f} |", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
