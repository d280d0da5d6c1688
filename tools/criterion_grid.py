#!/usr/bin/env python3
"""The reference's own benchmark grid (compute/shared/src/benchmark.rs:28-83) on the HIP backend:
shapes [2^k, 2*2^k] for k = 3..11, the "compute" workload (perform_steps only, then a sync so
that the work is actually done), throughput in cells x steps per second as criterion reports it
(Throughput::Elements(rows * cols * steps), :55-59).  Steps: a subset of the reference's 1..256.

    python tools/criterion_grid.py [--cpu]     # --cpu adds the parallel(block(autovec)) port
    python tools/criterion_grid.py --full      # the reference's two "full" workloads instead (see below)

--full: benchmark.rs:86-93 `full_sync_workload` = perform_steps + make_result_view -- what the unmodified
`simulate` binary pays per image through the shim (a gs_sync, then a blocking download of V into a fresh
array) -- and :97-113 `full_gpu_future_workload` = prepare_steps + make_scalar_view_after as ONE transaction --
here prepare_steps + write_result_view_after into a page-locked image + download_wait (the form
grayscott_amd/simulate.py uses).  Throughput is still cells x steps per second, so the columns compare
directly with the "compute" table.

Prints a markdown table: like criterion, every benchmark first iterates for a warm-up time (0.5 s
here, 3 s in criterion's default) -- the library finishes its on-line tuning for the shape in that
time -- and then reports the median of 15 timed iterations.
"""
import argparse
import os

import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402


def full_workloads(a, steps_list):
    from grayscott_amd.simulation import pinned_empty

    print("| rows x cols | " + " | ".join(f"compute {s}" for s in steps_list) + " | " +
          " | ".join(f"full sync {s}" for s in steps_list) + " | " +
          " | ".join(f"full future {s}" for s in steps_list) + " | download alone, us | kernel |")
    print("|---|" + "---|" * (3 * len(steps_list) + 2))
    sim = Simulation.new(Parameters(), HipArgs(devices=[0], kernel=a.kernel))
    ctx = sim.context

    def timed(fn, iters=15, warm=0.3):
        t_end = time.perf_counter() + warm
        while time.perf_counter() < t_end:
            fn()
        times = []
        for _ in range(iters):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return statistics.median(times)

    for k in range(a.kmin, a.kmax + 1):
        size = 2 ** k
        shape = (size, 2 * size)
        species = sim.make_species(shape)
        cells = shape[0] * shape[1]
        image = pinned_empty(shape)

        def compute(steps):
            sim.perform_steps(species, steps)

        def full_sync(steps):
            sim.perform_steps(species, steps)
            species.make_result_view()            # fresh array + blocking download (concentration/mod.rs:261-275)

        def full_future(steps):
            sim.prepare_steps(species, steps)
            species.write_result_view_after(image)
            ctx.download_wait()

        for steps in steps_list:                  # let the on-line tuning finish on this shape first
            timed(lambda: compute(steps), iters=1, warm=0.4)
        row = [f"{shape[0]} x {shape[1]}"]
        for fn in (compute, full_sync, full_future):
            for steps in steps_list:
                row.append(f"{cells * steps / timed(lambda: fn(steps)) / 1e6:.1f}")
        row.append(f"{timed(lambda: species.make_result_view()) * 1e6:.0f}")
        row.append(ctx.info()[0])
        print("| " + " | ".join(row) + " |", flush=True)
    print("\n(Mcells x steps / s, medians of 15 iterations after a warm-up; `full sync` = perform_steps + make_result_view, "
          "`full future` = prepare_steps + write_result_view_after + download_wait into a page-locked image)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--graph", action="store_true", help="gs_options.use_graph = 1")
    ap.add_argument("--kernel", type=int, default=0, help="gs_kernel to force (0 = auto, 3 = temporal blocking, 5 = LDS tiles)")
    ap.add_argument("--full", action="store_true", help="the reference's full workloads (perform_steps + result view)")
    ap.add_argument("--kmin", type=int, default=3)
    ap.add_argument("--kmax", type=int, default=11)
    a = ap.parse_args()
    steps_list = [1, 16, 256]
    if a.full:
        return full_workloads(a, steps_list)
    print("| rows x cols | " + " | ".join(f"HIP {s} steps" for s in steps_list) +
          (" | CPU port 16 steps |" if a.cpu else " |"))
    print("|---|" + "---|" * (len(steps_list) + (1 if a.cpu else 0)))
    sim = Simulation.new(Parameters(), HipArgs(devices=[0], use_graph=1 if a.graph else 0, kernel=a.kernel))
    for k in range(a.kmin, a.kmax + 1):
        size = 2 ** k
        shape = (size, 2 * size)
        species = sim.make_species(shape)
        cells = shape[0] * shape[1]
        row = [f"{shape[0]} x {shape[1]}"]
        for steps in steps_list:
            times = []
            t_end = time.perf_counter() + 0.5
            while time.perf_counter() < t_end:
                sim.perform_steps(species, steps)
                sim.context.sync()
            for it in range(15):
                t0 = time.perf_counter()
                sim.perform_steps(species, steps)
                sim.context.sync()
                times.append(time.perf_counter() - t0)
            row.append(f"{cells * steps / statistics.median(times) / 1e6:.1f}")
        if a.cpu:
            from oracle import cpu_parallel

            try:
                threads = len(os.sched_getaffinity(0))
                quota = open("/sys/fs/cgroup/cpu.max").read().split()
                if quota[0] != "max":
                    threads = min(threads, max(1, int(int(quota[0]) / int(quota[1]))))
            except (OSError, ValueError, AttributeError):
                threads = os.cpu_count() or 1
            cpu = cpu_parallel.ParallelSimulation(shape[0], shape[1], num_threads=threads)
            cpu.perform_steps(16)
            times = []
            for _ in range(5):
                t0 = time.perf_counter()
                cpu.perform_steps(16)
                times.append(time.perf_counter() - t0)
            row.append(f"{cells * 16 / statistics.median(times) / 1e6:.1f}")
            cpu.close()
        row.append(sim.context.info()[0])
        print("| " + " | ".join(row) + " |", flush=True)
    print("\n(Mcells x steps / s; HIP timings include the host-side enqueue and one sync per iteration, "
          "as criterion's `b.iter(|| workload(...))` would)")


if __name__ == "__main__":
    main()
