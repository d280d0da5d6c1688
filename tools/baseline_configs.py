#!/usr/bin/env python3
"""the single-GPU BASELINE configurations with their prescribed step counts
(1080x1920: 1000 steps, 4096^2: 1000, 16384^2: 10000; 100-step warm-up, median of 5 un-profiled
repeats), next to the CPU side on this box's host cores: the strict naive restatement (1 thread and
all usable cores) and the restated parallel(block(autovec)) backend at 1080x1920, 2048x4096, 4096^2
and 16384^2, with core counts and cache-blocking sizes.  Prints markdown.

    python tools/baseline_configs.py
"""
import os
import statistics
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402
from oracle import cpu_oracle, cpu_parallel  # noqa: E402


def gpu_rows():
    print("| grid (rows x cols) | steps | kernel | median of 5, Mcells×steps/s | min .. max | ms/step |")
    print("|---|---|---|---|---|---|")
    for rows, cols, steps in ((1080, 1920, 1000), (2048, 4096, 1000), (4096, 4096, 1000), (8192, 4096, 1000), (16384, 16384, 10000)):
        sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
        scratch = sim.make_species([rows, cols])
        sim.perform_steps(scratch, 4000 if rows < 16384 else 400)       # finish the on-line tuning
        sim.context.sync()
        del scratch
        rates = []
        for _ in range(5):
            # (large Species: planes placed by measurement, as bench.py does -- profiles/r05_cross_lane.md, section 4)
            sp = sim.make_species([rows, cols])     # (the library default: Species of >= 2^26 cells are placed by measurement)
            sim.perform_steps(sp, 100)
            sim.context.sync()
            t0 = time.perf_counter()
            sim.perform_steps(sp, steps)
            sim.context.sync()
            rates.append(rows * cols * steps / (time.perf_counter() - t0) / 1e6)
            del sp
        med = statistics.median(rates)
        print(f"| {rows} x {cols} | {steps} | `{sim.context.info()[0]}` | **{med:,.0f}** | {min(rates):,.0f} .. {max(rates):,.0f} | "
              f"{rows * cols / med / 1e3:.4f} |", flush=True)
        sim.context.close()


def cpu_rows():
    threads = cpu_oracle.usable_cpus()
    l1, l2 = cpu_parallel.cache_sizes_per_thread()
    lscpu = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
    model = next((line.split(":", 1)[1].strip() for line in lscpu.splitlines() if line.startswith("Model name")), "?")
    print(f"\nHost: {model}; {os.cpu_count()} logical CPUs visible, {threads} usable (cgroup quota); SIMD width "
          f"{cpu_parallel.simd_width()} lanes; per-thread L1d / L2 = {l1} / {l2} B, so the port blocks at L1/2 = {l1 // 2} B "
          f"and L2/2 = {l2 // 2} B (compute/block/src/default.rs:30-48).\n")
    print("| CPU backend | grid | threads | steps timed | Mcells×steps/s |")
    print("|---|---|---|---|---|")
    for rows, cols in ((1080, 1920), (2048, 4096), (4096, 4096), (16384, 16384)):
        u0, v0 = oracle.init_species(rows, cols)
        for nthreads in ((1, threads) if rows == 1080 else (threads,)):
            steps = max(2, (4 if nthreads == 1 else 48) * 1080 * 1920 // (rows * cols))
            oracle.run(u0, v0, 1, nthreads=nthreads)
            t0 = time.perf_counter()
            oracle.run(u0, v0, steps, nthreads=nthreads)
            dt = time.perf_counter() - t0
            print(f"| strict naive restatement (`oracle/gs_oracle.c`) | {rows} x {cols} | {nthreads} | {steps} | "
                  f"{rows * cols * steps / dt / 1e6:,.0f} |", flush=True)
        del u0, v0
    for rows, cols, steps in ((1080, 1920, 2000), (2048, 4096, 400), (4096, 4096, 200), (16384, 16384, 40)):
        sim = cpu_parallel.ParallelSimulation(rows, cols, num_threads=threads, ftz=True)
        sim.perform_steps(2)
        t0 = time.perf_counter()
        sim.perform_steps(steps)
        dt = time.perf_counter() - t0
        print(f"| restated `parallel(block(autovec))` (`oracle/gs_cpu_parallel.c`) | {rows} x {cols} | {threads} | {steps} | "
              f"{rows * cols * steps / dt / 1e6:,.0f} |", flush=True)
        sim.close()


if __name__ == "__main__":
    gpu_rows()
    cpu_rows()
