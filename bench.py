#!/usr/bin/env python3
"""Benchmark of the Gray-Scott step path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one simulation time step of the whole grid (the reference's own throughput
unit is cells x steps, compute/shared/src/benchmark.rs:55-59; its "compute" workload is
perform_steps only, :77-83).  Inputs are resident in HBM before the timed region starts.

N = 1 : BASELINE.json's headline workload, 16384 x 16384 f32 (config 3), Species::new init,
        default feed/kill.  Rank 0 also times the CPU port of the reference's
        parallel(block(autovec)) backend on the host cores on a bounded sample.
N > 1 : launched by torchrun, one process per GPU; weak scaling with 2^28 cells per GPU:
        rows = 16384 * N over 16384 columns (N = 8: 65536 x 32768, BASELINE config 5), row
        slabs with ghost-row exchange through RCCL send/recv inside libgs_hip.so.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_CELL_STEP = 16          # read U,V + write U,V, 4 B each
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)
HBM_COPY_CEILING_GBS = 6290.0     # measured float4-copy ceiling, same table


def grid_for(n_gpus: int):
    if n_gpus == 8:
        return 65536, 32768       # BASELINE config 5
    return 16384 * n_gpus, 16384  # config 3 (N=1), config 4 shape (N=2), same cells per GPU


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU
    box exposes 256 logical CPUs but grants a 16-CPU share per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                if text[0] != "max":
                    n = min(n, max(1, int(int(text[0]) / int(text[1]))))
            else:
                quota = int(text[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(target_seconds: float = 15.0):
    """Times the CPU port (oracle/gs_cpu_parallel.c) of the reference's parallel backend on a
    bounded sample of the SAME workload: 16384 x 16384, Species::new init, a few steps."""
    from oracle import cpu_parallel

    rows, cols = 16384, 16384
    threads = usable_cpus()
    sim = cpu_parallel.ParallelSimulation(rows, cols, num_threads=threads, ftz=True)
    sim.perform_steps(1)                                  # touch pages / warm the thread team
    t0 = time.perf_counter()
    sim.perform_steps(1)
    one = time.perf_counter() - t0
    n = max(2, min(200, int(target_seconds / max(one, 1e-3))))
    t0 = time.perf_counter()
    sim.perform_steps(n)
    dt = time.perf_counter() - t0
    info = {
        "value": rows * cols * n / dt / 1e6,
        "unit": "Mcells×steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{rows}x{cols} f32, Species::new init, {n} steps of the parallel(block(autovec)) "
                  f"port (oracle/gs_cpu_parallel.c), SIMD width {cpu_parallel.simd_width()}, FTZ on, "
                  f"L1/L2 block {sim.l1_block_size}/{sim.l2_block_size} B, {dt:.1f} s",
    }
    sim.close()
    return info


def measured_traffic(kernel_name: str):
    """HBM bytes per launch from rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/summarize_profile.py from separate FETCH_SIZE / WRITE_SIZE runs); None if absent."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            data = json.load(f)
        return data.get(kernel_name.split("@")[0])   # None for a configuration that was not profiled
    except (OSError, ValueError):
        return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--rows", type=int, default=0, help="override the grid (diagnostics only)")
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational fused-flavour leg")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N > 1 on a 1-GPU box: all ranks share GPU 0 and torch.distributed uses gloo; needs "
                         "GS_RCCL_LIBRARY to name a transport that accepts several ranks per device "
                         "(tests/cpp/shm_transport.cpp).  Checks the code path, the numbers mean nothing")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from grayscott_amd import HipArgs, Parameters, Simulation, capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus N > 1 must be launched with torch.distributed.run "
                  "(one process per GPU)", file=sys.stderr)
            return 2
        args.gpus = world
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    if args.rehearsal:
        if world > 1 and not os.environ.get("GS_RCCL_LIBRARY"):
            print("bench.py: --rehearsal needs GS_RCCL_LIBRARY", file=sys.stderr)
            return 2
        local_rank = 0
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if args.rehearsal else "cuda"   # where the bootstrap / reduction tensors live

    unique_id = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        buf = torch.zeros(capi.GS_UNIQUE_ID_BYTES, dtype=torch.uint8, device=red_dev)
        if rank == 0:
            buf.copy_(torch.frombuffer(bytearray(capi.get_unique_id()), dtype=torch.uint8))
        dist.broadcast(buf, src=0)
        unique_id = bytes(buf.cpu().numpy().tobytes())

    rows, cols = grid_for(args.gpus)
    if args.rows and args.cols:
        rows, cols = args.rows, args.cols
    # The library's defaults, nothing pinned: one kernel launch per pass, so "launch" in the roofline
    # object is unambiguous and comparable with rocprofv3's per-kernel average.
    hip_args = HipArgs(devices=[local_rank], rank=rank, world=world, unique_id=unique_id)
    sim = Simulation.new(Parameters(), hip_args)
    ctx = sim.context
    if world == 1:
        # gs_run chooses unit height / fused steps / columns per lane on line, from timed passes of
        # the simulation itself (per context and shape).  Let it finish on a scratch set of planes,
        # so that neither the W warm-up steps nor the K timed ones contain tuning passes whatever
        # W and K are.  (Multi-process contexts do not tune.)
        scratch = sim.make_species([rows, cols])
        sim.perform_steps(scratch, 400)
        ctx.sync()
        del scratch
    species = sim.make_species([rows, cols])        # Species::new on the device, HBM-resident

    def barrier():
        if world > 1:
            dist.barrier()

    sim.perform_steps(species, args.warmup)
    ctx.sync()
    barrier()
    torch.cuda.synchronize()
    _, launches0 = ctx.info()
    t0 = time.perf_counter()
    ctx.timer_start()                               # HIP events on the library's own stream
    sim.perform_steps(species, args.steps)
    event_ms = ctx.timer_stop()
    _, launches1 = ctx.info()
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([wall, event_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, event_ms = float(t[0]), float(t[1])

    kernel_name, _ = ctx.info()
    cells = rows * cols
    value = cells * args.steps / wall / 1e6
    # roofline of the dominant kernel: algorithmic bytes per launch / average launch duration.
    # One pass = one launch of the step kernel over one GPU's slab (plus, on a slab chain, the
    # small boundary-band launch); it advances `steps / passes` time steps (temporal blocking).
    launches = launches1 - launches0
    passes = launches if args.gpus == 1 else launches // 2
    extra = None
    if args.gpus == 1 and not args.no_extra:
        # informational: the fused-tap flavour (GS_MATH_FUSED: bit-identical wherever no sub-normal
        # intermediate occurs, |diff| <= 1e-37 elsewhere -- inside north_star's 1e-5 tolerance)
        sim_c = Simulation.new(Parameters(), HipArgs(devices=[local_rank], math=capi.GS_MATH_FUSED))
        species_c = sim_c.make_species([rows, cols])
        sim_c.perform_steps(species_c, max(args.warmup, 400))
        sim_c.context.sync()
        tc = time.perf_counter()
        sim_c.perform_steps(species_c, args.steps)
        sim_c.context.sync()
        tc = time.perf_counter() - tc
        extra = {"kernel": sim_c.context.info()[0], "value": rows * cols * args.steps / tc / 1e6}
        sim_c.context.close()
        del species_c, sim_c
    launch_ms = event_ms / passes
    per_launch_bytes = BYTES_PER_CELL_STEP * (cells / args.gpus) * args.steps / passes
    achieved = per_launch_bytes / (launch_ms * 1e-3) / 1e9
    result = {
        # BASELINE.json's metric, verbatim; `value` is its first quantity, the `roofline` object
        # carries the second (achieved GB/s and fraction of the HBM roof)
        "metric": "Mcells×steps/s and achieved HBM GB/s (% of roofline), 16384² f32 grid",
        "value": value,
        "unit": "Mcells×steps/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not args.rehearsal else "synthetic (REHEARSAL: ranks share one GPU, not a measurement)",
        "config": {
            "workload": f"{rows}x{cols} f32 (rows x cols), Species::new init, default feed/kill, "
                        f"double-buffered U/V in HBM",
            "cells_per_gpu": cells // args.gpus,
            "kernel": kernel_name,
            "launches_per_pass": 1 if args.gpus == 1 else 2,
            "partition": "single GPU" if args.gpus == 1 else
                         f"{args.gpus} row slabs, RCCL send/recv ghost rows",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
            "launch_ms": launch_ms,
            "launches": passes,
            "steps_per_launch": args.steps / passes,
            "algorithmic_bytes_per_launch": per_launch_bytes,
            # PMC figure of the committed profile of this kernel on this per-GPU grid (null otherwise)
            "traffic": measured_traffic(kernel_name) if cells // args.gpus == 16384 * 16384 else None,
        },
    }
    if extra is not None:
        result["fused_flavour"] = extra
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
