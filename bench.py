#!/usr/bin/env python3
"""Benchmark of the Gray-Scott step path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--grid RxC]

One "step" = one simulation time step of the whole grid (the reference's own throughput
unit is cells x steps, compute/shared/src/benchmark.rs:55-59; its "compute" workload is
perform_steps only, :77-83).  Inputs are resident in HBM before the timed region starts.

N = 1 : BASELINE.json's headline workload, 16384 x 16384 f32 (config 3), Species::new init,
        default feed/kill.  Rank 0 also times the same kernel on a developed spot pattern, the
        fused-tap flavour, and the CPU ports on the host cores on a bounded sample.
N > 1 : launched by torchrun, one process per GPU, row slabs with ghost-row exchange through RCCL
        send/recv inside libgs_hip.so.
        --scaling weak (default): 2^28 cells per GPU -- rows = 16384 * N over 16384 columns
            (N = 2: 32768 x 16384, BASELINE config 4; N = 8: 65536 x 32768, config 5);
        --scaling strong: one grid for every N -- 32768 x 16384 (config 4 "across 2 then 4") for
            N <= 4, 65536 x 32768 (config 5) for N = 8; --grid overrides it.

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
# VALU issue roof for plain f32 ops: 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz (half the
# 157.3 TFLOP/s FMA peak of the same table: the strict kernel issues no FMA)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
# arithmetic the reference's update needs per cell-step when each op is one instruction (taps:
# 4 corners x (sub, mul, add) + 4 sides x (sub with div:2, add), two species; reaction: 13)
USEFUL_VALU_PER_CELL_STEP = 53


def grid_for(n_gpus: int, scaling: str):
    if scaling == "strong":
        return (65536, 32768) if n_gpus > 4 else (32768, 16384)   # BASELINE configs 5 / 4
    if n_gpus == 8:
        return 65536, 32768       # BASELINE config 5
    return 16384 * n_gpus, 16384  # config 3 (N=1), config 4 (N=2), same cells per GPU


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


def cpu_baseline(target_seconds: float = 12.0):
    """Times the CPU side on a bounded sample of the SAME workload (16384 x 16384, Species::new
    init, a few steps) on every core this process may use: the port of the reference's
    parallel(block(autovec)) backend (oracle/gs_cpu_parallel.c) -- the reported baseline -- and
    the strict restatement of its naive backend (oracle/gs_oracle.c, OpenMP over rows) beside it."""
    import numpy as np

    import oracle
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
        "logical_cpus": os.cpu_count(),
    }
    sim.close()
    # the strict naive restatement (the parity oracle) on the same cores, a few steps
    u, v = oracle.init_species(rows, cols)
    t0 = time.perf_counter()
    u, v = oracle.run(u, v, 1, ftz=True, nthreads=threads)
    one = time.perf_counter() - t0
    m = max(1, min(20, int(6.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    oracle.run(u, v, m, ftz=True, nthreads=threads)
    dt = time.perf_counter() - t0
    info["naive"] = {"value": rows * cols * m / dt / 1e6, "unit": "Mcells×steps/s", "cores": threads,
                     "sample": f"{m} steps of the strict naive restatement (oracle/gs_oracle.c), {dt:.1f} s"}
    del u, v, np
    return info


def measured_counters(kernel_name: str, rows: int, cols: int, tuned):
    """Per-launch PMC figures of the committed rocprofv3 profile of this kernel on this grid
    (profiles/counters.json, a list written by tools/summarize_profile.py from separate --pmc passes; every
    entry names the layout it was measured with):
    {"traffic": HBM bytes, "valu_insts": SQ_INSTS_VALU wave-instructions, "launch_ms": rocprofv3's average
    launch duration, "rows_per_unit", "cols_per_lane", "steps_per_pass", "source"}; {} when no profile of
    this kernel on this grid is committed."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        with open(path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        return {}
    label = kernel_name.split("@")[0]
    best = {}
    for e in data if isinstance(data, list) else []:
        if e.get("kernel") == label and e.get("rows") == rows and e.get("cols") == cols:
            if not best or e.get("rows_per_unit") == tuned[0]:
                best = e
    return best


def scaled_valu_insts(pmc, tuned):
    """SQ_INSTS_VALU of the committed profile, re-scaled when this run's tuner picked another unit height of
    the same lane layout: a unit of h rows computes 4 h + 12 level-rows for 4 h stored ones (the 2K apron rows
    of the level pipeline), everything else is the same instruction stream.  Returns (instructions, how)."""
    insts = pmc.get("valu_insts")
    if not insts:
        return None, None
    h0, h = pmc.get("rows_per_unit"), tuned[0]
    if not h0 or not h or h0 == h:
        return insts, "measured (profile of this layout)"
    if pmc.get("cols_per_lane") != tuned[2] or pmc.get("steps_per_pass") != tuned[1]:
        return None, f"profile is for {pmc.get('cols_per_lane')} col/lane, {pmc.get('steps_per_pass')} steps/pass"
    k = tuned[1] or 4
    return insts * ((k * h + k * (k - 1)) / (k * h)) / ((k * h0 + k * (k - 1)) / (k * h0)), \
        f"scaled from the profile's {h0}-row units to this run's {h}-row units"


def developed_species(sim, rows, cols, develop_steps=4000):
    """A pattern-forming state instead of the reference's benchmark input: U = 1, V = 0 with one
    12 x 12 seed (U = 0.5, V = 0.25) per 40 000 cells and 1 % noise, advanced `develop_steps` steps
    (tools/pattern_rate.py, profiles/r01_soak.md: spots replicate until they fill the grid)."""
    import numpy as np

    from grayscott_amd import Evolving, HipConcentration, Species

    rng = np.random.default_rng(2024)
    u0 = np.ones((rows, cols), np.float32)
    v0 = np.zeros((rows, cols), np.float32)
    for _ in range(max(4, rows * cols // 40000)):
        r, c = int(rng.integers(0, max(1, rows - 12))), int(rng.integers(0, max(1, cols - 12)))
        u0[r:r + 12, c:c + 12] = 0.5
        v0[r:r + 12, c:c + 12] = 0.25
    u0 += rng.random(u0.shape, dtype=np.float32) * np.float32(0.01)
    v0 += rng.random(v0.shape, dtype=np.float32) * np.float32(0.01)
    ctx = sim.context
    u = Evolving([HipConcentration(ctx, u0.shape), HipConcentration(ctx, u0.shape)])
    v = Evolving([HipConcentration(ctx, u0.shape), HipConcentration(ctx, u0.shape)])
    u.in_out()[0].upload(ctx, u0)
    v.in_out()[0].upload(ctx, v0)
    del u0, v0
    species = Species(ctx, u, v)
    sim.perform_steps(species, develop_steps)
    return species


NOMINAL_SCLK_MHZ = 2400.0  # the clock VALU_PEAK_TLANEOPS is priced at


def sample_clock_and_power(work, device: int):
    """Medians of rocm-smi's shader clock (MHz) and socket power (W) sampled while `work()` runs; None when
    rocm-smi is missing or says nothing useful (informational fields, never part of `value`)."""
    import re
    import shutil
    import statistics
    import subprocess
    import threading

    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    sclk, power, stop = [], [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                out = subprocess.run([smi, "-d", str(device), "--showclocks", "--showpower"], capture_output=True,
                                     text=True, timeout=10).stdout
            except Exception:
                return
            m = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", out)
            if m:
                sclk.append(float(m.group(1)))
            m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
            if m:
                power.append(float(m.group(1)))
            stop.wait(0.2)

    thread = threading.Thread(target=sampler, daemon=True)
    thread.start()
    try:
        work()
    finally:
        stop.set()
        thread.join(timeout=15)
    busy = [c for c in sclk if c > 1000.0]          # samples taken while the kernel ran
    if not busy:
        return None
    return {"sclk_MHz": statistics.median(busy), "power_W": statistics.median(power) if power else None, "samples": len(busy)}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--grid", default="", help="ROWSxCOLS: override the grid of the chosen scaling mode")
    ap.add_argument("--rows", type=int, default=0, help="override the grid (diagnostics only)")
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each, back to back; `value` is their median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the informational legs (fused flavour, developed pattern)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N > 1 on a 1-GPU box: all ranks share GPU 0 and torch.distributed uses gloo; needs "
                         "GS_RCCL_LIBRARY to name a transport that accepts several ranks per device "
                         "(tests/cpp/shm_transport.cpp).  Checks the code path, the numbers mean nothing")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from grayscott_amd import HipArgs, Parameters, Simulation, capi
    from grayscott_amd import dist as gsd

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

    rows, cols = grid_for(args.gpus, args.scaling)
    if args.grid:
        rows, cols = (int(x) for x in args.grid.lower().split("x"))
    if args.rows and args.cols:
        rows, cols = args.rows, args.cols
    # The library's defaults, nothing pinned: one kernel launch per pass, so "launch" in the roofline
    # object is unambiguous and comparable with rocprofv3's per-kernel average.
    hip_args = HipArgs(devices=[local_rank], rank=rank, world=world, unique_id=unique_id)
    sim = Simulation.new(Parameters(), hip_args)
    ctx = sim.context
    cells = rows * cols
    cells_per_gpu = cells / args.gpus
    single = args.gpus == 1
    with_extra = single and not args.no_extra

    # Everything the timed regions touch exists BEFORE the first of them: the timed Species (Species::new on
    # the device, HBM-resident) and, for the co-headline, the developed pattern.  Nothing is allocated, freed
    # or filled between tuning and timing -- round 2's line read 8 % low because 4 GiB of planes were created
    # in that gap and the first launches after it ran on an idle chip's clocks.
    species = sim.make_species([rows, cols])
    sp_dev = developed_species(sim, rows, cols) if with_extra else None     # 4000 steps: also tunes the context
    tuned = (0, 0, 0)
    if world == 1:
        # gs_run chooses unit height / fused steps / columns per lane on line, from timed passes of the
        # simulation itself (per context and shape).  It finishes here, on passes of the timed Species, so
        # that neither the W warm-up steps nor the K timed ones contain tuning passes whatever W and K are.
        for _ in range(8):
            tuned = ctx.get_tuned(rows, cols)
            if tuned[0] > 0:
                break
            sim.perform_steps(species, 400)
    else:
        # A slab chain does not tune on line: rank 0 tunes on a throw-away single slab of the slab's
        # shape and every rank is handed the same configuration (grayscott_amd/dist.py).
        tuned = gsd.share_tuning(sim, rows // world, cols, rank, world, device=red_dev, local_device=local_rank)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_run(sp, steps):
        """(wall seconds, HIP-event ms, passes) of `steps` steps bracketed as the contract says."""
        ctx.sync()
        barrier()
        torch.cuda.synchronize()
        p0 = ctx.stats()["passes"]
        t0 = time.perf_counter()
        ctx.timer_start()                               # HIP events on the library's own stream
        sim.prepare_steps(sp, steps)
        ms = ctx.timer_stop()
        ctx.sync()
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t0, ms, ctx.stats()["passes"] - p0

    def repeated(sp, steps, repeats):
        """`repeats` timed regions of `steps` steps each, back to back; per region the maximum over ranks."""
        runs = []
        for _ in range(repeats):
            wall, ms, passes = timed_run(sp, steps)
            if world > 1:
                t = torch.tensor([wall, ms], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                wall, ms = float(t[0]), float(t[1])
            runs.append((wall, ms, passes))
        return runs

    def median_run(runs):
        return sorted(runs, key=lambda r: r[0])[len(runs) // 2]

    def warm(sp):
        """At least 120 ms of untimed steps right before a warm-up, no host gap: the timed launches meet the
        clocks and caches of a running simulation, which is what the number claims to describe.  (A chip that
        comes out of an idle stretch -- the tuning's waits, a clock sample -- needs ~50 ms of load before its rate
        settles: with 18 ms here the first five 5-ms regions read 1-4 % low, profiles/r03_sweeps.md section 6.)"""
        rate = 1.2e12 if cells_per_gpu >= (1 << 24) else 5.0e11        # cell-steps per second, a high guess
        n = (max(24, int(0.12 * rate / cells_per_gpu)) + 11) // 12 * 12
        sim.perform_steps(sp, n)                        # whole passes only, whatever the tuner fuses (2, 3 or 4 steps)
        return n

    # A rehearsal of the timed region first: the first torch.cuda.synchronize() / barrier / HIP-event calls of a
    # process may initialise things lazily, and a chip that idles for 2 ms runs its next ~10 ms at lower clocks
    # (tools/region_startup.py: a 5-pass region after 2 ms of idle reads 5 % low).
    # The W warm-up steps come first and the long untimed phase after them, directly before the timed regions: a W
    # that is not a whole number of passes (the driver's 5) ends in a single-step launch of another kernel, and the
    # chip, which sits on its power limit, answers that 0.7 ms change of load with a 20 ms dip -- the first four
    # 5-ms regions read 1-5 % low with W = 5 and not with W = 0, 4 or 8 (profiles/r03_sweeps.md, section 6).
    timed_run(species, args.steps)
    sim.perform_steps(species, args.warmup)
    extra_warm_steps = warm(species)
    runs = repeated(species, args.steps, args.repeats)
    wall, event_ms, passes = median_run(runs)
    walls = [r[0] for r in runs]

    per_rank, comm = [], []
    if world > 1:
        # Per rank: its own launch time, and -- in an untimed repeat with HIP events on the halo and compute
        # streams (gs_ctx_set_pass_timing) -- whether the boundary band + ghost-row exchange hid behind the
        # interior kernel.  Then what RCCL itself says about the communicator, and where every rank runs.
        _, my_ms, my_passes = median_run([timed_run(species, args.steps) for _ in range(3)])
        n_timed = min(64, max(1, my_passes))
        ctx.set_pass_timing(n_timed)
        timed_run(species, args.steps)
        st = ctx.stats()
        ctx.set_pass_timing(0)
        tp = max(1, st["timed_passes"])
        mine = torch.tensor([my_ms / max(1, my_passes), st["halo_ms"] / tp, st["interior_ms"] / tp,
                             st["halo_exposed_ms"] / tp, float(st["timed_passes"])] +
                            [float(x) for x in ctx.comm_info()] + [float(local_rank)],
                            dtype=torch.float64, device=red_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        for r in allr:
            x = [float(v) for v in r.cpu()]
            per_rank.append({"launch_ms": x[0], "halo_stream_ms_per_pass": x[1], "interior_ms_per_pass": x[2],
                             "halo_exposed_ms_per_pass": x[3], "timed_passes": int(x[4]),
                             "rccl_rank": int(x[6]), "rccl_device": int(x[7]), "local_rank": int(x[8])})
            comm.append(int(x[5]))

    kernel_name, _ = ctx.info()
    value = cells * args.steps / wall / 1e6
    steps_per_launch = args.steps / passes
    pmc = measured_counters(kernel_name, int(rows // args.gpus), cols, tuned)
    valu_insts, valu_how = scaled_valu_insts(pmc, tuned)

    def roofline_of(event_ms, passes):
        """The roofline object of one timed region (its own launch time; the committed profile's counters)."""
        launch_s = event_ms * 1e-3 / passes
        algo_bytes = BYTES_PER_CELL_STEP * cells_per_gpu * steps_per_launch
        algo_gbs = algo_bytes / launch_s / 1e9
        traffic = pmc.get("traffic")
        hbm_physical = traffic / launch_s / 1e9 / HBM_PEAK_GBS if traffic else None
        valu_rate = valu_insts * 64 / launch_s / 1e12 if valu_insts else None
        useful_rate = USEFUL_VALU_PER_CELL_STEP * cells_per_gpu * steps_per_launch / launch_s / 1e12
        # Which roof binds: with K >= 3 steps fused per HBM pass the kernel moves ~16 B per cell for K
        # steps and is bound by VALU issue; a single-step pass is bound by HBM.
        valu_bound = steps_per_launch >= 3
        if valu_bound:
            # issued VALU lane-instructions (PMC SQ_INSTS_VALU x 64, committed profile of this layout) per
            # launch time against the chip's plain-f32 issue rate; without a matching profile, the useful
            # instructions alone (computed from this run: a lower bound of what was issued)
            achieved = valu_rate if valu_rate else useful_rate
            frac = achieved / VALU_PEAK_TLANEOPS
        else:
            achieved = traffic / launch_s / 1e9 if traffic else algo_gbs
            frac = hbm_physical if hbm_physical else algo_gbs / HBM_PEAK_GBS
        return {
            "bound": "valu-issue" if valu_bound else "hbm",
            "achieved": achieved,
            "peak": VALU_PEAK_TLANEOPS if valu_bound else HBM_PEAK_GBS,
            "unit": "T lane-ops/s" if valu_bound else "GB/s",
            "frac": frac,
            "frac_source": (valu_how if valu_rate else "useful instructions only (no profile of this layout committed)")
                           if valu_bound else ("PMC traffic" if traffic else "algorithmic bytes"),
            "valu": valu_rate / VALU_PEAK_TLANEOPS if valu_rate else None,
            "useful_valu": useful_rate / VALU_PEAK_TLANEOPS,
            "hbm_physical": hbm_physical,
            # SURVEY section 8(d)'s algorithmic figure (16 B per cell-step): a throughput in GB/s-equivalents,
            # NOT a fraction of the HBM roof once K steps share one HBM pass (it exceeds the peak by design)
            "algorithmic_GBps": algo_gbs,
            "algorithmic_frac": algo_gbs / HBM_PEAK_GBS,
            "algorithmic_frac_of_copy_ceiling": algo_gbs / HBM_COPY_CEILING_GBS,
            "launch_ms": launch_s * 1e3,
            # rocprofv3's average duration of the same kernel in the committed profile (profiling lowers clocks)
            "profile_launch_ms": pmc.get("launch_ms"),
            "launches": passes,
            "steps_per_launch": steps_per_launch,
            "algorithmic_bytes_per_launch": algo_bytes,
            "traffic": traffic,                      # HBM bytes per launch, PMC (null: not profiled)
            "valu_insts_per_launch": valu_insts,     # SQ_INSTS_VALU per launch, PMC (null: not profiled)
            "counters_source": pmc.get("source"),
            "counters_layout": ({"rows_per_unit": pmc.get("rows_per_unit"), "steps_per_pass": pmc.get("steps_per_pass"),
                                 "cols_per_lane": pmc.get("cols_per_lane")} if pmc else None),
        }

    roofline = roofline_of(event_ms, passes)
    extra, developed, clocks = None, None, None
    if with_extra:
        # co-headline: the same kernel, same context, same launches on a developed spot pattern (the chip
        # sustains a lower clock on non-trivial operands; BASELINE.md asks for "random/real data not zeros")
        sim.perform_steps(sp_dev, args.warmup)
        warm(sp_dev)
        runs_dev = repeated(sp_dev, args.steps, args.repeats)
        w_dev, ms_dev, p_dev = median_run(runs_dev)
        developed = {"value": cells * args.steps / w_dev / 1e6,
                     "value_min": cells * args.steps / max(r[0] for r in runs_dev) / 1e6,
                     "value_max": cells * args.steps / min(r[0] for r in runs_dev) / 1e6,
                     "repeats": len(runs_dev),
                     "roofline": {k: v for k, v in roofline_of(ms_dev, p_dev).items()
                                  if k in ("bound", "achieved", "peak", "unit", "frac", "frac_source", "valu", "useful_valu",
                                           "hbm_physical", "algorithmic_GBps", "launch_ms")}}
        # informational: shader clock and socket power while the same kernel runs (rocm-smi samples next to
        # an untimed repeat of the timed run; the VALU roof is priced at the nominal 2.4 GHz, the chip
        # sustains less on its power limit)
        clocks = sample_clock_and_power(lambda: timed_run(species, max(args.steps, 8000)), local_rank)
        # informational: the fused-tap flavour (GS_MATH_FUSED: bit-identical wherever no sub-normal
        # intermediate occurs, |diff| <= 1e-37 elsewhere -- inside north_star's 1e-5 tolerance)
        sim_c = Simulation.new(Parameters(), HipArgs(devices=[local_rank], math=capi.GS_MATH_FUSED))
        species_c = sim_c.make_species([rows, cols])
        sim_c.perform_steps(species_c, max(args.warmup, 400))
        tc = time.perf_counter()
        sim_c.perform_steps(species_c, args.steps)
        tc = time.perf_counter() - tc
        extra = {"kernel": sim_c.context.info()[0], "value": rows * cols * args.steps / tc / 1e6}
        sim_c.context.close()
        del species_c, sim_c
    result = {
        # BASELINE.json's metric, verbatim; `value` is its first quantity, the `roofline` object
        # carries the second
        "metric": "Mcells×steps/s and achieved HBM GB/s (% of roofline), 16384² f32 grid",
        "value": value,                                   # the median of `repeats` timed regions
        "unit": "Mcells×steps/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not args.rehearsal else "synthetic (REHEARSAL: ranks share one GPU, not a measurement)",
        "repeats": len(runs),
        "value_min": cells * args.steps / max(walls) / 1e6,
        "value_max": cells * args.steps / min(walls) / 1e6,
        "values": [round(cells * args.steps / w / 1e6) for w in walls],     # every timed region, in order
        # untimed steps between the W warm-up steps and the first timed region (whole passes, >= 120 ms)
        "untimed_steps_after_warmup": extra_warm_steps,
        "config": {
            "workload": f"{rows}x{cols} f32 (rows x cols), Species::new init, default feed/kill, "
                        f"double-buffered U/V in HBM",
            "grid": [rows, cols],
            "cells_per_gpu": cells // args.gpus,
            "kernel": kernel_name,
            "tuned": {"rows_per_unit": tuned[0], "steps_per_pass": tuned[1], "cols_per_lane": tuned[2]},
            "launches_per_pass": 1 if single else 2,
            "partition": "single GPU" if single else
                         f"{args.gpus} row slabs, RCCL send/recv ghost rows",
        },
        "roofline": roofline,
    }
    if per_rank:
        result["rccl_ranks"] = comm[0]
        result["ranks"] = per_rank
    if clocks:
        roofline["sclk_MHz_under_load"] = clocks["sclk_MHz"]
        roofline["socket_power_W_under_load"] = clocks["power_W"]
        if roofline["valu"] and roofline["bound"] == "valu-issue":
            roofline["valu_at_sustained_clock"] = roofline["valu"] / (clocks["sclk_MHz"] / NOMINAL_SCLK_MHZ)
    if developed is not None:
        result["value_developed_pattern"] = developed["value"]
        result["developed_pattern"] = developed
    if extra is not None:
        result["fused_flavour"] = extra
    if rank == 0 and single and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
