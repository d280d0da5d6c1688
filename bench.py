#!/usr/bin/env python3
"""Benchmark of the Gray-Scott step path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--grid RxC]

One "step" = one simulation time step of the whole grid (the reference's own throughput
unit is cells x steps, compute/shared/src/benchmark.rs:55-59; its "compute" workload is
perform_steps only, :77-83).  Inputs are resident in HBM before the timed region starts.

N = 1 : BASELINE.json's headline workload, 16384 x 16384 f32 (config 3), Species::new init,
        default feed/kill.  Rank 0 also times the same kernel on a developed spot pattern, the
        fused-tap flavour, and the CPU ports on the host cores on a bounded sample.
N > 1 : one process per GPU (torchrun; `python bench.py --gpus N` without WORLD_SIZE starts that torchrun
        itself, as a CHILD process, before anything touches a GPU), row slabs with ghost-row exchange
        through RCCL send/recv inside libgs_hip.so.  Every rank runs a watchdog: a stage that exceeds its
        bound prints one JSON line {"error", "rank", "stage"} and the process exits non-zero.
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
    if not h:
        return insts, f"profile of {h0}-row units; this run's layout is not known (nothing tuned or pinned)"
    if not h0 or h0 == h:
        return insts, "measured (profile of this layout)"
    if pmc.get("cols_per_lane") != tuned[2] or pmc.get("steps_per_pass") != tuned[1]:
        return None, f"profile is for {pmc.get('cols_per_lane')} col/lane, {pmc.get('steps_per_pass')} steps/pass"
    k = tuned[1] or 4
    return insts * ((k * h + k * (k - 1)) / (k * h)) / ((k * h0 + k * (k - 1)) / (k * h0)), \
        f"scaled from the profile's {h0}-row units to this run's {h}-row units"


def developed_start(rows, cols):
    """Start of a pattern-forming run instead of the reference's benchmark input: U = 1, V = 0 with one
    12 x 12 seed (U = 0.5, V = 0.25) per 40 000 cells and 1 % noise (tools/pattern_rate.py, profiles/r01_soak.md:
    4000 steps later spots fill the grid).  Returns dense host arrays (u0, v0)."""
    import numpy as np

    rng = np.random.default_rng(2024)
    u0 = np.ones((rows, cols), np.float32)
    v0 = np.zeros((rows, cols), np.float32)
    for _ in range(max(4, rows * cols // 40000)):
        r, c = int(rng.integers(0, max(1, rows - 12))), int(rng.integers(0, max(1, cols - 12)))
        u0[r:r + 12, c:c + 12] = 0.5
        v0[r:r + 12, c:c + 12] = 0.25
    u0 += rng.random(u0.shape, dtype=np.float32) * np.float32(0.01)
    v0 += rng.random(v0.shape, dtype=np.float32) * np.float32(0.01)
    return u0, v0


def upload_species(sim, u0, v0):
    """A Species of `sim`'s context whose input planes hold (u0, v0); `steps_done` counts what it has run."""
    from grayscott_amd import Evolving, HipConcentration, Species

    ctx = sim.context
    u = Evolving([HipConcentration(ctx, u0.shape), HipConcentration(ctx, u0.shape)])
    v = Evolving([HipConcentration(ctx, u0.shape), HipConcentration(ctx, u0.shape)])
    u.in_out()[0].upload(ctx, u0)
    v.in_out()[0].upload(ctx, v0)
    species = Species(ctx, u, v)
    species.steps_done = 0
    return species


NOMINAL_SCLK_MHZ = 2400.0  # the clock VALU_PEAK_TLANEOPS is priced at


def sample_clock_and_power(work, device: int, cell_steps: float = 0.0):
    """Medians of rocm-smi's shader clock (MHz) and socket power (W) sampled while `work()` runs, the board's
    power cap, and -- from the card's accumulated-energy counter, first and last sample taken while the kernel
    ran -- the average power over that window and the energy per cell-step (`work` returns (wall seconds, ...)
    for `cell_steps` cell-steps, so pJ per cell-step = watts x seconds / cell-steps).  None when rocm-smi is
    missing or says nothing useful (informational fields, never part of `value`)."""
    import re
    import shutil
    import statistics
    import subprocess
    import threading

    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    sclk, power, energy, stop = [], [], [], threading.Event()
    cap = [None]

    def sampler():
        try:
            out = subprocess.run([smi, "-d", str(device), "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", out)
            if m:
                cap[0] = float(m.group(1))
        except Exception:
            pass
        while not stop.is_set():
            t0 = time.monotonic()
            try:
                out = subprocess.run([smi, "-d", str(device), "--showclocks", "--showpower", "--showenergycounter"],
                                     capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            t1 = time.monotonic()
            busy = False
            m = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", out)
            if m:
                busy = float(m.group(1)) > 1000.0
                sclk.append(float(m.group(1)))
            m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
            if m:
                power.append(float(m.group(1)))
            m = re.search(r"Accumulated Energy \(uJ\):\s*([0-9.]+)", out)
            if m and busy:
                energy.append((0.5 * (t0 + t1), float(m.group(1))))
            stop.wait(0.2)

    thread = threading.Thread(target=sampler, daemon=True)
    thread.start()
    try:
        ret = work()
    finally:
        stop.set()
        thread.join(timeout=15)
    busy = [c for c in sclk if c > 1000.0]          # samples taken while the kernel ran
    if not busy:
        return None
    out = {"sclk_MHz": statistics.median(busy), "power_W": statistics.median(power) if power else None,
           "samples": len(busy), "power_cap_W": cap[0]}
    # the last sample may have been taken after the kernel ended: leave it out when there are enough
    win = energy[:-1] if len(energy) >= 4 else energy
    if len(win) >= 2 and win[-1][0] - win[0][0] > 0.5 and cell_steps > 0 and ret:
        watts = (win[-1][1] - win[0][1]) * 1e-6 / (win[-1][0] - win[0][0])
        out["energy_W"] = watts
        out["energy_window_s"] = win[-1][0] - win[0][0]
        out["energy_pJ_per_cell_step"] = watts * ret[0] / cell_steps * 1e12
    return out


def planes_equal(a, b) -> bool:
    """Bit-for-bit equality of two HipConcentrations of one shape, compared on the device (the planes are
    1 GiB each at 16384^2): int32 views, so that NaNs and signed zeros count as what they are."""
    import torch

    ok = True
    for (_, _, x), (_, _, y) in zip(a.torch_views(), b.torch_views()):
        ok = ok and bool(torch.equal(x.view(torch.int32), y.view(torch.int32)))
    return ok


def verify_single_gpu(sim_s, sp_s, species, sp_dev_s, sp_dev, rows, cols, timed_kernel):
    """In-run proof that the timed launches did the work, and the HBM-bound single-step leg north_star asks the
    rocprof evidence for.  A second context pinned to the single-step stream kernel (one launch = one step =
    one read and one write of U and V: 16 B per cell-step of HBM traffic) starts from the same Species::new,
    is timed over 5 regions of steps (`single_step`), then runs on to exactly the number of steps the timed
    Species has taken -- tuning passes, warm-ups and every timed region included -- and both planes must be
    equal bit for bit; the same for the developed pattern (same upload, same step count)."""
    import statistics

    ctx_s = sim_s.context
    cells = rows * cols
    n_region = max(40, min(400, int(0.08 * 3.5e11 / cells)))        # ~80 ms per region
    n_region = min(n_region, max(1, (species.steps_done - 40) // 6))
    sim_s.perform_steps(sp_s, n_region)                              # untimed: clocks, first touches
    sp_s.steps_done += n_region
    rates, launch_ms = [], []
    for _ in range(5):
        ctx_s.sync()
        t0 = time.perf_counter()
        ctx_s.timer_start()
        sim_s.prepare_steps(sp_s, n_region)
        ms = ctx_s.timer_stop()
        ctx_s.sync()
        wall = time.perf_counter() - t0
        sp_s.steps_done += n_region
        rates.append(cells * n_region / wall / 1e6)
        launch_ms.append(ms / n_region)
    rate = statistics.median(rates)
    step_ms = statistics.median(launch_ms)
    gbs = BYTES_PER_CELL_STEP * cells / (step_ms * 1e-3) / 1e9
    label = ctx_s.info()[0]
    pmc = measured_counters(label, rows, cols, (0, 0, 0))
    single_step = {
        "kernel": label,
        "value": rate, "unit": "Mcells×steps/s", "values": [round(r) for r in rates], "steps_per_region": n_region,
        "launch_ms": step_ms,                                    # HIP events on the library's stream, per launch
        "hbm_GBps": gbs,                                         # algorithmic: 16 B per cell-step, one step per launch
        "frac_of_8TBps": gbs / HBM_PEAK_GBS,
        "frac_of_copy_ceiling": gbs / HBM_COPY_CEILING_GBS,
        "traffic": pmc.get("traffic"),                           # HBM bytes per launch, PMC of the committed profile
        "hbm_physical_GBps": pmc["traffic"] / (step_ms * 1e-3) / 1e9 if pmc.get("traffic") else None,
        "profile_launch_ms": pmc.get("launch_ms"), "counters_source": pmc.get("source"),
    }
    # The same kernel on two more, separately allocated sets of planes: where four 1 GiB allocations land in HBM decides
    # which of three levels (~330 / 350 / 375 k at 16384^2) this HBM-bound kernel reads, from box to box and from one
    # Species to the next (profiles/r04_sweeps.md, section 8).  `value` above is the Species the replay uses.
    placements = [cells / (step_ms * 1e-3) / 1e6]        # (HIP-event rates, like the two below)
    for _ in range(2):
        extra = sim_s.make_species([rows, cols])
        sim_s.perform_steps(extra, n_region)
        r3 = []
        for _ in range(3):
            ctx_s.timer_start()
            sim_s.prepare_steps(extra, n_region)
            r3.append(cells * n_region / (ctx_s.timer_stop() * 1e-3) / 1e6)
        ctx_s.sync()
        placements.append(statistics.median(r3))
        for c in extra.u._pair + extra.v._pair:
            c.destroy()
    single_step["by_plane_placement"] = [round(x) for x in placements]
    single_step["frac_of_8TBps_best_placement"] = max(placements) * 1e6 * BYTES_PER_CELL_STEP / 1e9 / HBM_PEAK_GBS
    left = species.steps_done - sp_s.steps_done
    if left < 0:
        raise RuntimeError(f"the replay is ahead of the timed Species ({sp_s.steps_done} > {species.steps_done} steps)")
    sim_s.perform_steps(sp_s, left)
    sp_s.steps_done += left
    species.context().sync()
    a, b = species.in_out(), sp_s.in_out()
    verified = {"against": f"single-step kernel {label} in a second context, same start", "timed_kernel": timed_kernel,
                "steps": species.steps_done,
                "equal": planes_equal(a[0], b[0]) and planes_equal(a[1], b[1])}
    if sp_dev is not None and sp_dev_s is not None:
        sim_s.perform_steps(sp_dev_s, sp_dev.steps_done)
        a, b = sp_dev.in_out(), sp_dev_s.in_out()
        verified["developed_pattern"] = {"steps": sp_dev.steps_done,
                                         "equal": planes_equal(a[0], b[0]) and planes_equal(a[1], b[1])}
    return single_step, verified


def range_checksums(view, block: int = 2048):
    """Two wrapping int64 sums (plain, position-weighted) of the bit patterns of every `block` rows of a plane
    view: a checksum of checksums for planes that live on different GPUs."""
    import torch

    out = []
    for k0 in range(0, view.shape[0], block):
        x = view[k0:k0 + block].view(torch.int32).to(torch.int64)
        w = (torch.arange(x.numel(), device=x.device, dtype=torch.int64) % 65521 + 1).reshape(x.shape)
        out.append((int(x.sum()), int((x * w).sum())))
    return out


def fill_noise(species, ctx, block: int = 2048):
    """Random U in [0, 1), V in [0, 0.5) in every cell, written on the device through the planes' pointers: a function
    of the global row block alone, so that every rank of a chain and a single-GPU replay hold the same start."""
    import torch

    in_u, in_v, _, _ = species.in_out()
    cols = in_u.shape()[1]
    for si, conc in enumerate((in_u, in_v)):
        for row0, rows, view in conc.torch_views():
            for k in range(row0 // block, (row0 + rows + block - 1) // block):
                g = torch.Generator(device=view.device)
                g.manual_seed(1_000_003 * (k + 1) + si)
                x = torch.rand((block, cols), generator=g, device=view.device, dtype=torch.float32)
                lo, hi = max(k * block, row0), min((k + 1) * block, row0 + rows)
                view[lo - row0:hi - row0].copy_((x if si == 0 else x * 0.5)[lo - k * block:hi - k * block])
        torch.cuda.synchronize()
        conc.mark_written(ctx)


def verify_slab_chain(sim, species, rows, cols, rank, world, local_rank, rehearsal):
    """N > 1: every rank checksums the rows it holds; rank 0 replays the WHOLE grid alone (a single slab on its
    own GPU: 288 GB hold BASELINE config 5 several times over) for as many steps as the chain has taken and
    checksums the same row ranges.  Equal sums = the exchanged ghost rows carried the right data on every seam.
    Twice: the timed Species (the reference's input: signal on the seam under the seed only), and 203 steps (a
    remainder pass, full passes) from random data everywhere, so that EVERY seam carries signal from step one."""
    import torch.distributed as dist

    from grayscott_amd import HipArgs, Parameters, Simulation

    def local_sums(sp):
        sp.context().sync()
        out = []
        for conc in sp.in_out()[:2]:
            for row0, nrows, view in conc.torch_views():
                out.append((row0, nrows, range_checksums(view)))
        return out

    noise_steps = 203
    noisy = sim.make_species([rows, cols])
    fill_noise(noisy, sim.context)
    sim.perform_steps(noisy, noise_steps)
    gathered = [None] * world
    dist.all_gather_object(gathered, (local_sums(species), local_sums(noisy)))
    result = None
    if rank == 0:
        solo = Simulation.new(Parameters(), HipArgs(devices=[local_rank]))

        def compare(which, whole):
            views = [conc.torch_views()[0][2] for conc in whole.in_out()[:2]]
            bad, blocks = [], 0
            for r, both in enumerate(gathered):
                parts = both[which]
                per_plane = len(parts) // 2
                for i, (row0, nrows, sums) in enumerate(parts):
                    ref = range_checksums(views[i // per_plane][row0:row0 + nrows])
                    blocks += len(ref)
                    if ref != sums and r not in bad:
                        bad.append(r)
            return bad, blocks

        whole = solo.make_species([rows, cols])
        solo.perform_steps(whole, species.steps_done)
        bad, blocks = compare(0, whole)
        fill_noise(whole, solo.context)
        solo.perform_steps(whole, noise_steps)
        bad_n, _ = compare(1, whole)
        solo.context.close()
        result = {"against": "single-GPU run of the whole grid on rank 0 (row-block checksums of U and V)",
                  "steps": species.steps_done, "equal": not bad and not bad_n, "blocks": blocks, "mismatching_ranks": bad,
                  "random_start": {"steps": noise_steps, "equal": not bad_n, "mismatching_ranks": bad_n}}
    return result


class Watchdog:
    """Per-rank stage timer.  `with wd.stage(name, seconds):` arms a bound; a daemon thread that finds it
    exceeded prints ONE JSON line {"error", "rank", "stage", "bound_s"} and ends the process with exit code 3
    (os._exit: the main thread may sit in ncclCommInitRank or a stream wait that never returns).  No restart,
    no re-exec: torchrun sees the non-zero exit and takes the other ranks down.
    GS_BENCH_WATCHDOG_S caps every bound (tests use a few seconds)."""

    EXIT_CODE = 3

    def __init__(self, rank: int = 0, out=None):
        import threading

        self.rank = rank
        self.out = out or sys.stdout
        self._lock = threading.Lock()
        self._stage = None          # (name, deadline, bound)
        cap = os.environ.get("GS_BENCH_WATCHDOG_S", "")
        self._cap = float(cap) if cap else None
        self._thread = threading.Thread(target=self._watch, daemon=True)
        self._thread.start()

    def _watch(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                st = self._stage
            if st and time.monotonic() > st[1]:
                line = json.dumps({"error": f"stage '{st[0]}' exceeded its bound of {st[2]:.0f} s",
                                   "rank": self.rank, "stage": st[0], "bound_s": st[2]})
                try:
                    self.out.write(line + "\n")
                    self.out.flush()
                finally:
                    os._exit(self.EXIT_CODE)

    def stage(self, name: str, seconds: float):
        wd = self
        bound = min(seconds, self._cap) if self._cap else seconds

        class _Stage:
            def __enter__(self_inner):
                with wd._lock:
                    wd._stage = (name, time.monotonic() + bound, bound)
                fault = os.environ.get("GS_BENCH_FAULT", "")       # "stall:RANK:STAGE" (tests)
                if fault.startswith("stall:"):
                    _, r, st = fault.split(":")
                    if int(r) == wd.rank and st == name:
                        time.sleep(1e6)

            def __exit__(self_inner, *exc):
                with wd._lock:
                    wd._stage = None
                return False

        return _Stage()


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: start `torch.distributed.run` as a
    CHILD process -- before this process has touched a GPU or loaded libgs_hip.so -- relay its output (the one
    JSON line) and return its exit code.  Never an exec of this process."""
    import socket
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rehearsal and not env.get("GS_RCCL_LIBRARY"):
        # all ranks share GPU 0: RCCL refuses that, the library binds the shared-memory transport double
        import shutil

        out_dir = os.path.join(ROOT, "gpurun_out", "rehearsal")
        os.makedirs(out_dir, exist_ok=True)
        lib = os.path.join(out_dir, "libshm_transport.so")
        cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        r = subprocess.run([cc, "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                            os.path.join(ROOT, "tests", "cpp", "shm_transport.cpp"), "-o", lib, "-lrt", "-lpthread"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            print("bench.py: building the rehearsal transport failed:\n" + r.stdout + r.stderr, file=sys.stderr)
            return 2
        env["GS_RCCL_LIBRARY"] = lib
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    limit = float(os.environ.get("GS_BENCH_CHILD_TIMEOUT_S", "1500"))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)       # inherits stdout / stderr
    try:
        return child.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal

        print(json.dumps({"error": f"the torchrun child exceeded {limit:.0f} s", "rank": -1, "stage": "child"}))
        try:
            os.killpg(child.pid, signal.SIGKILL)     # the session this process started, nothing else
        except OSError:
            pass
        child.wait()
        return 3


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
                    help="skip the informational legs (fused flavour, developed pattern, clock / energy samples)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the replay that compares the timed planes with an independent run (and, on one GPU, "
                         "the single-step HBM leg that is part of it)")
    ap.add_argument("--kernel", choices=("auto", "stream"), default="auto",
                    help="diagnostics / profiles: `stream` pins the single-step HBM-bound kernel for the whole protocol")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N > 1 on a 1-GPU box: all ranks share GPU 0 and torch.distributed uses gloo; the library "
                         "binds a transport that accepts several ranks per device (tests/cpp/shm_transport.cpp, "
                         "built on the fly unless GS_RCCL_LIBRARY names one).  Checks the code path, the numbers "
                         "mean nothing")
    args = ap.parse_args()

    # N > 1 without a torchrun environment: this process is only the launcher of the torchrun CHILD -- decided
    # before torch.cuda, the process group or libgs_hip.so exist in this process.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wd = Watchdog(rank)

    with wd.stage("import", 600):
        import torch
        import torch.distributed as dist

        from grayscott_amd import HipArgs, Parameters, Simulation, capi
        from grayscott_amd import dist as gsd

    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    if args.rehearsal:
        if world > 1 and not os.environ.get("GS_RCCL_LIBRARY"):
            print("bench.py: --rehearsal under torchrun needs GS_RCCL_LIBRARY", file=sys.stderr)
            return 2
        local_rank = 0
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if args.rehearsal else "cuda"   # where the bootstrap / reduction tensors live

    rows, cols = grid_for(args.gpus, args.scaling)
    if args.grid:
        rows, cols = (int(x) for x in args.grid.lower().split("x"))
    if args.rows and args.cols:
        rows, cols = args.rows, args.cols

    # communicator creation: the process group, rank 0's RCCL unique id, ncclCommInitRank inside gs_ctx_create
    with wd.stage("init", 300):
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
        # The library's defaults, nothing pinned: one kernel launch per pass, so "launch" in the roofline
        # object is unambiguous and comparable with rocprofv3's per-kernel average.
        hip_args = HipArgs(devices=[local_rank], rank=rank, world=world, unique_id=unique_id,
                           kernel=capi.GS_KERNEL_STREAM if args.kernel == "stream" else capi.GS_KERNEL_AUTO)
        sim = Simulation.new(Parameters(), hip_args)
    ctx = sim.context
    cells = rows * cols
    cells_per_gpu = cells / args.gpus
    single = args.gpus == 1
    with_extra = single and not args.no_extra
    verify = not args.no_verify

    def run(sp, steps):
        """perform_steps, counted: the replay at the end repeats exactly the steps a Species has taken."""
        sim.perform_steps(sp, steps)
        sp.steps_done += steps

    # Everything the timed regions touch exists BEFORE the first of them: the timed Species (Species::new on
    # the device, HBM-resident) and, for the co-headline, the developed pattern.  Nothing is allocated, freed
    # or filled between tuning and timing -- round 2's line read 8 % low because 4 GiB of planes were created
    # in that gap and the first launches after it ran on an idle chip's clocks.  The same holds for the planes
    # of the replay (single GPU: a second context pinned to the single-step stream kernel).
    with wd.stage("setup", 900):
        species = sim.make_species([rows, cols])
        species.steps_done = 0
        sp_dev, sim_s, sp_s, sp_dev_s = None, None, None, None
        if single and verify:
            sim_s = Simulation.new(Parameters(), HipArgs(devices=[local_rank], kernel=capi.GS_KERNEL_STREAM))
            sp_s = sim_s.make_species([rows, cols])
            sp_s.steps_done = 0
        if with_extra:
            u0, v0 = developed_start(rows, cols)
            sp_dev = upload_species(sim, u0, v0)
            if sim_s is not None:
                sp_dev_s = upload_species(sim_s, u0, v0)
            del u0, v0
            run(sp_dev, 4000)                            # develops the pattern; also tunes the context
        tuned = (0, 0, 0)
        if world == 1:
            # gs_run chooses unit height / fused steps / columns per lane on line, from timed passes of the
            # simulation itself (per context and shape).  It finishes here, on passes of the timed Species, so
            # that neither the W warm-up steps nor the K timed ones contain tuning passes whatever W and K are.
            for _ in range(8):
                tuned = ctx.get_tuned(rows, cols)
                if tuned[0] > 0:
                    break
                run(species, 400)
        else:
            # A slab chain does not tune on line: rank 0 tunes on a throw-away single slab of the slab's
            # shape and every rank is handed the same configuration (grayscott_amd/dist.py).
            tuned = gsd.share_tuning(sim, rows // world, cols, rank, world, device=red_dev, local_device=local_rank)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_run(sp, steps):
        """(wall seconds, HIP-event ms, passes) of `steps` steps, bracketed as the contract says: barrier +
        synchronize, clock, the steps, synchronize, clock, barrier.  The closing barrier is NOT inside the wall
        time (it is an all-reduce of its own: 50-100 us per 5 ms region): the job's time is the maximum over
        ranks of these walls, taken by repeated()."""
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
        wall = time.perf_counter() - t0
        barrier()
        sp.steps_done += steps
        return wall, ms, ctx.stats()["passes"] - p0

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
        run(sp, n)                                      # whole passes only, whatever the tuner fuses (2, 3 or 4 steps)
        return n

    # A rehearsal of the timed region first: the first torch.cuda.synchronize() / barrier / HIP-event calls of a
    # process may initialise things lazily, and a chip that idles for 2 ms runs its next ~10 ms at lower clocks
    # (tools/region_startup.py: a 5-pass region after 2 ms of idle reads 5 % low).
    # The W warm-up steps come first and the long untimed phase after them, directly before the timed regions: a W
    # that is not a whole number of passes (the driver's 5) ends in a single-step launch of another kernel, and the
    # chip, which sits on its power limit, answers that 0.7 ms change of load with a 20 ms dip -- the first four
    # 5-ms regions read 1-5 % low with W = 5 and not with W = 0, 4 or 8 (profiles/r03_sweeps.md, section 6).
    with wd.stage("timed", 600):
        timed_run(species, args.steps)
        run(species, args.warmup)
        extra_warm_steps = warm(species)
        untimed_before_first = species.steps_done       # tuning + rehearsal + W + warm: everything before region 0
        runs = repeated(species, args.steps, args.repeats)
    wall, event_ms, passes = median_run(runs)
    walls = [r[0] for r in runs]

    per_rank, comm = [], []
    if world > 1:
        # Per rank: its own launch time, and -- in an untimed repeat with HIP events on the halo and compute
        # streams (gs_ctx_set_pass_timing) -- whether the boundary band + ghost-row exchange hid behind the
        # interior kernel.  Then what RCCL itself says about the communicator, and where every rank runs.
        with wd.stage("per-rank", 600):
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
    extra, developed, clocks, clocks_dev = None, None, None, None
    if with_extra:
        with wd.stage("extras", 900):
            # co-headline: the same kernel, same context, same launches on a developed spot pattern (the chip
            # sustains a lower clock on non-trivial operands; BASELINE.md asks for "random/real data not zeros")
            run(sp_dev, args.warmup)
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
            # informational: shader clock, socket power and energy per cell-step while the same kernel runs
            # (rocm-smi samples next to an untimed repeat of >= 3 s; the VALU roof is priced at the nominal
            # 2.4 GHz, the chip sustains less on its power limit), on both inputs
            long_steps = max(args.steps, int(3.2 * value * 1e6 / cells) // 12 * 12)
            clocks = sample_clock_and_power(lambda: timed_run(species, long_steps), local_rank, cells * long_steps)
            clocks_dev = sample_clock_and_power(lambda: timed_run(sp_dev, long_steps), local_rank, cells * long_steps)
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

    verified, single_step = None, None
    if verify:
        with wd.stage("verify", 900):
            try:
                if single:
                    single_step, verified = verify_single_gpu(sim_s, sp_s, species, sp_dev_s, sp_dev, rows, cols, kernel_name)
                else:
                    verified = verify_slab_chain(sim, species, rows, cols, rank, world, local_rank, args.rehearsal)
            except Exception as e:                      # the line is still worth printing
                verified = {"error": f"{type(e).__name__}: {e}"}
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
        "value_first_region": cells * args.steps / walls[0] / 1e6,          # the plain "warm up, then time K" reading
        # untimed steps between the W warm-up steps and the first timed region (whole passes, >= 120 ms) ...
        "untimed_steps_after_warmup": extra_warm_steps,
        # ... and ALL steps this Species took before region 0: on-line tuning, the rehearsal region, W, the above
        "untimed_steps_before_first_region": untimed_before_first,
        "timing": "wall clock per region between barrier + synchronize brackets, closing barrier outside; "
                  "maximum over ranks; median over regions",
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
        roofline["power_cap_W"] = clocks.get("power_cap_W")
        if roofline["valu"] and roofline["bound"] == "valu-issue":
            roofline["valu_at_sustained_clock"] = roofline["valu"] / (clocks["sclk_MHz"] / NOMINAL_SCLK_MHZ)
        if clocks.get("energy_pJ_per_cell_step"):
            result["energy_pJ_per_cell_step"] = clocks["energy_pJ_per_cell_step"]
            roofline["energy_W_from_counter"] = clocks.get("energy_W")
        cap, pw = clocks.get("power_cap_W"), clocks.get("energy_W") or clocks["power_W"]
        if cap and pw and pw >= 0.96 * cap and roofline["bound"] == "valu-issue":
            # the package sits on its power limit: what a faster instruction stream gains, the clock gives back
            roofline["bound"] = "power-capped valu"
    if developed is not None:
        result["value_developed_pattern"] = developed["value"]
        result["developed_pattern"] = developed
        if clocks_dev:
            developed["sclk_MHz_under_load"] = clocks_dev["sclk_MHz"]
            developed["socket_power_W_under_load"] = clocks_dev["power_W"]
            developed["energy_pJ_per_cell_step"] = clocks_dev.get("energy_pJ_per_cell_step")
    if extra is not None:
        result["fused_flavour"] = extra
    if single_step is not None:
        result["single_step"] = single_step
    if verified is not None:
        result["verified"] = verified
    if rank == 0 and single and not args.no_cpu_baseline:
        with wd.stage("cpu_baseline", 600):
            result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False), flush=True)
    with wd.stage("teardown", 120):
        if sim_s is not None:
            sim_s.context.close()
        ctx.close()
        if world > 1:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
