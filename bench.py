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

from benchkit.harness import (BYTES_PER_CELL_STEP, HBM_COPY_CEILING_GBS, HBM_PEAK_GBS, NOMINAL_SCLK_MHZ,  # noqa: E402,F401
                              USEFUL_VALU_PER_CELL_STEP, VALU_PEAK_TLANEOPS, Watchdog, grid_for, self_launch, usable_cpus)
from benchkit.legs import (add_clocks, cpu_baseline, developed_start, fused_flavour_leg, measured_counters,  # noqa: E402,F401
                           planes_equal, roofline_object, sample_clock_and_power, scaled_valu_insts, upload_species,
                           verify_single_gpu)
from benchkit.multigpu import (fill_noise, peer_chain_leg, per_rank_report, range_checksums,  # noqa: E402,F401
                               verify_slab_chain)


def parse_args():
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
    ap.add_argument("--extra", action="store_true",
                    help="N = 1: also the fused-tap flavour and the single-step kernel on two more sets of planes")
    ap.add_argument("--no-extra", action="store_true",
                    help="N = 1: skip the developed-pattern co-headline and the clock / power / energy samples")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the replay that compares the timed planes with an independent run (and, on one GPU, "
                         "the single-step HBM leg that is part of it)")
    ap.add_argument("--place-candidates", type=int, default=-1,
                    help="most extra blocks gs_fields_place may draw for every Species (default -1: the library's default -- "
                         "make_species places Species of >= 2^26 cells per process with at most 12)")
    ap.add_argument("--no-place", action="store_true",
                    help="planes as hipMalloc hands them out (--hip-place-candidates 0); `value` then describes a "
                         "configuration the library does not run by default")
    ap.add_argument("--no-peer-chain", action="store_true",
                    help="N > 1: skip rank 0's in-process chain over all GPUs (hipMemcpyPeerAsync, no RCCL) after the timed job")
    ap.add_argument("--bootstrap", choices=("nccl", "gloo"), default="nccl",
                    help="N > 1: backend of the torch.distributed group that carries the unique id, barriers and timing "
                         "reductions (the ghost rows always travel through the library's own RCCL communicator).  nccl: "
                         "barriers on the GPU, and torch's communicator shares the RCCL instance the library binds; gloo: "
                         "the library's communicator is the only one in the process")
    ap.add_argument("--kernel", choices=("auto", "stream"), default="auto",
                    help="diagnostics / profiles: `stream` pins the single-step HBM-bound kernel for the whole protocol")
    ap.add_argument("--rehearsal", action="store_true",
                    help="N > 1 on a 1-GPU box: all ranks share GPU 0 and torch.distributed uses gloo; the library "
                         "binds a transport that accepts several ranks per device (tests/cpp/shm_transport.cpp, "
                         "built on the fly unless GS_RCCL_LIBRARY names one).  Checks the code path, the numbers "
                         "mean nothing")
    return ap.parse_args()


def placement_report(ctx, species, sp_dev, place, slab_rows, cols):
    """What gs_fields_place did for the Species of the timed context: `transient_GiB` is the most memory it held beyond the
    planes themselves at any moment (it draws one block of a plane's size at a time and frees what it does not keep)."""
    probes, drawn = ctx.place_stats()
    plane_gib = (slab_rows + 8) * ((cols + 63) // 64 * 64) * 4 / 2 ** 30
    placed = [sp for sp in (species, sp_dev) if sp is not None and getattr(sp, "placement", None)]
    cap = 12 if place is None else place
    return {"default": place is None, "max_extra_blocks": cap,
            # (the deep stage: only when `max_extra_blocks` draws found one region only and more than half of the device's
            # memory is free -- one probe per block)
            "deep_stage_max_extra_blocks": min(124, 4 * cap), "deep_stage_used": any(sp.placement_drawn > cap for sp in placed),
            "species_placed": len(placed), "probes": probes, "extra_blocks_drawn": drawn,
            "extra_blocks_drawn_per_species": [sp.placement_drawn for sp in placed],
            "transient_GiB": round(max([sp.placement_drawn for sp in placed] or [0]) * plane_gib, 2),
            # (mean ms of the probe pass over the two slots' (U, V) pairs: before, after)
            "timed_species_probe_ms": getattr(species, "placement", None),
            "developed_species_probe_ms": getattr(sp_dev, "placement", None) if sp_dev is not None else None}


def main() -> int:
    args = parse_args()
    # N > 1 without a torchrun environment: this process is only the launcher of the torchrun CHILD -- decided
    # before torch.cuda, the process group or libgs_hip.so exist in this process.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wd = Watchdog(rank)
    stage_s = {}                                     # seconds per stage of this rank (rank 0's go into the line)

    class stage:                                     # a watchdog stage that also keeps its wall time
        def __init__(self, name, bound):
            self.name, self.inner = name, wd.stage(name, bound)

        def __enter__(self):
            self.t0 = time.perf_counter()
            return self.inner.__enter__()

        def __exit__(self, *exc):
            stage_s[self.name] = round(stage_s.get(self.name, 0.0) + time.perf_counter() - self.t0, 2)
            return self.inner.__exit__(*exc)

    # ---- the libraries, in the order that decides which copies libgs_hip.so binds (gs_hip.h: gs_runtime_info) ----
    with stage("import", 600):
        import torch
        import torch.distributed as dist

        from grayscott_amd import HipArgs, Parameters, Simulation, capi
        from grayscott_amd import dist as gsd

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
    gloo = args.rehearsal or args.bootstrap == "gloo"
    red_dev = "cpu" if gloo else "cuda"              # where the bootstrap / reduction tensors live

    rows, cols = grid_for(args.gpus, args.scaling)
    if args.grid:
        rows, cols = (int(x) for x in args.grid.lower().split("x"))
    if args.rows and args.cols:
        rows, cols = args.rows, args.cols
    cells = rows * cols
    cells_per_gpu = cells / world
    single = world == 1
    with_extra = single and not args.no_extra
    verify = not args.no_verify

    # ---- N > 1: the process group, rank 0's RCCL unique id, ncclCommInitRank inside gs_ctx_create ----------------
    with stage("init", 300):
        unique_id = None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if gloo:
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
        sim = Simulation.new(Parameters(), HipArgs(devices=[local_rank], rank=rank, world=world, unique_id=unique_id,
                                                   kernel=capi.GS_KERNEL_STREAM if args.kernel == "stream" else capi.GS_KERNEL_AUTO))
        runtime = capi.runtime_info(load_rccl=world > 1)
        runtime["bootstrap"] = ("gloo" if gloo else "nccl") if world > 1 else None
        runtime["torch"] = torch.__version__
    ctx = sim.context

    def run(sp, steps):
        """perform_steps, counted: the replay at the end repeats exactly the steps a Species has taken."""
        sim.perform_steps(sp, steps)
        sp.steps_done += steps

    # ---- everything the timed regions touch exists BEFORE the first of them -------------------------------------
    # The timed Species (Species::new on the device, HBM-resident) and, for the co-headline, the developed pattern.
    # Nothing is allocated, freed or filled between tuning and timing -- round 2's line read 8 % low because 4 GiB of
    # planes were created in that gap and the first launches after it ran on an idle chip's clocks.  The same holds for
    # the planes of the replay (single GPU: a second context pinned to the single-step stream kernel).
    # Placement by measurement (gs_fields_place) is the LIBRARY'S default for Species of >= 2^26 cells per process, in every
    # host mirror (make_species): the bench passes nothing and measures what a reference-side caller with default CliArgs
    # gets.  `value_unplaced` and `single_step.unplaced_frac_of_8TBps` are the same kernels on planes as hipMalloc hands
    # them out (profiles/r06_placement.md).
    place = 0 if args.no_place else (args.place_candidates if args.place_candidates >= 0 else None)
    with stage("setup", 900):
        species = sim.make_species([rows, cols], place_candidates=place)
        species.steps_done = 0
        sp_dev, sim_s, sp_s, sp_dev_s = None, None, None, None
        if single and verify:
            sim_s = Simulation.new(Parameters(), HipArgs(devices=[local_rank], kernel=capi.GS_KERNEL_STREAM))
            sp_s = sim_s.make_species([rows, cols], place_candidates=place)
            sp_s.steps_done = 0
        if with_extra:
            u0, v0 = developed_start(rows, cols)
            sp_dev = upload_species(sim, u0, v0, place)
            if sim_s is not None:
                sp_dev_s = upload_species(sim_s, u0, v0, 0)      # (the replay of the pattern is not timed)
            del u0, v0
            run(sp_dev, 4000)                            # develops the pattern; also tunes the context
        tuned = (0, 0, 0, 0)
        if single:
            # gs_run chooses unit height / fused steps / columns per lane / tap sharing on line, from timed passes of
            # the simulation itself (per context and shape).  It finishes here, on passes of the timed Species, so
            # that neither the W warm-up steps nor the K timed ones contain tuning passes whatever W and K are.
            for _ in range(8):
                tuned = ctx.get_tuned(rows, cols)
                if tuned[0] > 0:
                    break
                run(species, 400)
        else:
            # A slab chain does not tune on line: rank 0 tunes on a throw-away single slab of the slab's
            # shape and every rank is handed the same configuration (grayscott_amd/dist.py).
            tuned = gsd.share_tuning(sim, rows // world, cols, rank, world, device=red_dev, local_device=local_rank,
                                     place_candidates=12 if place is None else place)

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
        settles: with 18 ms here the first five 5-ms regions read 1-4 % low, profiles/archive/r03_sweeps.md section 6.)"""
        rate = 1.2e12 if cells_per_gpu >= (1 << 24) else 5.0e11        # cell-steps per second, a high guess
        n = (max(24, int(0.12 * rate / cells_per_gpu)) + 11) // 12 * 12
        run(sp, n)                                      # whole passes only, whatever the tuner fuses (2, 3 or 4 steps)
        return n

    # ---- the timed regions -------------------------------------------------------------------------------------
    # A rehearsal of the timed region first: the first torch.cuda.synchronize() / barrier / HIP-event calls of a
    # process may initialise things lazily, and a chip that idles for 2 ms runs its next ~10 ms at lower clocks
    # (tools/region_startup.py: a 5-pass region after 2 ms of idle reads 5 % low).
    # The W warm-up steps come first and the long untimed phase after them, directly before the timed regions: a W
    # that is not a whole number of passes (the driver's 5) ends in a single-step launch of another kernel, and the
    # chip, which sits on its power limit, answers that 0.7 ms change of load with a 20 ms dip -- the first four
    # 5-ms regions read 1-5 % low with W = 5 and not with W = 0, 4 or 8 (profiles/archive/r03_sweeps.md, section 6).
    with stage("timed", 600):
        timed_run(species, args.steps)
        run(species, args.warmup)
        extra_warm_steps = warm(species)
        untimed_before_first = species.steps_done       # tuning + rehearsal + W + warm: everything before region 0
        runs = repeated(species, args.steps, args.repeats)
    wall, event_ms, passes = median_run(runs)
    walls = [r[0] for r in runs]

    per_rank, rccl_ranks = [], None
    if world > 1:
        with stage("per-rank", 600):
            per_rank, rccl_ranks = per_rank_report(sim, species, args.steps, timed_run, world, local_rank, red_dev)

    kernel_name, _ = ctx.info()
    value = cells * args.steps / wall / 1e6
    steps_per_launch = args.steps / passes
    pmc = measured_counters(kernel_name, int(rows // world), cols, tuned)
    valu_insts, valu_how = scaled_valu_insts(pmc, tuned)

    def roofline_of(ms, n_passes, data="Species::new"):
        if data == "Species::new":
            return roofline_object(ms, n_passes, steps_per_launch, cells_per_gpu, pmc, valu_insts, valu_how, kernel_name)
        pmc_d = measured_counters(kernel_name, int(rows // world), cols, tuned, data)      # the profile of this input, if committed
        insts_d, how_d = scaled_valu_insts(pmc_d, tuned)
        return roofline_object(ms, n_passes, steps_per_launch, cells_per_gpu, pmc_d, insts_d, how_d, kernel_name)

    roofline = roofline_of(event_ms, passes)
    fused, developed, clocks, clocks_dev = None, None, None, None
    if with_extra:
        with stage("developed", 900):
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
                         "roofline": {k: v for k, v in roofline_of(ms_dev, p_dev, "developed").items()
                                      if k in ("bound", "achieved", "peak", "unit", "frac", "frac_source", "valu", "valu_source", "useful_valu",
                                               "hbm_physical", "traffic", "algorithmic_GBps", "launch_ms", "counters_source")}}
        unplaced = None
        if getattr(species, "placement", None):
            with stage("unplaced", 600):
                # the same launches on a Species::new whose planes lie where hipMalloc put them (placement switched off)
                sp_un = sim.make_species([rows, cols], place_candidates=0)
                sp_un.steps_done = 0
                run(sp_un, args.warmup)
                warm(sp_un)
                runs_un = repeated(sp_un, args.steps, min(3, args.repeats))
                unplaced = cells * args.steps / median_run(runs_un)[0] / 1e6
                for c in sp_un.u._pair + sp_un.v._pair:
                    c.destroy()
        with stage("energy", 900):
            # informational: shader clock, socket power and energy per cell-step while the same kernel runs
            # (rocm-smi samples next to an untimed repeat of >= 3 s; the VALU roof is priced at the nominal
            # 2.4 GHz, the chip sustains less on its power limit), on both inputs
            long_steps = max(args.steps, int(3.2 * value * 1e6 / cells) // 12 * 12)
            clocks = sample_clock_and_power(lambda: timed_run(species, long_steps), local_rank, cells * long_steps)
            clocks_dev = sample_clock_and_power(lambda: timed_run(sp_dev, long_steps), local_rank, cells * long_steps)
    if single and args.extra:
        with stage("fused", 600):
            fused = fused_flavour_leg(rows, cols, args.steps, args.warmup, local_rank)

    verified, single_step, peer_chain = None, None, None
    if verify:
        with stage("verify", 900):
            try:
                if single:
                    single_step, verified = verify_single_gpu(sim_s, sp_s, species, sp_dev_s, sp_dev, rows, cols, kernel_name,
                                                              extra_placements=(2 if args.extra else 1) if getattr(sp_s, "placement", None) else 0)
                else:
                    verified = verify_slab_chain(sim, species, rows, cols, rank, world, local_rank, args.rehearsal)
            except Exception as e:                      # the line is still worth printing
                verified = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 and not args.no_peer_chain:
        # the other ranks wait at the barrier (their GPUs idle but for rank 0's slabs)
        with stage("peer-chain", 900):
            if rank == 0:
                # in a thread of its own with a bound of its own: a second figure, never the line's fate -- if it does
                # not come back, the line says so and the other ranks are released all the same
                import threading

                box = {}

                def leg():
                    try:
                        box["out"] = peer_chain_leg(rows, cols, world, args.steps, tuned, args.rehearsal)
                    except Exception as e:
                        box["out"] = {"error": f"{type(e).__name__}: {e}"}

                th = threading.Thread(target=leg, daemon=True)
                th.start()
                th.join(float(os.environ.get("GS_BENCH_PEER_CHAIN_S", "300")))
                peer_chain = box.get("out", {"error": "the in-process chain did not finish within its bound"})
            barrier()

    result = {
        # BASELINE.json's metric, verbatim; `value` is its first quantity, the `roofline` object
        # carries the second
        "metric": "Mcells×steps/s and achieved HBM GB/s (% of roofline), 16384² f32 grid",
        "value": value,                                   # the median of `repeats` timed regions
        "unit": "Mcells×steps/s",
        "n_gpus": world,
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
            "cells_per_gpu": cells // world,
            "kernel": kernel_name,
            "tuned": {"rows_per_unit": tuned[0], "steps_per_pass": tuned[1], "cols_per_lane": tuned[2],
                      "share_taps": {1: "within lanes", 2: "off", 3: "across lanes"}.get(tuned[3])},
            "placement": placement_report(ctx, species, sp_dev, place, rows // world, cols),
            "launches_per_pass": 1 if single else 2,
            "partition": "single GPU" if single else f"{world} row slabs, RCCL send/recv ghost rows",
        },
        "roofline": roofline,
        # which HIP runtime and RCCL the library is bound to in this process (gs_runtime_info), and what carried the
        # barriers: torch is imported first, so both are the copies the torch wheel bundles
        "runtime": runtime,
    }
    if per_rank:
        result["rccl_ranks"] = rccl_ranks
        result["ranks"] = per_rank
    if clocks:
        add_clocks(roofline, result, clocks)
    if single_step and roofline.get("hbm_physical") and single_step.get("frac_of_8TBps"):
        # the marching kernel's physical HBM rate against what the HBM-bound single-step kernel reads on planes placed the
        # same way: off the power cap and within a fifth of it, the launch is held back by where its planes lie
        # (profiles/r05_cross_lane.md, section 4), whatever the VALU fraction says
        ratio = roofline["hbm_physical"] / single_step["frac_of_8TBps"]
        roofline["hbm_physical_over_single_step_leg"] = ratio
        if roofline["bound"] == "valu-issue" and ratio >= 0.8:
            roofline["bound"] = "hbm of these planes"
    if with_extra and unplaced is not None:
        result["value_unplaced"] = unplaced             # Species::new, planes as hipMalloc hands them out (not the default)
    if developed is not None:
        result["value_developed_pattern"] = developed["value"]
        result["developed_pattern"] = developed
        if clocks_dev:
            cap_d, pw_d = clocks_dev.get("power_cap_W"), clocks_dev.get("energy_W") or clocks_dev["power_W"]
            if cap_d and pw_d and pw_d >= 0.96 * cap_d and developed["roofline"]["bound"] == "valu-issue":
                developed["roofline"]["bound"] = "power-capped valu"
            developed["roofline"]["frac_at_sustained_clock"] = developed["roofline"]["frac"] / (clocks_dev["sclk_MHz"] / NOMINAL_SCLK_MHZ)
            developed["sclk_MHz_under_load"] = clocks_dev["sclk_MHz"]
            developed["socket_power_W_under_load"] = clocks_dev["power_W"]
            developed["energy_pJ_per_cell_step"] = clocks_dev.get("energy_pJ_per_cell_step")
    for key, leg in (("fused_flavour", fused), ("single_step", single_step), ("verified", verified), ("peer_chain", peer_chain)):
        if leg is not None:
            result[key] = leg
    if ctx.stats().get("window_fallbacks"):
        result["window_fallbacks"] = ctx.stats()["window_fallbacks"]     # a timing that contains one measured a stall
    if rank == 0 and single and not args.no_cpu_baseline:
        with stage("cpu_baseline", 600):
            result["cpu_baseline"] = cpu_baseline()
    result["stage_seconds"] = stage_s
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False), flush=True)

    def failed(v):
        """A verification record that does not say "equal": missing, an error, or a mismatch (nested records too)."""
        if v is None or "error" in v or v.get("equal") is False:
            return True
        return any(isinstance(x, dict) and ("equal" in x or "error" in x) and failed(x) for x in v.values())

    # (rank 0 holds the records; the other ranks of a chain report through it)
    bad = rank == 0 and verify and (failed(verified) or (peer_chain is not None and "verified" in peer_chain and failed(peer_chain["verified"])))
    with wd.stage("teardown", 120):
        if sim_s is not None:
            sim_s.context.close()
        ctx.close()
        if world > 1:
            dist.destroy_process_group()
    if bad and rank == 0:
        print("bench.py: the timed planes do not equal the replay (see `verified` in the line)", file=sys.stderr)
    return 3 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
