"""The legs of bench.py's single-GPU line beside the timed regions: the CPU baseline (the only user of oracle/ outside
tests and smoke()), the committed PMC counters, the developed-pattern input, clock / power / energy sampling, the
single-step HBM leg with the in-run replay that proves the timed launches did the work."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .harness import (BYTES_PER_CELL_STEP, HBM_COPY_CEILING_GBS, HBM_PEAK_GBS, NOMINAL_SCLK_MHZ, USEFUL_VALU_PER_CELL_STEP,
                      USEFUL_VALU_PER_CELL_STEP_SHARED, USEFUL_VALU_PER_CELL_STEP_SHARED_ACROSS, VALU_PEAK_TLANEOPS, usable_cpus)


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


def measured_counters(kernel_name: str, rows: int, cols: int, tuned, data: str = "Species::new"):
    """Per-launch PMC figures of the committed rocprofv3 profile of this kernel on this grid
    (profiles/counters.json, a list written by tools/summarize_profile.py from separate --pmc passes; every
    entry names the layout it was measured with):
    {"traffic": HBM bytes, "valu_insts": SQ_INSTS_VALU wave-instructions, "launch_ms": rocprofv3's average
    launch duration, "rows_per_unit", "cols_per_lane", "steps_per_pass", "source"}; {} when no profile of
    this kernel on this grid is committed."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        with open(path) as f:
            entries = json.load(f)
    except (OSError, ValueError):
        return {}
    label = kernel_name.split("@")[0]
    best, rank = {}, -1
    for e in entries if isinstance(entries, list) else []:
        if e.get("kernel") == label and e.get("rows") == rows and e.get("cols") == cols:
            # the profile of this input and this layout first; then this layout; then this input
            r = 2 * (e.get("rows_per_unit") == tuned[0]) + (e.get("input", "Species::new") == data)
            if r > rank:
                best, rank = e, r
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
    12 x 12 seed (U = 0.5, V = 0.25) per 40 000 cells and 1 % noise (tools/pattern_rate.py, profiles/archive/r01_soak.md:
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


def upload_species(sim, u0, v0, place_candidates=None):
    """A Species of `sim`'s context whose input planes hold (u0, v0); `steps_done` counts what it has run.
    `place_candidates`: as make_species (None = the library's default: large Species are placed by measurement)."""
    ctx = sim.context
    species = sim.make_species(list(u0.shape), place_candidates=place_candidates)
    in_u, in_v, _, _ = species.in_out()
    in_u.upload(ctx, u0)
    in_v.upload(ctx, v0)
    species.steps_done = 0
    return species


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


def verify_single_gpu(sim_s, sp_s, species, sp_dev_s, sp_dev, rows, cols, timed_kernel, extra_placements=0):
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
    for _ in range(extra_placements):
        extra = sim_s.make_species([rows, cols], place_candidates=0)     # planes as hipMalloc hands them out
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
    if getattr(sp_s, "placement", None):
        first, best = sp_s.placement
        probes, drawn = ctx_s.place_stats()
        single_step["placement"] = {"how": "gs_fields_place (the library's default for Species of >= 2^26 cells): U's and V's planes in "
                                           "blocks of different physical regions; probe = a pass that reads and writes back a slot's two planes",
                                    "first_blocks_probe_ms": first, "chosen_blocks_probe_ms": best,
                                    "probes": probes, "extra_blocks_drawn": drawn}
    if extra_placements:
        # the same kernel on planes as hipMalloc hands them out (what a caller gets with --hip-place-candidates 0)
        single_step["unplaced_frac_of_8TBps"] = [round(x * 1e6 * BYTES_PER_CELL_STEP / 1e9 / HBM_PEAK_GBS, 4) for x in placements[1:]]
    single_step["by_plane_placement"] = [round(x) for x in placements]
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


def useful_valu_per_cell_step(kernel_name: str) -> int:
    """Arithmetic instructions per cell-step of the kernel's form of the reference's update, each operation one
    instruction: 53 as the reference writes it (compute/naive/src/lib.rs:63-79), 46 with full difference sharing at 2
    columns per lane (the `.ds` variants: the N / NW / NE taps are the negated S / SE / SW taps of the row above), 41 when
    the differences that cross a lane boundary are also formed once (the `.dx` variants)."""
    if ".dx" in kernel_name:
        return USEFUL_VALU_PER_CELL_STEP_SHARED_ACROSS
    return USEFUL_VALU_PER_CELL_STEP_SHARED if ".ds" in kernel_name else USEFUL_VALU_PER_CELL_STEP


def roofline_object(event_ms, passes, steps_per_launch, cells_per_gpu, pmc, valu_insts, valu_how, kernel_name):
    """The roofline object of one timed region: its own launch time (HIP events on the library's stream), the committed
    profile's counters (`pmc`: measured_counters(); `valu_insts`, `valu_how`: scaled_valu_insts())."""
    launch_s = event_ms * 1e-3 / passes
    algo_bytes = BYTES_PER_CELL_STEP * cells_per_gpu * steps_per_launch
    algo_gbs = algo_bytes / launch_s / 1e9
    traffic = pmc.get("traffic")
    hbm_physical = traffic / launch_s / 1e9 / HBM_PEAK_GBS if traffic else None
    valu_rate = valu_insts * 64 / launch_s / 1e12 if valu_insts else None
    useful = useful_valu_per_cell_step(kernel_name)
    useful_rate = useful * cells_per_gpu * steps_per_launch / launch_s / 1e12
    # Which roof binds: with K >= 3 steps fused per HBM pass the kernel moves ~16 B per cell for K
    # steps and is bound by VALU issue; a single-step pass is bound by HBM.
    valu_bound = steps_per_launch >= 3
    # the same rate priced in the REFERENCE's form of the update (53 operations per cell-step, one instruction each:
    # compute/naive/src/lib.rs:63-79): the share of the VALU roof a kernel that formed every tap afresh would need
    reference_rate = USEFUL_VALU_PER_CELL_STEP * cells_per_gpu * steps_per_launch / launch_s / 1e12
    if valu_bound:
        # `frac` (round 6 on): the reference-form rate against the plain-f32 issue roof.  It is computed from this run's
        # launch time alone, and it does not fall when the kernel finds a way to issue fewer instructions for the same
        # update (the issued fraction `valu` did: 0.80 -> 0.70 while the rate rose 11 %).  `valu` (issued, PMC
        # SQ_INSTS_VALU x 64 of the committed profile of this layout) and `useful_valu` (the arithmetic of the
        # kernel's own form of the update) stand beside it.
        achieved = reference_rate
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
        "frac_definition": ("53 operations per cell-step (the reference's form of the update, compute/naive/src/lib.rs:63-79) x cells x "
                            "steps per launch / launch_ms / peak (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz)") if valu_bound else
                           "HBM bytes per launch / launch_ms / 8 TB/s",
        "frac_source": "this run's launch time (HIP events on the library's stream); no counters needed"
                       if valu_bound else ("PMC traffic" if traffic else "algorithmic bytes"),
        "valu_source": (valu_how if valu_rate else "no profile of this layout committed") if valu_bound else None,
        "valu": valu_rate / VALU_PEAK_TLANEOPS if valu_rate else None,
        "useful_valu": useful_rate / VALU_PEAK_TLANEOPS,
        "useful_valu_per_cell_step": useful,
        "reference_form_valu": reference_rate / VALU_PEAK_TLANEOPS,          # (= frac when the VALU roof binds)
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


def add_clocks(roofline, result, clocks):
    """Fold a sample_clock_and_power() result into the roofline object (and name the bound from it)."""
    roofline["sclk_MHz_under_load"] = clocks["sclk_MHz"]
    roofline["socket_power_W_under_load"] = clocks["power_W"]
    roofline["power_cap_W"] = clocks.get("power_cap_W")
    if roofline["bound"] == "valu-issue":
        # the VALU roof at the clock the board sustains under this kernel (it sits on its power cap), next to the nominal one
        rel = clocks["sclk_MHz"] / NOMINAL_SCLK_MHZ
        roofline["peak_at_sustained_clock"] = roofline["peak"] * rel
        roofline["frac_at_sustained_clock"] = roofline["frac"] / rel
        if roofline["valu"]:
            roofline["valu_at_sustained_clock"] = roofline["valu"] / rel
    if clocks.get("energy_pJ_per_cell_step"):
        result["energy_pJ_per_cell_step"] = clocks["energy_pJ_per_cell_step"]
        roofline["energy_W_from_counter"] = clocks.get("energy_W")
    cap, pw = clocks.get("power_cap_W"), clocks.get("energy_W") or clocks["power_W"]
    if cap and pw and pw >= 0.96 * cap and roofline["bound"] == "valu-issue":
        # the package sits on its power limit: what a faster instruction stream gains, the clock gives back
        roofline["bound"] = "power-capped valu"


def fused_flavour_leg(rows, cols, steps, warmup, device):
    """Informational: the fused-tap flavour (GS_MATH_FUSED: bit-identical wherever no sub-normal intermediate
    occurs, |diff| <= 1e-37 elsewhere -- inside north_star's 1e-5 tolerance) on the same grid."""
    from grayscott_amd import HipArgs, Parameters, Simulation, capi

    sim_c = Simulation.new(Parameters(), HipArgs(devices=[device], math=capi.GS_MATH_FUSED))
    species_c = sim_c.make_species([rows, cols])
    sim_c.perform_steps(species_c, max(warmup, 400))
    tc = time.perf_counter()
    sim_c.perform_steps(species_c, steps)
    tc = time.perf_counter() - tc
    out = {"kernel": sim_c.context.info()[0], "value": rows * cols * steps / tc / 1e6}
    sim_c.context.close()
    return out
