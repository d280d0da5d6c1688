"""The N > 1 side of bench.py: row-block checksums of planes that live on different GPUs, the whole-grid replay on
rank 0 that proves the seams, and the in-process peer-copy chain -- the route the reference's
single-process binaries would take to several GPUs (rust/compute_hip: --hip-devices 0,1,...)."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def per_rank_report(sim, species, steps, timed_run, world, local_rank, red_dev):
    """Per rank: its own launch time, and -- in an untimed repeat with HIP events on the halo and compute streams
    (gs_ctx_set_pass_timing) -- whether the boundary band + ghost-row exchange hid behind the interior kernel.  Then what
    RCCL itself says about the communicator, and where every rank runs.  Returns (list of per-rank dicts, RCCL's rank
    count)."""
    import torch
    import torch.distributed as dist

    ctx = sim.context
    runs = sorted((timed_run(species, steps) for _ in range(3)), key=lambda r: r[0])
    _, my_ms, my_passes = runs[1]
    ctx.set_pass_timing(min(64, max(1, my_passes)))
    timed_run(species, steps)
    st = ctx.stats()
    ctx.set_pass_timing(0)
    tp = max(1, st["timed_passes"])
    mine = torch.tensor([my_ms / max(1, my_passes), st["halo_ms"] / tp, st["interior_ms"] / tp,
                         st["halo_exposed_ms"] / tp, float(st["timed_passes"])] +
                        [float(x) for x in ctx.comm_info()] + [float(local_rank)], dtype=torch.float64, device=red_dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    per_rank, comm = [], []
    for r in allr:
        x = [float(v) for v in r.cpu()]
        per_rank.append({"launch_ms": x[0], "halo_stream_ms_per_pass": x[1], "interior_ms_per_pass": x[2],
                         "halo_exposed_ms_per_pass": x[3], "timed_passes": int(x[4]),
                         "rccl_rank": int(x[6]), "rccl_device": int(x[7]), "local_rank": int(x[8])})
        comm.append(int(x[5]))
    return per_rank, comm[0]


def peer_chain_leg(rows, cols, world, steps, tuned, rehearsal):
    """Rank 0 alone, after the RCCL chain has finished and verified: the SAME grid as an in-process chain of `world`
    slabs over devices 0 .. world - 1, ghost rows moved by hipMemcpyPeerAsync on the halo streams, no RCCL -- the route
    the reference's single-process binaries take to several GPUs (`--hip-devices 0,1,...` of rust/compute_hip;
    precedent for one process driving overlapping sub-grids: compute/shared/src/cpu.rs:111-154).  Untimed by the
    contract (`value` is the RCCL chain's): a second scaling figure, and the one that stands if RCCL's first contact
    fails.  Checked against a single-GPU run of the whole grid (row-block checksums)."""
    import statistics

    from grayscott_amd import HipArgs, Parameters, Simulation

    devices = [0] * world if rehearsal else list(range(world))
    sim = Simulation.new(Parameters(), HipArgs(devices=devices))
    ctx = sim.context
    if tuned and tuned[0] > 0:
        ctx.set_tuned(rows // world, cols, *tuned)
    species = sim.make_species([rows, cols])
    done = 0

    def run(n):
        nonlocal done
        sim.perform_steps(species, n)
        done += n

    cells = rows * cols
    run(max(steps, 24))                                             # first touches, ghost refresh, clocks
    run((max(24, int(0.12 * 1.2e12 * world / cells)) + 11) // 12 * 12)
    rates = []
    for _ in range(5):
        ctx.sync()
        t0 = time.perf_counter()
        sim.prepare_steps(species, steps)
        ctx.sync()
        rates.append(cells * steps / (time.perf_counter() - t0) / 1e6)
        done += steps
    out = {"route": f"one process, {world} slabs on devices {devices}, ghost rows by hipMemcpyPeerAsync (no RCCL)",
           "value": statistics.median(rates), "unit": "Mcells×steps/s", "values": [round(r) for r in rates],
           "steps_per_region": steps, "kernel": ctx.info()[0]}
    ctx.sync()
    sums = []
    for conc in species.in_out()[:2]:
        for row0, nrows, view in conc.torch_views():
            sums.append((row0, nrows, range_checksums(view)))
    solo = Simulation.new(Parameters(), HipArgs(devices=[0]))
    whole = solo.make_species([rows, cols])
    solo.perform_steps(whole, done)
    views = [conc.torch_views()[0][2] for conc in whole.in_out()[:2]]
    per_plane = len(sums) // 2
    equal = all(range_checksums(views[i // per_plane][row0:row0 + nrows]) == s for i, (row0, nrows, s) in enumerate(sums))
    out["verified"] = {"against": "single-GPU run of the whole grid (row-block checksums of U and V)", "steps": done, "equal": equal}
    solo.context.close()
    ctx.close()
    return out
