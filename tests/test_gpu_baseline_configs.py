"""BASELINE configs 4 and 5 at their full sizes on ONE MI355X (288 GB holds them several times over).

    config 4: 32768 x 16384 f32 row-split across 2, then 4 slabs
    config 5: 65536 x 32768 f32 across 8 slabs

The slab chain runs exactly what those configs run on 2 / 4 / 8 GPUs minus xGMI: rank-local slabs of 2^28
(2^27) cells with 4 ghost rows, the boundary-band kernel on the halo stream, K-row exchanges every K steps
(device-to-device copies inside a process, ncclSend / ncclRecv pairs between processes -- here through the
shared-memory double of tests/cpp/shm_transport.cpp, because RCCL refuses two ranks on one device), the
interior kernel on the compute stream, the tuning handed over from a single slab of the slab's shape as
bench.py does it (grayscott_amd/dist.py: share_tuning), global row offsets beyond 2^31 elements.

Every word of U and V is compared ON THE DEVICE (torch views of gs_field_device_ptr: the planes are 2 and 8
GiB each) with a single-slab run of the same grid, which the other GPU tests tie to the oracle; the rows
around every seam and the grid's corners are also compared with the ORACLE itself on crops (a crop's inner
region, `steps` cells away from the crop's artificial edges, is exact whatever lies outside).  Data
everywhere: random U, V written through the device pointers (gs_field_mark_written).

The pool admits at most 6 processes on a GPU, so the 8-slab chain of config 5 runs in-process and as
4 processes x 2 slabs (both kinds of exchange in one chain); 8 processes cannot be started here.

Reference precedent for overlapping sub-grids: compute/shared/src/cpu.rs:111-154; arithmetic:
compute/naive/src/lib.rs:42-83.
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHUNK = 2048        # rows per block of generated data / per checksum
STEPS_A, STEPS_B, SINGLE = 19, 8, 2   # 19 = a 3-step pass + 4 full passes; 8 = 2 full passes; 2 single steps
TOTAL = STEPS_A + STEPS_B + SINGLE


class _DeviceArray:
    """A strided f32 device array for torch.as_tensor (__cuda_array_interface__, version 3)."""

    def __init__(self, address, rows, cols, pitch):
        self.__cuda_array_interface__ = {"shape": (rows, cols), "typestr": "<f4", "data": (address, False),
                                         "version": 3, "strides": (pitch * 4, 4)}


def plane_views(conc):
    """[(global row0, rows, torch view [rows, cols])] of a HipConcentration's local slabs (zero-copy)."""
    import torch

    cols = conc.shape()[1]
    out = []
    for address, pitch, row0, rows, device in conc.device_slabs():
        out.append((row0, rows, torch.as_tensor(_DeviceArray(address, rows, cols, pitch), device=f"cuda:{device}")))
    return out


def rows_view(conc, r0, r1):
    """torch view of global rows [r0, r1) -- they must lie in one local slab."""
    for row0, rows, view in plane_views(conc):
        if row0 <= r0 and r1 <= row0 + rows:
            return view[r0 - row0:r1 - row0]
    raise AssertionError(f"rows [{r0}, {r1}) straddle slabs")


def random_chunk(k, cols, species_index):
    """Block k (CHUNK rows) of the global random start: U ~ [0, 1), V ~ [0, 0.5), a function of k alone."""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(1_000_003 * (k + 1) + species_index)
    x = torch.rand((CHUNK, cols), generator=g, device="cuda", dtype=torch.float32)
    return x if species_index == 0 else x * 0.5


def fill_random(species, ctx):
    """Write the global random start into the local rows of `species`' input planes through their device
    pointers."""
    import torch

    in_u, in_v, _, _ = species.in_out()
    cols = in_u.shape()[1]
    for si, conc in enumerate((in_u, in_v)):
        for row0, rows, view in plane_views(conc):
            for k in range(row0 // CHUNK, (row0 + rows + CHUNK - 1) // CHUNK):
                lo, hi = max(k * CHUNK, row0), min((k + 1) * CHUNK, row0 + rows)
                view[lo - row0:hi - row0].copy_(random_chunk(k, cols, si)[lo - k * CHUNK:hi - k * CHUNK])
        torch.cuda.synchronize()
        conc.mark_written(ctx)


def run_schedule(sim, species):
    """The sequence every variant runs: a short pass + full passes, full passes again (no ghost refresh in
    between), then single steps."""
    sim.perform_steps(species, STEPS_A)
    sim.perform_steps(species, STEPS_B)
    for _ in range(SINGLE):
        sim.perform_step(species)
    sim.context.sync()      # gs_step only enqueues: the torch views below read on another stream


def expected_passes(k):
    """Passes of run_schedule on a chain whose full passes fuse k steps."""
    return sum((1 if n % k else 0) + n // k for n in (STEPS_A, STEPS_B)) + SINGLE


def checksums(conc):
    """Two wrapping int64 sums per CHUNK-row block of the local rows (every word counts; the second is
    position-weighted): {block index: (s1, s2)}."""
    import torch

    out = {}
    for row0, rows, view in plane_views(conc):
        assert row0 % CHUNK == 0 and rows % CHUNK == 0
        n = CHUNK * view.shape[1]
        w = (torch.arange(n, device=view.device, dtype=torch.int64) % 65521 + 1).reshape(CHUNK, -1)
        for k in range(rows // CHUNK):
            x = view[k * CHUNK:(k + 1) * CHUNK].contiguous().view(torch.int32).to(torch.int64)
            out[row0 // CHUNK + k] = (int(x.sum()), int((x * w).sum()))
    return out


def oracle_crops(rows, cols, seams):
    """Crops [R0, R1) x [C0, C1) around every seam (and the grid's top and bottom edges) at the left edge,
    in the middle and at the right edge of the grid."""
    half = 3 * TOTAL + 8
    crops = []
    for s in [0] + list(seams) + [rows]:
        R0, R1 = max(0, s - half), min(rows, s + half)
        for C0 in (0, (cols // 2 - 448) // 64 * 64 + 13, cols - 900):
            crops.append((R0, R1, C0, min(cols, C0 + 900)))
    return crops


def crop_to_host(conc, crop):
    """Dense host copy of a crop (it may straddle slabs)."""
    R0, R1, C0, C1 = crop
    out = np.empty((R1 - R0, C1 - C0), np.float32)
    for row0, rows, view in plane_views(conc):
        lo, hi = max(R0, row0), min(R1, row0 + rows)
        if lo < hi:
            out[lo - R0:hi - R0] = view[lo - row0:hi - row0, C0:C1].cpu().numpy()
    return out


def check_crops_against_oracle(start, got, rows, cols, what):
    """start / got: {crop: (U, V)} before and after TOTAL steps.  The oracle advances the crop as if it were a
    grid of its own; where a crop edge is not a grid edge the outermost TOTAL cells are not comparable."""
    import oracle
    from tests.helpers import assert_bits_equal

    for crop, (u0, v0) in start.items():
        R0, R1, C0, C1 = crop
        ru, rv = oracle.run(u0, v0, TOTAL, ftz=True)
        sl = (slice(TOTAL if R0 > 0 else 0, (R1 - R0) - (TOTAL if R1 < rows else 0)),
              slice(TOTAL if C0 > 0 else 0, (C1 - C0) - (TOTAL if C1 < cols else 0)))
        assert sl[0].stop - sl[0].start >= 8 and sl[1].stop - sl[1].start >= 8
        gu, gv = got[crop]
        assert_bits_equal(gu[sl], ru[sl], f"{what}: U of crop {crop} vs oracle")
        assert_bits_equal(gv[sl], rv[sl], f"{what}: V of crop {crop} vs oracle")


def _make_chain(rows, cols, devices, rank=0, world=1, unique_id=None):
    """A slab-chain context with the tuning of a single slab of the slab's shape, as bench.py sets it up."""
    from grayscott_amd import HipArgs, Parameters, Simulation
    from grayscott_amd import dist as gsd

    sim = Simulation.new(Parameters(), HipArgs(devices=devices, rank=rank, world=world, unique_id=unique_id))
    n_slabs = len(devices) * world
    tuned = gsd.share_tuning(sim, rows // n_slabs, cols, rank, world, device="cpu", tune_steps=400)
    return sim, tuned


def _release(sim, *species):
    for sp in species:
        for c in sp.u._pair + sp.v._pair:
            c.destroy()
    sim.context.close()


@pytest.mark.parametrize("rows,cols,n_slabs", [
    (32768, 16384, 2),       # config 4, "across 2"
    (32768, 16384, 4),       # config 4, "then 4"
    (65536, 32768, 8),       # config 5
])
def test_inprocess_chain_full_size_vs_single_slab_and_oracle(built, rows, cols, n_slabs):
    import torch

    from grayscott_amd import HipArgs, Parameters, Simulation

    seams = [k * rows // n_slabs for k in range(1, n_slabs)]
    crops = oracle_crops(rows, cols, seams)

    single = Simulation.new(Parameters(), HipArgs(devices=[0]))
    ref = single.make_species([rows, cols])
    fill_random(ref, single.context)
    start = {c: (crop_to_host(ref.in_out()[0], c), crop_to_host(ref.in_out()[1], c)) for c in crops}
    run_schedule(single, ref)
    label = single.context.info()[0]
    assert label.startswith("stream") or label.startswith("tb-k"), label

    chain, tuned = _make_chain(rows, cols, [0] * n_slabs)
    assert tuned[0] > 0 and 1 <= tuned[1] <= 4, tuned
    sp = chain.make_species([rows, cols])
    fill_random(sp, chain.context)
    for c in crops:   # same start in both (the generator is a function of the row block alone)
        assert np.array_equal(crop_to_host(sp.in_out()[0], c).view(np.uint32), start[c][0].view(np.uint32))
    before = chain.context.stats()
    run_schedule(chain, sp)
    st = chain.context.stats()
    # one blocking refresh per written plane and no other: not after the short pass, not between the calls,
    # not before the single steps (every pass exchanges ghost-depth rows)
    assert st["ghost_refreshes"] - before["ghost_refreshes"] == 2, (before, st)
    assert st["steps"] - before["steps"] == TOTAL and st["passes"] - before["passes"] == expected_passes(tuned[1]), (before, st)

    # every word, on the device
    torch.cuda.synchronize()
    for name, a, b in (("U", ref.in_out()[0], sp.in_out()[0]), ("V", ref.in_out()[1], sp.in_out()[1])):
        (_, _, whole), = plane_views(a)
        for row0, nrows, view in plane_views(b):
            same = torch.equal(view.view(torch.int32), whole[row0:row0 + nrows].view(torch.int32))
            assert same, f"{name}: slab at row {row0} of the {n_slabs}-slab chain differs from the single-slab run"
        assert bool(torch.isfinite(whole[::97]).all())
    # seams, edges and corners against the oracle
    got = {c: (crop_to_host(sp.in_out()[0], c), crop_to_host(sp.in_out()[1], c)) for c in crops}
    check_crops_against_oracle(start, got, rows, cols, f"{rows}x{cols} over {n_slabs} slabs")
    _release(chain, sp)
    _release(single, ref)


def fill_seeded(species, ctx):
    """A pattern-forming start written on the device, a function of the row block alone (so every context gets the
    same data): U = 1, V = 0, 12 x 12 seeds (U = 0.5, V = 0.25) on a 12-cell lattice with one seed per 40 000 cells
    on average, 1 % noise everywhere -- the start of tools/soak.py and bench.py's developed pattern."""
    import torch

    in_u, in_v, _, _ = species.in_out()
    cols = in_u.shape()[1]
    for si, conc in enumerate((in_u, in_v)):
        for row0, rows, view in plane_views(conc):
            for k in range(row0 // CHUNK, (row0 + rows + CHUNK - 1) // CHUNK):
                g = torch.Generator(device="cuda")
                g.manual_seed(7_000_003 * (k + 1))
                coarse = torch.rand((CHUNK // 12 + 1, cols // 12 + 1), generator=g, device="cuda") < 144.0 / 40000.0
                mask = coarse.repeat_interleave(12, 0).repeat_interleave(12, 1)[:CHUNK, :cols]
                g.manual_seed(9_000_011 * (k + 1) + si)
                noise = torch.rand((CHUNK, cols), generator=g, device="cuda", dtype=torch.float32) * 0.01
                base = torch.where(mask, 0.5, 1.0) if si == 0 else torch.where(mask, 0.25, 0.0)
                lo, hi = max(k * CHUNK, row0), min((k + 1) * CHUNK, row0 + rows)
                view[lo - row0:hi - row0].copy_((base.to(torch.float32) + noise)[lo - k * CHUNK:hi - k * CHUNK])
        torch.cuda.synchronize()
        conc.mark_written(ctx)


def test_two_slab_chain_2000_steps_of_a_developing_pattern(built):
    """BASELINE config 4's grid (32768 x 16384) over 2 slabs, a pattern-forming start, 2000 steps in uneven calls:
    the seam under chaotic dynamics -- one stale or misplaced ghost row anywhere in ~500 exchanges changes bits
    that the dynamics then spread -- against the single-slab run of the same grid, every word of U and V.
    Spec: compute/naive/src/lib.rs:42-83; overlapping sub-grids: compute/shared/src/cpu.rs:111-154."""
    import torch

    from grayscott_amd import HipArgs, Parameters, Simulation

    rows, cols, n_slabs = 32768, 16384, 2
    calls = (997, 1003)
    single = Simulation.new(Parameters(), HipArgs(devices=[0]))
    ref = single.make_species([rows, cols])
    fill_seeded(ref, single.context)
    for n in calls:
        single.perform_steps(ref, n)
    chain, tuned = _make_chain(rows, cols, [0] * n_slabs)
    sp = chain.make_species([rows, cols])
    fill_seeded(sp, chain.context)
    before = chain.context.stats()
    for n in calls:
        chain.perform_steps(sp, n)
    st = chain.context.stats()
    assert st["steps"] - before["steps"] == sum(calls)
    assert st["ghost_refreshes"] - before["ghost_refreshes"] == 2, (before, st)     # the two written planes, once
    torch.cuda.synchronize()
    for name, a, b in (("U", ref.in_out()[0], sp.in_out()[0]), ("V", ref.in_out()[1], sp.in_out()[1])):
        (_, _, whole), = plane_views(a)
        for row0, nrows, view in plane_views(b):
            assert torch.equal(view.view(torch.int32), whole[row0:row0 + nrows].view(torch.int32)), \
                f"{name}: slab at row {row0} differs from the single-slab run after {sum(calls)} steps"
        assert bool(torch.isfinite(whole[::61]).all())
    (_, _, v), = plane_views(ref.in_out()[1])
    assert float(v[16300:16460].max()) > 0.3           # the pattern is alive on the seam
    _release(chain, sp)
    _release(single, ref)


# ---- N processes over the transport double -------------------------------------------------------------

def _free_port():
    from tests.helpers import free_port

    return free_port()


def _worker(rank, world, port, rows, cols, local_slabs, out_dir, transport_lib):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", GS_RCCL_LIBRARY=transport_lib)
    import torch
    import torch.distributed as dist

    from grayscott_amd import HipArgs, Parameters, Simulation
    from grayscott_amd import dist as gsd

    info = gsd.bootstrap(backend="gloo", device="cpu")
    torch.cuda.set_device(0)
    n_slabs = world * local_slabs
    sim = Simulation.new(Parameters(), HipArgs(devices=[0] * local_slabs, rank=info.rank, world=info.world,
                                               unique_id=info.unique_id))
    tuned = gsd.share_tuning(sim, rows // n_slabs, cols, rank, world, device="cpu", tune_steps=400)
    sp = sim.make_species([rows, cols])
    r0, r1 = sp.in_out()[0].local_rows()
    assert (r0, r1) == (rank * rows // world, (rank + 1) * rows // world)
    fill_random(sp, sim.context)
    before = sim.context.stats()
    run_schedule(sim, sp)
    st = {k: v - before[k] for k, v in sim.context.stats().items()}
    st["expected_passes"] = expected_passes(tuned[1]) if tuned[1] else -1
    mine = {"tuned": tuned, "stats": st, "rows": (r0, r1),
            "sums": [checksums(sp.in_out()[0]), checksums(sp.in_out()[1])]}
    # the rows next to every seam this rank owns a side of, dense, for a word-by-word comparison on rank 0
    edge = {}
    for k in range(1, n_slabs):
        s = k * rows // n_slabs
        for lo, hi in ((s - 8, s), (s, s + 8)):
            if r0 <= lo and hi <= r1:
                edge[(lo, hi)] = [rows_view(sp.in_out()[i], lo, hi).cpu().numpy() for i in (0, 1)]
    mine["edge"] = edge
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank == 0:
        # the single-slab run of the same grid, in this process
        single = Simulation.new(Parameters(), HipArgs(devices=[0]))
        ref = single.make_species([rows, cols])
        fill_random(ref, single.context)
        seams = [k * rows // n_slabs for k in range(1, n_slabs)]
        crops = oracle_crops(rows, cols, seams)[3:6]          # around the first seam: left, middle, right
        start = {c: (crop_to_host(ref.in_out()[0], c), crop_to_host(ref.in_out()[1], c)) for c in crops}
        run_schedule(single, ref)
        want = [checksums(ref.in_out()[0]), checksums(ref.in_out()[1])]
        problems = []
        for r, part in enumerate(parts):
            if part["stats"]["ghost_refreshes"] != 2 or part["stats"]["steps"] != TOTAL or \
                    part["stats"]["passes"] != part["stats"]["expected_passes"]:
                problems.append(f"rank {r}: stats {part['stats']}")
            for i, name in enumerate("UV"):
                for block, sums in part["sums"][i].items():
                    if tuple(sums) != tuple(want[i][block]):
                        problems.append(f"rank {r}: {name} rows [{block * CHUNK}, {(block + 1) * CHUNK}) differ from the single-slab run")
                for (lo, hi), arrs in part["edge"].items():
                    if not np.array_equal(arrs[i].view(np.uint32), rows_view(ref.in_out()[i], lo, hi).cpu().numpy().view(np.uint32)):
                        problems.append(f"rank {r}: {name} rows [{lo}, {hi}) next to a seam differ")
        covered = sorted(b for part in parts for b in part["sums"][0])
        if covered != list(range(rows // CHUNK)):
            problems.append("the ranks' blocks do not cover the grid")
        got = {c: (crop_to_host(ref.in_out()[0], c), crop_to_host(ref.in_out()[1], c)) for c in crops}
        try:
            check_crops_against_oracle(start, got, rows, cols, "single-slab reference")
        except AssertionError as e:
            problems.append(str(e))
        open(os.path.join(out_dir, "result.txt"), "w").write("\n".join(problems) if problems else "ok")
        _release(single, ref)
    dist.barrier()
    _release(sim, sp)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,local_slabs,rows,cols", [
    (2, 1, 32768, 16384),        # config 4 across 2 processes
    (4, 1, 32768, 16384),        # config 4 across 4 processes
    (4, 2, 65536, 32768),        # config 5: 8 slabs as 4 processes x 2 (the pool admits 6 GPU processes)
])
def test_process_chain_full_size_vs_single_slab(tmp_path, built, world, local_slabs, rows, cols):
    import torch.multiprocessing as mp

    lib = _build_transport()
    mp.spawn(_worker, args=(world, _free_port(), rows, cols, local_slabs, str(tmp_path), lib), nprocs=world, join=True)
    assert open(tmp_path / "result.txt").read() == "ok"


def _build_transport():
    import subprocess

    from grayscott_amd import _build

    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libshm_transport.so")
    src = os.path.join(ROOT, "tests", "cpp", "shm_transport.cpp")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.run([_build.hipcc(), "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                        src, "-o", lib, "-lrt", "-lpthread"], check=True)
    return lib
