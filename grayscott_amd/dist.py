"""Multi-process plumbing around the C ABI: one process per GPU, ``torch.distributed`` only
for the bootstrap and for barriers / timing reductions.  The per-step ghost-row exchange is
NOT here: it runs inside ``libgs_hip.so`` (RCCL ncclSend/ncclRecv on a side HIP stream,
``gs_api.cpp: push_halo``).

What is here:

* ``slab_range``         the row partition (mirrors ``gs_field_create``: slab k of S owns rows
                         ``[k*R//S, (k+1)*R//S)``), the multi-GPU analogue of the reference's
                         in-process ``SimulateCpu::split_grid`` (compute/shared/src/cpu.rs:111-154);
* ``bootstrap``          rank / world / local-rank discovery from the torchrun environment and
                         the out-of-band broadcast of rank 0's RCCL unique id;
* ``exchange_ghost_rows`` a statement of the exchange PROTOCOL over ``torch.distributed``
                         point-to-point ops (gloo on CPU tensors, nccl on GPU tensors).  The CPU
                         tests run it with world_size 2..3 to pin the protocol the library
                         implements: first owned row -> upper neighbour's bottom ghost, last
                         owned row -> lower neighbour's top ghost, chain ends untouched;
* ``gather_rows``        collect a row-distributed field on rank 0 (verification only).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

UNIQUE_ID_BYTES = 128


def slab_range(rows: int, n_slabs: int, k: int) -> Tuple[int, int]:
    """Rows ``[r0, r1)`` of the global grid owned by slab ``k`` of ``n_slabs``."""
    if not (0 <= k < n_slabs):
        raise ValueError(f"slab {k} out of range for {n_slabs} slabs")
    if rows < n_slabs:
        raise ValueError(f"{rows} rows cannot be split over {n_slabs} slabs")
    return k * rows // n_slabs, (k + 1) * rows // n_slabs


@dataclass
class RankInfo:
    rank: int
    world: int
    local_rank: int
    unique_id: Optional[bytes]


def env_rank() -> Tuple[int, int, int]:
    """(RANK, WORLD_SIZE, LOCAL_RANK) as torchrun exports them; (0, 1, 0) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def bootstrap(backend: Optional[str] = None, id_source: Optional[Callable[[], bytes]] = None,
              device: Optional[str] = None) -> RankInfo:
    """Initialise ``torch.distributed`` (if WORLD_SIZE > 1) and broadcast rank 0's unique id.

    ``id_source`` produces the 128-byte id on rank 0 (default: ``capi.get_unique_id``, i.e.
    ``ncclGetUniqueId``); ``backend`` defaults to nccl (= RCCL) when a GPU is visible, else gloo;
    ``device`` is where the broadcast buffer lives ("cuda" for nccl, "cpu" for gloo).
    """
    rank, world, local_rank = env_rank()
    if world == 1:
        return RankInfo(0, 1, local_rank, None)
    import torch
    import torch.distributed as dist

    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    if device is None:
        device = "cuda" if backend == "nccl" else "cpu"
    buf = torch.zeros(UNIQUE_ID_BYTES, dtype=torch.uint8, device=device)
    if rank == 0:
        if id_source is None:
            from . import capi

            id_source = capi.get_unique_id
        raw = id_source()
        if len(raw) != UNIQUE_ID_BYTES:
            raise ValueError(f"unique id must be {UNIQUE_ID_BYTES} bytes, got {len(raw)}")
        buf.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
    dist.broadcast(buf, src=0)
    return RankInfo(rank, world, local_rank, bytes(buf.cpu().numpy().tobytes()))


def share_tuning(sim, slab_rows: int, cols: int, rank: int, world: int, device: str = "cpu",
                 tune_steps: int = 400, local_device: int = 0, place_candidates: int = 0) -> Tuple[int, int, int, int]:
    """Give every process of a slab chain the same tuned kernel configuration.

    Multi-process contexts do not tune on line (a timing window would have to be collective).
    Rank 0 lets a throw-away single-slab context of the slab's shape run ``tune_steps`` steps, reads
    what ``gs_run`` chose (``gs_ctx_get_tuned``) and broadcasts the four integers; every rank hands them
    to its own context (``gs_ctx_set_tuned``).  They must agree: the exchange is ``fuse_steps`` rows deep.
    Returns (rows per unit, steps per pass, columns per lane, share_taps); zeros mean "nothing chosen, defaults".
    """
    from .simulation import HipArgs, Parameters, Simulation

    choice = [0, 0, 0, 0]
    if rank == 0:
        args = sim.context.args
        scratch = Simulation.new(sim.params, HipArgs(devices=[local_device], math=args.math, kernel=args.kernel,
                                                     fuse_steps=args.fuse_steps, boundary=args.boundary,
                                                     general_kernels=args.general_kernels, share_taps=args.share_taps))
        # (placed like the slabs it stands for: on badly placed planes the tuner measures the HBM, not the kernel)
        species = scratch.make_species([slab_rows, cols], place_candidates=place_candidates)
        for _ in range(8):                       # long calls wait for their tuning phases
            scratch.perform_steps(species, tune_steps)
            choice = list(scratch.context.get_tuned(slab_rows, cols))
            if choice[0] > 0:
                break
        del species
        scratch.context.close()
    if world > 1:
        import torch
        import torch.distributed as dist

        t = torch.tensor(choice, dtype=torch.int32, device=device)
        dist.broadcast(t, src=0)
        choice = [int(x) for x in t.cpu()]
    if choice[0] > 0:
        sim.context.set_tuned(slab_rows, cols, *choice)
    return tuple(choice)


def exchange_ghost_rows(planes: Sequence, rank: int, world: int) -> None:
    """Protocol statement: refresh the ghost rows of row-distributed planes.

    Each plane is a 2-D torch tensor ``[rows_local + 2, cols]`` whose row 0 / row -1 are the
    ghost rows above / below.  Rank r sends its first owned row (index 1) to rank r-1, which
    stores it in its bottom ghost row, and its last owned row (index -2) to rank r+1, which
    stores it in its top ghost row.  Rank 0 has no upper and rank world-1 no lower neighbour:
    their outer ghost rows are never written (global edges use naive's clipped window).
    """
    import torch.distributed as dist

    ops = []
    for p in planes:
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, p[1].contiguous(), rank - 1))
            ops.append(dist.P2POp(dist.irecv, p[0], rank - 1))
        if rank < world - 1:
            ops.append(dist.P2POp(dist.isend, p[-2].contiguous(), rank + 1))
            ops.append(dist.P2POp(dist.irecv, p[-1], rank + 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def gather_rows(local: np.ndarray, rank: int, world: int) -> Optional[np.ndarray]:
    """Concatenate row-distributed dense arrays on rank 0 (None elsewhere)."""
    if world == 1:
        return local
    import torch
    import torch.distributed as dist

    parts: List = [None] * world if rank == 0 else None
    dist.gather_object(np.ascontiguousarray(local), parts, dst=0)
    return np.concatenate(parts, axis=0) if rank == 0 else None


def max_over_ranks(value: float, world: int, device: str = "cpu") -> float:
    if world == 1:
        return value
    import torch
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
