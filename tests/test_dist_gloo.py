"""N > 1 host path on CPU: world_size 2 and 3 over the gloo backend (no GPU needed).

Covers what runs OUTSIDE libgs_hip.so in a multi-process job -- the torchrun bootstrap
(rank discovery, broadcast of rank 0's 128-byte unique id), the row partition, result
gathering, the max-over-ranks timing reduction -- and pins the ghost-row exchange PROTOCOL
the library implements with RCCL (``gs_api.cpp: push_halo``): each rank keeps its row slab
plus ghost rows, steps it with the oracle as the stand-in for the kernel, exchanges boundary
rows with its chain neighbours, and the gathered result must be bit-identical to the
single-domain oracle.  (The same slab semantics run on a real GPU, through the C ABI, in
tests/test_gpu_parity.py::test_row_slabs_bit_identical_to_single.)
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    from tests.helpers import free_port

    return free_port()


def _worker(rank, world, port, rows, cols, steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    import oracle
    from grayscott_amd import dist as gsd

    torch.set_num_threads(1)
    fake_id = bytes(range(128))
    info = gsd.bootstrap(backend="gloo", id_source=lambda: fake_id)
    assert (info.rank, info.world) == (rank, world)
    assert info.unique_id == fake_id                      # every rank received rank 0's id

    r0, r1 = gsd.slab_range(rows, world, rank)
    n = r1 - r0
    top, bottom = rank > 0, rank < world - 1
    gu, gv = oracle.init_species(rows, cols)              # every rank can build the global init
    rng = np.random.default_rng(3)                        # plus noise so that every row matters
    gu = (gu * rng.random((rows, cols), dtype=np.float32)).astype(np.float32)
    gv = (gv + rng.random((rows, cols), dtype=np.float32) * np.float32(0.3)).astype(np.float32)

    # local planes with ghost rows: [n + 2, cols]; ghost contents start as garbage on purpose
    def local(g):
        t = torch.full((n + 2, cols), float("nan"), dtype=torch.float32)
        t[1:-1] = torch.from_numpy(g[r0:r1])
        return t

    u, v = local(gu), local(gv)
    gsd.exchange_ghost_rows([u, v], rank, world)          # gs_field_finalize
    p = oracle.default_params()
    for _ in range(steps):
        # the kernel stand-in: naive rule on the window [r0 - top, r1 + bottom) of the grid;
        # array edges coincide with global edges exactly where no neighbour exists
        lo, hi = (0 if top else 1), (n + 2 if bottom else n + 1)
        wu = np.ascontiguousarray(u[lo:hi].numpy())
        wv = np.ascontiguousarray(v[lo:hi].numpy())
        ou, ov = np.empty_like(wu), np.empty_like(wv)
        first = 1 if top else 0
        oracle.step_rows(wu, wv, ou, ov, p, first, first + n, ftz=True, nthreads=1)
        nu = torch.full_like(u, float("nan"))
        nv = torch.full_like(v, float("nan"))
        nu[1:-1] = torch.from_numpy(ou[first:first + n])
        nv[1:-1] = torch.from_numpy(ov[first:first + n])
        gsd.exchange_ghost_rows([nu, nv], rank, world)
        u, v = nu, nv
    # chain ends never receive: their outer ghost rows must still be untouched
    if not top:
        assert torch.isnan(u[0]).all()
    if not bottom:
        assert torch.isnan(u[-1]).all()

    full_u = gsd.gather_rows(u[1:-1].numpy(), rank, world)
    full_v = gsd.gather_rows(v[1:-1].numpy(), rank, world)
    worst = gsd.max_over_ranks(float(rank + 1), world)
    assert worst == float(world)
    if rank == 0:
        ref_u, ref_v = oracle.run(gu, gv, steps, ftz=True, nthreads=2)
        ok = (full_u.tobytes() == ref_u.tobytes()) and (full_v.tobytes() == ref_v.tobytes())
        open(os.path.join(out_dir, "ok" if ok else "mismatch"), "w").close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,cols,steps", [(2, 64, 96, 12), (3, 50, 40, 9), (2, 2, 17, 3)])
def test_ghost_row_protocol_over_gloo(tmp_path, world, rows, cols, steps):
    import oracle

    oracle.build()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, rows, cols, steps, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists(), "distributed result differs from the single-domain oracle"


def test_slab_range_partition():
    from grayscott_amd.dist import slab_range

    for rows, s in ((16384, 1), (32768, 2), (32768, 4), (65536, 8), (1080, 7), (5, 5)):
        edges = [slab_range(rows, s, k) for k in range(s)]
        assert edges[0][0] == 0 and edges[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        assert all(b > a for a, b in edges)
    assert slab_range(65536, 8, 3) == (24576, 32768)   # 2^28 cells per GPU at 32768 columns
    with pytest.raises(ValueError):
        slab_range(3, 4, 0)


def test_bootstrap_single_process(monkeypatch):
    from grayscott_amd import dist as gsd

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    info = gsd.bootstrap()
    assert (info.rank, info.world, info.unique_id) == (0, 1, None)
