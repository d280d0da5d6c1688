"""The multi-process leg (one process per rank, RCCL ncclSend/ncclRecv inside libgs_hip.so).

A 1-GPU box cannot give every rank its own GPU, so both ranks are put on device 0; RCCL may
refuse that ("duplicate GPU"), in which case the test is skipped and the in-process slab tests
(tests/test_gpu_parity.py::test_row_slabs_*) remain the coverage of the exchange schedule.
With >= 2 visible GPUs each rank takes its own device.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows, cols, steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from grayscott_amd import GsError, HipArgs, Parameters, Simulation, capi
    from grayscott_amd import dist as gsd

    info = gsd.bootstrap(backend="gloo", device="cpu")     # unique id travels over gloo
    ndev = capi.device_count()
    device = rank if ndev >= world else 0
    try:
        sim = Simulation.new(Parameters(), HipArgs(devices=[device], rank=info.rank, world=info.world,
                                                   unique_id=info.unique_id))
    except GsError as e:
        open(os.path.join(out_dir, f"skip{rank}"), "w").write(str(e))
        return
    species = sim.make_species([rows, cols])
    r0, r1 = gsd.slab_range(rows, world, rank)
    assert species.u.in_out()[0].local_rows() == (r0, r1)
    sim.perform_steps(species, steps)          # fused passes + K-row RCCL exchanges
    for _ in range(3):
        sim.perform_step(species)              # single steps + 1-row exchanges
    in_u, in_v, _, _ = species.in_out()
    u = gsd.gather_rows(in_u.make_scalar_view(sim.context), rank, world)
    v = gsd.gather_rows(in_v.make_scalar_view(sim.context), rank, world)
    if rank == 0:
        np.save(os.path.join(out_dir, "u.npy"), u)
        np.save(os.path.join(out_dir, "v.npy"), v)
    dist.barrier()
    sim.context.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_ranks_match_oracle(tmp_path, built, world):
    import oracle

    rows, cols, steps = 96, 300, 22
    mp.spawn(_worker, args=(world, _free_port(), rows, cols, steps, str(tmp_path)), nprocs=world, join=True)
    skips = [p for p in os.listdir(tmp_path) if p.startswith("skip")]
    if skips:
        pytest.skip("RCCL refused the rank layout on this box: " + open(tmp_path / skips[0]).read()[:200])
    u0, v0 = oracle.init_species(rows, cols)
    ref_u, ref_v = oracle.run(u0, v0, steps + 3)
    assert np.load(tmp_path / "u.npy").tobytes() == ref_u.tobytes()
    assert np.load(tmp_path / "v.npy").tobytes() == ref_v.tobytes()
