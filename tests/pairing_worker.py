"""Worker of tests/test_gpu_multiprocess.py::test_library_pairing_of_a_bench_rank_runs_on_one_gpu: started in a FRESH
interpreter (`python -c`), so that the order in which torch and libgs_hip.so come up is the order this file says.
Nothing heavy is imported at module level."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pairing_worker(out_dir, order, port):
    """One process, the libraries brought up in the given order, then everything a rank of `bench.py --gpus N` does with
    them at world size 1: torch's process group on the nccl backend (a real communicator: one all-reduce), the library's
    own one-rank communicator moving K-row messages on a high-priority stream (gs_rccl_selftest), steps of a context."""
    import json

    sys.path.insert(0, ROOT)
    assert "torch" not in sys.modules and "grayscott_amd" not in sys.modules, "the worker must start from a bare interpreter"
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("GS_RCCL_LIBRARY", None)
    result = {"order": order}
    seen = []
    try:
        def bring_up_torch():
            import torch
            import torch.distributed as dist

            torch.cuda.set_device(0)
            dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
            t = torch.ones(1024, device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()
            assert float(t.sum()) == 1024.0
            result["torch"] = torch.__version__

        def bring_up_library():
            from grayscott_amd import HipArgs, Parameters, Simulation, capi

            assert capi.device_count() >= 1
            sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
            sp = sim.make_species([64, 128])
            sim.perform_steps(sp, 3)
            sim.context.close()

        for what in order.split("+"):
            if what == "library":
                result["torch_imported_before_library"] = "torch" in sys.modules
            (bring_up_torch if what == "torch" else bring_up_library)()
            seen.append(what)
        import numpy as np

        import oracle
        from grayscott_amd import HipArgs, Parameters, Simulation, capi

        for floats in (1, 4 * 16384, 4 * 32768):
            capi.rccl_selftest(0, floats)
        result["runtime"] = capi.runtime_info(load_rccl=True)
        # a context handed a unique id, as every rank of a chain is (world = 1: no neighbour to talk to)
        sim = Simulation.new(Parameters(), HipArgs(devices=[0], rank=0, world=1, unique_id=capi.get_unique_id()))
        sp = sim.make_species([96, 300])
        sim.perform_steps(sp, 22)
        u0, v0 = oracle.init_species(96, 300)
        ref_u, ref_v = oracle.run(u0, v0, 22)
        in_u, in_v, _, _ = sp.in_out()
        result["bit_exact"] = bool(np.array_equal(in_u.make_scalar_view(sim.context).view(np.uint32), ref_u.view(np.uint32)) and
                                   np.array_equal(in_v.make_scalar_view(sim.context).view(np.uint32), ref_v.view(np.uint32)))
        sim.context.close()
        if "torch" in order:
            import torch.distributed as dist

            dist.barrier()
            dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 -- the parent reports it
        result["error"] = f"{type(e).__name__}: {e}"
    result["seen"] = seen
    json.dump(result, open(os.path.join(out_dir, "result.json"), "w"))
