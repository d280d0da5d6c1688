#!/usr/bin/env python3
"""Does the ghost-row exchange get onto the chip while the interior kernel of a slab holds every wave slot?  (VERDICT round 5,
"next round" item 6; design row SURVEY.md section 8(e); precedent compute/shared/src/cpu.rs:111-154.)

One GPU is all this pool gives a builder, and RCCL refuses two ranks on one device -- but what decides whether a pass hides
its exchange is visible on one GPU: a ONE-rank communicator sends the ghost message of a 16384-column slab (4 rows x 2 species
x 2 directions = 4 messages of 65536 f32 in one group) to itself on a high-priority stream, created exactly like a slab's halo
stream, (a) on an idle chip and (b) while the marching kernel of a 2^28-cell slab runs on the compute stream of a context --
multi-round launches that refill every wave slot the moment it frees.  Reported per exchange: host time from the first
enqueue to the end of the wait (what a pass's boundary band has to hide), device time between events around the exchange
on its stream, and the same for device-to-device copies of the same bytes (the in-process chain's route, hipMemcpyAsync
here, hipMemcpyPeerAsync between GPUs).  Run it once with and once without --torch-first: the two library pairings a
process can have (gs_hip.h: gs_runtime_info).

    python tools/rccl_under_load.py [--torch-first] [--rows 16384 --cols 16384 --reps 40]
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--torch-first", action="store_true")
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--cols", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    if a.torch_first:
        import torch

        torch.cuda.init()
    from grayscott_amd import HipArgs, Parameters, Simulation, capi

    lib = capi.load()
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    sp = sim.make_species([a.rows, a.cols])
    sim.perform_steps(sp, 2400)                       # on-line tuning done: what follows are the production launches
    ctx = sim.context
    ctx.timer_start()
    sim.prepare_steps(sp, 400)
    pass_ms = ctx.timer_stop() / 100                  # one pass = 4 steps
    ctx.sync()
    floats = 4 * a.cols                               # 4 ghost rows of one species, one direction
    out = {"runtime": capi.runtime_info(load_rccl=True), "torch_first": a.torch_first, "grid": [a.rows, a.cols],
           "kernel": ctx.info()[0], "interior_pass_ms": pass_ms, "message_floats": floats, "messages_per_exchange": 4, "routes": {}}
    for mode, name in ((0, "rccl send/recv to self, one group"), (1, "device-to-device copies")):
        probe = ctypes.c_void_p()
        capi.check(lib.gs_debug_exchange_probe_create(0, mode, 4, floats, ctypes.byref(probe)))
        host, dev = ctypes.c_float(0), ctypes.c_float(0)

        def exchange():
            capi.check(lib.gs_debug_exchange_probe_run(probe, ctypes.byref(host), ctypes.byref(dev)))
            return float(host.value), float(dev.value)

        for _ in range(5):
            exchange()
        idle = [exchange() for _ in range(a.reps)]
        # under load: ~1 s of interior passes enqueued, the exchanges issued while they run (each waits for its own end only)
        steps = int(1.2e3 / pass_ms) * 4
        sim.prepare_steps(sp, steps)
        time.sleep(0.05)                              # the queue is running
        t0 = time.perf_counter()
        loaded = []
        while len(loaded) < a.reps and time.perf_counter() - t0 < 0.9:
            loaded.append(exchange())
            time.sleep(0.002)
        still_running = time.perf_counter() - t0 < steps / 4 * pass_ms * 1e-3 - 0.05
        ctx.sync()
        capi.check(lib.gs_debug_exchange_probe_destroy(probe))

        def med(rows, i):
            return statistics.median(r[i] for r in rows)

        def worst(rows, i):
            return max(r[i] for r in rows)

        out["routes"][name] = {
            "idle_chip": {"host_ms": med(idle, 0), "device_ms": med(idle, 1), "host_ms_max": worst(idle, 0)},
            "under_interior_kernel": {"host_ms": med(loaded, 0), "device_ms": med(loaded, 1), "host_ms_max": worst(loaded, 0),
                                      "device_ms_max": worst(loaded, 1), "exchanges": len(loaded),
                                      "interior_still_running_at_the_last": still_running},
            # an exchange that had to wait for the interior launch to drain would take about one pass
            "host_ms_over_interior_pass": med(loaded, 0) / pass_ms,
        }
    print(json.dumps(out), flush=True)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
