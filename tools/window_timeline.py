#!/usr/bin/env python3
"""Where a super-step of the persistent window kernel goes (diagnostic build):

    python tools/ab_build.py wintrace -DGS_WIN_TRACE=1
    GS_HIP_LIBRARY=grayscott_amd/variants/libgs_hip_wintrace.so python tools/window_timeline.py ROWS COLS [key=value ...]

Wave 0 of every workgroup stamps the 100 MHz real-time counter at seven points of each of its last 8 super-steps
(gs_window_kernel.h: GS_WIN_TRACE).  Prints, per kind of window, the phases of a super-step: the K steps, ring
stores + drain, barrier, flag + poll (= waiting for the slowest neighbour), barrier, apron loads."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    kw = {"devices": [0], "kernel": capi.GS_KERNEL_WINDOW}
    steps = 404
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        if k == "steps":
            steps = int(v)
        else:
            kw[k] = int(v)
    sim = Simulation.new(Parameters(), HipArgs(**kw))
    sp = sim.make_species([rows, cols])
    sim.perform_steps(sp, steps)
    sim.context.timer_start()
    sim.prepare_steps(sp, steps)
    ms = sim.context.timer_stop()
    sim.context.sync()
    lib = capi.load()
    read = lib.gs_debug_win_trace_read_strict
    read.restype = ctypes.c_int32
    read.argtypes = [ctypes.c_void_p]
    buf = np.zeros((1024, 8, 8), np.uint64)
    assert read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    print(f"grid {rows}x{cols}  kernel {sim.context.info()[0]}  {kw}: {steps} steps in {ms * 1e3:.1f} us = {ms * 1e3 / steps:.3f} us per step"
          f" = {rows * cols * steps / ms / 1e3:.0f} Mcells*steps/s")
    t = buf[:, :, :7].astype(np.int64) * 0.01          # microseconds
    kind = buf[:, :, 7].astype(np.int64)
    live = (buf[:, :, 0] > 0) & (buf[:, :, 6] > 0)
    names = ["steps", "ring stores + drain", "barrier", "flag + poll", "barrier", "apron loads"]
    kinds = {0: "interior", 1: "general", 2: "left edge", 3: "right edge", 4: "top / bottom", 5: "left corner", 6: "right corner", 7: "masked interior"}

    def pct(x):
        return " / ".join(f"{np.percentile(x, q):6.2f}" for q in (0, 10, 50, 90, 100))

    for kd in sorted(set(kind[live].tolist())):
        m = live & (kind == kd)
        print(f"--- {kinds.get(kd, kd)} windows: {int(m.any(axis=1).sum())} workgroups, {int(m.sum())} super-steps; percentiles 0/10/50/90/100 [us]")
        for i, nm in enumerate(names):
            print(f"    {nm:22s} {pct((t[:, :, i + 1] - t[:, :, i])[m])}")
        print(f"    {'whole super-step':22s} {pct((t[:, :, 6] - t[:, :, 0])[m])}")
    m = live
    print(f"all windows: super-step {pct((t[:, :, 6] - t[:, :, 0])[m])}; start-to-start period {pct((t[:, 1:, 0] - t[:, :-1, 0])[live[:, 1:] & live[:, :-1]])}")
    sim.context.close()


if __name__ == "__main__":
    main()
