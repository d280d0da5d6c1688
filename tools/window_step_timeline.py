#!/usr/bin/env python3
"""Where a STEP of the persistent window kernel goes, wave by wave (diagnostic build):

    python tools/ab_build.py winsteptrace -DGS_WIN_TRACE=2
    GS_HIP_LIBRARY=grayscott_amd/variants/libgs_hip_winsteptrace.so python tools/window_step_timeline.py ROWS COLS [steps=N]

Every wave of the first 256 workgroups stamps the 100 MHz real-time counter at four points of each of the launch's last
four steps (gs_window_kernel.h: GS_WIN_TRACE == 2): step begins, at the barrier, past the barrier, the neighbouring
waves' rows are in registers.  Prints the phases of a step over all waves, the spread of the waves of a workgroup at each
point, and the waves of one workgroup inside the grid one by one (SIMD = wave % 4)."""
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
    show = 100
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        if k == "steps":
            steps = int(v)
        elif k == "show":
            show = int(v)
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
    buf = np.zeros((256, 16, 4, 4), np.uint64)
    assert read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    print(f"grid {rows}x{cols}  kernel {sim.context.info()[0]}  {kw}: {steps} steps in {ms * 1e3:.1f} us = {ms * 1e3 / steps:.3f} us per step"
          f" = {rows * cols * steps / ms / 1e3:.0f} Mcells*steps/s")
    # the launch's last four steps are one super-step (steps % 4 == 0): steps 0..3 in order of (step & 3)
    assert steps % 4 == 0
    t = buf.astype(np.int64) * 0.01                       # [wg, wave, step, slot] in microseconds
    live = (buf > 0).all(axis=(2, 3))                     # [wg, wave]: waves that stamped everything (shared-difference steps)
    print(f"{int(live.sum())} waves of {int(live.any(axis=1).sum())} workgroups stamped all four steps")

    def pct(x):
        return " / ".join(f"{np.percentile(x, q):6.2f}" for q in (0, 10, 50, 90, 100))

    names = ["begin -> at the barrier", "waiting at the barrier", "rows above / below from LDS", "last two rows (to next begin)"]
    print("phases of a step over all waves, percentiles 0/10/50/90/100 [us]")
    for i in range(3):
        print(f"    {names[i]:32s} {pct((t[:, :, :, i + 1] - t[:, :, :, i])[live])}")
    print(f"    {names[3]:32s} {pct((t[:, :, 1:, 0] - t[:, :, :-1, 3])[live])}")
    print(f"    {'whole step (begin to begin)':32s} {pct((t[:, :, 1:, 0] - t[:, :, :-1, 0])[live])}")
    full = live.all(axis=1)                               # workgroups whose 16 waves all stamped
    print(f"spread over the 16 waves of a workgroup (max - min), {int(full.sum())} workgroups:")
    for i, nm in enumerate(["begin", "at the barrier", "past the barrier", "rows in registers"]):
        x = t[full][:, :, :, i]
        print(f"    {nm:32s} {pct(x.max(axis=1) - x.min(axis=1))}")
    x = t[full]
    print(f"    last wave at the barrier -> first wave past it   {pct(x[:, :, :, 2].min(axis=1) - x[:, :, :, 1].max(axis=1))}")
    wgs = np.nonzero(full)[0]
    if len(wgs):
        wg = int(wgs[min(show, len(wgs) - 1)]) if show < len(wgs) else int(wgs[len(wgs) // 2])
        t0 = t[wg, :, 0, 0].min()
        print(f"workgroup {wg}, times since its first wave began the super-step's first step [us]: begin | barrier | past | rows in")
        for w in range(16):
            print(f"    wave {w:2d} (SIMD {w % 4}): " + "   ".join(" ".join(f"{t[wg, w, s, i] - t0:6.2f}" for i in range(4)) for s in range(4)))
    sim.context.close()


if __name__ == "__main__":
    main()
