#!/usr/bin/env python3
"""A super-step of the persistent window kernel, wave by wave (diagnostic build):

    python tools/ab_build.py winwavetrace -DGS_WIN_TRACE=3
    GS_HIP_LIBRARY=grayscott_amd/variants/libgs_hip_winwavetrace.so python tools/window_wave_timeline.py ROWS COLS [steps=N] [show=WG]

Every wave of the first 256 workgroups stamps the 100 MHz real-time counter at three points of each of its last four
super-steps (gs_window_kernel.h: GS_WIN_TRACE == 3): the super-step begins, its steps are done (the ring stores follow), the
apron is in; and leaves the number of polls the apron took.  Prints those phases over all waves, by position of the wave in
its window, and one workgroup wave by wave (SIMD = wave % 4)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    kw = {"devices": [0], "kernel": capi.GS_KERNEL_WINDOW}
    steps, show = 404, None
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
    # the launch's last super-step has no exchange: super-steps 0 .. 2 of the four are whole
    t = buf[:, :, :, :3].astype(np.int64) * 0.01         # [wg, wave, super-step, begin / steps done / apron in] in microseconds
    polls = buf[:, :, :3, 3].astype(np.int64)
    live = (buf[:, :, :3, :3] > 0).all(axis=(2, 3))      # [wg, wave]

    def pct(x):
        return " / ".join(f"{np.percentile(x, q):6.2f}" for q in (0, 10, 50, 90, 100))

    print(f"{int(live.sum())} waves of {int(live.any(axis=1).sum())} workgroups; percentiles 0/10/50/90/100 [us]")
    print(f"    the K steps                      {pct((t[:, :, :3, 1] - t[:, :, :3, 0])[live])}")
    print(f"    ring stores, polls until apron   {pct((t[:, :, :3, 2] - t[:, :, :3, 1])[live])}")
    print(f"    whole super-step (begin to begin) {pct((t[:, :, 1:3, 0] - t[:, :, 0:2, 0])[live])}")
    print(f"    polls per exchange: " + ", ".join(f"{n}: {int((polls[live] == n).sum())}" for n in range(1, 6)) + f", more: {int((polls[live] > 5).sum())}")
    print("by the wave's place in its window (median): steps | exchange | polls | begin after the workgroup's first wave")
    for w in range(16):
        m = live[:, w]
        if not m.any():
            continue
        first = np.where(live[:, :, None], t[:, :, :3, 0], np.inf).min(axis=1)       # [wg, super-step]
        print(f"    wave {w:2d}: {np.median((t[:, w, :3, 1] - t[:, w, :3, 0])[m]):6.2f} | {np.median((t[:, w, :3, 2] - t[:, w, :3, 1])[m]):6.2f} | "
              f"{np.mean(polls[:, w][m]):5.2f} | {np.median((t[:, w, :3, 0] - first)[m]):6.2f}")
    full = np.nonzero(live.all(axis=1))[0]
    if len(full):
        wg = show if show is not None else int(full[len(full) // 2])
        t0 = t[wg, :, 0, 0].min()
        print(f"workgroup {wg}, times since its first wave began the first of the super-steps [us]: begin  steps done  apron in (polls)")
        for w in range(16):
            print(f"    wave {w:2d} (SIMD {w % 4}): " + "    ".join(
                f"{t[wg, w, q, 0] - t0:6.2f} {t[wg, w, q, 1] - t0:6.2f} {t[wg, w, q, 2] - t0:6.2f} ({polls[wg, w, q]})" for q in range(3)))
    sim.context.close()


if __name__ == "__main__":
    main()
