#!/usr/bin/env python3
"""Where the waves of the persistent window kernel spend a launch (diagnostic build):

    python tools/ab_build.py winbudget -DGS_WIN_TRACE=4
    GS_HIP_LIBRARY=grayscott_amd/variants/libgs_hip_winbudget.so python tools/window_wave_budget.py ROWS COLS [steps=N]

Every wave adds up, in shader clocks, what it spends waiting for the waves above and below it inside the steps, between its
last step of a super-step and its apron, and in all (gs_window_kernel.h: GS_WIN_TRACE == 4).  The rest -- issuing its own
instructions, or held up by the other three waves of its SIMD -- is what is left."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    kw = {"devices": [0], "kernel": capi.GS_KERNEL_WINDOW}
    steps = 1000
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        if k == "steps":
            steps = int(v)
        else:
            kw[k] = int(v)
    lib = capi.load()
    read = lib.gs_debug_win_trace_read_strict
    read.restype = ctypes.c_int32
    read.argtypes = [ctypes.c_void_p]
    sim = Simulation.new(Parameters(), HipArgs(**kw))
    sp = sim.make_species([rows, cols])
    sim.perform_steps(sp, steps)
    before = np.zeros(1024 * 8 * 8, np.uint64)                   # (only the first 256 * 16 * 4 words are used)
    assert read(before.ctypes.data_as(ctypes.c_void_p)) == 0
    sim.context.timer_start()
    sim.prepare_steps(sp, steps)
    ms = sim.context.timer_stop()
    sim.context.sync()
    big = np.zeros(1024 * 8 * 8, np.uint64)
    assert read(big.ctypes.data_as(ctypes.c_void_p)) == 0
    after = big[:256 * 16 * 4].reshape(256, 16, 4)
    d = (after - before[:256 * 16 * 4].reshape(256, 16, 4)).astype(np.float64)
    print(f"grid {rows}x{cols}  kernel {sim.context.info()[0]}: {steps} steps in {ms * 1e3:.1f} us = {rows * cols * steps / ms / 1e3:.0f} Mcells*steps/s")
    live = d[:, :, 2] > 0
    tot = d[:, :, 2]
    clk = np.median(tot[live]) / (ms * 1e3)
    print(f"{int(live.sum())} waves; a wave's run: median {np.median(tot[live]) / clk:.1f} us of the launch's {ms * 1e3:.1f} (counter: {clk:.0f} ticks per us)")
    supers = steps // 4
    print("wave | waiting for neighbouring waves' rows | ring to apron | the rest (own issue, SIMD shared with three waves) | waits that waited, of", steps)
    for w in range(16):
        m = live[:, w]
        if not m.any():
            continue
        a, x, t, n = (np.median(d[:, w, i][m]) for i in (0, 1, 2, 3))
        print(f"  {w:2d} | {a / t * 100:5.1f} % = {a / clk / steps:5.2f} us per step | {x / t * 100:5.1f} % = {x / clk / max(supers - 1, 1):5.2f} us per exchange | "
              f"{(t - a - x) / t * 100:5.1f} % = {(t - a - x) / clk / steps:5.2f} us per step | {n:6.0f}")
    a, x, t = d[:, :, 0][live].sum(), d[:, :, 1][live].sum(), tot[live].sum()
    print(f"all waves: rows of neighbouring waves {a / t * 100:.1f} %, ring to apron {x / t * 100:.1f} %, the rest {(t - a - x) / t * 100:.1f} %")
    sim.context.close()


if __name__ == "__main__":
    main()
