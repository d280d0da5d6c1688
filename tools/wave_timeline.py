#!/usr/bin/env python3
"""Per-wave timeline of one pass of the temporally blocked kernel (diagnostic build).

    python tools/ab_build.py trace -DGS_TB_TRACE=1
    GS_HIP_LIBRARY=grayscott_amd/variants/libgs_hip_trace.so python tools/wave_timeline.py ROWS COLS [key=value ...]

key=value pairs are HipArgs fields (rows_per_block=36 cols_per_lane=2 fuse_steps=4 ...); without them the
context tunes itself first.  Every wave of the LAST pass leaves five timestamps of the 100 MHz real-time counter
(entry, tick 3 = first rows used, tick 2K = level pipeline full, tick nticks - 2K = last level-0 row taken,
exit), its hardware id and its unit (gs_march.h: GS_TB_TRACE).  Prints where the time of a pass goes:
the span of the launch against the pass period, when waves start and end, how long the fill / steady / drain
phases of a unit take, and how many waves are resident over time.
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation, capi  # noqa: E402


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    kw = {"devices": [0]}
    developed = False
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        if k == "developed":            # developed=1: bench.py's developed spot pattern instead of Species::new
            developed = int(v) != 0
            continue
        kw[k] = int(v)
    sim = Simulation.new(Parameters(), HipArgs(**kw))
    if developed:
        import bench

        sp = bench.upload_species(sim, *bench.developed_start(rows, cols))
        sim.perform_steps(sp, 4000)
    else:
        sp = sim.make_species([rows, cols])
    for _ in range(8):
        sim.perform_steps(sp, 400)
        if "rows_per_block" in kw or sim.context.get_tuned(rows, cols)[0] > 0:
            break
    lib = capi.load()
    read = lib.gs_debug_trace_read_op
    read.restype = ctypes.c_int32
    read.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    units = 1 << 17
    buf = np.zeros((units, 8), np.uint64)
    # period of a pass from HIP events
    sim.perform_steps(sp, 400)
    sim.context.timer_start()
    p0 = sim.context.stats()["passes"]
    sim.prepare_steps(sp, 800)
    ms = sim.context.timer_stop()
    passes = sim.context.stats()["passes"] - p0
    sim.context.sync()
    assert read(buf.ctypes.data_as(ctypes.c_void_p), units, 1) == 0
    sim.prepare_steps(sp, 800 // passes * 3)      # three passes; the last one's records stay
    sim.context.sync()
    assert read(buf.ctypes.data_as(ctypes.c_void_p), units, 0) == 0
    label = sim.context.info()[0]
    rec = buf[buf[:, 4] > 0]
    n = len(rec)
    t = rec[:, :5].astype(np.int64)
    base = t[:, 0].min()
    t = (t - base) * 0.01                                   # microseconds since the first wave's entry
    nrows = (rec[:, 6] & np.uint64(0x3fffffff)).astype(np.int64)
    edge = (rec[:, 6] & np.uint64(0x40000000)) != 0
    hw = rec[:, 5] & np.uint64(0xffffffff)
    xcc = (rec[:, 5] >> np.uint64(32)).astype(np.int64)
    cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)   # layout of HW_ID on gfx9: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(np.int64)
    simd = ((hw >> np.uint64(4)) & np.uint64(0x3)).astype(np.int64)
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    period = ms * 1e3 / passes
    span = t[:, 4].max()

    def pct(x, qs=(0, 10, 50, 90, 100)):
        return " / ".join(f"{np.percentile(x, q):.2f}" for q in qs)

    print(f"grid {rows}x{cols}  kernel {label}  tuned {sim.context.get_tuned(rows, cols)}  {kw}")
    print(f"pass period (HIP events, {passes} passes): {period:.2f} us = {rows * cols * (800 // passes) / period:.0f} Mcells*steps/s"
          f" at {800 // passes} steps per pass")
    print(f"waves recorded: {n} ({int(edge.sum())} edge units); launch span first entry -> last exit: {span:.2f} us; "
          f"gap to the period: {period - span:.2f} us")
    # in-kernel clock: shader cycles (s_memtime) over real time (s_memrealtime, 100 MHz) between a wave's entry
    # and exit (MI355X_MICROARCH.md, "DVFS give-back" item 6); waves shorter than 20 us are left out
    cycles = (rec[:, 7] >> np.uint64(32)).astype(np.float64)
    dur_us = (rec[:, 4].astype(np.int64) - rec[:, 0].astype(np.int64)) * 0.01
    ok = (dur_us > 20.0) & (cycles > 0)
    if ok.any():
        print(f"in-kernel clock (cycles / real time per wave), MHz, percentiles 0/10/50/90/100: {pct(cycles[ok] / dur_us[ok])}"
              f"  ({int(ok.sum())} waves)")
    print(f"distinct CUs {len(np.unique(cu_key))}, SIMDs {len(np.unique(simd_key))}; waves per SIMD min/median/max: "
          f"{pct(np.bincount(np.unique(simd_key, return_inverse=True)[1]), (0, 50, 100))}")
    print("percentiles 0/10/50/90/100 [us]:")
    print(f"  entry                      {pct(t[:, 0])}")
    print(f"  exit                       {pct(t[:, 4])}")
    for name, mask in (("interior", ~edge), ("edge", edge)):
        if mask.sum() == 0:
            continue
        tt = t[mask]
        print(f"  {name} units ({int(mask.sum())}, {int(np.median(nrows[mask]))} rows): ")
        print(f"    entry -> tick 3          {pct(tt[:, 1] - tt[:, 0])}")
        print(f"    tick 3 -> pipeline full  {pct(tt[:, 2] - tt[:, 1])}")
        print(f"    steady march             {pct(tt[:, 3] - tt[:, 2])}")
        print(f"    drain                    {pct(tt[:, 4] - tt[:, 3])}")
        print(f"    whole unit               {pct(tt[:, 4] - tt[:, 0])}")
    # resident waves over time, in 20 bins
    bins = np.linspace(0, span, 21)
    line = []
    for a, b in zip(bins[:-1], bins[1:]):
        mid = 0.5 * (a + b)
        line.append(int(((t[:, 0] <= mid) & (t[:, 4] > mid)).sum()))
    print("resident waves at 20 instants across the span:", " ".join(str(x) for x in line))
    # second-round waves: entries later than the first exits
    first_exit = t[:, 4].min()
    late = t[:, 0] > first_exit
    print(f"first exit at {first_exit:.2f} us; waves entering after it (a second round): {int(late.sum())}")
    per_simd_end = np.zeros(simd_key.max() + 1)
    np.maximum.at(per_simd_end, simd_key, t[:, 4])
    used = np.unique(simd_key)
    print(f"per-SIMD time of last exit [us]: {pct(per_simd_end[used])}")
    sim.context.close()


if __name__ == "__main__":
    main()
