#!/usr/bin/env python3
"""Is the throughput data-independent?  Times 1000-step runs of the 16384^2 grid on two states in
one context: the reference's benchmark input (Species::new: one seed in a uniform field) and a
field with many seeds that develops into a full spot pattern.  (It is not: see
profiles/archive/r01_soak.md.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import soak
from grayscott_amd import HipArgs, Parameters, Simulation
rows = cols = 16384
rng = np.random.default_rng(2024)
u0 = np.ones((rows, cols), np.float32); v0 = np.zeros((rows, cols), np.float32)
for _ in range(rows * cols // 40000):
    r, c = int(rng.integers(0, rows - 12)), int(rng.integers(0, cols - 12))
    u0[r:r + 12, c:c + 12] = 0.5; v0[r:r + 12, c:c + 12] = 0.25
u0 += (rng.random(u0.shape, dtype=np.float32) * np.float32(0.01)).astype(np.float32)
v0 += (rng.random(v0.shape, dtype=np.float32) * np.float32(0.01)).astype(np.float32)
sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
sp = soak.species_from(sim, u0, v0)
ref = sim.make_species([rows, cols])
for s in (sp, ref):
    sim.perform_steps(s, 1000)
sim.context.sync()
for label, s in (("Species::new state", ref), ("seeded, step 1000", sp)):
    ts = []
    for _ in range(3):
        sim.context.timer_start(); sim.perform_steps(s, 1000); ts.append(sim.context.timer_stop())
    print(f"{label:22s} {rows*cols*1000/min(ts)/1e6:.0f} k Mcells*steps/s", flush=True)
for _ in range(8):
    sim.perform_steps(sp, 1000)
ts = []
for _ in range(3):
    sim.context.timer_start(); sim.perform_steps(sp, 1000); ts.append(sim.context.timer_stop())
v = sp.in_out()[1].make_scalar_view(sim.context)
print(f"seeded, step 13000      {rows*cols*1000/min(ts)/1e6:.0f} k Mcells*steps/s; cells with V > 0.1: {100*np.count_nonzero(v>0.1)/v.size:.1f} %")
ts = []
for _ in range(3):
    sim.context.timer_start(); sim.perform_steps(ref, 1000); ts.append(sim.context.timer_stop())
print(f"Species::new state again {rows*cols*1000/min(ts)/1e6:.0f} k Mcells*steps/s")
