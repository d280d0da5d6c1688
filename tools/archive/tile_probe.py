#!/usr/bin/env python3
"""A few hundred launches of the LDS-window kernel on one mid-size grid, for rocprofv3 (kernel trace or PMC pass):
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES -- python3 tools/archive/tile_probe.py 128 256
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import HipArgs, Parameters, Simulation  # noqa: E402

rows, cols = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
sp = sim.make_species([rows, cols])
for _ in range(4):
    sim.perform_steps(sp, steps)
print(sim.context.info())
