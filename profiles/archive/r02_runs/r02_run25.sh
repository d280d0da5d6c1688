#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run25
mkdir -p "$OUT"
cd "$ROOT"
S="timeout -k 10 300 python tools/sweep.py --rounds 5"
$S --rows 1024 --cols 2048 --steps 1000 rows_per_block=8,cols_per_lane=1 rows_per_block=6,cols_per_lane=2 rows_per_block=8,cols_per_lane=2 rows_per_block=12,cols_per_lane=2 rows_per_block=16,cols_per_lane=2 rows_per_block=4,cols_per_lane=1 rows_per_block=6,cols_per_lane=1 rows_per_block=12,cols_per_lane=1 rows_per_block=16,cols_per_lane=1 rows_per_block=8,cols_per_lane=1,fuse_steps=3 rows_per_block=6,cols_per_lane=1,fuse_steps=3 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
export GS_HIP_TRACE_TUNER=1
timeout -k 10 600 python - > "$OUT/tuned.log" 2> "$OUT/tuner_trace.log" <<'PY'
import time
from grayscott_amd import HipArgs, Parameters, Simulation
for rows, cols, steps in ((1024,2048,1000),(512,1024,1000)):
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    sc = sim.make_species([rows, cols]); sim.perform_steps(sc, 8000); sim.context.sync(); del sc
    rates=[]
    for _ in range(3):
        sp = sim.make_species([rows, cols]); sim.perform_steps(sp, 100); sim.context.sync()
        t0=time.perf_counter(); sim.perform_steps(sp, steps); sim.context.sync()
        rates.append(rows*cols*steps/(time.perf_counter()-t0)/1e6); del sp
    print(rows, cols, sim.context.info()[0], [round(r) for r in rates], flush=True)
PY
cat "$OUT/tuned.log"
