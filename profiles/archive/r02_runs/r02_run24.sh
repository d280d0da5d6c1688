#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run24
mkdir -p "$OUT"
cd "$ROOT"
export GS_HIP_TRACE_TUNER=1
timeout -k 10 600 python - > "$OUT/tuned.log" 2> "$OUT/tuner_trace.log" <<'PY'
import time
from grayscott_amd import HipArgs, Parameters, Simulation
for rows, cols, steps in ((1080,1920,1000),(2048,4096,1000),(4096,4096,1000),(8192,8192,1000),(16384,16384,2000)):
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    sc = sim.make_species([rows, cols]); sim.perform_steps(sc, 4000 if rows < 16384 else 800); sim.context.sync(); del sc
    rates=[]
    for _ in range(3):
        sp = sim.make_species([rows, cols]); sim.perform_steps(sp, 100); sim.context.sync()
        t0=time.perf_counter(); sim.perform_steps(sp, steps); sim.context.sync()
        rates.append(rows*cols*steps/(time.perf_counter()-t0)/1e6); del sp
    # untuned default of a fresh context
    sim2 = Simulation.new(Parameters(), HipArgs(devices=[0], no_tune=1))
    sp = sim2.make_species([rows, cols]); sim2.perform_steps(sp, 100); sim2.context.sync()
    t0=time.perf_counter(); sim2.perform_steps(sp, steps); sim2.context.sync()
    un = rows*cols*steps/(time.perf_counter()-t0)/1e6
    print(rows, cols, sim.context.info()[0], [round(r) for r in rates], "untuned", sim2.context.info()[0], round(un), flush=True)
PY
cat "$OUT/tuned.log"
unset GS_HIP_TRACE_TUNER
timeout -k 10 600 python tools/criterion_grid.py > "$OUT/criterion.md" 2>&1; tail -12 "$OUT/criterion.md"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py -m gpu -x -q 2>&1 | tail -2 | tee -a "$OUT/pytest.log"
