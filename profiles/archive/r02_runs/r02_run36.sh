#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run41
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
    sim2 = Simulation.new(Parameters(), HipArgs(devices=[0], no_tune=1))
    sp = sim2.make_species([rows, cols]); sim2.perform_steps(sp, 100); sim2.context.sync()
    t0=time.perf_counter(); sim2.perform_steps(sp, steps); sim2.context.sync()
    un = rows*cols*steps/(time.perf_counter()-t0)/1e6
    print(rows, cols, sim.context.info()[0], [round(r) for r in rates], "untuned", sim2.context.info()[0], round(un), flush=True)
PY
cat "$OUT/tuned.log"; unset GS_HIP_TRACE_TUNER
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee "$OUT/pytest.log"
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], round(d['value']), d['roofline']['launch_ms'])" | tee -a "$OUT/bench.log"
done
