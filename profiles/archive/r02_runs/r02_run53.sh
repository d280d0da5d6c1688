#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run53
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident or golden or unusual" 2>&1 | tail -3 | tee "$OUT/pytest.log"
timeout -k 10 300 python - <<'PY' 2>&1 | tee -a "$OUT/log"
import time, statistics
from grayscott_amd import HipArgs, Parameters, Simulation, capi
for shape in [(8,16),(16,16),(16,32),(24,32),(32,32),(32,48)]:
    sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
    sp = sim.make_species(list(shape))
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end: sim.perform_steps(sp, 256)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); sim.perform_steps(sp, 256); ts.append(time.perf_counter() - t0)
    print(shape, shape[0]*shape[1], sim.context.info()[0], round(shape[0]*shape[1]*256/statistics.median(ts)/1e6), flush=True)
    sim.context.close()
PY
