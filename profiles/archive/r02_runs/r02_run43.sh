#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run43
mkdir -p "$OUT"; rm -f "$OUT"/*.log
cd "$ROOT"
python - > "$OUT/run.log" 2>&1 <<'PY' &
import time
from grayscott_amd import HipArgs, Parameters, Simulation
sim = Simulation.new(Parameters(), HipArgs(devices=[0]))
sp = sim.make_species([16384, 16384])
t0 = time.time()
print("start", t0, flush=True)
for i in range(10):
    sim.perform_steps(sp, 8000)
    print("chunk", i, time.time() - t0, flush=True)
PY
PID=$!
for i in $(seq 1 50); do
  echo "t=$(date +%s.%N)" >> "$OUT/smi_load.log"
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" >> "$OUT/smi_load.log"
  sleep 0.5
done
wait $PID
cat "$OUT/run.log"
grep -i "sclk" "$OUT/smi_load.log" | sort | uniq -c | sort -rn | head -20
grep -i "Power (W)" "$OUT/smi_load.log" | sort | uniq -c | sort -rn | head -12
