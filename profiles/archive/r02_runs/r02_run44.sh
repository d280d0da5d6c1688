#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run44
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python tools/soak.py --steps 10000 2>&1 | tail -4 | tee "$OUT/soak.log"
timeout -k 10 300 python tools/soak.py --rows 4096 --cols 4096 --steps 20000 2>&1 | tail -4 | tee -a "$OUT/soak.log"
timeout -k 10 300 python tools/soak.py --rows 512 --cols 1024 --steps 50000 2>&1 | tail -4 | tee -a "$OUT/soak.log"
timeout -k 10 300 python tools/soak.py --rows 1080 --cols 1920 --steps 30000 2>&1 | tail -4 | tee -a "$OUT/soak.log"
for g in 512x1024 1080x1920 4096x4096; do
  timeout -k 10 300 python bench.py --grid $g --steps 2000 --warmup 200 --no-cpu-baseline --no-extra 2>&1 | tail -1 | cut -c1-900 | tee -a "$OUT/bench_small.log"
done
