#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run12
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -4 "$OUT/pytest.log"
timeout -k 10 300 python tools/sweep.py --steps 96 --rounds 5 rows_per_block=256,cols_per_lane=2 rows_per_block=192,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 rows_per_block=128,cols_per_lane=4 rows_per_block=128,cols_per_lane=1 2>&1 | tee "$OUT/sweep.log"
timeout -k 10 300 python tools/sweep.py --rows 4096 --cols 4096 --steps 400 --rounds 5 rows_per_block=64,cols_per_lane=2 rows_per_block=32,cols_per_lane=2 rows_per_block=32,cols_per_lane=1 2>&1 | tee -a "$OUT/sweep.log"
timeout -k 10 600 python bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"; tail -1 "$OUT/bench.json" | cut -c1-900
