#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run35
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py -m gpu -x -q 2>&1 | tail -3 | tee "$OUT/pytest.log"
S="timeout -k 10 300 python tools/sweep.py --rounds 5"
for sp in 0 1; do
  echo "== GS_HIP_EDGE_SPLIT=$sp" | tee -a "$OUT/sweep.log"
  export GS_HIP_EDGE_SPLIT=$sp
  $S --rows 4096 --cols 4096 --steps 400 rows_per_block=18,cols_per_lane=2 rows_per_block=20,cols_per_lane=2 rows_per_block=36,cols_per_lane=2 rows_per_block=39,cols_per_lane=2 rows_per_block=40,cols_per_lane=2 rows_per_block=42,cols_per_lane=2 rows_per_block=30,cols_per_lane=1 rows_per_block=32,cols_per_lane=1 rows_per_block=60,cols_per_lane=1 rows_per_block=64,cols_per_lane=1 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
  $S --rows 2048 --cols 4096 --steps 400 rows_per_block=18,cols_per_lane=2 rows_per_block=20,cols_per_lane=2 rows_per_block=21,cols_per_lane=2 rows_per_block=30,cols_per_lane=1 rows_per_block=32,cols_per_lane=1 rows_per_block=34,cols_per_lane=1 rows_per_block=16,cols_per_lane=1 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
  $S --rows 1080 --cols 1920 --steps 1000 rows_per_block=8,cols_per_lane=1 rows_per_block=16,cols_per_lane=1 rows_per_block=9,cols_per_lane=1 rows_per_block=10,cols_per_lane=1 rows_per_block=16,cols_per_lane=2 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
  $S --rows 8192 --cols 8192 --steps 200 rows_per_block=64,cols_per_lane=2 rows_per_block=48,cols_per_lane=2 rows_per_block=70,cols_per_lane=2 rows_per_block=75,cols_per_lane=2 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
done
