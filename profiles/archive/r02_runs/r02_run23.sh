#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run23
mkdir -p "$OUT"
cd "$ROOT"
S="timeout -k 10 300 python tools/sweep.py --rounds 5"
$S --rows 4096 --cols 4096 --steps 400 rows_per_block=32,cols_per_lane=2 rows_per_block=34,cols_per_lane=2 rows_per_block=36,cols_per_lane=2 rows_per_block=40,cols_per_lane=2 rows_per_block=48,cols_per_lane=2 rows_per_block=18,cols_per_lane=2 rows_per_block=24,cols_per_lane=2 rows_per_block=32,cols_per_lane=1 rows_per_block=56,cols_per_lane=1 rows_per_block=60,cols_per_lane=1 rows_per_block=64,cols_per_lane=1 rows_per_block=28,cols_per_lane=1 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
$S --rows 8192 --cols 8192 --steps 200 rows_per_block=64,cols_per_lane=2 rows_per_block=70,cols_per_lane=2 rows_per_block=47,cols_per_lane=2 rows_per_block=48,cols_per_lane=2 rows_per_block=139,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
$S --rows 2048 --cols 4096 --steps 400 rows_per_block=8,cols_per_lane=2 rows_per_block=16,cols_per_lane=2 rows_per_block=18,cols_per_lane=2 rows_per_block=20,cols_per_lane=2 rows_per_block=16,cols_per_lane=1 rows_per_block=28,cols_per_lane=1 rows_per_block=30,cols_per_lane=1 rows_per_block=32,cols_per_lane=1 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
