#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run54
mkdir -p "$OUT"
cd "$ROOT"
S="timeout -k 10 300 python tools/sweep.py --rounds 5"
$S --rows 4096 --cols 4096 --steps 400 rows_per_block=32,cols_per_lane=1 rows_per_block=39,cols_per_lane=2,split=2 rows_per_block=20,cols_per_lane=2,split=2 rows_per_block=78,cols_per_lane=2,split=2 rows_per_block=32,cols_per_lane=1,split=2 rows_per_block=64,cols_per_lane=1,split=2 rows_per_block=39,cols_per_lane=2,split=3 rows_per_block=39,cols_per_lane=2,split=4 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
$S --rows 8192 --cols 8192 --steps 200 rows_per_block=64,cols_per_lane=2 rows_per_block=64,cols_per_lane=2,split=2 rows_per_block=75,cols_per_lane=2,split=2 rows_per_block=128,cols_per_lane=2,split=2 rows_per_block=96,cols_per_lane=2,split=2 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
