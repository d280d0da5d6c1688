#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run13
mkdir -p "$OUT"
cd "$ROOT"
GS_HIP_TRACE_TUNER=1 timeout -k 10 300 python tools/sweep.py --steps 96 --rounds 5 rows_per_block=256,cols_per_lane=2 rows_per_block=192,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 rows_per_block=128,cols_per_lane=4 rows_per_block=128,cols_per_lane=1 2>&1 | grep -v "tuner" | tee "$OUT/sweep.log"
timeout -k 10 300 python tools/sweep.py --rows 4096 --cols 4096 --steps 400 --rounds 5 rows_per_block=64,cols_per_lane=2 rows_per_block=32,cols_per_lane=2 rows_per_block=32,cols_per_lane=1 2>&1 | tee -a "$OUT/sweep.log"
timeout -k 10 300 python tools/sweep.py --rows 8192 --cols 8192 --steps 200 --rounds 5 rows_per_block=128,cols_per_lane=2 rows_per_block=64,cols_per_lane=2 rows_per_block=48,cols_per_lane=2 2>&1 | tee -a "$OUT/sweep.log"
