#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run45
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python tools/sweep.py --steps 400 --rounds 7 rows_per_block=96,cols_per_lane=2 rows_per_block=101,cols_per_lane=2 rows_per_block=122,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=155,cols_per_lane=2 rows_per_block=192,cols_per_lane=2 rows_per_block=214,cols_per_lane=2 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
