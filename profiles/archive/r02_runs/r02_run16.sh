#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run16
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python tools/sweep.py --steps 96 --rounds 5 \
     rows_per_block=192,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 rows_per_block=128,cols_per_lane=2,boundary=1 rows_per_block=128,cols_per_lane=2,general_kernels=1 rows_per_block=128,cols_per_lane=1 rows_per_block=128,cols_per_lane=4 2>&1 | tee -a "$OUT/sweep.log"
timeout -k 10 300 python tools/sweep.py --rows 4096 --cols 4096 --steps 400 --rounds 5 rows_per_block=32,cols_per_lane=2 rows_per_block=32,cols_per_lane=1 2>&1 | tee -a "$OUT/sweep.log"
timeout -k 10 300 python tools/sweep.py --rows 1080 --cols 1920 --steps 1000 --rounds 5 rows_per_block=8,cols_per_lane=1 2>&1 | tee -a "$OUT/sweep.log"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_sizes.py tests/test_gpu_property.py -m gpu -x -q 2>&1 | tail -3 | tee -a "$OUT/sweep.log"
