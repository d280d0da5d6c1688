#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run10
mkdir -p "$OUT"
cd "$ROOT"
for round in 1 2; do
for v in w3 w4; do
  echo "== variant $v (round $round)" | tee -a "$OUT/sweep.log"
  GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$v.so timeout -k 10 300 python tools/sweep.py --steps 96 --rounds 5 \
     rows_per_block=192,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 rows_per_block=64,cols_per_lane=2 rows_per_block=48,cols_per_lane=2 2>&1 | tee -a "$OUT/sweep.log"
done
done
GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_w4.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_sizes.py -m gpu -x -q 2>&1 | tail -3 | tee -a "$OUT/sweep.log"
