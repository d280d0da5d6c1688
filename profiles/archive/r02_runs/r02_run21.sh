#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run21
mkdir -p "$OUT"
cd "$ROOT"
export GS_HIP_NO_SPLIT_LAUNCH=1
for v in lf0 lf1 lf1r; do
  echo "== variant $v" | tee -a "$OUT/sweep.log"
  lib=$ROOT/grayscott_amd/variants/libgs_hip_$v.so
  GS_HIP_LIBRARY=$lib timeout -k 10 300 python tools/sweep.py --steps 96 --rounds 5 rows_per_block=192,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 rows_per_block=64,cols_per_lane=2 rows_per_block=128,cols_per_lane=1 rows_per_block=128,cols_per_lane=4 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
  GS_HIP_LIBRARY=$lib timeout -k 10 300 python tools/sweep.py --rows 8192 --cols 8192 --steps 200 --rounds 5 rows_per_block=64,cols_per_lane=2 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
  GS_HIP_LIBRARY=$lib timeout -k 10 300 python tools/sweep.py --rows 4096 --cols 4096 --steps 400 --rounds 5 rows_per_block=32,cols_per_lane=2 rows_per_block=32,cols_per_lane=1 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
done
GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_lf1.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_sizes.py tests/test_gpu_property.py -m gpu -x -q 2>&1 | tail -2 | tee -a "$OUT/sweep.log"
