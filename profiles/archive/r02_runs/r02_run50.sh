#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run50
mkdir -p "$OUT"
cd "$ROOT"
for v in "" ab4 ab2; do
  echo "== variant ${v:-shipped}" | tee -a "$OUT/sweep.log"
  if [ -n "$v" ]; then export GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$v.so; fi
  timeout -k 10 300 python tools/sweep.py --rows 4096 --cols 4096 --steps 400 --rounds 5 rows_per_block=39,cols_per_lane=2 rows_per_block=19,cols_per_lane=2 rows_per_block=32,cols_per_lane=1 rows_per_block=64,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
done
