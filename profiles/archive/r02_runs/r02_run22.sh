#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run22
mkdir -p "$OUT"
cd "$ROOT"
for v in base ab2 ab3 ab4 base; do
  echo "== variant $v" | tee -a "$OUT/sweep.log"
  lib=$ROOT/grayscott_amd/variants/libgs_hip_$v.so
  GS_HIP_LIBRARY=$lib timeout -k 10 300 python tools/sweep.py --steps 96 --rounds 5 rows_per_block=96,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=128,cols_per_lane=1 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
done
