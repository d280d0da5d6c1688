#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run17
mkdir -p "$OUT"
cd "$ROOT"
# all-edge grid: 240 columns = 2 strips of 120, both touch a column edge
for lib in main old; do
  l=$ROOT/grayscott_amd/libgs_hip.so; [ $lib = old ] && l=$ROOT/grayscott_amd/variants/libgs_hip_nb.so
  echo "== $lib" | tee -a "$OUT/sweep.log"
  GS_HIP_LIBRARY=$l timeout -k 10 300 python tools/sweep.py --rows 1048576 --cols 240 --steps 96 --rounds 3 rows_per_block=128,cols_per_lane=2 rows_per_block=128,cols_per_lane=2,boundary=1 2>&1 | tee -a "$OUT/sweep.log"
  GS_HIP_LIBRARY=$l timeout -k 10 300 python tools/sweep.py --rows 1048576 --cols 360 --steps 96 --rounds 3 rows_per_block=128,cols_per_lane=2 2>&1 | tee -a "$OUT/sweep.log"
done
