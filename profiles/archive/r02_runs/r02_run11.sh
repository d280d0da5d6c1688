#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run11
mkdir -p "$OUT"
cd "$ROOT"
for t in "" "4,4,0,0" "4,3,0,0" "2,3,4,3" "2,3,8,3" "4,3,8,3" "2,6,8,3" "3,3,6,3" "2,3,4,3"; do
  echo "== taper '$t'" | tee -a "$OUT/sweep.log"
  GS_HIP_TAPER="$t" timeout -k 10 300 python tools/sweep.py --steps 96 --rounds 5 \
     rows_per_block=256,cols_per_lane=2 rows_per_block=192,cols_per_lane=2 rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
done
GS_HIP_TAPER="2,3,8,3" timeout -k 10 600 python -m pytest tests/test_gpu_timed_sizes.py -m gpu -x -q 2>&1 | tail -2 | tee -a "$OUT/sweep.log"
