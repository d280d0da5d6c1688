#!/bin/bash
# Round-2 experiment 1 (run through gpurun): parity of the compact-row kernel, VALU issue rate by
# occupancy, and A/B timing of the row layouts (libgs_hip variants built by tools/ab_build.py).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_exp1
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -3 "$OUT/pytest.log"
timeout -k 10 300 tools/ubench/valu_rate2 > "$OUT/valu_rate2.log" 2>&1; cat "$OUT/valu_rate2.log"
for round in 1 2; do
for v in old c0 c1 c2; do
  echo "== variant $v (round $round)" | tee -a "$OUT/sweep.log"
  GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$v.so timeout -k 10 300 python tools/sweep.py --steps 48 --rounds 5 \
     rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 rows_per_block=64,cols_per_lane=2 rows_per_block=48,cols_per_lane=2 \
     rows_per_block=128,cols_per_lane=4 rows_per_block=128,cols_per_lane=1 2>&1 | tee -a "$OUT/sweep.log"
done
done
