#!/bin/bash
# Round-2 experiment 2: ds_bpermute exchange vs DPP, compact rows, late fetch.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_exp2
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 tools/ubench/valu_rate2 > "$OUT/valu_rate2.log" 2>&1; cat "$OUT/valu_rate2.log"
for v in old ol bo bc bc0; do
  echo "== variant $v" | tee -a "$OUT/sweep.log"
  GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$v.so timeout -k 10 300 python tools/sweep.py --steps 48 --rounds 5 \
     rows_per_block=128,cols_per_lane=2 rows_per_block=96,cols_per_lane=2 rows_per_block=64,cols_per_lane=2 \
     rows_per_block=128,cols_per_lane=4 rows_per_block=64,cols_per_lane=4 rows_per_block=128,cols_per_lane=1 2>&1 | tee -a "$OUT/sweep.log"
done
for v in old bc; do
  echo "== variant $v parity (tb tests)" | tee -a "$OUT/sweep.log"
  GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a "$OUT/sweep.log"
done
