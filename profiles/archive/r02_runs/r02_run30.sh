#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run30
mkdir -p "$OUT"
cd "$ROOT"
for v in "" ta1 ta2 ta3; do
  echo "== variant ${v:-shipped}" | tee -a "$OUT/sweep.log"
  if [ -n "$v" ]; then export GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_$v.so; fi
  timeout -k 10 300 python tools/tile_sweep.py 128x256 512x1024 2>&1 | tee -a "$OUT/sweep.log"
done
