#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run32
mkdir -p "$OUT"
cd "$ROOT"
for f in 0 65000 84000; do
  echo "== LDS floor $f" | tee -a "$OUT/sweep.log"
  GS_HIP_TILE_LDS_FLOOR=$f timeout -k 10 300 python tools/tile_sweep.py 128x256 256x512 512x1024 2>&1 | tee -a "$OUT/sweep.log"
done
