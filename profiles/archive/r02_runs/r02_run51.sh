#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run51
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python tools/tile_sweep.py 720x1280 1024x1024 768x1536 1080x1920 1200x1600 1024x2048 2>&1 | tee "$OUT/tile_sweep.md"
