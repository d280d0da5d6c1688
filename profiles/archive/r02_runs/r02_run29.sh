#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run29
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python tools/tile_sweep.py 96x192 192x384 384x768 768x1536 720x1280 1024x1024 1080x1920 100x3000 3000x100 2>&1 | tee "$OUT/tile_sweep.md"
