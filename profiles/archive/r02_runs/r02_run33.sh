#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run33
mkdir -p "$OUT"
cd "$ROOT"
export GS_HIP_LIBRARY=$ROOT/grayscott_amd/variants/libgs_hip_tf1.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile" 2>&1 | tail -5 | tee "$OUT/pytest.log"
timeout -k 10 400 python tools/tile_sweep.py 64x128 128x256 256x512 512x1024 2>&1 | tee -a "$OUT/sweep.log"
