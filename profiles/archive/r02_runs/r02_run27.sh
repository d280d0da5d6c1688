#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run47
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile" 2>&1 | tail -15 | tee "$OUT/pytest.log"
timeout -k 10 600 python tools/tile_sweep.py 2>&1 | tee "$OUT/tile_sweep.md"
