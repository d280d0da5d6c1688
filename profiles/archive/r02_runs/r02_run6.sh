#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run6
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -3 "$OUT/pytest.log"
timeout -k 10 900 python tools/tile_sweep.py 2>&1 | tee "$OUT/tile_sweep.log"
