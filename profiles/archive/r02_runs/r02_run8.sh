#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run8
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py tests/test_gpu_property.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -4 "$OUT/pytest.log"
timeout -k 10 600 python tools/sweep.py --steps 96 --rounds 5 rows_per_block=96,cols_per_lane=2 slabs=2,rows_per_block=96,cols_per_lane=2 slabs=4,rows_per_block=96,cols_per_lane=2 slabs=2,rows_per_block=64,cols_per_lane=2 2>&1 | tee "$OUT/sweep.log"
GS_HIP_NO_DIRECT_GHOSTS=1 timeout -k 10 600 python tools/sweep.py --steps 96 --rounds 5 slabs=2,rows_per_block=96,cols_per_lane=2 slabs=4,rows_per_block=96,cols_per_lane=2 2>&1 | tee -a "$OUT/sweep.log"
