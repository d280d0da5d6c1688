#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run4
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
for k in 3 5; do
  echo "== kernel $k" | tee -a "$OUT/grid.log"
  timeout -k 10 600 python tools/criterion_grid.py --kernel $k --kmin 5 2>&1 | tee -a "$OUT/grid.log"
done
