#!/bin/bash
# final records of the round: full GPU suite, smoke, bench line, profile of the shipped state, criterion grid
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02c}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -4 "$OUT/pytest.log"
timeout -k 10 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"; tail -1 "$OUT/bench.json" | cut -c1-1200
GS_HIP_ROWS_PER_BLOCK=${GS_HIP_ROWS_PER_BLOCK:-128} timeout -k 10 900 bash tools/profile_gpu.sh $TAG 400 > "$OUT/profile.log" 2>&1; echo "profile rc=$?"; tail -2 "$OUT/profile.log"
timeout -k 10 600 python tools/criterion_grid.py --cpu > "$OUT/criterion_grid.md" 2> "$OUT/criterion_grid.log"; echo "grid rc=$?"; cat "$OUT/criterion_grid.md"
timeout -k 10 900 python tools/baseline_configs.py > "$OUT/baseline_configs.md" 2> "$OUT/baseline_configs.log"; echo "configs rc=$?"; head -8 "$OUT/baseline_configs.md"
