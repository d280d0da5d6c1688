#!/bin/bash
# round-2 records: full GPU suite, bench line, BASELINE configs, criterion grid
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run9
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -4 "$OUT/pytest.log"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"; tail -1 "$OUT/bench.json" | cut -c1-1500
timeout -k 10 900 python tools/baseline_configs.py > "$OUT/baseline_configs.md" 2> "$OUT/baseline_configs.log"; echo "baseline rc=$?"; cat "$OUT/baseline_configs.md"
timeout -k 10 600 python tools/criterion_grid.py --cpu > "$OUT/criterion_grid.md" 2> "$OUT/criterion_grid.log"; echo "grid rc=$?"; cat "$OUT/criterion_grid.md"
