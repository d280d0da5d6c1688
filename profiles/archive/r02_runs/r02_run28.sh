#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run28
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q 2>&1 | tail -60 > "$OUT/pytest.log"; tail -30 "$OUT/pytest.log"
timeout -k 10 600 python tools/criterion_grid.py > "$OUT/criterion.md" 2>&1; tail -12 "$OUT/criterion.md"
