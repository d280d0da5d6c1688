#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run42
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 400 python tools/sweep.py --steps 96 --rounds 5 no_tune=1 slabs=2,no_tune=1 slabs=4,no_tune=1 slabs=8,no_tune=1 2>&1 | grep -v "^grid" | tee -a "$OUT/sweep.log"
timeout -k 10 600 bash tools/rehearsal.sh 2>&1 | tail -12 | tee -a "$OUT/rehearsal.log"
