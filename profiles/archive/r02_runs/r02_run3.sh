#!/bin/bash
# full GPU test-suite, default bench line, profile of the shipped kernel, multi-process rehearsal
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run3
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"; tail -1 "$OUT/bench.json"; tail -3 "$OUT/bench.log"
GS_HIP_ROWS_PER_BLOCK=96 timeout -k 10 900 bash tools/profile_gpu.sh r02a 400 > "$OUT/profile.log" 2>&1; echo "profile rc=$?"; tail -3 "$OUT/profile.log"
timeout -k 10 900 bash tools/rehearsal.sh 60 > "$OUT/rehearsal.log" 2>&1; echo "rehearsal rc=$?"; tail -6 "$OUT/rehearsal.log"
