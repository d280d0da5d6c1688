#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run34
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2 3 4; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['kernel'], round(d['value']), d['roofline']['launch_ms'])" | tee -a "$OUT/bench.log"
done
timeout -k 10 900 python -m pytest tests/test_gpu_timed_sizes.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | tee "$OUT/pytest.log"
