#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3 (run through gpurun):
#   0. an un-profiled run              the layout gs_run's tuner picks on this box (unit height, steps per
#                                      pass, columns per lane); the profiled runs are pinned to it, so that
#                                      every profiled launch is the same kernel configuration
#   1. --kernel-trace --stats          per-kernel time (average launch duration)
#   2. --pmc FETCH_SIZE                HBM read traffic   (its own pass: TCC has 4 slots,
#   3. --pmc WRITE_SIZE                HBM write traffic   FETCH_SIZE costs 3, WRITE_SIZE 2)
#   4. --pmc SQ_INSTS_VALU ...         issued VALU wave-instructions, wave cycles, stall shares (8 SQ slots)
# PMC passes use --kernel-trace only (no sys/hip/hsa tracing), as the pool requires.
# Raw CSVs land under gpurun_out/prof_$TAG/; tools/summarize_profile.py turns them into the
# summaries committed under profiles/.
#   tools/profile_gpu.sh TAG [STEPS] ; GS_BENCH_EXTRA="--rows 4096 --cols 4096" for another grid
set -eo pipefail
TAG=${1:-r01}
STEPS=${2:-200}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
EXTRA=${GS_BENCH_EXTRA:-}
python3 "$ROOT/bench.py" --steps "$STEPS" --warmup 20 --no-cpu-baseline --no-extra --no-verify $EXTRA > "$OUT/bench_unprofiled.json" 2> "$OUT/unprofiled.log"
tail -1 "$OUT/bench_unprofiled.json"
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
b = json.loads(open(out + "/bench_unprofiled.json").read().strip().splitlines()[-1])
json.dump(b["config"]["tuned"], open(out + "/layout.json", "w"))
t = b["config"]["tuned"]
open(out + "/layout.env", "w").write(
    f"export GS_HIP_ROWS_PER_BLOCK={t['rows_per_unit']} GS_HIP_FUSE_STEPS={t['steps_per_pass']} GS_HIP_COLS_PER_LANE={t['cols_per_lane']}"
    f" GS_HIP_SHARE_TAPS={ {'off': 2, 'across lanes': 3}.get(t.get('share_taps'), 1) }\n"
    if t["rows_per_unit"] > 0 else "")
PY
source "$OUT/layout.env"
cat "$OUT/layout.json"; echo
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
    python3 "$ROOT/bench.py" --steps "$STEPS" --warmup 20 --no-cpu-baseline --no-extra --no-verify $EXTRA > "$OUT/bench_stats.json" 2> "$OUT/stats.log"
tail -1 "$OUT/bench_stats.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o bench -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-extra --no-verify $EXTRA > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o bench -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-extra --no-verify $EXTRA > "$OUT/bench_write.json" 2> "$OUT/write.log"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES \
    --output-format csv -d "$OUT/sq" -o bench -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-extra --no-verify $EXTRA > "$OUT/bench_sq.json" 2> "$OUT/sq.log"
find "$OUT" -name '*.csv' | head -20
