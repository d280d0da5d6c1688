#!/bin/bash
# HBM read traffic (FETCH_SIZE) of the headline launches under two settings of one environment variable:
#   tools/archive/fetch_ab.sh VAR VALUE_A VALUE_B      (run through gpurun; layout pinned as tools/profile_sq_detail.sh)
set -eo pipefail
VAR=$1; A=$2; B=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/fetch_ab_$VAR
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
export GS_HIP_ROWS_PER_BLOCK=122 GS_HIP_FUSE_STEPS=4 GS_HIP_COLS_PER_LANE=2
for v in "$A" "$B"; do
  export $VAR=$v
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$v" -o bench -- \
      python3 "$ROOT/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-extra > "$OUT/bench_$v.json" 2> "$OUT/fetch_$v.log"
  python3 - "$OUT/fetch_$v/bench_counter_collection.csv" "$VAR=$v" <<'PY'
import csv, statistics, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "gs_step_tb_k" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print(f"{sys.argv[2]}: FETCH_SIZE median {statistics.median(vals):.0f} KiB over {len(vals)} launches -> reads = {2 * statistics.median(vals) * 1024 / 2**30:.3f} GiB per launch")
PY
done
