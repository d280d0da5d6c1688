#!/bin/bash
# Extra SQ counter passes over bench.py's headline launches (run through gpurun): what else besides VALU the
# waves issue and what they wait for.  Same rules as tools/profile_gpu.sh (PMC passes with --kernel-trace only,
# the program itself after --, layout pinned to the tuner's usual choice).
#   tools/profile_sq_detail.sh TAG [bench args...]
set -eo pipefail
TAG=${1:-sqd}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
export GS_HIP_ROWS_PER_BLOCK=${GS_HIP_ROWS_PER_BLOCK:-122} GS_HIP_FUSE_STEPS=4 GS_HIP_COLS_PER_LANE=2
pass() { # name counters...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o bench -- \
      python3 "$ROOT/bench.py" --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-extra --no-verify $BENCH_ARGS > "$OUT/bench_$name.json" 2> "$OUT/$name.log"
}
BENCH_ARGS="$*"
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVE_CYCLES
python3 - "$OUT" <<'PY'
import csv, statistics, sys, os
out = sys.argv[1]
for name in "abc":
    path = os.path.join(out, name, "bench_counter_collection.csv")
    if not os.path.exists(path):
        print(name, "no counters collected"); continue
    vals = {}
    for row in csv.DictReader(open(path)):
        if "gs_step_tb" in row["Kernel_Name"]:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in sorted(vals.items()):
        print(f"{name} {k:28s} median {statistics.median(v):16.0f}  launches {len(v)}")
PY
