#!/bin/bash
# rocprofv3 passes over ANY python tool of this repository (run through gpurun), one kernel of interest:
#   tools/profile_kernel.sh TAG tools/run_steps.py --rows 1080 --cols 1920 --steps 1000 --calls 5
#   0. un-profiled run                   the tool's own line (rate)
#   1. --kernel-trace --stats            per-kernel time
#   2. --pmc FETCH_SIZE / 3. --pmc WRITE_SIZE      HBM traffic (separate passes; MI355X_MICROARCH.md "HBM")
#   4.-6. three SQ passes                instruction mix, what waves wait for, LDS bank conflicts
# PMC passes use --kernel-trace only; the program itself (python3 ...) follows `--`.  Raw CSVs under gpurun_out/prof_TAG/;
# tools/summarize_kernel_profile.py TAG NEEDLE turns them into profiles/TAG_summary.md.
set -eo pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PROG=("$@"); PROG[0]="$ROOT/${PROG[0]}"
python3 "${PROG[@]}" > "$OUT/unprofiled.json" 2> "$OUT/unprofiled.log"; tail -1 "$OUT/unprofiled.json"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "${PROG[@]}" > "$OUT/stats.json" 2> "$OUT/stats.log"
tail -1 "$OUT/stats.json"
pass() { # name counters...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o run -- python3 "${PROG[@]}" > "$OUT/$name.json" 2> "$OUT/$name.log"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_WAVE_CYCLES
find "$OUT" -name '*.csv' | head -20
