#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run7
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python tools/sweep.py --steps 96 --rounds 5 rows_per_block=96,cols_per_lane=2 slabs=2,rows_per_block=96,cols_per_lane=2 slabs=4,rows_per_block=96,cols_per_lane=2 2>&1 | tee "$OUT/sweep.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/trace2" -o t -- python3 "$ROOT/tools/sweep.py" --steps 96 --rounds 2 slabs=2,rows_per_block=96,cols_per_lane=2 > "$OUT/trace2.log" 2>&1
head -12 "$OUT"/trace2/*kernel_stats.csv
head -8 "$OUT"/trace2/*memory_copy_stats.csv 2>/dev/null
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace2/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# print a window of 30 kernels in steady state
mid = len(rows) * 3 // 4
for r in rows[mid:mid + 30]:
    print(f'{(int(r["Start_Timestamp"]) - t0)/1e3:12.1f} us  dur {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3:8.1f} us  q{r.get("Queue_Id","?")}  {r["Kernel_Name"][:60]}  grid {r.get("Grid_Size_X", r.get("Grid_Size","?"))}')
PY
