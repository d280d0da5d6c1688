#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run31
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for sz in "128 256" "512 1024"; do
  tag=$(echo $sz | tr ' ' x)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$tag" -o p -- python3 "$ROOT/tools/tile_probe.py" $sz > "$OUT/stats_$tag.log" 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_$tag" -o p -- python3 "$ROOT/tools/tile_probe.py" $sz > "$OUT/pmc_$tag.log" 2>&1
done
find "$OUT" -name "*kernel_stats.csv" | while read f; do echo "$f"; head -4 "$f"; done
python3 - <<'PY'
import csv, glob, statistics, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/r02_run31/pmc_*/**/*counter_collection.csv', recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tile' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-3] if 'pmc_' in f else f, {k: (statistics.median(v), len(v)) for k, v in d.items()})
PY
