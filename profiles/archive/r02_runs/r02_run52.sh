#!/bin/bash
# rocprofv3 evidence for the mid-size kernels: kernel trace + stats and an SQ counter pass per grid
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02_run52
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for sz in "16 32" "32 64" "128 256" "256 512" "512 1024"; do
  tag=$(echo $sz | tr ' ' x)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$tag" -o p -- python3 "$ROOT/tools/tile_probe.py" $sz 2048 > "$OUT/stats_$tag.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_$tag" -o p -- python3 "$ROOT/tools/tile_probe.py" $sz 2048 > "$OUT/pmc_$tag.log" 2>&1
done
python3 - <<'PY' | tee "$OUT/summary.md"
import csv, glob, statistics, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r02_run52"
print("| grid | kernel | launches | avg µs | min µs | max µs | waves | VALU instr per wave | LDS instr per wave | wave lifetime µs (SQ_WAVE_CYCLES x 4 / 2.4 GHz) | waiting (`s_waitcnt`) | issue stalls |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for tag in ("16x32", "32x64", "128x256", "256x512", "512x1024"):
    st = None
    for f in glob.glob(f"{root}/stats_{tag}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gs_run_" in r["Name"]:
                st = r
    d = collections.defaultdict(list)
    for f in glob.glob(f"{root}/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gs_run_" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: statistics.median(v) for k, v in d.items()}
    if not st or not m:
        print(f"| {tag} | (no data) |")
        continue
    name = st["Name"].split("::")[-1].split("(")[0]
    w = m["SQ_WAVES"]
    print(f"| {tag} | `{name}` | {st['Calls']} | {float(st['AverageNs'])/1e3:.2f} | {float(st['MinNs'])/1e3:.2f} | {float(st['MaxNs'])/1e3:.2f} | {w:.0f} | "
          f"{m['SQ_INSTS_VALU']/w:.0f} | {m['SQ_INSTS_LDS']/w:.0f} | {m['SQ_WAVE_CYCLES']/w*4/2400:.2f} | {100*m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.0f} % | {100*m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.0f} % |")
PY
