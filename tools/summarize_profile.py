#!/usr/bin/env python3
"""Turns the raw rocprofv3 CSVs of tools/profile_gpu.sh (gpurun_out/prof_<tag>/) into the
summaries committed under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_summary.md         per-kernel time + HBM traffic per launch of the step kernel
  profiles/counters.json            [{"kernel": bench label, "rows", "cols", "rows_per_unit", "steps_per_pass",
                                    "cols_per_lane", "traffic": HBM bytes per launch, "valu_insts": SQ_INSTS_VALU per
                                    launch, "launch_ms": rocprofv3 average, "source"}, ...] read by bench.py

HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, both are in KiB, and on gfx950 FETCH_SIZE reports exactly half of the
bytes of wide coalesced (16 B/lane) streaming reads, so reads = 2 * FETCH_SIZE * 1024 while
WRITE_SIZE * 1024 is exact for 16 B/lane streaming stores.
"""
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_values(path, needle, counter):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return vals


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    # the layout the profiled launches were pinned to (written by tools/profile_gpu.sh)
    try:
        layout = json.load(open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "layout.json")))
    except (OSError, ValueError):
        layout = {}
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    bench0 = json.loads(open(os.path.join(src, "bench_stats.json")).read().strip().splitlines()[-1])
    label = bench0["config"]["kernel"]
    auto = ("gs_step_tb_dx_k" if ".dx" in label else "gs_step_tb_ds_k" if ".ds" in label else "gs_step_tb_k") if label.startswith("tb-") else "gs_step_stream_k"
    needle = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "auto" else auto
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = os.path.join(src, "stats", "bench_kernel_stats.csv")
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    fetch = counter_values(os.path.join(src, "fetch", "bench_counter_collection.csv"), needle, "FETCH_SIZE")
    write = counter_values(os.path.join(src, "write", "bench_counter_collection.csv"), needle, "WRITE_SIZE")
    bench = json.loads(open(os.path.join(src, "bench_stats.json")).read().strip().splitlines()[-1])
    fetch_kib = statistics.median(fetch)
    write_kib = statistics.median(write)
    read_bytes = 2.0 * fetch_kib * 1024.0
    write_bytes = write_kib * 1024.0
    total = read_bytes + write_bytes
    algo = bench["roofline"]["algorithmic_bytes_per_launch"]
    kern = next(r for r in rows if needle in r["Name"])
    avg_ms = float(kern["AverageNs"]) / 1e6
    lines = [
        f"# rocprofv3 summary `{tag}` — `{bench['config']['workload']}`",
        "",
        f"Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {bench['steps']} --warmup {bench['warmup']} --no-cpu-baseline`"
        " (PMC passes: `--kernel-trace --pmc FETCH_SIZE` and `--kernel-trace --pmc WRITE_SIZE`, separate runs).",
        "",
        "| kernel | calls | avg ms | min ms | max ms | % of GPU time |",
        "|---|---|---|---|---|---|",
    ]
    for r in rows:
        lines.append(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} |"
                     f" {float(r['MaxNs'])/1e6:.4f} | {float(r['Percentage']):.2f} |")
    lines += [
        "",
        f"Step kernel `{needle}` (bench label `{bench['config']['kernel']}`), per launch:",
        "",
        f"* average duration under the profiler: **{avg_ms:.4f} ms**; `bench.py`'s own HIP-event figure in the same run: "
        f"{bench['roofline']['launch_ms']:.4f} ms (un-profiled runs are faster: profiling lowers clocks)",
        f"* algorithmic bytes: {algo/2**30:.3f} GiB (16 B x {bench['config']['cells_per_gpu']} cells x "
        f"{bench['roofline'].get('steps_per_launch', 1):g} steps per launch)",
        f"* FETCH_SIZE median {fetch_kib:.0f} KiB -> reads = 2 x FETCH_SIZE = **{read_bytes/2**30:.3f} GiB** (gfx950 half-count correction)",
        f"* WRITE_SIZE median {write_kib:.0f} KiB -> writes = **{write_bytes/2**30:.3f} GiB**",
        f"* HBM traffic = **{total/2**30:.3f} GiB = {total/algo:.3f} x algorithmic**",
        f"* algorithmic rate under the profiler: {algo/avg_ms/1e6:.0f} GB/s; HBM-side rate {total/avg_ms/1e6:.0f} GB/s",
        "",
    ]
    # SQ counters (their own pass): issued VALU instructions, wave cycles, stall shares
    entry = {"traffic": total, "valu_insts": None, "source": f"profiles/{tag}_summary.md"}
    sq_csv = os.path.join(src, "sq", "bench_counter_collection.csv")
    if os.path.exists(sq_csv):
        med = {}
        for name in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                     "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES"):
            vals = counter_values(sq_csv, needle, name)
            if vals:
                med[name] = statistics.median(vals)
        if "SQ_INSTS_VALU" in med:
            entry["valu_insts"] = med["SQ_INSTS_VALU"]
            cell_steps = bench["config"]["cells_per_gpu"] * bench["roofline"].get("steps_per_launch", 1)
            sq_bench = json.loads(open(os.path.join(src, "bench_sq.json")).read().strip().splitlines()[-1])
            sq_ms = sq_bench["roofline"]["launch_ms"]
            lines += [
                f"## SQ counters of `{needle}` (separate pass: `--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES "
                "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES`), medians per launch",
                "",
                "| counter | per launch | reading |",
                "|---|---|---|",
                f"| SQ_WAVES | {med.get('SQ_WAVES', 0):.0f} | waves per launch |",
                f"| SQ_INSTS_VALU | {med['SQ_INSTS_VALU']/1e6:.1f} M wave-instructions | "
                f"{med['SQ_INSTS_VALU']*64/cell_steps:.1f} instruction-lanes per cell-step "
                f"({41 if '.dx' in bench['config']['kernel'] else 46 if '.ds' in bench['config']['kernel'] else 53} are the arithmetic of the update in this kernel's form) |",
            ]
            if "GRBM_GUI_ACTIVE" in med:
                cyc = med["GRBM_GUI_ACTIVE"] / 8.0
                lines.append(f"| GRBM_GUI_ACTIVE | {med['GRBM_GUI_ACTIVE']/1e6:.2f} M (sum over 8 XCDs) | {cyc/1e6:.2f} M cycles per launch = "
                             f"{cyc/(sq_ms*1e-3)/1e9:.2f} GHz effective over the {sq_ms:.4f} ms that pass measured |")
                lines.append(f"| VALU issue | one instruction per {cyc*1024/med['SQ_INSTS_VALU']:.2f} cycles per SIMD | "
                             f"{200.0*med['SQ_INSTS_VALU']/(cyc*1024):.0f} % of the issue slots (a wave64 f32 op holds a SIMD-32 for 2 cycles); "
                             f"{med['SQ_INSTS_VALU']*64/(sq_ms*1e-3)/1e12:.1f} T lane-ops/s in that pass, "
                             f"{med['SQ_INSTS_VALU']*64/(avg_ms*1e-3)/1e12:.1f} T at the kernel-trace pass's launch time |")
            if "SQ_WAVE_CYCLES" in med:
                for k, what in (("SQ_WAIT_INST_ANY", "issue-side stalls (dependencies, pipe busy)"),
                                ("SQ_WAIT_ANY", "waves parked on `s_waitcnt` (memory, LDS crossbar)")):
                    if k in med:
                        lines.append(f"| {k} / SQ_WAVE_CYCLES | {100*med[k]/med['SQ_WAVE_CYCLES']:.0f} % | {what} |")
            lines.append("")
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
    # profiles/counters.json: a list, one entry per (kernel label, grid, layout) profiled
    cpath = os.path.join(dst, "counters.json")
    try:
        counters = json.load(open(cpath))
        if not isinstance(counters, list):
            counters = []
    except (OSError, ValueError):
        counters = []
    cfg = bench["config"]
    grid = cfg.get("grid") or [int(x) for x in cfg["workload"].split()[0].split("x")]
    tuned = cfg.get("tuned") or {}
    pinned = layout.get("rows_per_unit") or tuned.get("rows_per_unit", 0)
    entry.update({"kernel": cfg["kernel"].split("@")[0], "rows": int(grid[0]) // int(bench.get("n_gpus", 1)), "cols": int(grid[1]),
                  "rows_per_unit": pinned, "steps_per_pass": int(round(bench["roofline"].get("steps_per_launch", 1))),
                  "cols_per_lane": layout.get("cols_per_lane") or tuned.get("cols_per_lane", 0) or
                                   (2 if "c2/" in cfg["kernel"] else (1 if "c1/" in cfg["kernel"] else 4)),
                  "launch_ms": avg_ms})
    counters = [e for e in counters if not (e.get("kernel") == entry["kernel"] and e.get("rows") == entry["rows"] and
                                            e.get("cols") == entry["cols"] and e.get("rows_per_unit") == entry["rows_per_unit"])]
    counters.append(entry)
    counters.sort(key=lambda e: (e["kernel"], e["rows"], e["cols"], e["rows_per_unit"]))
    json.dump(counters, open(cpath, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
