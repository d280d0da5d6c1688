#!/usr/bin/env python3
"""Summary of tools/profile_kernel.sh's passes for one kernel (name contains NEEDLE):

    python tools/summarize_kernel_profile.py TAG NEEDLE [--counters-entry]

writes profiles/TAG_kernel_stats.csv (rocprofv3's stats, verbatim) and profiles/TAG_summary.md, and with
--counters-entry adds / replaces the kernel's entry in profiles/counters.json (per launch: HBM bytes, SQ_INSTS_VALU,
rocprofv3's average duration).  HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE from
separate passes, both in KiB; FETCH_SIZE under-reports wide coalesced reads by exactly half on gfx950 (the kernels here
read with 8- and 16-byte accesses per lane)."""
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path, needle):
    vals = {}
    if not os.path.exists(path):
        return vals
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row["Kernel_Name"]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: statistics.median(v) for k, v in vals.items()}


def main():
    tag, needle = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    line = json.loads(open(os.path.join(src, "unprofiled.json")).read().strip().splitlines()[-1])
    prof = json.loads(open(os.path.join(src, "stats.json")).read().strip().splitlines()[-1])
    stats = os.path.join(src, "stats", "run_kernel_stats.csv")
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    kern = max((r for r in rows if needle in r["Name"]), key=lambda r: float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["Calls"]))
    avg_ms = float(kern["AverageNs"]) / 1e6
    # the MEDIAN duration of the kernel's dispatches (the tuning run before the timed calls is a launch of another
    # length): from the kernel trace of the same pass
    trace = os.path.join(src, "stats", "run_kernel_trace.csv")
    if os.path.exists(trace):
        durs = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(trace)) if needle in r["Kernel_Name"]]
        if durs:
            avg_ms = statistics.median(durs)
    c = {}
    for name in ("fetch", "write", "a", "b", "c"):
        c.update(counters(os.path.join(src, name, "run_counter_collection.csv"), needle))
    cells = line["rows"] * line["cols"]
    launches = max(1.0, line["launches_per_call"])
    steps_per_launch = line["steps_per_call"] / launches
    cell_steps = cells * steps_per_launch
    inp = line.get("input", "Species::new")
    out = [f"# rocprofv3 summary `{tag}` — `{line['kernel']}` on {line['rows']} x {line['cols']}, input: {inp}", "",
           f"Program: `python3 tools/run_steps.py --rows {line['rows']} --cols {line['cols']} --steps {line['steps_per_call']} "
           f"--calls {line['calls']}{' --developed' if inp == 'developed' else ''}` (planes placed by the library's default: probe pass "
           f"{line.get('placement')} ms before / after; {launches:g} launch(es) of `{needle}` per call, {steps_per_launch:g} time steps per launch); "
           f"un-profiled: **{line['Mcells_steps_per_s']:,.0f} Mcells×steps/s**, under `--kernel-trace --stats`: {prof['Mcells_steps_per_s']:,.0f}.", "",
           "| kernel | calls | avg ms | min ms | max ms | % of GPU time |", "|---|---|---|---|---|---|"]
    for r in rows:
        out.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['MinNs'])/1e6:.4f} | "
                   f"{float(r['MaxNs'])/1e6:.4f} | {float(r['Percentage']):.2f} |")
    out += ["", f"`{needle}`, per launch ({cell_steps:,.0f} cell-steps; median duration {avg_ms:.4f} ms = "
            f"{cell_steps / avg_ms / 1e3:,.0f} Mcells×steps/s under the profiler):", ""]
    entry = {"kernel": line["kernel"].split("@")[0], "rows": line["rows"], "cols": line["cols"], "launch_ms": avg_ms,
             "steps_per_pass": steps_per_launch, "rows_per_unit": (line.get("tuned") or [0])[0], "cols_per_lane": (line.get("tuned") or [0, 0, 0])[2],
             "traffic": None, "valu_insts": None, "source": f"profiles/{tag}_summary.md", "input": inp}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
        entry["traffic"] = rd + wr
        algo = 16.0 * cell_steps
        out += [f"* FETCH_SIZE {c['FETCH_SIZE']:,.0f} KiB -> reads = 2 x FETCH_SIZE = **{rd / 2**20:,.1f} MiB** (gfx950 half-count correction); "
                f"WRITE_SIZE {c['WRITE_SIZE']:,.0f} KiB -> writes = **{wr / 2**20:,.1f} MiB**",
                f"* HBM traffic {(rd + wr) / 2**20:,.1f} MiB = **{(rd + wr) / algo:.4f} x the algorithmic 16 B per cell-step** "
                f"({algo / 2**20:,.0f} MiB); HBM-side rate {(rd + wr) / (avg_ms * 1e-3) / 1e9:,.0f} GB/s = {(rd + wr) / (avg_ms * 1e-3) / 8e12:.3f} of 8 TB/s"]
    if "SQ_INSTS_VALU" in c:
        entry["valu_insts"] = c["SQ_INSTS_VALU"]
        out.append(f"* SQ_INSTS_VALU {c['SQ_INSTS_VALU'] / 1e6:,.1f} M wave-instructions = **{c['SQ_INSTS_VALU'] * 64 / cell_steps:.1f} "
                   f"instruction-lanes per cell-step**; {c['SQ_INSTS_VALU'] * 64 / (avg_ms * 1e-3) / 1e12:.1f} T lane-ops/s = "
                   f"**{c['SQ_INSTS_VALU'] * 64 / (avg_ms * 1e-3) / 78.6432e12:.3f} of the plain-f32 VALU roof** (78.64 T lane-ops/s) at the stats pass's duration")
        for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"):
            if k in c:
                out.append(f"* {k} {c[k]:,.0f} ({c[k] / c['SQ_INSTS_VALU']:.4f} per VALU instruction)")
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        out.append(f"* GRBM_GUI_ACTIVE {cyc / 1e6:.2f} M cycles per launch (sum over 8 XCDs / 8): VALU issue = "
                   f"**{200.0 * c['SQ_INSTS_VALU'] / (cyc * 1024):.0f} % of the issue slots** (a wave64 f32 op holds a SIMD for 2 cycles)")
    if "SQ_WAVE_CYCLES" in c:
        for k, what in (("SQ_WAIT_ANY", "wave-cycles parked on `s_waitcnt`"), ("SQ_WAIT_INST_ANY", "wave-cycles waiting for an issue slot / dependency"),
                        ("SQ_WAIT_INST_LDS", "wave-cycles waiting to issue an LDS instruction")):
            if k in c:
                out.append(f"* {k} / SQ_WAVE_CYCLES = **{100 * c[k] / c['SQ_WAVE_CYCLES']:.0f} %** ({what})")
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"* SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = **{100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} %** of the LDS pipe's "
                   f"active cycles are bank conflicts ({c['SQ_LDS_BANK_CONFLICT']:,.0f} of {c['SQ_LDS_IDX_ACTIVE']:,.0f})")
    out.append("")
    out.append("Raw medians per launch: " + ", ".join(f"{k} {v:,.0f}" for k, v in sorted(c.items())))
    out.append("")
    open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(out))
    print("\n".join(out))
    if "--counters-entry" in sys.argv:
        cpath = os.path.join(dst, "counters.json")
        try:
            data = json.load(open(cpath))
        except (OSError, ValueError):
            data = []
        data = [e for e in data if not (e.get("kernel") == entry["kernel"] and e.get("rows") == entry["rows"] and e.get("cols") == entry["cols"]
                                        and e.get("rows_per_unit") == entry["rows_per_unit"]
                                        and e.get("input", "Species::new") == entry["input"])]
        data.append(entry)
        data.sort(key=lambda e: (e["kernel"], e["rows"], e["cols"], e.get("rows_per_unit") or 0, e.get("input", "Species::new")))
        json.dump(data, open(cpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
