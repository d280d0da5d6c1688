#!/usr/bin/env python3
"""profiles/<tag>_energy.md from tools/energy_table.py's log (un-profiled: rate, clock, power, energy per
cell-step) and its two rocprofv3 PMC passes (per flavour: issued VALU instructions, effective clock).

    python tools/summarize_energy.py TAG RUN_DIR      (RUN_DIR = gpurun_out/<run>: energy_table.log,
                                                       energy_new/..counter_collection.csv, energy_developed/..)
Effective clock = GRBM_GUI_ACTIVE / 8 / launch duration (MI355X_MICROARCH.md, "DVFS give-back": the counter is the
sum over the 8 XCDs; PMC passes run slower than un-profiled ones -- their clock is quoted as such, not mixed in)."""
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"strict.op.dx": "gs_step_tb_dx_k_strict<4, 4>", "strict.op.ds": "gs_step_tb_ds_k_strict<4, 4>", "strict.op": "gs_step_tb_k_strict<4, 3, 2, 4>",
           "strict": "gs_step_tb_k_strict<4, 0, 2, 4>", "fused": "gs_step_tb_k_fused<4, 0, 2, 4>"}
USEFUL = {"strict.op.dx": 41, "strict.op.ds": 46, "strict.op": 53, "strict": 63, "fused": 47}   # the update's arithmetic as this flavour issues it


def pmc(run_dir, data):
    out = {}
    files = glob.glob(os.path.join(run_dir, f"energy_{data}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return out
    rows = list(csv.DictReader(open(files[0])))
    for flavour, needle in KERNELS.items():
        per = {}
        for r in rows:
            if needle in r["Kernel_Name"]:
                d = per.setdefault(r["Dispatch_Id"], {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
                d[r["Counter_Name"]] = float(r["Counter_Value"])
        disp = [d for d in per.values() if "SQ_INSTS_VALU" in d]
        # full passes only (4 steps): the most common instruction count
        if not disp:
            continue
        top = statistics.mode(round(d["SQ_INSTS_VALU"]) for d in disp)
        disp = [d for d in disp if round(d["SQ_INSTS_VALU"]) == top]
        med = lambda k: statistics.median(d[k] for d in disp if k in d)      # noqa: E731
        out[flavour] = {"launches": len(disp), "valu": med("SQ_INSTS_VALU"), "ns": med("ns"),
                        "gui": med("GRBM_GUI_ACTIVE") if any("GRBM_GUI_ACTIVE" in d for d in disp) else None,
                        "busy": med("SQ_BUSY_CYCLES") if any("SQ_BUSY_CYCLES" in d for d in disp) else None}
    return out


def main():
    tag, run_dir = sys.argv[1], sys.argv[2]
    rows = [json.loads(ln) for ln in open(os.path.join(run_dir, "energy_table.log")) if ln.startswith("{")]
    cells = 16384 * 16384
    lines = [f"# What a joule buys: rate, clock, power and energy per cell-step of the production kernel's flavours (`{tag}`)",
             "",
             "16384² f32, 4 steps per pass, 2 columns per lane, 122-row units pinned for every flavour; un-profiled windows of ≥ 4 s of",
             "back-to-back launches, the flavours interleaved in one process (`tools/energy_table.py`), `rocm-smi` sampled beside",
             "them: shader clock and socket power (medians), and the card's accumulated-energy counter between the first and",
             "the last sample taken while the kernel ran.  Issued VALU instructions and the effective clock under the profiler",
             "come from separate `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES",
             "SQ_ACTIVE_INST_VALU` passes of the same tool (`--profile`).  Power cap of the board: 1400 W.",
             "",
             "| input | flavour | Mcells×steps/s | sclk MHz (`rocm-smi`) | socket W (samples / energy counter) | **pJ per cell-step** | issued VALU lane-instr. per cell-step (useful) | pJ per issued lane-instr. | eff. clock in the PMC pass |",
             "|---|---|---|---|---|---|---|---|---|"]
    for data in ("new", "developed"):
        counters = pmc(run_dir, data)
        for flavour in KERNELS:
            rs = [r for r in rows if r["data"] == data and r["flavour"] == flavour]
            if not rs:
                continue
            med = lambda k: statistics.median(r[k] for r in rs if r.get(k) is not None)     # noqa: E731
            pj = med("pJ_per_cell_step")
            c = counters.get(flavour)
            ipc = c["valu"] * 64 / (cells * 4) if c else None
            clk = c["gui"] / 8 / c["ns"] if c and c.get("gui") else None
            lines.append(f"| {'Species::new' if data == 'new' else 'developed pattern'} | {flavour} | {med('Mcells_steps_per_s'):.0f} | "
                         f"{med('sclk_MHz'):.0f} | {med('power_W'):.0f} / {med('energy_W'):.0f} | **{pj:.0f}** | "
                         + (f"{ipc:.1f} ({USEFUL[flavour]})" if ipc else "–") + " | " + (f"{pj / ipc:.1f}" if ipc else "–") + " | "
                         + (f"{clk:.2f} GHz ({c['ns'] / 1e6:.3f} ms per launch)" if clk else "–") + " |")
    open(os.path.join(ROOT, "profiles", f"{tag}_energy_table.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
