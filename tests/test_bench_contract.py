"""The bench line contract (CPU-only check of the most recent recorded bench output under
profiles/): every key the driver and the judge read is present and well-formed."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_has_the_contract_keys():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_unprofiled.json")))
    assert files, "no recorded bench line under profiles/"
    line = open(files[-1]).read().strip().splitlines()[-1]
    b = json.loads(line)
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["unit"] == "Mcells×steps/s" and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["dtype"] == "f32" and b["data"] == "synthetic" and b["vs_baseline"] is None
    assert "workload" in b["config"] and "model" not in b["config"]
    r = b["roofline"]
    assert r["traffic"] is None or r["traffic"] > 0
    if "algorithmic_GBps" in r:
        # round 2 on: the object names the roof that binds.  The 16 B per cell-step figure of SURVEY 8(d)
        # is kept as a throughput (`algorithmic_*`), never as the fraction of a roof.
        assert r["bound"] in ("valu-issue", "power-capped valu", "hbm of these planes", "hbm")
        assert abs(r["algorithmic_GBps"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) \
            < 1e-6 * r["algorithmic_GBps"]
        assert abs(r["algorithmic_frac"] - r["algorithmic_GBps"] / 8000.0) < 1e-9
        assert 0 < r["useful_valu"] < 1
        if r["bound"] in ("valu-issue", "power-capped valu", "hbm of these planes"):
            assert r["steps_per_launch"] >= 3 and r["unit"] == "T lane-ops/s" and abs(r["peak"] - 78.6432) < 1e-3
            if "frac_definition" in r:
                # round 6 on: `frac` is the rate priced in the reference's form of the update (53 operations per cell-step)
                # against the VALU roof -- recomputable from the line alone, and it does not fall when the kernel issues
                # fewer instructions for the same update; the issued fraction (`valu`, from the committed profile of the
                # layout that ran) and the kernel's own arithmetic (`useful_valu`) stand beside it
                cell_steps = b["config"]["cells_per_gpu"] * r["steps_per_launch"]
                assert abs(r["achieved"] - 53 * cell_steps / (r["launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
                assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and abs(r["frac"] - r["reference_form_valu"]) < 1e-9
                assert 0 < r["useful_valu"] <= r["valu"] < r["frac"] < 1
                assert abs(r["valu"] - r["valu_insts_per_launch"] * 64 / (r["launch_ms"] * 1e-3) / 1e12 / r["peak"]) < 1e-9
                # the counters are those of the layout that ran: nothing is scaled from another unit height
                assert r["valu_source"] == "measured (profile of this layout)", r["valu_source"]
                assert r["counters_layout"]["rows_per_unit"] == b["config"]["tuned"]["rows_per_unit"]
                assert abs(r["peak_at_sustained_clock"] - r["peak"] * r["sclk_MHz_under_load"] / 2400.0) < 1e-6
                assert abs(r["frac_at_sustained_clock"] - r["frac"] * 2400.0 / r["sclk_MHz_under_load"]) < 1e-9
                # `value` is what the library's defaults give (placement by measurement included); the same launches on
                # planes as hipMalloc hands them out stand beside it
                pl = b["config"]["placement"]
                assert pl["default"] is True and pl["max_extra_blocks"] == 12
                # <= 12 GiB held for a moment, unless 4 + 12 consecutive blocks lay in one region (a fresh box can do that):
                # then, with more than half of the device's memory free, up to 48
                assert pl["transient_GiB"] <= (48.1 if pl.get("deep_stage_used") else 12.1), pl
                assert 0.8 * b["value"] < b["value_unplaced"] < 1.05 * b["value"]
                ss_pl = b["single_step"]["placement"]
                assert 0 < ss_pl["chosen_blocks_probe_ms"] <= ss_pl["first_blocks_probe_ms"] * 1.0001
                assert b["single_step"]["frac_of_8TBps"] >= 0.70, b["single_step"]["frac_of_8TBps"]      # north_star's target on the HBM-bound leg
                assert len(b["single_step"]["unplaced_frac_of_8TBps"]) >= 1
                d_r = b["developed_pattern"]["roofline"]
                assert d_r["counters_source"] != r["counters_source"] and d_r["valu_source"] == "measured (profile of this layout)"
            elif r["valu_insts_per_launch"]:
                assert abs(r["achieved"] - r["valu_insts_per_launch"] * 64 / (r["launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
                assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["frac"] == r["valu"] and 0 < r["frac"] < 1
                assert r["useful_valu"] <= r["valu"]
        if r["traffic"]:
            assert abs(r["hbm_physical"] - r["traffic"] / (r["launch_ms"] * 1e-3) / 8e12) < 1e-9 and 0 < r["hbm_physical"] < 1
        assert "value_developed_pattern" in b and 0 < b["value_developed_pattern"] <= 1.05 * b["value"]
        if "repeats" in b:
            # round 3 on: `value` is the median of `repeats` timed regions; the developed pattern is the co-headline
            # with a roofline of its own; the counters name the layout they were profiled with
            assert b["repeats"] >= 5 and b["value_min"] <= b["value"] <= b["value_max"]
            d = b["developed_pattern"]
            assert d["value"] == b["value_developed_pattern"] and d["repeats"] >= 5 and d["value_min"] <= d["value"] <= d["value_max"]
            assert d["roofline"]["bound"] in ("valu-issue", "power-capped valu", r["bound"]) and 0 < d["roofline"]["frac"] < r["frac"] + 0.05
            assert r["frac_source"] and (r["counters_layout"] is None or r["counters_layout"]["rows_per_unit"] > 0)
            assert r["profile_launch_ms"] is None or 0.8 < r["profile_launch_ms"] / r["launch_ms"] < 1.25
            assert b["config"]["grid"] == [16384, 16384] and b["config"]["tuned"]["rows_per_unit"] > 0
            if "values" in b:
                # end of round 3: every timed region in order; whole untimed passes between the W warm-up steps and
                # the first region (an odd W ends in a single-step launch that costs the next 20 ms 1-5 %)
                assert len(b["values"]) == b["repeats"] and abs(sorted(b["values"])[len(b["values"]) // 2] - b["value"]) <= 1
                assert b["untimed_steps_after_warmup"] % 12 == 0 and b["untimed_steps_after_warmup"] >= 24
                assert min(b["values"]) > 0.97 * b["value"]       # no region of the recorded run stands out
                if "verified" in b:
                    # round 4 on: the line proves its own work -- the timed planes equal an independent replay with the
                    # single-step kernel (both inputs), whose timing is the HBM-bound leg north_star asks for; energy
                    # per cell-step from the card's counter; the closing barrier is outside the timed wall
                    v = b["verified"]
                    assert v["equal"] is True and v["steps"] >= b["untimed_steps_before_first_region"] + b["steps"] * b["repeats"]
                    assert v["developed_pattern"]["equal"] is True and v["developed_pattern"]["steps"] >= 4000
                    ss = b["single_step"]
                    assert ss["kernel"].startswith("stream") and len(ss["values"]) == 5
                    assert abs(ss["hbm_GBps"] - 16 * 16384 * 16384 / (ss["launch_ms"] * 1e-3) / 1e9) < 1e-6 * ss["hbm_GBps"]
                    assert abs(ss["frac_of_8TBps"] - ss["hbm_GBps"] / 8000.0) < 1e-9 and 0.6 < ss["frac_of_8TBps"] < 1.0
                    assert 0.8 < ss["value"] * 1e6 * 16 / 1e9 / ss["hbm_GBps"] <= 1.0005      # wall rate <= event rate
                    assert r["bound"] in ("valu-issue", "power-capped valu", "hbm of these planes")
                    if "hbm_physical_over_single_step_leg" in r:     # end of round 5: the marching kernel feels its planes
                        assert abs(r["hbm_physical_over_single_step_leg"] - r["hbm_physical"] / ss["frac_of_8TBps"]) < 1e-9
                        assert (r["bound"] == "hbm of these planes") == (
                            r["hbm_physical_over_single_step_leg"] >= 0.8 and r["bound"] != "power-capped valu")
                    assert 800 < b["energy_pJ_per_cell_step"] < 3000
                    assert b["developed_pattern"]["energy_pJ_per_cell_step"] > b["energy_pJ_per_cell_step"]
                    assert "closing barrier outside" in b["timing"] and b["value_first_region"] > 0.9 * b["value"]
                    assert b["untimed_steps_before_first_region"] >= b["untimed_steps_after_warmup"] + b["warmup"] + b["steps"]
                    if "runtime" in b:
                        # round 5 on: the line says which HIP runtime (and, for N > 1, which RCCL) the library was bound
                        # to, what every stage cost, how the HBM leg's planes were placed, and prices its rate both in the
                        # kernel's own form of the update (46 instructions with full difference sharing) and in the reference's
                        rt = b["runtime"]
                        assert os.path.basename(rt["hip"]).startswith("libamdhip64.so") and rt["hip_runtime_version"] > 70000000 and rt["torch"]
                        assert rt["rccl"] is None and rt["bootstrap"] is None          # N = 1: nothing loads RCCL
                        assert b["stage_seconds"]["timed"] > 0 and sum(b["stage_seconds"].values()) < 120
                        pl = ss["placement"]
                        if "chosen_blocks_ms_per_step" in pl:      # round 5's form (single-step probes over 4-subsets)
                            assert 0 < pl["chosen_blocks_ms_per_step"] <= pl["first_blocks_ms_per_step"] * 1.0001
                            assert abs(ss["frac_of_8TBps"] - pl["chosen_blocks_frac_of_8TBps"]) < 0.03
                        assert r["useful_valu_per_cell_step"] == (41 if ".dx" in b["config"]["kernel"] else 46 if ".ds" in b["config"]["kernel"] else 53)
                        assert abs(r["reference_form_valu"] - r["useful_valu"] * 53 / r["useful_valu_per_cell_step"]) < 1e-9
                        assert b["config"]["tuned"]["share_taps"] in (True, False, "within lanes", "across lanes", "off")
    else:  # round 1 format
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        # achieved = algorithmic bytes per launch / launch duration
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = b["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # value consistent with ms_per_step on the named workload
    assert "16384x16384" in b["config"]["workload"]
    assert abs(b["value"] - 16384 * 16384 / (b["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * b["value"]
    if files[-1] >= os.path.join(ROOT, "profiles", "r01e"):
        assert b["metric"] == baseline["metric"]
