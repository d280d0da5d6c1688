"""bench.py's multi-GPU path end to end on a ONE-GPU box (`--rehearsal`: the ranks share GPU 0, the library binds
the shared-memory transport double instead of librccl, torch.distributed runs on gloo), started the way a driver
might start it: `python bench.py --gpus N` with no torchrun around it.

* the launcher path: the parent starts torchrun as a child and relays the one JSON line;
* the line's `verified` object: rank 0 replays the whole grid alone and the ranks' row-block checksums match;
* a forced stall of one rank inside the timed stage: the watchdog prints an error line and the job exits non-zero
  (instead of hanging until the driver's limit).

Multi-GPU spec: SURVEY.md section 8(e); precedent for overlapping sub-grids: compute/shared/src/cpu.rs:111-154.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GS_RCCL_LIBRARY"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e,
                          capture_output=True, text=True, timeout=timeout)


def test_gpus_2_rehearsal_without_torchrun_prints_a_verified_line(built):
    # (--place-candidates 3: every rank places its slab's planes by measurement, as the full-size run does)
    r = _bench(["--gpus", "2", "--rehearsal", "--steps", "24", "--warmup", "5", "--repeats", "3", "--grid", "4096x2048",
                "--place-candidates", "3"])
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and len(line["ranks"]) == 2
    pl = line["config"]["placement"]
    assert pl["default"] is False and pl["max_extra_blocks"] == 3 and pl["species_placed"] >= 1 and pl["probes"] >= 6, pl
    assert 0 < pl["timed_species_probe_ms"][1] <= pl["timed_species_probe_ms"][0] * 1.0001 and pl["extra_blocks_drawn"] <= 3 * 3, pl
    assert line["config"]["grid"] == [4096, 2048] and "REHEARSAL" in line["data"]
    v = line["verified"]
    assert v["equal"] is True and v["mismatching_ranks"] == [] and v["blocks"] >= 4 and v["steps"] > 24, v
    assert v["random_start"] == {"steps": 203, "equal": True, "mismatching_ranks": []}, v     # signal on every seam
    assert "closing barrier outside" in line["timing"]
    # the second route to several GPUs, run by rank 0 alone after the chain: one process, the slabs' ghost rows by
    # device-to-device copies (here: both slabs on GPU 0), checked against a single-slab run
    pc = line["peer_chain"]
    assert "error" not in pc and pc["value"] > 0 and pc["verified"]["equal"] is True and len(pc["values"]) == 5, pc
    # ... and which libraries the ranks were bound to
    rt = line["runtime"]
    assert rt["hip"] and rt["hip_runtime_version"] > 0 and rt["bootstrap"] == "gloo" and rt["rccl_named_by_GS_RCCL_LIBRARY"] is True, rt
    assert line["stage_seconds"]["timed"] > 0
    assert line["value"] > 0 and line["value_first_region"] > 0 and line["untimed_steps_before_first_region"] >= 24 + 5


def test_a_stalled_rank_ends_the_job_with_an_error_line(built):
    r = _bench(["--gpus", "2", "--rehearsal", "--steps", "24", "--warmup", "5", "--repeats", "3", "--grid", "4096x2048"],
               env={"GS_BENCH_FAULT": "stall:1:timed", "GS_BENCH_WATCHDOG_S": "45"}, timeout=600)
    assert r.returncode != 0, (r.stdout[-2000:], r.stderr[-2000:])
    errors = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{") and '"error"' in ln]
    assert errors and all(e["stage"] == "timed" for e in errors), r.stdout[-2000:]
    assert not any('"metric"' in ln for ln in r.stdout.splitlines())      # no bench line from a broken job
