"""bench.py's harness around the timed regions, without a GPU: the per-rank watchdog (a stage that exceeds its
bound prints one JSON line and the process exits non-zero), the fault hook the GPU tests use to force such a
stall, and the launcher path of `python bench.py --gpus N` (a torchrun CHILD, started before this process touches
a GPU; never an exec)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _python(code, env=None, timeout=120):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_watchdog_ends_a_stage_that_exceeds_its_bound():
    r = _python("import time, bench\nwd = bench.Watchdog(rank=3)\n"
                "with wd.stage('quick', 30):\n    pass\n"
                "with wd.stage('comm-init', 0.5):\n    time.sleep(30)\nprint('not reached')\n")
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["rank"] == 3 and line["stage"] == "comm-init" and "error" in line and line["bound_s"] == 0.5
    assert "not reached" not in r.stdout


def test_watchdog_leaves_a_stage_that_ends_in_time_alone():
    r = _python("import time, bench\nwd = bench.Watchdog(rank=0)\n"
                "with wd.stage('a', 1.0):\n    time.sleep(0.2)\ntime.sleep(1.5)\nprint('done')\n")
    assert r.returncode == 0 and r.stdout.strip() == "done", (r.returncode, r.stdout, r.stderr)


def test_fault_hook_stalls_the_named_rank_and_stage_only():
    code = ("import bench\nwd = bench.Watchdog(rank=1)\n"
            "with wd.stage('init', 2):\n    pass\nprint('init passed', flush=True)\n"
            "with wd.stage('timed', 2):\n    pass\nprint('timed passed')\n")
    r = _python(code, env={"GS_BENCH_FAULT": "stall:1:timed", "GS_BENCH_WATCHDOG_S": "1"})
    assert r.returncode == 3 and "init passed" in r.stdout and "timed passed" not in r.stdout
    assert json.loads(r.stdout.strip().splitlines()[-1])["stage"] == "timed"
    r = _python(code, env={"GS_BENCH_FAULT": "stall:0:timed", "GS_BENCH_WATCHDOG_S": "1"})
    assert r.returncode == 0 and "timed passed" in r.stdout


def test_gpus_n_without_torchrun_starts_a_torchrun_child():
    """No GPU here: every child rank reaches bench.py's own "no GPU visible" exit, which proves that the parent
    started torch.distributed.run with N ranks and relayed its failure as a non-zero exit code."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "0"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("no GPU visible") >= 2, r.stderr[-2000:]
