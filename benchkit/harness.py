"""bench.py's harness around the timed regions: grids per N, the per-rank watchdog, the torchrun child of
`python bench.py --gpus N`, the CPU share of this process.  No GPU work, nothing of the product path."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


BYTES_PER_CELL_STEP = 16          # read U,V + write U,V, 4 B each
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)
HBM_COPY_CEILING_GBS = 6290.0     # measured float4-copy ceiling, same table
# VALU issue roof for plain f32 ops: 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz (half the
# 157.3 TFLOP/s FMA peak of the same table: the strict kernel issues no FMA)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
# arithmetic the reference's update needs per cell-step when each op is one instruction (taps:
# 4 corners x (sub, mul, add) + 4 sides x (sub with div:2, add), two species; reaction: 13)
USEFUL_VALU_PER_CELL_STEP = 53
# ... and with full difference sharing at 2 columns per lane (grayscott_amd/csrc/gs_march.h: cells_vshare): per lane-row and species
# 14 x 2 + 5 tap instructions instead of 20 x 2, the same 13 for the reaction
USEFUL_VALU_PER_CELL_STEP_SHARED = 46
# ... and with the three differences that cross a lane boundary computed by one lane only (cells_xshare): 82 per lane-row
USEFUL_VALU_PER_CELL_STEP_SHARED_ACROSS = 41
NOMINAL_SCLK_MHZ = 2400.0  # the clock VALU_PEAK_TLANEOPS is priced at


def grid_for(n_gpus: int, scaling: str):
    if scaling == "strong":
        return (65536, 32768) if n_gpus > 4 else (32768, 16384)   # BASELINE configs 5 / 4
    if n_gpus == 8:
        return 65536, 32768       # BASELINE config 5
    return 16384 * n_gpus, 16384  # config 3 (N=1), config 4 (N=2), same cells per GPU


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU
    box exposes 256 logical CPUs but grants a 16-CPU share per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                if text[0] != "max":
                    n = min(n, max(1, int(int(text[0]) / int(text[1]))))
            else:
                quota = int(text[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


class Watchdog:
    """Per-rank stage timer.  `with wd.stage(name, seconds):` arms a bound; a daemon thread that finds it
    exceeded prints ONE JSON line {"error", "rank", "stage", "bound_s"} and ends the process with exit code 3
    (os._exit: the main thread may sit in ncclCommInitRank or a stream wait that never returns).  No restart,
    no re-exec: torchrun sees the non-zero exit and takes the other ranks down.
    GS_BENCH_WATCHDOG_S caps every bound (tests use a few seconds)."""

    EXIT_CODE = 3

    def __init__(self, rank: int = 0, out=None):
        import threading

        self.rank = rank
        self.out = out or sys.stdout
        self._lock = threading.Lock()
        self._stage = None          # (name, deadline, bound)
        cap = os.environ.get("GS_BENCH_WATCHDOG_S", "")
        self._cap = float(cap) if cap else None
        self._thread = threading.Thread(target=self._watch, daemon=True)
        self._thread.start()

    def _watch(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                st = self._stage
            if st and time.monotonic() > st[1]:
                line = json.dumps({"error": f"stage '{st[0]}' exceeded its bound of {st[2]:.0f} s",
                                   "rank": self.rank, "stage": st[0], "bound_s": st[2]})
                try:
                    self.out.write(line + "\n")
                    self.out.flush()
                finally:
                    os._exit(self.EXIT_CODE)

    def stage(self, name: str, seconds: float):
        wd = self
        bound = min(seconds, self._cap) if self._cap else seconds

        class _Stage:
            def __enter__(self_inner):
                with wd._lock:
                    wd._stage = (name, time.monotonic() + bound, bound)
                fault = os.environ.get("GS_BENCH_FAULT", "")       # "stall:RANK:STAGE" (tests)
                if fault.startswith("stall:"):
                    _, r, st = fault.split(":")
                    if int(r) == wd.rank and st == name:
                        time.sleep(1e6)

            def __exit__(self_inner, *exc):
                with wd._lock:
                    wd._stage = None
                return False

        return _Stage()


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: start `torch.distributed.run` as a
    CHILD process -- before this process has touched a GPU or loaded libgs_hip.so -- relay its output (the one
    JSON line) and return its exit code.  Never an exec of this process."""
    import socket
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rehearsal and not env.get("GS_RCCL_LIBRARY"):
        # all ranks share GPU 0: RCCL refuses that, the library binds the shared-memory transport double
        import shutil

        out_dir = os.path.join(ROOT, "gpurun_out", "rehearsal")
        os.makedirs(out_dir, exist_ok=True)
        lib = os.path.join(out_dir, "libshm_transport.so")
        cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        r = subprocess.run([cc, "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                            os.path.join(ROOT, "tests", "cpp", "shm_transport.cpp"), "-o", lib, "-lrt", "-lpthread"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            print("bench.py: building the rehearsal transport failed:\n" + r.stdout + r.stderr, file=sys.stderr)
            return 2
        env["GS_RCCL_LIBRARY"] = lib
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    limit = float(os.environ.get("GS_BENCH_CHILD_TIMEOUT_S", "1500"))
    child = subprocess.Popen(cmd, env=env, start_new_session=True)       # inherits stdout / stderr
    try:
        return child.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal

        print(json.dumps({"error": f"the torchrun child exceeded {limit:.0f} s", "rank": -1, "stage": "child"}))
        try:
            os.killpg(child.pid, signal.SIGKILL)     # the session this process started, nothing else
        except OSError:
            pass
        child.wait()
        return 3
