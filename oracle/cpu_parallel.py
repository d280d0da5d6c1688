"""ctypes binding of ``libgs_cpu_parallel.so`` (oracle/gs_cpu_parallel.c): the CPU TIMING
baseline -- a restatement of the reference's ``parallel(block(autovec))`` backend
(compute/parallel, compute/block, compute/autovec; file:line in the C file's header).

MEASUREMENT INFRASTRUCTURE ONLY: used by ``bench.py``'s ``cpu_baseline`` leg and by tests.
It is not the parity target (zero-halo boundary rule, FMA association).
"""
from __future__ import annotations

import ctypes
import glob
import os
import re

import numpy as np

from .cpu_oracle import Params, build, default_params

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgs_cpu_parallel.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        sz = ctypes.c_size_t
        lib.gs_par_width.restype = ctypes.c_int
        lib.gs_par_create.argtypes = [ctypes.POINTER(Params), sz, sz, sz, sz, sz, ctypes.c_int, ctypes.c_int]
        lib.gs_par_create.restype = ctypes.c_void_p
        lib.gs_par_destroy.argtypes = [ctypes.c_void_p]
        lib.gs_par_destroy.restype = None
        lib.gs_par_perform_steps.argtypes = [ctypes.c_void_p, sz]
        lib.gs_par_perform_steps.restype = None
        lib.gs_par_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        lib.gs_par_read.restype = None
        _LIB = lib
    return _LIB


def simd_width() -> int:
    return int(_lib().gs_par_width())


def _parse_size(text: str) -> int:
    m = re.match(r"(\d+)([KMG]?)", text.strip())
    return int(m.group(1)) * {"": 1, "K": 1024, "M": 1024 ** 2, "G": 1024 ** 3}[m.group(2)]


def _count_cpus(cpu_list: str) -> int:
    n = 0
    for part in cpu_list.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            n += int(b) - int(a) + 1
        elif part:
            n += 1
    return max(n, 1)


def cache_sizes_per_thread():
    """(L1d, L2) data-cache bytes per hardware thread of cpu0 -- what hwloc's
    ``smallest_data_cache_sizes_per_thread`` gives the reference (parallel/src/block.rs:20-38).
    Falls back to the reference's own fallbacks (16 KiB L1; L2 = L1) when sysfs has nothing."""
    out = {}
    for d in glob.glob("/sys/devices/system/cpu/cpu0/cache/index*"):
        try:
            typ = open(os.path.join(d, "type")).read().strip()
            if typ not in ("Data", "Unified"):
                continue
            level = int(open(os.path.join(d, "level")).read())
            size = _parse_size(open(os.path.join(d, "size")).read())
            share = _count_cpus(open(os.path.join(d, "shared_cpu_list")).read())
            out[level] = size // share
        except (OSError, ValueError, AttributeError):
            continue
    l1 = out.get(1, 32 * 1024 // 2)
    l2 = out.get(2, l1)
    return l1, l2


class ParallelSimulation:
    """``compute_parallel::Simulation`` restated; Species::new state is created inside."""

    def __init__(self, rows: int, cols: int, params: Params | None = None, num_threads: int = 0,
                 l1_block_size: int | None = None, l2_block_size: int | None = None,
                 seq_block_size: int | None = None, ftz: bool = True):
        l1, l2 = cache_sizes_per_thread()
        self.l1_block_size = l1_block_size or l1 // 2          # MultiCore defaults, block.rs:27-33
        self.l2_block_size = l2_block_size or l2 // 2
        self.seq_block_size = seq_block_size or self.l2_block_size  # parallel/src/lib.rs:73-81
        self.num_threads = num_threads or (os.cpu_count() or 1)
        self.rows, self.cols = rows, cols
        self._p = params or default_params()
        self._h = _lib().gs_par_create(ctypes.byref(self._p), rows, cols, self.l1_block_size,
                                       self.l2_block_size, self.seq_block_size, self.num_threads, int(ftz))
        if not self._h:
            raise ValueError(f"rows ({rows}) must be a multiple of the SIMD width {simd_width()}")

    def perform_steps(self, steps: int) -> None:
        _lib().gs_par_perform_steps(self._h, steps)

    def read(self, species: int) -> np.ndarray:
        out = np.empty((self.rows, self.cols), np.float32)
        _lib().gs_par_read(self._h, species, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        return out

    def close(self) -> None:
        if self._h:
            _lib().gs_par_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()
