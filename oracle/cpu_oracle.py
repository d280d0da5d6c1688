"""ctypes binding of ``libgs_oracle.so`` (oracle/gs_oracle.c).  TEST INFRASTRUCTURE ONLY.

The functions mirror the reference items they restate:

* ``init_species``  -> ``Species::new``                 data/src/concentration/mod.rs:36-59
* ``step``          -> naive ``perform_step``           compute/naive/src/lib.rs:42-83
* ``run``           -> ``Simulate::perform_steps``      compute/shared/src/cpu.rs:30-42
* ``set_ftz``       -> ``DenormalsFlusher``             compute/shared/src/lib.rs:161-180
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(ctypes.Structure):
    """``Parameters`` (data/src/parameters.rs:13-33) as a plain C struct."""

    _fields_ = [
        ("w", (ctypes.c_float * 3) * 3),
        ("du", ctypes.c_float),
        ("dv", ctypes.c_float),
        ("feed", ctypes.c_float),
        ("kill", ctypes.c_float),
        ("dt", ctypes.c_float),
    ]

    def weights(self) -> np.ndarray:
        return np.array([[self.w[i][j] for j in range(3)] for i in range(3)], dtype=np.float32)

    def set_weights(self, w) -> None:
        for i in range(3):
            for j in range(3):
                self.w[i][j] = float(w[i][j])


def build(force: bool = False) -> None:
    """Compile the checker libraries with the committed Makefile (gcc only)."""
    targets = [os.path.join(_HERE, n) for n in ("libgs_oracle.so", "libgs_cpu_parallel.so")]
    if force or not all(os.path.exists(t) for t in targets):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgs_oracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        pp = ctypes.POINTER(Params)
        sz = ctypes.c_size_t
        lib.gs_oracle_default_params.argtypes = [pp]
        lib.gs_oracle_default_params.restype = None
        lib.gs_oracle_set_ftz.argtypes = [ctypes.c_int]
        lib.gs_oracle_set_ftz.restype = ctypes.c_int
        lib.gs_oracle_seed_ranges.argtypes = [sz, sz, ctypes.POINTER(sz)]
        lib.gs_oracle_seed_ranges.restype = None
        lib.gs_oracle_init.argtypes = [fp, fp, sz, sz]
        lib.gs_oracle_init.restype = None
        lib.gs_oracle_step_rows.argtypes = [fp, fp, fp, fp, sz, sz, pp, sz, sz, ctypes.c_int,
                                            ctypes.c_int]
        lib.gs_oracle_step_rows.restype = None
        lib.gs_oracle_run.argtypes = [fp, fp, fp, fp, sz, sz, pp, sz, ctypes.c_int, ctypes.c_int]
        lib.gs_oracle_run.restype = ctypes.c_int
        lib.gs_oracle_set_boundary.argtypes = [ctypes.c_int]
        lib.gs_oracle_set_boundary.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU
    box shows 256 CPUs to a container that is allowed 16; OpenMP would start 256 threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _threads(nthreads: int, cells: int) -> int:
    """0 = automatic: one thread per 16k cells, at most the usable CPUs."""
    if nthreads > 0:
        return nthreads
    return max(1, min(usable_cpus(), cells // 16384))


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def default_params() -> Params:
    p = Params()
    _lib().gs_oracle_default_params(ctypes.byref(p))
    return p


def set_ftz(on: bool) -> bool:
    """Set/clear MXCSR.FTZ on the calling thread; returns the previous state."""
    return bool(_lib().gs_oracle_set_ftz(int(on)))


def seed_ranges(rows: int, cols: int):
    out = (ctypes.c_size_t * 4)()
    _lib().gs_oracle_seed_ranges(rows, cols, out)
    return (int(out[0]), int(out[1])), (int(out[2]), int(out[3]))


def init_species(rows: int, cols: int):
    """``Species::new([rows, cols])`` -> (U, V) as dense float32 arrays."""
    u = np.empty((rows, cols), np.float32)
    v = np.empty((rows, cols), np.float32)
    _lib().gs_oracle_init(_fp(u), _fp(v), rows, cols)
    return u, v


def step_rows(u, v, out_u, out_v, params: Params, r0: int, r1: int, ftz: bool = True,
              nthreads: int = 0) -> None:
    """One naive step over output rows [r0, r1) of dense [rows, cols] arrays."""
    rows, cols = u.shape
    assert v.shape == u.shape == out_u.shape == out_v.shape
    _lib().gs_oracle_step_rows(_fp(u), _fp(v), _fp(out_u), _fp(out_v), rows, cols,
                               ctypes.byref(params), r0, r1, int(ftz), _threads(nthreads, rows * cols))


CLIPPED, ZERO_HALO = 0, 1   # boundary rules: naive's clipped window (parity target) / zero halo


class _Boundary:
    """``with _Boundary(rule):`` runs the enclosed steps with that boundary rule."""

    def __init__(self, rule: int):
        self.rule = rule

    def __enter__(self):
        self.was = _lib().gs_oracle_set_boundary(self.rule)

    def __exit__(self, *exc):
        _lib().gs_oracle_set_boundary(self.was)


def step(u, v, params: Params | None = None, ftz: bool = True, nthreads: int = 0, boundary: int = CLIPPED):
    """One step of the whole grid; returns new (U, V)."""
    params = params or default_params()
    u = np.ascontiguousarray(u, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    ou, ov = np.empty_like(u), np.empty_like(v)
    with _Boundary(boundary):
        step_rows(u, v, ou, ov, params, 0, u.shape[0], ftz, nthreads)
    return ou, ov


def run(u, v, steps: int, params: Params | None = None, ftz: bool = True, nthreads: int = 0,
        boundary: int = CLIPPED):
    """``perform_steps``: returns (U, V) after ``steps`` steps (inputs are not modified)."""
    params = params or default_params()
    u0 = np.array(u, np.float32, order="C", copy=True)
    v0 = np.array(v, np.float32, order="C", copy=True)
    u1, v1 = np.empty_like(u0), np.empty_like(v0)
    rows, cols = u0.shape
    with _Boundary(boundary):
        slot = _lib().gs_oracle_run(_fp(u0), _fp(u1), _fp(v0), _fp(v1), rows, cols,
                                    ctypes.byref(params), steps, int(ftz), _threads(nthreads, rows * cols))
    return (u1, v1) if slot else (u0, v0)
