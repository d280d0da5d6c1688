"""Independent numpy restatement of the reference's naive step.  TEST INFRASTRUCTURE ONLY.

Written separately from ``gs_oracle.c`` (whole-array masked taps instead of a
per-cell clipped loop) so that the two can pin each other bit for bit; the
reference itself provides no golden values for this path ("parity unpinned").

Follows compute/naive/src/lib.rs:42-83: for output cell (r, c) the window is
rows ``max(r-1,0) .. min(r+2,R)``, cols ``max(c-1,0) .. min(c+2,C)``; the fold
visits it row-major and takes ``weights[i][j]`` with (i, j) counted from the
window's top-left corner (:63-71).  Every numpy float32 op rounds once, like the
Rust scalar ops (no contraction).  MXCSR.FTZ, when set on this thread through
``oracle.set_ftz``, applies to numpy's SSE/AVX loops as well.
"""
from __future__ import annotations

import numpy as np

DEFAULT_WEIGHTS = np.array([[0.25, 0.5, 0.25], [0.5, 0.0, 0.5], [0.25, 0.5, 0.25]], np.float32)


def default_params() -> dict:
    """``Parameters::default()`` (data/src/parameters.rs:72-83, weights :116-122)."""
    f = np.float32
    return dict(w=DEFAULT_WEIGHTS.copy(), du=f(0.1), dv=f(0.05), feed=f(0.014), kill=f(0.054),
                dt=f(1.0))


def init_species(rows: int, cols: int):
    """``Species::new`` (data/src/concentration/mod.rs:36-59)."""
    u = np.ones((rows, cols), np.float32)
    v = np.zeros((rows, cols), np.float32)
    r0, r1 = max(rows * 7 // 16 - 4, 0), max(rows * 8 // 16 - 4, 0)
    c0, c1 = cols * 7 // 16, cols * 8 // 16
    u[r0:r1, c0:c1] = 0.0
    v[r0:r1, c0:c1] = 1.0
    return u, v


def _shifted(a: np.ndarray, di: int, dj: int):
    """(values of a[r+di, c+dj] where that exists, mask of where it exists)."""
    rows, cols = a.shape
    val = np.zeros_like(a)
    ok = np.zeros(a.shape, bool)
    rs, re = max(0, -di), min(rows, rows - di)
    cs, ce = max(0, -dj), min(cols, cols - dj)
    if rs < re and cs < ce:
        val[rs:re, cs:ce] = a[rs + di:re + di, cs + dj:ce + dj]
        ok[rs:re, cs:ce] = True
    return val, ok


def step(u: np.ndarray, v: np.ndarray, params: dict | None = None):
    p = params or default_params()
    w = np.asarray(p["w"], np.float32)
    u = np.asarray(u, np.float32)
    v = np.asarray(v, np.float32)
    rows, cols = u.shape
    # offset of the centre inside the clipped window: 0 on the first row/col, else 1
    oi = (np.arange(rows) > 0).astype(np.intp)[:, None]
    oj = (np.arange(cols) > 0).astype(np.intp)[None, :]
    acc_u = np.zeros_like(u)
    acc_v = np.zeros_like(v)
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            su, ok = _shifted(u, di, dj)
            sv, _ = _shifted(v, di, dj)
            weight = w[np.clip(oi + di, 0, 2), np.clip(oj + dj, 0, 2)]
            weight = np.broadcast_to(weight, u.shape)
            new_u = acc_u + weight * (su - u)
            new_v = acc_v + weight * (sv - v)
            acc_u = np.where(ok, new_u, acc_u)
            acc_v = np.where(ok, new_v, acc_v)
    uv_square = u * v * v
    du = p["du"] * acc_u - uv_square + p["feed"] * (np.float32(1.0) - u)
    dv = p["dv"] * acc_v + uv_square - (np.float32(p["feed"]) + np.float32(p["kill"])) * v
    out_u = u + du * p["dt"]
    out_v = v + dv * p["dt"]
    assert out_u.dtype == np.float32 and out_v.dtype == np.float32
    return out_u, out_v


def run(u, v, steps: int, params: dict | None = None):
    """``perform_steps`` (compute/shared/src/cpu.rs:30-42)."""
    for _ in range(steps):
        u, v = step(u, v, params)
    return u, v



def step_zero_halo(u: np.ndarray, v: np.ndarray, params: dict | None = None):
    """The reference's other boundary rule -- full window, centred weights, zeros outside the grid
    (Vulkan sampler: compute/gpu/naive/src/pipeline.rs:105-113, fold main.comp:37-44; SIMD halo:
    data/src/concentration/simd/mod.rs:281-326) -- with naive's operation order.  Written with a
    zero-padded copy instead of the C checker's per-tap bounds test."""
    p = params or default_params()
    w = np.asarray(p["w"], np.float32)
    u = np.asarray(u, np.float32)
    v = np.asarray(v, np.float32)
    rows, cols = u.shape
    pu = np.zeros((rows + 2, cols + 2), np.float32)
    pv = np.zeros((rows + 2, cols + 2), np.float32)
    pu[1:-1, 1:-1] = u
    pv[1:-1, 1:-1] = v
    acc_u = np.zeros_like(u)
    acc_v = np.zeros_like(v)
    for i in range(3):
        for j in range(3):
            acc_u = acc_u + w[i, j] * (pu[i:i + rows, j:j + cols] - u)
            acc_v = acc_v + w[i, j] * (pv[i:i + rows, j:j + cols] - v)
    uv_square = u * v * v
    du = p["du"] * acc_u - uv_square + p["feed"] * (np.float32(1.0) - u)
    dv = p["dv"] * acc_v + uv_square - (np.float32(p["feed"]) + np.float32(p["kill"])) * v
    out_u = u + du * p["dt"]
    out_v = v + dv * p["dt"]
    assert out_u.dtype == np.float32 and out_v.dtype == np.float32
    return out_u, out_v
