"""ctypes binding of ``libgs_hip.so`` -- the same C ABI (``include/gs_hip.h``) that the
reference-side Rust shim binds (``rust/compute_hip/src/ffi.rs``).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this
module raises (``GsError``).  Nothing here imports ``oracle``.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GS_HIP_LIBRARY names another build of the same ABI (A/B timing of kernel variants, tools/ab_build.py);
# it is loaded instead of, never as a fallback for, the in-tree library.
LIB_PATH = os.environ.get("GS_HIP_LIBRARY") or os.path.join(_HERE, "libgs_hip.so")

GS_OK = 0
GS_ERR_INVALID = -1
GS_ERR_HIP = -2
GS_ERR_RCCL = -3
GS_ERR_NO_DEVICE = -4
GS_ERR_UNSUPPORTED = -5
GS_ERR_NOMEM = -6

GS_MATH_STRICT, GS_MATH_FUSED = 0, 1
GS_KERNEL_AUTO, GS_KERNEL_SIMPLE, GS_KERNEL_STREAM, GS_KERNEL_TB, GS_KERNEL_LDS, GS_KERNEL_TILE, GS_KERNEL_WINDOW = 0, 1, 2, 3, 4, 5, 6
GS_BOUNDARY_CLIPPED, GS_BOUNDARY_ZERO_HALO = 0, 1
GS_UNIQUE_ID_BYTES = 128

# Every symbol include/gs_hip.h declares; tests check that the library exports them all.
EXPORTS = (
    "gs_default_params", "gs_default_options", "gs_abi_version", "gs_last_error",
    "gs_device_count", "gs_get_unique_id", "gs_ctx_create", "gs_ctx_destroy",
    "gs_ctx_set_params", "gs_field_create", "gs_field_destroy", "gs_field_shape",
    "gs_field_local_rows", "gs_field_raw_shape", "gs_field_fill", "gs_field_fill_slice",
    "gs_field_finalize", "gs_field_upload", "gs_field_download", "gs_field_device_ptr",
    "gs_step", "gs_run", "gs_sync", "gs_timer_start", "gs_timer_stop", "gs_ctx_info",
    "gs_host_alloc", "gs_host_free", "gs_field_download_async", "gs_download_wait",
    "gs_ctx_get_tuned", "gs_ctx_set_tuned", "gs_ctx_comm_info", "gs_field_colormap",
    "gs_ctx_stats", "gs_ctx_set_pass_timing", "gs_field_mark_written", "gs_rccl_selftest",
    "gs_runtime_info", "gs_fields_place", "gs_debug_dyn_lds_key", "gs_debug_window_plan",
    "gs_debug_place_stats", "gs_debug_exchange_probe_create", "gs_debug_exchange_probe_run",
    "gs_debug_exchange_probe_destroy", "gs_download_wait_but",
)


class GsError(RuntimeError):
    """A non-zero ``gs_status`` (the Rust shim maps the same codes to ``HipError``)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"gs_hip error {code}: {message}")
        self.code = code
        self.message = message


class GsParams(ctypes.Structure):
    """``gs_params`` = ``Parameters`` (data/src/parameters.rs:13-33)."""

    _fields_ = [
        ("w", (ctypes.c_float * 3) * 3),
        ("du", ctypes.c_float),
        ("dv", ctypes.c_float),
        ("feed", ctypes.c_float),
        ("kill", ctypes.c_float),
        ("dt", ctypes.c_float),
    ]


class GsOptions(ctypes.Structure):
    _fields_ = [
        ("math", ctypes.c_int32),
        ("kernel", ctypes.c_int32),
        ("rows_per_block", ctypes.c_int32),
        ("fuse_steps", ctypes.c_int32),
        ("use_graph", ctypes.c_int32),
        ("pitch_pad", ctypes.c_int32),
        ("split", ctypes.c_int32),
        ("general_kernels", ctypes.c_int32),
        ("cols_per_lane", ctypes.c_int32),
        ("boundary", ctypes.c_int32),
        ("no_tune", ctypes.c_int32),
        ("tile_shape", ctypes.c_int32),
        ("share_taps", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 3),
    ]


class GsStats(ctypes.Structure):
    """``gs_stats`` (include/gs_hip.h)."""

    _fields_ = [
        ("passes", ctypes.c_uint64),
        ("steps", ctypes.c_uint64),
        ("launches", ctypes.c_uint64),
        ("ghost_refreshes", ctypes.c_uint64),
        ("timed_passes", ctypes.c_uint64),
        ("halo_ms", ctypes.c_float),
        ("interior_ms", ctypes.c_float),
        ("halo_exposed_ms", ctypes.c_float),
        ("reserved", ctypes.c_float),
        ("window_fallbacks", ctypes.c_uint64),
    ]


_lib = None


def load() -> ctypes.CDLL:
    """Load ``libgs_hip.so`` (building is ``__graft_entry__.build()``'s job); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GsError(GS_ERR_NO_DEVICE,
                      f"{LIB_PATH} is missing: build it with `python -m grayscott_amd._build` "
                      "(there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, u64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64, ctypes.c_float
    P = ctypes.POINTER
    sig = {
        "gs_default_params": (None, [P(GsParams)]),
        "gs_default_options": (None, [P(GsOptions)]),
        "gs_abi_version": (i32, []),
        "gs_last_error": (ctypes.c_char_p, []),
        "gs_device_count": (i32, [P(i32)]),
        "gs_get_unique_id": (i32, [vp]),
        "gs_ctx_create": (i32, [P(vp), P(GsParams), P(GsOptions), P(i32), i32, i32, i32, vp]),
        "gs_ctx_destroy": (i32, [vp]),
        "gs_ctx_set_params": (i32, [vp, P(GsParams)]),
        "gs_field_create": (i32, [vp, P(vp), u64, u64]),
        "gs_field_destroy": (i32, [vp, vp]),
        "gs_field_shape": (i32, [vp, P(u64), P(u64)]),
        "gs_field_local_rows": (i32, [vp, P(u64), P(u64)]),
        "gs_field_raw_shape": (i32, [vp, P(u64), P(u64)]),
        "gs_field_fill": (i32, [vp, vp, f32]),
        "gs_field_fill_slice": (i32, [vp, vp, u64, u64, u64, u64, f32]),
        "gs_field_finalize": (i32, [vp, vp]),
        "gs_field_upload": (i32, [vp, vp, vp]),
        "gs_field_download": (i32, [vp, vp, vp]),
        "gs_field_device_ptr": (i32, [vp, i32, P(vp), P(u64), P(u64), P(u64), P(i32)]),
        "gs_step": (i32, [vp, vp, vp, vp, vp]),
        "gs_run": (i32, [vp, vp, vp, vp, vp, u64, P(i32)]),
        "gs_sync": (i32, [vp]),
        "gs_timer_start": (i32, [vp]),
        "gs_timer_stop": (i32, [vp, P(f32)]),
        "gs_ctx_info": (i32, [vp, ctypes.c_char_p, ctypes.c_size_t, P(u64)]),
        "gs_host_alloc": (i32, [P(vp), u64]),
        "gs_host_free": (i32, [vp]),
        "gs_field_download_async": (i32, [vp, vp, vp]),
        "gs_download_wait": (i32, [vp]),
        "gs_ctx_get_tuned": (i32, [vp, u64, u64, P(i32), P(i32), P(i32), P(i32)]),
        "gs_ctx_set_tuned": (i32, [vp, u64, u64, i32, i32, i32, i32]),
        "gs_ctx_comm_info": (i32, [vp, P(i32), P(i32), P(i32)]),
        "gs_field_colormap": (i32, [vp, vp, f32, vp, i32, vp]),
        "gs_ctx_stats": (i32, [vp, P(GsStats)]),
        "gs_ctx_set_pass_timing": (i32, [vp, i32]),
        "gs_field_mark_written": (i32, [vp, vp]),
        "gs_rccl_selftest": (i32, [i32, u64]),
        "gs_runtime_info": (i32, [i32, ctypes.c_char_p, ctypes.c_size_t]),
        "gs_fields_place": (i32, [vp, P(vp), i32, P(f32), P(f32)]),
        "gs_debug_place_stats": (i32, [vp, P(u64), P(u64)]),
        "gs_download_wait_but": (i32, [vp, i32]),
        "gs_debug_exchange_probe_create": (i32, [i32, i32, i32, u64, P(vp)]),
        "gs_debug_exchange_probe_run": (i32, [vp, P(f32), P(f32)]),
        "gs_debug_exchange_probe_destroy": (i32, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != GS_OK:
        msg = load().gs_last_error()
        raise GsError(status, msg.decode(errors="replace") if msg else "unknown error")


def default_params() -> GsParams:
    p = GsParams()
    load().gs_default_params(ctypes.byref(p))
    return p


def default_options() -> GsOptions:
    o = GsOptions()
    load().gs_default_options(ctypes.byref(o))
    return o


def device_count() -> int:
    n = ctypes.c_int32(0)
    status = load().gs_device_count(ctypes.byref(n))
    return int(n.value) if status == GS_OK else 0


def rccl_selftest(device: int = 0, floats: int = 1 << 16) -> None:
    """``gs_rccl_selftest``: raises ``GsError`` when RCCL cannot be loaded or a one-rank send / receive fails."""
    check(load().gs_rccl_selftest(device, floats))


def runtime_info(load_rccl: bool = False) -> dict:
    """``gs_runtime_info``: the HIP runtime and the RCCL this process's libgs_hip.so is bound to (paths, versions)."""
    import json

    buf = ctypes.create_string_buffer(2048)
    check(load().gs_runtime_info(1 if load_rccl else 0, buf, len(buf)))
    return json.loads(buf.value.decode())


def get_unique_id() -> bytes:
    buf = ctypes.create_string_buffer(GS_UNIQUE_ID_BYTES)
    check(load().gs_get_unique_id(buf))
    return buf.raw
