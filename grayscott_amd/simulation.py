"""Host-side mirror of the reference's backend interface, over the C ABI.

Same names, argument meaning and error behaviour as the Rust items they mirror, so that
the parity tests read like tests of a reference backend:

=========================  ==========================================================
here                       reference (/root/reference/...)
=========================  ==========================================================
``Parameters``             data/src/parameters.rs:13-33, ``Default`` :72-83
``HipConcentration``       ``Concentration`` trait, data/src/concentration/mod.rs:198-296
``Evolving`` / ``Species`` data/src/concentration/mod.rs:17-187
``HipArgs``                ``SimulateBase::CliArgs`` (defaults + env), compute/shared/src/lib.rs:20-25
``Simulation``             ``SimulateBase`` / ``SimulateCreate`` / ``Simulate``,
                           compute/shared/src/lib.rs:19-58
=========================  ==========================================================

The Rust shim a maintainer would add (rust/compute_hip) has exactly this shape; this
module is what the Python harness (tests, bench) uses in its place because the image has
no Rust toolchain.  All arithmetic happens in ``libgs_hip.so``; there is no CPU path.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .capi import GsError

STENCIL_WEIGHTS = ((0.25, 0.5, 0.25), (0.5, 0.0, 0.5), (0.25, 0.5, 0.25))  # parameters.rs:116-122

# The reference picks its stencil at compile time with cargo features (data/Cargo.toml:28-58,
# parameters.rs:91-122); here every set is a run-time value of ``Parameters.weights`` (the
# reference's ``weights-runtime`` feature).  Only the power-of-two sets can use GS_MATH_FUSED.
STENCILS = {
    "oono-puri": STENCIL_WEIGHTS,                                                    # default
    "5points": ((0.0, 1.0, 0.0), (1.0, 0.0, 1.0), (0.0, 1.0, 0.0)),                 # weights-5points
    "patrakarttunen": ((1 / 6, 4 / 6, 1 / 6), (4 / 6, 0.0, 4 / 6), (1 / 6, 4 / 6, 1 / 6)),   # weights-patrakarttunen
    "pretty": ((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), (1.0, 1.0, 1.0)),                  # weights-pretty
}


@dataclass
class Parameters:
    """``Parameters`` with ``Parameters::default()`` values (parameters.rs:72-83)."""

    weights: Tuple[Tuple[float, float, float], ...] = STENCIL_WEIGHTS
    diffusion_rate_u: float = 0.1
    diffusion_rate_v: float = 0.05
    feed_rate: float = 0.014
    kill_rate: float = 0.054
    time_step: float = 1.0

    @classmethod
    def with_stencil(cls, name: str, **kw) -> "Parameters":
        """Default parameters with one of the reference's named stencils (``STENCILS``)."""
        return cls(weights=STENCILS[name], **kw)

    def to_c(self) -> capi.GsParams:
        p = capi.GsParams()
        for i in range(3):
            for j in range(3):
                p.w[i][j] = self.weights[i][j]
        p.du, p.dv = self.diffusion_rate_u, self.diffusion_rate_v
        p.feed, p.kill, p.dt = self.feed_rate, self.kill_rate, self.time_step
        return p


def _env_int(name: str, default: int) -> int:
    v = os.environ.get(name)
    return int(v) if v not in (None, "") else default


@dataclass
class HipArgs:
    """Backend ``CliArgs``: every field has a default and an environment variable, as the
    reference requires so that its criterion harness can build a backend from the
    environment alone (compute/shared/src/lib.rs:20-25, benchmark.rs:36-40)."""

    devices: Sequence[int] = field(default_factory=lambda: [
        int(x) for x in os.environ.get("GS_HIP_DEVICES", "0").split(",") if x != ""])
    math: int = field(default_factory=lambda: _env_int("GS_HIP_MATH", capi.GS_MATH_STRICT))
    kernel: int = field(default_factory=lambda: _env_int("GS_HIP_KERNEL", capi.GS_KERNEL_AUTO))
    rows_per_block: int = field(default_factory=lambda: _env_int("GS_HIP_ROWS_PER_BLOCK", 0))
    fuse_steps: int = field(default_factory=lambda: _env_int("GS_HIP_FUSE_STEPS", 0))
    use_graph: int = field(default_factory=lambda: _env_int("GS_HIP_USE_GRAPH", 0))
    pitch_pad: int = field(default_factory=lambda: _env_int("GS_HIP_PITCH_PAD", 0))
    split: int = field(default_factory=lambda: _env_int("GS_HIP_SPLIT", 0))
    general_kernels: int = field(default_factory=lambda: _env_int("GS_HIP_GENERAL_KERNELS", 0))
    cols_per_lane: int = field(default_factory=lambda: _env_int("GS_HIP_COLS_PER_LANE", 0))
    boundary: int = field(default_factory=lambda: _env_int("GS_HIP_BOUNDARY", capi.GS_BOUNDARY_CLIPPED))
    no_tune: int = field(default_factory=lambda: _env_int("GS_HIP_NO_TUNE", 0))
    tile_shape: int = field(default_factory=lambda: _env_int("GS_HIP_TILE_SHAPE", 0))
    share_taps: int = field(default_factory=lambda: _env_int("GS_HIP_SHARE_TAPS", 0))
    # not a gs_options field: the most extra blocks gs_fields_place may draw for a Species that make_species creates
    # (every Species of >= PLACE_MIN_CELLS cells per process on a context with one slab per process; 0 = no placement)
    place_candidates: int = field(default_factory=lambda: _env_int("GS_HIP_PLACE_CANDIDATES", 12))
    rank: int = 0
    world: int = 1
    unique_id: Optional[bytes] = None

    def to_c(self) -> capi.GsOptions:
        o = capi.default_options()
        o.math, o.kernel = self.math, self.kernel
        o.rows_per_block, o.fuse_steps = self.rows_per_block, self.fuse_steps
        o.use_graph, o.pitch_pad = self.use_graph, self.pitch_pad
        o.split = self.split
        o.general_kernels = self.general_kernels
        o.cols_per_lane = self.cols_per_lane
        o.boundary = self.boundary
        o.no_tune = self.no_tune
        o.tile_shape = self.tile_shape
        o.share_taps = self.share_taps
        return o


# make_species places a Species by measurement from this many cells per process on (planes of 256 MiB): below, the
# planes largely stay in the 256 MB last-level cache and where they lie in HBM does not show
PLACE_MIN_CELLS = 1 << 26


class HipContext:
    """``Concentration::Context`` of ``HipConcentration``: devices, streams, row partition
    and (multi-process) the RCCL communicator -- one ``gs_ctx``."""

    def __init__(self, params: Parameters, args: Optional[HipArgs] = None):
        args = args or HipArgs()
        lib = capi.load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        cp, co = params.to_c(), args.to_c()
        devs = (ctypes.c_int32 * len(args.devices))(*args.devices)
        uid = ctypes.create_string_buffer(args.unique_id, capi.GS_UNIQUE_ID_BYTES) \
            if args.unique_id else None
        capi.check(lib.gs_ctx_create(ctypes.byref(self._h), ctypes.byref(cp), ctypes.byref(co),
                                     devs, len(args.devices), args.rank, args.world, uid))
        self.args = args

    @property
    def handle(self):
        if not self._h:
            raise GsError(capi.GS_ERR_INVALID, "context already destroyed")
        return self._h

    def set_params(self, params: Parameters) -> None:
        cp = params.to_c()
        capi.check(self._lib.gs_ctx_set_params(self.handle, ctypes.byref(cp)))

    def sync(self) -> None:
        capi.check(self._lib.gs_sync(self.handle))

    def download_wait(self, in_flight: int = 0) -> None:
        """Wait for the asynchronous downloads enqueued so far (not for later steps); ``in_flight=1``: for all but the
        newest (``gs_download_wait_but``: two images on their way, the PCIe link never idles between them)."""
        capi.check(self._lib.gs_download_wait_but(self.handle, int(in_flight)))

    def timer_start(self) -> None:
        capi.check(self._lib.gs_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = ctypes.c_float(0)
        capi.check(self._lib.gs_timer_stop(self.handle, ctypes.byref(ms)))
        return float(ms.value)

    def get_tuned(self, slab_rows: int, cols: int) -> Tuple[int, int, int, int]:
        """(rows per unit, steps fused per pass, columns per lane, share_taps: 1 = on, 2 = off, 3 = across lanes too) chosen for slabs of
        this shape; zeros when nothing was chosen yet (``gs_ctx_get_tuned``)."""
        a, b, c, d = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        capi.check(self._lib.gs_ctx_get_tuned(self.handle, slab_rows, cols, ctypes.byref(a), ctypes.byref(b),
                                              ctypes.byref(c), ctypes.byref(d)))
        return int(a.value), int(b.value), int(c.value), int(d.value)

    def set_tuned(self, slab_rows: int, cols: int, rows_per_block: int, fuse_steps: int, cols_per_lane: int,
                  share_taps: int = 0) -> None:
        capi.check(self._lib.gs_ctx_set_tuned(self.handle, slab_rows, cols, rows_per_block, fuse_steps,
                                              cols_per_lane, share_taps))

    def place_stats(self) -> Tuple[int, int]:
        """(pair probes timed, extra blocks drawn) by ``gs_fields_place`` on this context so far."""
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        capi.check(self._lib.gs_debug_place_stats(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def comm_info(self) -> Tuple[int, int, int]:
        """(ranks, rank, device) as RCCL reports them for this context's communicator; (0, -1, -1)
        for a single process."""
        a, b, c = ctypes.c_int32(0), ctypes.c_int32(-1), ctypes.c_int32(-1)
        capi.check(self._lib.gs_ctx_comm_info(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def stats(self) -> dict:
        """``gs_ctx_stats``: passes / steps / launches / ghost_refreshes since the context was created and,
        for the passes timed with ``set_pass_timing``, the halo-stream and interior-kernel times (ms)."""
        st = capi.GsStats()
        capi.check(self._lib.gs_ctx_stats(self.handle, ctypes.byref(st)))
        return {name: getattr(st, name) for name, _ in capi.GsStats._fields_ if name != "reserved"}

    def set_pass_timing(self, passes: int) -> None:
        capi.check(self._lib.gs_ctx_set_pass_timing(self.handle, passes))

    def info(self) -> Tuple[str, int]:
        buf = ctypes.create_string_buffer(64)
        n = ctypes.c_uint64(0)
        capi.check(self._lib.gs_ctx_info(self.handle, buf, 64, ctypes.byref(n)))
        return buf.value.decode(), int(n.value)

    def close(self) -> None:
        if self._h:
            self._lib.gs_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pinned_empty(shape: Sequence[int]) -> np.ndarray:
    """float32 array in page-locked host memory (``gs_host_alloc``) for overlapped downloads.
    The allocation is released when the last view of it is garbage-collected."""
    import weakref

    lib = capi.load()
    count = int(np.prod(shape))
    ptr = ctypes.c_void_p()
    capi.check(lib.gs_host_alloc(ctypes.byref(ptr), max(count, 1) * 4))
    buf = (ctypes.c_float * max(count, 1)).from_address(ptr.value)
    root = np.frombuffer(buf, dtype=np.float32)        # every view's .base; does not own the memory
    weakref.finalize(root, lib.gs_host_free, ctypes.c_void_p(ptr.value))
    return root[:count].reshape(shape)


class HipConcentration:
    """One species plane in HBM: the ``Concentration`` implementation of this backend."""

    def __init__(self, context: HipContext, shape: Sequence[int]):
        rows, cols = int(shape[0]), int(shape[1])
        self._ctx = context
        self._h = ctypes.c_void_p()
        capi.check(context._lib.gs_field_create(context.handle, ctypes.byref(self._h), rows, cols))
        self._shape = (rows, cols)

    # -- constructors (Concentration::default / zeros / ones, mod.rs:205-218) -------------
    @classmethod
    def default(cls, context: HipContext, shape) -> "HipConcentration":
        return cls(context, shape)

    @classmethod
    def zeros(cls, context: HipContext, shape) -> "HipConcentration":
        return cls(context, shape)  # planes are created zero-filled

    @classmethod
    def ones(cls, context: HipContext, shape) -> "HipConcentration":
        c = cls(context, shape)
        capi.check(context._lib.gs_field_fill(context.handle, c._h, 1.0))
        return c

    @property
    def handle(self):
        if not self._h:
            raise GsError(capi.GS_ERR_INVALID, "concentration already destroyed")
        return self._h

    def shape(self) -> Tuple[int, int]:
        return self._shape

    def raw_shape(self) -> Tuple[int, int]:
        r, p = ctypes.c_uint64(), ctypes.c_uint64()
        capi.check(self._ctx._lib.gs_field_raw_shape(self.handle, ctypes.byref(r), ctypes.byref(p)))
        return int(r.value), int(p.value)

    def local_rows(self) -> Tuple[int, int]:
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        capi.check(self._ctx._lib.gs_field_local_rows(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def fill_slice(self, context: HipContext, slice_: Sequence[range], value: float) -> None:
        """``fill_slice(ctx, [rows, cols], value)`` with half-open ranges (mod.rs:230-243)."""
        rr, cc = slice_
        capi.check(context._lib.gs_field_fill_slice(context.handle, self.handle, rr.start, rr.stop,
                                                    cc.start, cc.stop, value))

    def finalize(self, context: HipContext) -> None:
        capi.check(context._lib.gs_field_finalize(context.handle, self.handle))

    def upload(self, context: HipContext, host: np.ndarray) -> None:
        """Test/bench helper: overwrite this process's rows from a dense float32 array."""
        r0, r1 = self.local_rows()
        host = np.ascontiguousarray(host, np.float32)
        if host.shape != (r1 - r0, self._shape[1]):
            raise AssertionError(f"upload shape {host.shape} != local shape {(r1 - r0, self._shape[1])}")
        capi.check(context._lib.gs_field_upload(context.handle, self.handle,
                                                host.ctypes.data_as(ctypes.c_void_p)))

    def device_slabs(self):
        """``gs_field_device_ptr`` for every local slab: ``(address, pitch in f32, global row0, rows, device)``
        tuples, top to bottom -- for zero-copy consumers / producers (a producer calls ``mark_written``)."""
        out = []
        lib = self._ctx._lib
        i = 0
        while True:
            ptr, pitch, r0, rows = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
            dev = ctypes.c_int32()
            if lib.gs_field_device_ptr(self.handle, i, ctypes.byref(ptr), ctypes.byref(pitch), ctypes.byref(r0),
                                       ctypes.byref(rows), ctypes.byref(dev)) != capi.GS_OK:
                break
            out.append((int(ptr.value or 0), int(pitch.value), int(r0.value), int(rows.value), int(dev.value)))
            i += 1
        return out

    def torch_views(self):
        """Zero-copy torch views of the local slabs: ``[(global row0, rows, tensor [rows, cols])]`` with the row
        pitch as stride (``__cuda_array_interface__``); for on-device comparisons and producers (a producer calls
        ``mark_written``).  The library's streams are not torch's: ``context.sync()`` before reading."""
        import torch

        class _DeviceArray:
            def __init__(self, address, rows, cols, pitch):
                self.__cuda_array_interface__ = {"shape": (rows, cols), "typestr": "<f4", "data": (address, False),
                                                 "version": 3, "strides": (pitch * 4, 4)}

        cols = self._shape[1]
        return [(row0, rows, torch.as_tensor(_DeviceArray(address, rows, cols, pitch), device=f"cuda:{device}"))
                for address, pitch, row0, rows, device in self.device_slabs() if rows > 0 and cols > 0]

    def mark_written(self, context: HipContext) -> None:
        """``gs_field_mark_written``: cells were written through ``device_slabs`` addresses."""
        capi.check(context._lib.gs_field_mark_written(context.handle, self.handle))

    def make_scalar_view(self, context: HipContext) -> np.ndarray:
        """Owned dense copy of this process's rows (mod.rs:261-275)."""
        r0, r1 = self.local_rows()
        out = np.empty((r1 - r0, self._shape[1]), np.float32)
        self.write_scalar_view(context, out)
        return out

    def write_scalar_view(self, context: HipContext, target: np.ndarray) -> None:
        """``write_scalar_view``; like ``validate_write`` (mod.rs:291-295) a shape mismatch is
        a programming error and asserts."""
        r0, r1 = self.local_rows()
        assert target.shape == (r1 - r0, self._shape[1]), (target.shape, (r1 - r0, self._shape[1]))
        assert target.dtype == np.float32 and target.flags.c_contiguous
        capi.check(context._lib.gs_field_download(context.handle, self.handle,
                                                  target.ctypes.data_as(ctypes.c_void_p)))

    def write_scalar_view_after(self, context: HipContext, target: np.ndarray) -> None:
        """``write_scalar_view_after`` (data/src/concentration/gpu/image/mod.rs:196-206): enqueue
        the download behind the steps already enqueued and return at once; ``target`` (ideally
        from ``pinned_empty``) is valid after ``context.download_wait()``."""
        r0, r1 = self.local_rows()
        assert target.shape == (r1 - r0, self._shape[1]), (target.shape, (r1 - r0, self._shape[1]))
        assert target.dtype == np.float32 and target.flags.c_contiguous
        capi.check(context._lib.gs_field_download_async(context.handle, self.handle,
                                                        target.ctypes.data_as(ctypes.c_void_p)))

    def colormap(self, context: HipContext, palette: np.ndarray, scale: float = 2.0) -> np.ndarray:
        """The pixels ``data-to-pics`` makes of this plane (data-to-pics/src/main.rs:139-144):
        ``palette[clamp(floor(scale * value * n), 0, n - 1)]`` as uint8 ``[rows, cols, 3]``; ``palette`` is
        ``[n, 3]`` uint8 (the reference: the 256 colours of ``colorous::INFERNO``), ``scale`` its
        ``AMPLITUDE_SCALE`` = 1 / 0.5 (ui/src/lib.rs:117-123)."""
        palette = np.ascontiguousarray(palette, np.uint8)
        assert palette.ndim == 2 and palette.shape[1] == 3 and len(palette) >= 1
        r0, r1 = self.local_rows()
        out = np.empty((r1 - r0, self._shape[1], 3), np.uint8)
        capi.check(context._lib.gs_field_colormap(context.handle, self.handle, scale,
                                                  palette.ctypes.data_as(ctypes.c_void_p), len(palette),
                                                  out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def destroy(self) -> None:
        if self._h and self._ctx._h:
            self._ctx._lib.gs_field_destroy(self._ctx._h, self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Evolving:
    """Input/output pair of one species (mod.rs:140-187): slot 0 is the input."""

    def __init__(self, pair: List[HipConcentration]):
        self._pair = pair

    @classmethod
    def zeros_out(cls, context, shape):
        return cls([HipConcentration.default(context, shape), HipConcentration.zeros(context, shape)])

    @classmethod
    def ones_out(cls, context, shape):
        return cls([HipConcentration.default(context, shape), HipConcentration.ones(context, shape)])

    def in_out(self):
        return self._pair[0], self._pair[1]

    def out(self):
        return self._pair[1]

    def shape(self):
        return self._pair[0].shape()

    def raw_shape(self):
        return self._pair[0].raw_shape()

    def flip(self, context) -> None:
        self._pair[1].finalize(context)
        self._pair.reverse()


class Species:
    """``Species<HipConcentration>``; ``Species.new`` = ``Species::new`` (mod.rs:36-59)."""

    def __init__(self, context: HipContext, u: Evolving, v: Evolving):
        self._context, self.u, self.v = context, u, v

    @classmethod
    def new(cls, context: HipContext, shape: Sequence[int], place_candidates: int = 0) -> "Species":
        """``Species::new`` (data/src/concentration/mod.rs:36-59).  ``place_candidates`` > 0 (not in the reference): the
        four planes are then placed by measurement with at most n extra blocks drawn (``Species.place``)."""
        shape = (int(shape[0]), int(shape[1]))
        u = Evolving.ones_out(context, shape)
        v = Evolving.zeros_out(context, shape)
        num_range, frac, row_shift = (7, 8), 16, 4
        sl = []
        for i in (0, 1):
            shift = row_shift if i == 0 else 0
            start, end = (max(shape[i] * num_range[j] // frac - shift, 0) for j in (0, 1))
            sl.append(range(start, end))
        u.out().fill_slice(context, sl, 0.0)
        v.out().fill_slice(context, sl, 1.0)
        s = cls(context, u, v)
        s.flip()
        s.placement, s.placement_drawn = None, 0
        if place_candidates > 0:
            s.place(place_candidates)
        return s

    def place(self, candidates: int) -> Tuple[float, float]:
        """Placement by measurement (``gs_fields_place``): planes of one physical region of HBM that a pass writes together
        are slow, so U's and V's planes are given blocks of different regions -- drawing at most ``candidates`` extra blocks,
        moving the planes that have to move with their contents.  Returns and remembers (``placement``) the mean time, in
        ms, of the probe pass (a step's traffic on a slot's two planes) over the blocks they had and the blocks they have now."""
        in_u, in_v, out_u, out_v = self.in_out()
        arr = (ctypes.c_void_p * 4)(in_u.handle, in_v.handle, out_u.handle, out_v.handle)
        first, best = ctypes.c_float(0), ctypes.c_float(0)
        ctx = self._context
        drawn0 = ctx.place_stats()[1]
        capi.check(ctx._lib.gs_fields_place(ctx.handle, arr, int(candidates), ctypes.byref(first), ctypes.byref(best)))
        self.placement = (float(first.value), float(best.value))
        self.placement_drawn = ctx.place_stats()[1] - drawn0      # extra blocks held for a moment by THIS call
        return self.placement

    def context(self) -> HipContext:
        return self._context

    def shape(self):
        return self.u.shape()

    def raw_shape(self):
        return self.u.raw_shape()

    def in_out(self):
        in_u, out_u = self.u.in_out()
        in_v, out_v = self.v.in_out()
        return in_u, in_v, out_u, out_v

    def flip(self) -> None:
        self.u.flip(self._context)
        self.v.flip(self._context)

    def access_result(self, f: Callable):
        return f(self.v._pair[0], self._context)

    def make_result_view(self) -> np.ndarray:
        return self.access_result(lambda v, ctx: v.make_scalar_view(ctx))

    def write_result_view(self, target: np.ndarray) -> None:
        self.access_result(lambda v, ctx: v.write_scalar_view(ctx, target))

    def write_result_view_after(self, target: np.ndarray) -> None:
        """Asynchronous form used by the driver loop (simulate/src/main.rs:99-106)."""
        self.access_result(lambda v, ctx: v.write_scalar_view_after(ctx, target))


class Simulation:
    """The backend: ``SimulateBase + SimulateCreate + Simulate``."""

    CliArgs = HipArgs
    Concentration = HipConcentration
    Error = GsError

    def __init__(self, params: Parameters, args: Optional[HipArgs] = None):
        self.params = params
        self.context = HipContext(params, args)

    @classmethod
    def new(cls, params: Parameters, args: Optional[HipArgs] = None) -> "Simulation":
        """``SimulateCreate::new(params, args)`` (compute/shared/src/lib.rs:42-45)."""
        return cls(params, args)

    def make_species(self, shape: Sequence[int], place_candidates: Optional[int] = None) -> Species:
        """``SimulateBase::make_species`` (lib.rs:33-34).  ``place_candidates``: None = the library's default -- a Species
        of >= 2^26 cells per process on a context with one slab per process is placed by measurement with at most
        ``HipArgs.place_candidates`` extra blocks (``--hip-place-candidates`` / GS_HIP_PLACE_CANDIDATES, default 12; 0 = never)
        --, 0 = no placement, n > 0 = placed whatever its size."""
        if place_candidates is None:
            args = self.context.args
            cells = int(shape[0]) * int(shape[1]) // max(1, args.world)
            place_candidates = args.place_candidates if len(args.devices) == 1 and cells >= PLACE_MIN_CELLS else 0
        return Species.new(self.context, shape, place_candidates)

    def perform_steps(self, species: Species, steps: int) -> None:
        """``Simulate::perform_steps`` (lib.rs:48-58): ``steps`` steps; on return they are DONE and
        the input slots of ``species`` hold the final state.  Synchronous like every backend of the
        reference -- its GPU backends end ``perform_steps_impl`` with
        ``.then_signal_fence_and_flush()?.wait(None)?`` (compute/shared/src/gpu/mod.rs:77-91)."""
        self.prepare_steps(species, steps)
        self.context.sync()

    def prepare_steps(self, species: Species, steps: int) -> None:
        """The asynchronous form, ``SimulateGpu::prepare_steps`` (compute/shared/src/gpu/mod.rs:
        70-75): enqueue ``steps`` steps and return; whatever is enqueued next on this context (more
        steps, ``write_result_view_after``) runs behind them, a download or ``context.sync()`` waits.
        HIP streams order the work, so there is no future object to pass along."""
        in_u, in_v, out_u, out_v = species.in_out()
        slot = ctypes.c_int32(0)
        lib = self.context._lib
        capi.check(lib.gs_run(self.context.handle, in_u.handle, in_v.handle, out_u.handle,
                              out_v.handle, int(steps), ctypes.byref(slot)))
        if slot.value == 1:  # odd number of steps: the newest state sits in the output slot
            species.u._pair.reverse()
            species.v._pair.reverse()

    def perform_step(self, species: Species) -> None:
        """One ``gs_step`` then ``species.flip()`` -- the ``SimulateStep`` form (cpu.rs:21-42)."""
        in_u, in_v, out_u, out_v = species.in_out()
        capi.check(self.context._lib.gs_step(self.context.handle, in_u.handle, in_v.handle,
                                             out_u.handle, out_v.handle))
        species.flip()
