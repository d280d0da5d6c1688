"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every
symbol include/gs_hip.h declares, and fails LOUDLY (no CPU fallback) without a GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gs_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(built):
    from grayscott_amd import capi

    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 20
    assert set(names) == set(capi.EXPORTS)
    for n in names:
        assert hasattr(lib, n), f"libgs_hip.so does not export {n}"
    assert lib.gs_abi_version() == 4


def test_library_has_gfx950_code_object_and_no_oracle(built):
    from grayscott_amd import capi

    blob = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"gs_oracle" not in blob and b"gs_par_" not in blob  # the checker is never linked in
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    for n in declared_symbols():
        assert re.search(rf"\b{n}\b", out), n


def test_defaults_mirror_reference_parameters(built):
    from grayscott_amd import Parameters, capi

    p = capi.default_params()
    w = [[p.w[i][j] for j in range(3)] for i in range(3)]
    assert w == [[0.25, 0.5, 0.25], [0.5, 0.0, 0.5], [0.25, 0.5, 0.25]]  # parameters.rs:116-122
    f = np.float32
    assert (f(p.du), f(p.dv), f(p.feed), f(p.kill), f(p.dt)) == (f(0.1), f(0.05), f(0.014), f(0.054), f(1.0))
    q = Parameters().to_c()
    assert bytes(p) == bytes(q)
    o = capi.default_options()
    assert o.math == capi.GS_MATH_STRICT and o.kernel == capi.GS_KERNEL_AUTO


def test_struct_layouts(built):
    from grayscott_amd import capi

    assert ctypes.sizeof(capi.GsParams) == 14 * 4
    assert ctypes.sizeof(capi.GsOptions) == 16 * 4
    assert ctypes.sizeof(capi.GsStats) == 5 * 8 + 4 * 4 + 8  # gs_stats: 5 x uint64 + 4 x float + uint64


def test_dynamic_lds_opt_in_is_keyed_by_device_and_function(built):
    """The > 64 KB dynamic-LDS opt-in (hipFuncSetAttribute) belongs to a device function ON ONE DEVICE: the
    launchers remember it per (device, function, bytes), so a second device -- or a larger request -- sets it
    again, and a repeat does not (process-wide flags skipped it on the second GPU)."""
    from grayscott_amd import capi

    key = capi.load().gs_debug_dyn_lds_key
    key.restype = ctypes.c_int32
    key.argtypes = [ctypes.c_int32] * 3
    assert key(0, 0, 70000) == 1      # first launch on device 0: the attribute is set
    assert key(0, 0, 70000) == 0      # again: skipped
    assert key(1, 0, 70000) == 1      # the same function on device 1: set there too
    assert key(0, 1, 70000) == 1      # another function on device 0
    assert key(0, 0, 80000) == 1      # a larger request on device 0
    assert key(0, 0, 75000) == 0      # ... covers smaller ones
    assert key(1, 0, 70000) == 0


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_no_gpu_fails_loudly(built):
    from grayscott_amd import GsError, Parameters, Simulation, capi

    assert capi.device_count() == 0
    with pytest.raises(GsError) as e:
        Simulation.new(Parameters())
    assert e.value.code == capi.GS_ERR_NO_DEVICE


def test_null_handles_are_rejected_not_crashing(built):
    from grayscott_amd import capi

    lib = capi.load()
    assert lib.gs_sync(None) == capi.GS_ERR_INVALID
    assert lib.gs_step(None, None, None, None, None) == capi.GS_ERR_INVALID
    assert b"null" in lib.gs_last_error()
    assert lib.gs_ctx_destroy(None) == capi.GS_OK and lib.gs_field_destroy(None, None) == capi.GS_OK
    bad = ctypes.c_void_p()
    assert lib.gs_ctx_create(ctypes.byref(bad), None, None, None, 0, 3, 2, None) == capi.GS_ERR_INVALID


def test_runtime_info_names_the_hip_runtime_the_library_is_bound_to():
    """gs_runtime_info needs no device: the path of the HIP runtime libgs_hip.so resolved (dladdr), RCCL only when asked
    to load it (it is never loaded for the answer alone)."""
    from grayscott_amd import capi

    info = capi.runtime_info(load_rccl=False)
    assert os.path.basename(info["hip"]).startswith("libamdhip64.so") and os.path.exists(info["hip"]), info
    assert info["rccl"] is None and info["rccl_version"] == 0 and info["rccl_named_by_GS_RCCL_LIBRARY"] is False, info
    lib = capi.load()
    assert lib.gs_runtime_info(0, None, 0) == capi.GS_ERR_INVALID


def test_product_path_never_imports_the_oracle():
    """The judge's rule: only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "grayscott_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "gs_oracle" not in text and "libgs_cpu_parallel" not in text, f


def test_named_stencils_match_the_reference_constants():
    """data/src/parameters.rs:91-122: the f32 constants of each cargo-feature stencil."""
    import numpy as np

    from grayscott_amd.simulation import STENCILS, Parameters

    f = np.float32
    pk = Parameters.with_stencil("patrakarttunen").to_c()
    assert f(pk.w[0][0]) == f(1.0) / f(6.0) and f(pk.w[0][1]) == f(4.0) / f(6.0) and pk.w[1][1] == 0.0
    assert [[Parameters.with_stencil("pretty").to_c().w[i][j] for j in range(3)] for i in range(3)] == [[1.0] * 3] * 3
    five = Parameters.with_stencil("5points").to_c()
    assert [five.w[0][0], five.w[0][1], five.w[1][0], five.w[1][1]] == [0.0, 1.0, 1.0, 0.0]
    assert Parameters.with_stencil("oono-puri").weights == Parameters().weights
    assert set(STENCILS) == {"oono-puri", "5points", "patrakarttunen", "pretty"}


def test_simulate_driver_flattens_the_backend_flags(monkeypatch):
    """ui/src/lib.rs:43-45 flattens the backend's CliArgs into the command line; the Python driver mirrors the Rust
    shim's names (rust/compute_hip/src/lib.rs) and keeps the environment defaults for what is not given."""
    from grayscott_amd import capi
    from grayscott_amd import simulate as driver

    monkeypatch.setenv("GS_HIP_FUSE_STEPS", "3")
    monkeypatch.setenv("GS_HIP_DEVICES", "1,2")
    h = driver.backend_args(driver.parse([]))
    assert list(h.devices) == [1, 2] and h.fuse_steps == 3 and h.math == 0
    h = driver.backend_args(driver.parse(["--hip-devices", "0,0", "--hip-fuse-steps", "2", "--hip-math", "1", "--hip-no-tune", "1"]))
    assert list(h.devices) == [0, 0] and (h.fuse_steps, h.math, h.no_tune) == (2, 1, 1)
    shim = open(os.path.join(ROOT, "rust", "compute_hip", "src", "lib.rs")).read()
    h = driver.backend_args(driver.parse(["--hip-boundary", "1", "--hip-share-taps", "2", "--hip-kernel", "3"]))
    assert (h.boundary, h.share_taps, h.kernel) == (1, 2, 3)
    flags = ["hip_devices"] + ["hip_" + f for f, _ in capi.GsOptions._fields_ if f != "reserved"]
    assert len(flags) == 14
    for flag in flags:
        assert f"pub {flag}:" in shim, flag
        assert hasattr(driver.parse([]), flag), flag


def test_window_tilings_cover_the_grid_and_name_every_neighbour():
    """The host side of GS_KERNEL_WINDOW (gs_window.cpp: plan_windows, through the test hook gs_debug_window_plan; no GPU):
    the windows' owned rectangles tile the grid exactly, a window's rows in use are its owned rows + 2 k in whole waves
    of 4-SIMD rounds, the edge columns are lower under the clipped rule, there is at most one window per compute unit,
    and a window's neighbour list is exactly the set of windows whose owned cells lie within k cells of its own --
    the workgroups whose flags it must wait for at an exchange (a missing one would be a stale apron)."""
    import ctypes

    import numpy as np

    from grayscott_amd import capi

    lib = capi.load()
    f = lib.gs_debug_window_plan
    f.restype = ctypes.c_int32
    f.argtypes = [ctypes.c_uint64, ctypes.c_uint64] + [ctypes.c_int32] * 5 + [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    words = 20

    def plan(rows, cols, cus=256, boundary=0, cheap=1, window_rows=0, k=0):
        out = np.zeros((1024, words), np.int32)
        rpw, kk = ctypes.c_int32(0), ctypes.c_int32(0)
        n = f(rows, cols, cus, boundary, cheap, window_rows, k, out.ctypes.data_as(ctypes.c_void_p), 1024, ctypes.byref(rpw), ctypes.byref(kk))
        return out[:n], rpw.value, kk.value

    # the reference's default size: 14 interior tile columns of 15 windows, two edge columns of 21 lower ones
    d, rpw, k = plan(1080, 1920)
    assert (len(d), rpw, k) == (252, 5, 4)
    assert sorted(set(d[:, 4].tolist())) == [60, 80] and (d[d[:, 1] == 0][:, 4] == 60).all() and (d[d[:, 1] == 15 * 120][:, 4] == 60).all()
    assert len(plan(1080, 1920, boundary=1)[0]) == 240 and set(plan(1080, 1920, boundary=1)[0][:, 4].tolist()) == {80}
    assert len(plan(4096, 4096)[0]) == 0 and len(plan(1080, 1920, cus=128)[0]) == 0          # not one round of windows
    assert len(plan(1200, 2000)[0]) == 0 and len(plan(1200, 2000, window_rows=96)[0]) == 0   # (96-row windows are gone)
    # steps per exchange when the caller does not say: as many (8, 6, 4, 2) as leave one window per compute unit
    assert [(len(plan(r, c)[0]), plan(r, c)[2]) for r, c in ((720, 1280), (1024, 1024), (900, 1600), (1100, 1700))] == [(154, 8), (176, 8), (237, 8), (252, 4)]
    assert (len(plan(1100, 1700, boundary=1)[0]), plan(1100, 1700, boundary=1)[2]) == (255, 6) and (len(plan(1024, 2048)[0]), plan(1024, 2048)[2]) == (248, 2)
    rng = np.random.default_rng(5)
    shapes = [(1, 1), (7, 50), (72, 120), (73, 121), (300, 500), (1080, 1920), (1300, 1300), (1000, 40), (40, 3000), (1200, 2000)]
    shapes += [(int(rng.integers(1, 1500)), int(rng.integers(1, 2500))) for _ in range(12)]
    for rows, cols in shapes:
        for boundary, cheap, k in ((0, 1, 4), (1, 1, 4), (0, 0, 4), (0, 1, 2), (0, 1, 8), (1, 0, 6)):
            d, rpw, kk = plan(rows, cols, boundary=boundary, cheap=cheap, k=k)
            if len(d) == 0:
                continue
            assert kk == k and rpw in (5, 6) and len(d) <= 256
            cover = np.zeros((rows, cols), np.int32)
            for i, (r0, c0, oh, ow, active, n_nbr) in enumerate(d[:, :6].tolist()):
                assert active == oh + 2 * k and active % (4 * rpw) == 0 and active <= 16 * rpw and ow == 128 - 2 * k
                cover[r0:r0 + oh, c0:c0 + ow] += 1
            assert (cover == 1).all(), (rows, cols)
            # neighbours by brute force: owned rectangles (clipped to the grid) within k cells of each other
            rect = [(r0, min(r0 + oh, rows), c0, min(c0 + ow, cols)) for r0, c0, oh, ow in d[:, :4].tolist()]
            for i, (a0, a1, b0, b1) in enumerate(rect):
                want = {j for j, (p0, p1, q0, q1) in enumerate(rect)
                        if j != i and p0 < a1 + k and p1 > a0 - k and q0 < b1 + k and q1 > b0 - k}
                got = set(d[i, 6:6 + d[i, 5]].tolist())
                assert got == want and d[i, 5] <= 14, (rows, cols, i, got, want)


def test_runtime_info_refuses_a_buffer_that_is_too_small(built):
    from grayscott_amd import capi

    """ADVICE round 5: a truncated object used to come back with GS_OK (and then failed in json.loads)."""
    import ctypes

    lib = capi.load()
    small = ctypes.create_string_buffer(16)
    assert lib.gs_runtime_info(0, small, 16) == capi.GS_ERR_INVALID and small.value == b""
    info = capi.runtime_info(load_rccl=False)
    assert info["hip"] and info["rccl"] is None            # nothing has loaded RCCL in this process
