"""The C++ host-side mirror (include/grayscott_hip.hpp) over the C ABI.

CPU: the header compiles with plain g++ against gs_hip.h, links libgs_hip.so, and the
program fails loudly (HipError, GS_ERR_NO_DEVICE) without a GPU.  GPU: the reference's
driver sequence new -> make_species -> perform_steps -> write_result_view, bit-exact."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(built, tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "host_mirror"
    libdir = os.path.join(ROOT, "grayscott_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp"), "-o", str(out),
           "-L", libdir, "-lgs_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(out)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_cpp_mirror_builds_and_fails_loudly_without_gpu(exe, tmp_path):
    r = subprocess.run([exe, "8", "8", "1", str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 14 and "HipError" in r.stderr  # GS_ERR_NO_DEVICE, no CPU fallback


@pytest.mark.gpu
def test_cpp_mirror_matches_oracle(exe, tmp_path):
    import oracle

    rows, cols, steps = 72, 200, 33
    out = tmp_path / "o.bin"
    r = subprocess.run([exe, str(rows), str(cols), str(steps), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    data = np.fromfile(out, np.float32).reshape(3, rows, cols)
    u0, v0 = oracle.init_species(rows, cols)
    ref_u, ref_v = oracle.run(u0, v0, steps)
    assert data[0].tobytes() == ref_u.tobytes() and data[1].tobytes() == ref_v.tobytes()
    assert data[2].tobytes() == oracle.run(u0, v0, steps // 2)[1].tobytes()  # the asynchronous image
