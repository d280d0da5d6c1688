"""the simulate-equivalent driver loop with overlapped result download."""
import numpy as np
import pytest

import oracle
from grayscott_amd import HipArgs, Parameters, Simulation, pinned_empty
from grayscott_amd import simulate as driver
from tests.helpers import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("slabs", [1, 3])
def test_async_download_is_a_snapshot(built, slabs):
    """The image enqueued after N steps holds the state after exactly N steps although more
    steps (which overwrite that plane) are enqueued right behind it."""
    rows, cols = 192, 520
    sim = Simulation.new(Parameters(), HipArgs(devices=[0] * slabs))
    species = sim.make_species([rows, cols])
    images = [pinned_empty((rows, cols)) for _ in range(3)]
    done = []
    for i, n in enumerate((8, 5, 12)):
        sim.perform_steps(species, n)
        species.write_result_view_after(images[i])
        done.append(n + (done[-1] if done else 0))
    sim.perform_steps(species, 40)                       # keeps the GPU busy behind the copies
    sim.context.download_wait()
    u0, v0 = oracle.init_species(rows, cols)
    for image, n in zip(images, done):
        assert_bits_equal(image, oracle.run(u0, v0, n)[1], f"image after {n} steps")
    assert_bits_equal(species.make_result_view(), oracle.run(u0, v0, done[-1] + 40)[1], "final state")


def test_driver_loop_matches_oracle(built, tmp_path):
    out = tmp_path / "out.npy"
    args = driver.parse(["-n", "6", "-e", "9", "-r", "120", "-c", "250", "-k", "0.06", "-f", "0.03",
                         "-t", "0.5", "-o", str(out), "--output-buffer", "2"])
    info = driver.run(args)
    assert info["images"] == 6 and info["steps_per_image"] == 9
    data = np.load(out)
    assert data.shape == (6, 120, 250) and data.dtype == np.float32
    p = oracle.default_params()
    p.kill, p.feed, p.dt = 0.06, 0.03, 0.5
    u, v = oracle.init_species(120, 250)
    for i in range(6):
        u, v = oracle.run(u, v, 9, p)
        assert_bits_equal(data[i], v, f"image {i}")


def test_driver_loop_writes_the_reference_container(built, tmp_path):
    """-o *.h5: dataset "matrix" [nbimage, rows, cols] f32 in [1, rows, cols] chunks (data/src/hdf5.rs:36-63),
    read back with our parser and, where an HDF5 installation exists, with the real h5dump."""
    import os
    import subprocess

    from grayscott_amd import hdf5_min

    out = tmp_path / "out.h5"
    info = driver.run(driver.parse(["-n", "70", "-e", "5", "-r", "64", "-c", "200", "-o", str(out)]))
    assert info["images"] == 70
    data = hdf5_min.read(str(out))
    assert data.shape == (70, 64, 200)
    u, v = oracle.init_species(64, 200)
    for i in range(70):
        u, v = oracle.run(u, v, 5)
        assert_bits_equal(np.asarray(data[i]), v, f"image {i}")
    h5dump = os.path.join(os.environ.get("HDF5_DIR", "/opt/conda"), "bin", "h5dump")
    if os.path.exists(h5dump):
        header = subprocess.run([h5dump, "-p", "-H", str(out)], capture_output=True, text=True, check=True).stdout
        assert "CHUNKED ( 1, 64, 200 )" in header and "( 70, 64, 200 )" in header and "H5T_IEEE_F32LE" in header
        dump = tmp_path / "dump.bin"
        subprocess.run([h5dump, "-d", "/matrix", "-b", "LE", "-o", str(dump), str(out)], capture_output=True, check=True)
        assert np.array_equal(np.fromfile(dump, "<f4").reshape(70, 64, 200), np.asarray(data))


def test_driver_defaults_match_reference_cli():
    a = driver.parse([])
    assert a.output == "output.h5"                                                  # ui/src/lib.rs:72-75
    assert (a.nbrow, a.nbcol, a.nbimage, a.output_buffer) == (1080, 1920, 1000, 2)  # ui/src/lib.rs:32-38, main.rs:29-43
    assert a.nbextrastep is None and driver.simulation_parameters(a) == Parameters()


def test_driver_backend_flags_reach_the_library(built, tmp_path):
    """The backend's parameters flattened into the command line (ui/src/lib.rs:43-45: `#[command(flatten)] backend:
    Simulation::CliArgs`), same names as the Rust shim's HipArgs: a 3-slab chain, pinned layout, bit-exact."""
    out = tmp_path / "out.npy"
    args = driver.parse(["-n", "3", "-e", "7", "-r", "150", "-c", "260", "-o", str(out), "--hip-devices", "0,0,0",
                         "--hip-fuse-steps", "2", "--hip-rows-per-block", "8", "--hip-cols-per-lane", "1", "--hip-no-tune", "1"])
    h = driver.backend_args(args)
    assert list(h.devices) == [0, 0, 0] and (h.fuse_steps, h.rows_per_block, h.cols_per_lane, h.no_tune) == (2, 8, 1, 1)
    driver.run(args)
    data = np.load(out)
    u, v = oracle.init_species(150, 260)
    for i in range(3):
        u, v = oracle.run(u, v, 7)
        assert_bits_equal(data[i], v, f"image {i}")
