"""Static checks of the reference-side binding (rust/compute_hip), which this image cannot compile
(no Rust toolchain): the FFI declarations agree with include/gs_hip.h, the manifest has what the
reference's own backends need (compute/block/Cargo.toml:8,14-24), perform_steps is synchronous as in
compute/shared/src/gpu/mod.rs:77-91, and the workspace patch applies to the reference checkout."""
import os
import re
import shutil
import subprocess

import pytest

from grayscott_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "rust", "compute_hip")
REF = "/root/reference"


def _read(*parts):
    return open(os.path.join(*parts)).read()


def test_ffi_matches_the_header():
    ffi = _read(RUST, "src", "ffi.rs")
    header = _read(ROOT, "include", "gs_hip.h")
    for fn in re.findall(r"pub fn (gs_\w+)\(", ffi):
        assert fn in capi.EXPORTS and re.search(r"\b%s\(" % fn, header), fn
    # gs_options: same fields, same order, same total size as the C struct (16 x int32)
    rust_fields = re.findall(r"pub (\w+): (?:i32|\[i32; (\d+)\])", ffi[ffi.index("pub struct gs_options"):ffi.index("pub struct gs_ctx")])
    c_body = header[header.index("typedef struct gs_options {"):header.index("} gs_options;")]
    c_fields = re.findall(r"int32_t (\w+)(?:\[(\d+)\])?;", c_body)
    assert [f for f, _ in rust_fields] == [f for f, _ in c_fields]
    assert sum(int(n or 1) for _, n in rust_fields) == sum(int(n or 1) for _, n in c_fields) == 16
    assert [f for f, _ in capi.GsOptions._fields_] == [f for f, _ in c_fields]
    assert "ABI version %d" % capi.load().gs_abi_version() in ffi


def test_manifest_and_bench_entry():
    toml = _read(RUST, "Cargo.toml")
    assert re.search(r'clap = \{ workspace = true, features = \["env"\] \}', toml)            # #[arg(env = ...)]
    assert re.search(r'\[dev-dependencies\]\s*\ncompute = \{ workspace = true, features = \["criterion"\] \}', toml)
    assert re.search(r'\[\[bench\]\]\s*\nname = "compute_hip"\s*\nharness = false', toml)
    bench = _read(RUST, "benches", "compute_hip.rs")
    assert "compute::cpu_benchmark!(compute_hip);" in bench
    lib = _read(RUST, "src", "lib.rs")
    # every field of gs_options is a flag of HipArgs -- `--hip-<field>`, env GS_HIP_<FIELD>, defaulted
    # (compute/shared/src/lib.rs:20-25) -- reaches the C struct, and is the variable grayscott_amd.HipArgs reads
    header = _read(ROOT, "include", "gs_hip.h")
    c_body = header[header.index("typedef struct gs_options {"):header.index("} gs_options;")]
    fields = [f for f in re.findall(r"int32_t (\w+)(?:\[\d+\])?;", c_body) if f != "reserved"]
    assert len(fields) == 13
    py = _read(ROOT, "grayscott_amd", "simulation.py")
    for f in ["devices"] + fields:
        m = re.search(r"#\[arg\(long, env = \"(GS_HIP_\w+)\"[^\]]*default_value(?:_t)? = [^\]]*\)\]\s*\n\s*pub hip_%s:" % f, lib)
        assert m and m.group(1) == "GS_HIP_" + f.upper(), f
        assert '"%s"' % m.group(1) in py, f
        if f != "devices":
            assert "opts.%s = args.hip_%s;" % (f, f) in lib, f
    # ... plus the one argument that is not an option of the context: make_species places the large Species it creates,
    # by default (12 extra blocks at most), in all three hosts
    assert re.search(r'#\[arg\(long, env = "GS_HIP_PLACE_CANDIDATES", default_value_t = 12\)\]\s*\n\s*pub hip_place_candidates: i32', lib)
    assert '"GS_HIP_PLACE_CANDIDATES", 12)' in py
    assert "int32_t place_candidates = 12;" in _read(ROOT, "include", "grayscott_hip.hpp")
    make = lib[lib.index("fn make_species"):lib.index("impl SimulateCreate for Simulation")]
    assert "Species::new(self.context.clone(), shape)?" in make and "ffi::gs_fields_place(" in make
    assert "shape[0] as u64 * shape[1] as u64 >= 1u64 << 26" in make          # planes of >= 256 MiB only
    ffi = _read(RUST, "src", "ffi.rs")
    assert re.search(r"pub fn gs_fields_place\(\s*ctx: \*mut gs_ctx,\s*planes: \*const \*mut gs_field,\s*candidates: i32,\s*first_ms: \*mut f32,"
                     r"\s*best_ms: \*mut f32,\s*\) -> i32;", ffi)


def test_perform_steps_waits_and_prepare_steps_does_not():
    lib = _read(RUST, "src", "lib.rs")
    body = lib[lib.index("impl Simulate for Simulation"):lib.index("impl Simulation {")]
    assert "self.prepare_steps(species, steps)?" in body and "ffi::gs_sync(self.context.0)" in body
    prep = lib[lib.index("pub fn prepare_steps"):lib.index("pub fn perform_step(")]
    assert "ffi::gs_run(" in prep and "gs_sync" not in prep


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="needs the reference checkout and patch(1)")
def test_workspace_patch_applies_to_the_reference(tmp_path):
    patch = os.path.join(ROOT, "rust", "grayscott_compute_hip.patch")
    files = re.findall(r"^\+\+\+ b/(\S+)", _read(patch), flags=re.M)
    assert "compute/selector/src/lib.rs" in files and "Cargo.toml" in files
    for f in files:
        os.makedirs(os.path.join(tmp_path, os.path.dirname(f)), exist_ok=True)
        shutil.copy(os.path.join(REF, f), os.path.join(tmp_path, f))
    subprocess.run(["patch", "-p1", "--forward", "-i", patch], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL)
    sel = _read(tmp_path, "compute/selector/src/lib.rs")
    assert sel.index('feature = "compute_hip"') < sel.index('feature = "compute_gpu_specialized"')
    assert 'compute_hip.path = "compute/hip"' in _read(tmp_path, "Cargo.toml")
