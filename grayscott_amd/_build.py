"""In-tree build of ``libgs_hip.so`` with hipcc for gfx950 (no cmake, no JIT cache).

The step kernels are compiled twice from one source (``gs_step_kernels.hip``):

* strict: ``-DGS_MATH_FUSED=0`` with f32 denormal mode "flush results, keep inputs"
  (``-fdenormal-fp-math-f32=preserve-sign,ieee`` -> ``.amdhsa_float_denorm_mode_32 1``),
  the GPU equivalent of the reference's MXCSR.FTZ ``DenormalsFlusher``;
* fused:  ``-DGS_MATH_FUSED=1`` with hipcc's default (keep denormals).

Both with ``-ffp-contract=off``: parity with the reference's naive backend is bit for
bit, so the compiler must never contract ``a*b+c`` on its own.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libgs_hip.so")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
          "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# Experiment hook (tools/sweep runs): extra hipcc flags for the kernel translation units.
EXTRA = os.environ.get("GS_HIP_EXTRA_FLAGS", "").split()

# The step kernels are built without the SLP vectoriser: v_pk_*_f32 has the lane throughput of the
# plain ops on gfx950 and the packing costs ~10 % extra v_mov (profiles/archive/r01_sweeps.md, runs 49-57).
KERNEL_FLAGS = ["-fno-slp-vectorize"]
STRICT = ["-DGS_MATH_FUSED=0", "-Xclang", "-fdenormal-fp-math-f32=preserve-sign,ieee"]

UNITS = [
    # (source, object, extra flags)
    ("gs_step_kernels.hip", "gs_step_strict.o", STRICT + KERNEL_FLAGS),
    # the parameter-specialised strict variants: their own translation unit, compiled in parallel
    ("gs_step_kernels.hip", "gs_step_strict_op.o", STRICT + ["-DGS_TB_OP_ONLY=1"] + KERNEL_FLAGS),
    ("gs_step_kernels.hip", "gs_step_fused.o", ["-DGS_MATH_FUSED=1"] + KERNEL_FLAGS),
    ("gs_util_kernels.hip", "gs_util.o", []),
    # the host side (contexts and schedule, planes, kernel configuration, the window kernel's runtime, RCCL): only the
    # C ABI of include/gs_hip.h is visible outside the library
    ("gs_api.cpp", "gs_api.o", ["-x", "hip", "-fvisibility=hidden"]),
    ("gs_fields.cpp", "gs_fields.o", ["-x", "hip", "-fvisibility=hidden"]),
    ("gs_tuner.cpp", "gs_tuner.o", ["-x", "hip", "-fvisibility=hidden"]),
    ("gs_window.cpp", "gs_window.o", ["-x", "hip", "-fvisibility=hidden"]),
    ("gs_rccl.cpp", "gs_rccl.o", ["-x", "hip", "-fvisibility=hidden"]),
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libgs_hip.so cannot be built")
    return exe


def _sources():
    out = [os.path.join(CSRC, n) for n in os.listdir(CSRC)]
    out.append(os.path.join(HERE, os.pardir, "include", "gs_hip.h"))
    out.append(os.path.abspath(__file__))
    return out


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(s) <= t for s in _sources())


def build(force: bool = False, verbose: bool = False, lib: str = LIB, build_dir: str = BUILD,
          extra_flags=()) -> str:
    """Compile every HIP translation unit and link ``libgs_hip.so``; returns its path.
    ``lib`` / ``build_dir`` / ``extra_flags`` build a variant next to it (tools/ab_build.py)."""
    if lib == LIB and not force and up_to_date():
        return LIB
    os.makedirs(build_dir, exist_ok=True)
    cc = hipcc()
    procs = []
    for src, obj, extra in UNITS:
        flags = list(COMMON) + ((EXTRA + list(extra_flags)) if src.endswith(".hip") else [])
        cmd = [cc] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", os.path.join(build_dir, obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
        if verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    objs = [os.path.join(build_dir, obj) for _, obj, _ in UNITS]
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stdout.decode(errors="replace")))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
