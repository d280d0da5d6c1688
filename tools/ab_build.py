#!/usr/bin/env python3
"""Build a variant of libgs_hip.so with extra hipcc flags for A/B timing:

    python tools/ab_build.py NAME [FLAG...]     ->  grayscott_amd/variants/libgs_hip_NAME.so

Run anything against it with GS_HIP_LIBRARY=<that path> (grayscott_amd/capi.py).  The variants are
git-ignored (*.so) but travel to the GPU box with gpurun.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grayscott_amd import _build  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
vdir = os.path.join(_build.HERE, "variants")
os.makedirs(vdir, exist_ok=True)
print(_build.build(force=True, lib=os.path.join(vdir, f"libgs_hip_{name}.so"),
                   build_dir=os.path.join(_build.HERE, "build", "variant_" + name), extra_flags=flags))
