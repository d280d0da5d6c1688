"""CPU checkers for the Gray-Scott step path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the reported CPU
baseline.  The product path (``grayscott_amd`` -> ``libgs_hip.so``) never does.

PARITY UNPINNED BY THE REFERENCE: the reference holds no golden vectors or
known-answer tests for this path and cannot be built in this image (Rust, no
toolchain).  The C restatement is pinned by hand-derived known answers and by
bit-for-bit agreement with the independently written numpy restatement
(``oracle.numpy_ref``); see ``oracle/gs_oracle.c`` for the file:line citations.
"""
from .cpu_oracle import (  # noqa: F401
    CLIPPED,
    ZERO_HALO,
    Params,
    build,
    default_params,
    init_species,
    run,
    seed_ranges,
    set_ftz,
    step,
    step_rows,
)
