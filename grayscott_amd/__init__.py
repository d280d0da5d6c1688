"""grayscott_amd -- MI355X-native (gfx950) compute backend for the Gray-Scott step path of
HadrienG2/grayscott: hand-written HIP kernels behind a C ABI (``include/gs_hip.h``), plus
the host-side mirror of the reference's backend interface used by the tests and the bench.

Importing this package does not touch the GPU; ``capi.load()`` raises if ``libgs_hip.so``
has not been built (``__graft_entry__.build()``), and there is no CPU fallback.
"""
from . import capi  # noqa: F401
from .capi import GsError  # noqa: F401
from .simulation import (  # noqa: F401
    Evolving,
    HipArgs,
    HipConcentration,
    HipContext,
    Parameters,
    Simulation,
    Species,
    pinned_empty,
)

__all__ = ["capi", "GsError", "Evolving", "HipArgs", "HipConcentration", "HipContext",
           "Parameters", "Simulation", "Species", "pinned_empty"]
