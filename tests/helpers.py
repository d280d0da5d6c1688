"""Shared helpers for the parity tests (GPU side goes through the C ABI only)."""
from __future__ import annotations

import numpy as np

import oracle
from grayscott_amd import (Evolving, HipArgs, HipConcentration, Parameters, Simulation, Species,
                           capi)

STRESS_SHAPES = [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (17, 33), (64, 128), (250, 130)]


def stress_fields(shape, seed):
    """U ~ Uniform[0,1), V ~ Uniform[0,0.5), numpy default_rng(seed)."""
    rng = np.random.default_rng(seed)
    u = rng.random(shape, dtype=np.float32)
    v = (rng.random(shape, dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    return u, v


def oracle_params(p: Parameters) -> oracle.Params:
    q = oracle.default_params()
    q.set_weights(p.weights)
    q.du, q.dv = p.diffusion_rate_u, p.diffusion_rate_v
    q.feed, q.kill, q.dt = p.feed_rate, p.kill_rate, p.time_step
    return q


def species_from_arrays(sim: Simulation, u0: np.ndarray, v0: np.ndarray, shape=None) -> Species:
    """``u0``/``v0`` hold this process's rows; ``shape`` is the global shape when they differ."""
    ctx = sim.context
    shape = shape or u0.shape
    u = Evolving([HipConcentration(ctx, shape), HipConcentration(ctx, shape)])
    v = Evolving([HipConcentration(ctx, shape), HipConcentration(ctx, shape)])
    u.in_out()[0].upload(ctx, u0)
    v.in_out()[0].upload(ctx, v0)
    return Species(ctx, u, v)


def gpu_run(u0, v0, steps, params: Parameters | None = None, args: HipArgs | None = None,
            stepwise: bool = False):
    """upload -> perform_steps -> download, all through libgs_hip.so."""
    params = params or Parameters()
    sim = Simulation.new(params, args or HipArgs(devices=[0]))
    species = species_from_arrays(sim, u0, v0)
    if stepwise:
        for _ in range(steps):
            sim.perform_step(species)
    else:
        sim.perform_steps(species, steps)
    in_u, in_v, _, _ = species.in_out()
    out = in_u.make_scalar_view(sim.context), in_v.make_scalar_view(sim.context)
    info = sim.context.info()
    sim.context.close()
    return out + (info,)


def assert_bits_equal(got: np.ndarray, ref: np.ndarray, what: str):
    if got.tobytes() != ref.tobytes():
        bad = np.flatnonzero(got.view(np.uint32).ravel() != ref.view(np.uint32).ravel())
        i = int(bad[0])
        r, c = divmod(i, got.shape[1])
        raise AssertionError(
            f"{what}: {bad.size} of {got.size} cells differ; first at ({r},{c}): "
            f"got {got[r, c]!r} ref {ref[r, c]!r}; max|d|={float(np.max(np.abs(got - ref)))}")


_handed_out = set()


def free_port() -> int:
    """A port nobody listens on and that this test process has not handed out before: the kernel gives a closed
    ephemeral port out again at once, and the rendezvous store of the previous test may still hold it (EADDRINUSE in
    the next test's TCPStore, seen once on the GPU box)."""
    import os
    import random
    import socket

    rng = random.Random(os.getpid() * 7919 + len(_handed_out))
    for _ in range(200):
        port = rng.randrange(20000, 45000)
        if port in _handed_out:
            continue
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
        _handed_out.add(port)
        return port
    raise RuntimeError("no free port found")
