"""``simulate``-equivalent driver loop on the HIP backend.

Mirrors /root/reference/simulate/src/main.rs:46-127: ``nbimage`` images, ``nbextrastep`` steps
between images, the V plane of every image handed to a writer thread through a bounded queue
with buffer recycling (main.rs:73-121), shared CLI flags of ui/src/lib.rs:18-46.  Like the
reference's ``async-gpu`` path (main.rs:99-106) the steps and the download of the result are
enqueued together: the download of image i overlaps the steps of image i+1.

    python -m grayscott_amd.simulate -n 100 -r 1080 -c 1920 -o out.h5

Output: the reference writes an HDF5 dataset ``matrix[nbimage, rows, cols]`` f32
(data/src/hdf5.rs:36-63).  ``-o name.h5`` (or ``.hdf5``) writes that dataset through the minimal
HDF5 writer of ``grayscott_amd/hdf5_min.py`` (libhdf5 / h5py are not in this image: see the status
note there); any other name gives a ``.npy`` file (numpy format 1.0, C order) with the same array.
"""
from __future__ import annotations

import argparse
import queue
import sys
import threading
import time

import numpy as np

from . import hdf5_min
from .simulation import HipArgs, Parameters, Simulation, pinned_empty


def parse(argv=None):
    ap = argparse.ArgumentParser(prog="simulate", description="Perform Gray-Scott simulation")
    ap.add_argument("-k", "--killrate", type=float, default=None)        # ui/src/lib.rs:20-22
    ap.add_argument("-f", "--feedrate", type=float, default=None)        # :24-26
    ap.add_argument("-e", "--nbextrastep", type=int, default=None)       # :28-30 (default 32, main.rs:52)
    ap.add_argument("-r", "--nbrow", type=int, default=1080)             # :32-34
    ap.add_argument("-c", "--nbcol", type=int, default=1920)             # :36-38
    ap.add_argument("-t", "--deltat", type=float, default=None)          # :40-42
    ap.add_argument("-n", "--nbimage", type=int, default=1000)           # main.rs:29-31
    ap.add_argument("-o", "--output", default="output.h5")               # main.rs:33-35, ui/src/lib.rs:72-75
    ap.add_argument("--output-buffer", type=int, default=2)              # main.rs:37-43
    # The backend's own parameters, flattened into the command line as the reference flattens
    # `Simulation::CliArgs` (ui/src/lib.rs:43-45, inside the SharedArgs that simulate/src/main.rs:25-27 flattens): the names, meanings and environment
    # variables of rust/compute_hip/src/lib.rs (HipArgs).  Defaults come from the environment (HipArgs' own).
    be = ap.add_argument_group("HIP backend")
    be.add_argument("--hip-devices", default=None, metavar="ID[,ID...]",
                    help="HIP devices that run the simulation, one row slab each, top to bottom [env GS_HIP_DEVICES, 0]")
    be.add_argument("--hip-math", type=int, default=None, help="0 = strict (bit-identical to compute_naive), 1 = fused taps [GS_HIP_MATH]")
    be.add_argument("--hip-rows-per-block", type=int, default=None, help="rows each wavefront marches over, 0 = chosen on line [GS_HIP_ROWS_PER_BLOCK]")
    be.add_argument("--hip-fuse-steps", type=int, default=None, help="time steps fused per pass over HBM, 1..4, 0 = chosen on line [GS_HIP_FUSE_STEPS]")
    be.add_argument("--hip-cols-per-lane", type=int, default=None, help="columns per lane: 4, 2 or 1, 0 = chosen on line [GS_HIP_COLS_PER_LANE]")
    be.add_argument("--hip-no-tune", type=int, default=None, help="1 = never time candidate configurations inside perform_steps [GS_HIP_NO_TUNE]")
    be.add_argument("--hip-kernel", type=int, default=None, help="step kernel (gs_kernel in gs_hip.h), 0 = by grid size and call length [GS_HIP_KERNEL]")
    be.add_argument("--hip-boundary", type=int, default=None, help="0 = compute_naive's clipped window, 1 = zero halo (the SIMD / Vulkan backends' rule) [GS_HIP_BOUNDARY]")
    be.add_argument("--hip-general-kernels", type=int, default=None, help="1 = never run the variants specialised for the default stencil and time step [GS_HIP_GENERAL_KERNELS]")
    be.add_argument("--hip-share-taps", type=int, default=None, help="full difference sharing: 0 = on (form 3) unless measured slower, 1 = within a lane only, 2 = off, 3 = across lanes too [GS_HIP_SHARE_TAPS]")
    be.add_argument("--hip-place-candidates", type=int, default=None, help="most extra blocks gs_fields_place may draw when a Species of >= 2^26 cells is placed by measurement (default 12; 0 = planes as hipMalloc hands them out) [GS_HIP_PLACE_CANDIDATES]")
    be.add_argument("--hip-split", type=int, default=None, help="row bands a single slab is scheduled as [GS_HIP_SPLIT]")
    be.add_argument("--hip-use-graph", type=int, default=None, help="1 = replay batches of 16 passes through a hipGraph [GS_HIP_USE_GRAPH]")
    be.add_argument("--hip-tile-shape", type=int, default=None, help="window of the LDS-window kernel: 1 = 32x64, 2 = 16x64, 3 = 64x64 [GS_HIP_TILE_SHAPE]")
    be.add_argument("--hip-pitch-pad", type=int, default=None, help="extra f32 of row pitch [GS_HIP_PITCH_PAD]")
    return ap.parse_args(argv)


def backend_args(args) -> HipArgs:
    """``HipArgs`` from the command line; what it leaves unset keeps its environment default."""
    h = HipArgs()
    if getattr(args, "hip_devices", None):
        h.devices = [int(x) for x in str(args.hip_devices).split(",") if x != ""]
    for name in ("math", "rows_per_block", "fuse_steps", "cols_per_lane", "no_tune", "kernel", "boundary", "general_kernels",
                 "share_taps", "split", "use_graph", "tile_shape", "pitch_pad", "place_candidates"):
        flag = "hip_" + name
        value = getattr(args, flag, None)
        if value is not None:
            setattr(h, name, value)
    return h


def simulation_parameters(args) -> Parameters:
    """``SharedArgs::simulation_parameters`` (ui/src/lib.rs:51-63)."""
    p = Parameters()
    if args.killrate is not None:
        p.kill_rate = args.killrate
    if args.feedrate is not None:
        p.feed_rate = args.feedrate
    if args.deltat is not None:
        p.time_step = args.deltat
    return p


def run(args, hip_args: HipArgs | None = None, out=None) -> dict:
    steps_per_image = args.nbextrastep if args.nbextrastep is not None else 32
    shape = (args.nbrow, args.nbcol)
    if args.output_buffer < 1:
        raise ValueError("--output-buffer must be at least 1")
    sim = Simulation.new(simulation_parameters(args), hip_args if hip_args is not None else backend_args(args))
    species = sim.make_species(shape)
    ctx = sim.context
    if out is None and args.output.lower().endswith((".h5", ".hdf5")):
        out = hdf5_min.create(args.output, (args.nbimage,) + shape)       # dataset "matrix" (hdf5.rs:24)
    elif out is None:
        out = np.lib.format.open_memmap(args.output, mode="w+", dtype=np.float32,
                                        shape=(args.nbimage,) + shape)

    # I/O thread: writes images down and recycles their buffers (main.rs:73-87)
    full: "queue.Queue" = queue.Queue(maxsize=args.output_buffer)
    free: "queue.Queue" = queue.Queue()
    for _ in range(args.output_buffer + 2):            # +2: the two images on their way from the device
        free.put(pinned_empty(shape))
    errors = []
    failed = threading.Event()

    def writer():
        try:
            index = 0
            while True:
                image = full.get()
                if image is None:
                    return
                out[index] = image
                index += 1
                free.put(image)
        except Exception as e:
            # The reference's main loop fails as soon as the writer is gone (`image_send.send(image)?`,
            # main.rs:120).  Here: flag it, wake a main loop that waits for a buffer, and keep draining
            # so that its full.put() never blocks; the loop below stops at its next iteration.
            errors.append(e)
            failed.set()
            free.put(None)
            while full.get() is not None:
                pass

    thread = threading.Thread(target=writer, daemon=True)
    thread.start()
    t0 = time.perf_counter()
    pending = []                                            # images on their way from the device, oldest first (at most 2)
    try:
        for _ in range(args.nbimage):
            image = free.get()
            if failed.is_set() or image is None:
                break
            sim.prepare_steps(species, steps_per_image)     # enqueued; nothing here waits for the device
            species.write_result_view_after(image)          # this image: staged + copied behind those steps while we go on
            pending.append(image)
            if len(pending) == 2:
                # two images in flight (the library stages them in two buffers in turn): the host copy of the newer one
                # overlaps the next steps AND the hand-over of the older one, so the PCIe link never idles between images
                ctx.download_wait(in_flight=1)              # the older image is complete ...
                full.put(pending.pop(0))                    # ... hand it to the I/O thread
        if pending and not failed.is_set():
            ctx.download_wait()
            for image in pending:
                full.put(image)
    finally:
        full.put(None)
        thread.join()
        ctx.sync()
    elapsed = time.perf_counter() - t0
    if errors:
        raise errors[0]
    if hasattr(out, "flush"):
        out.flush()
    cells = shape[0] * shape[1]
    info = {
        "images": args.nbimage, "steps_per_image": steps_per_image, "shape": shape, "seconds": elapsed,
        "mcells_steps_per_s": cells * steps_per_image * args.nbimage / elapsed / 1e6,
        "image_MB_per_s": cells * 4 * args.nbimage / elapsed / 1e6,
    }
    ctx.close()
    return info


def main(argv=None) -> int:
    args = parse(argv)
    info = run(args)
    print("simulate: {images} images x {steps_per_image} steps on {shape[0]}x{shape[1]} in {seconds:.3f} s "
          "({mcells_steps_per_s:.0f} Mcells*steps/s, {image_MB_per_s:.0f} MB/s of images)".format(**info),
          file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
