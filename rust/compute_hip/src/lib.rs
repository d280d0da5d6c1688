//! MI355X (gfx950) HIP implementation of Gray-Scott simulation
//!
//! Thin shim over libgs_hip.so (hand-written HIP kernels behind a C ABI).  Storage lives in
//! HBM as plain row-major f32 planes.  As in the Vulkan backends (compute/shared/src/gpu/mod.rs:
//! 70-91) there are two forms of the step call: `prepare_steps` enqueues all steps with one
//! asynchronous `gs_run` (the Vulkan backends batch the same way: one command buffer for N steps,
//! compute/gpu/naive/src/lib.rs:99-131) and `Simulate::perform_steps` is that plus a wait, so the
//! reference's criterion "compute" workload (compute/shared/src/benchmark.rs:77-83) times finished
//! steps.  Results match `compute_naive` bit for bit (clipped-window boundary rule,
//! FTZ-without-DAZ denormal handling).
//!
//! `HipContext` is an `Rc`, so `Species<HipConcentration>` is `!Send`: fine for `simulate`, whose
//! compute stays on the main thread (simulate/src/main.rs:73-121); `HipError` is `Send + Sync`
//! as `SimulateBase::Error` requires.

mod ffi;

use clap::Args;
use compute::{Simulate, SimulateBase, SimulateCreate};
use data::{
    concentration::{Concentration, Species},
    parameters::Parameters,
    Precision,
};
use ndarray::{Array2, ArrayView2, ArrayViewMut2};
use std::{ffi::CStr, ops::Range, ptr, rc::Rc};
use thiserror::Error;

/// Errors reported by libgs_hip.so (status code + gs_last_error() text)
#[derive(Clone, Debug, Error)]
#[error("gs_hip error {code}: {message}")]
pub struct HipError {
    pub code: i32,
    pub message: String,
}

fn check(code: i32) -> Result<(), HipError> {
    if code == ffi::GS_OK {
        return Ok(());
    }
    // SAFETY: gs_last_error returns a NUL-terminated thread-local string owned by the library
    let message = unsafe { CStr::from_ptr(ffi::gs_last_error()) }
        .to_string_lossy()
        .into_owned();
    Err(HipError { code, message })
}

/// CLI parameters of the HIP backend (all defaulted, all settable from the environment, as
/// compute/shared/src/lib.rs:20-25 requires)
#[derive(Args, Clone, Debug)]
pub struct HipArgs {
    /// HIP devices that run the simulation, one row slab each, top to bottom (ids may repeat)
    #[arg(long, env = "GS_HIP_DEVICES", value_delimiter = ',', default_value = "0")]
    pub hip_devices: Vec<i32>,

    /// Arithmetic flavour: 0 = strict (bit-identical to compute_naive), 1 = fused taps
    #[arg(long, env = "GS_HIP_MATH", default_value_t = 0)]
    pub hip_math: i32,

    /// Rows each wavefront marches over (0 = chosen on line)
    #[arg(long, env = "GS_HIP_ROWS_PER_BLOCK", default_value_t = 0)]
    pub hip_rows_per_block: i32,

    /// Time steps fused per pass over HBM, 1..4 (0 = chosen on line)
    #[arg(long, env = "GS_HIP_FUSE_STEPS", default_value_t = 0)]
    pub hip_fuse_steps: i32,

    /// Columns per lane of the temporally blocked kernel: 4, 2 or 1 (0 = chosen on line)
    #[arg(long, env = "GS_HIP_COLS_PER_LANE", default_value_t = 0)]
    pub hip_cols_per_lane: i32,

    /// 1 = never time candidate configurations inside perform_steps (untuned defaults, or the
    /// pinned values above)
    #[arg(long, env = "GS_HIP_NO_TUNE", default_value_t = 0)]
    pub hip_no_tune: i32,

    /// Step kernel (`gs_kernel` in gs_hip.h): 0 = chosen by grid size and call length, 1 = one thread
    /// per cell, 2 = single-step streaming, 3 = temporally blocked, 4 = LDS-staged, 5 = LDS-resident
    /// windows, 6 = one persistent launch per call on register-resident windows
    #[arg(long, env = "GS_HIP_KERNEL", default_value_t = 0)]
    pub hip_kernel: i32,

    /// Rule on the edges of the grid (`gs_boundary`): 0 = compute_naive's clipped window, 1 = zero halo
    /// (what the SIMD and Vulkan backends compute: data/src/concentration/simd/mod.rs:281-326)
    #[arg(long, env = "GS_HIP_BOUNDARY", default_value_t = 0)]
    pub hip_boundary: i32,

    /// 1 = never run the kernel variants specialised for the default stencil and time step (A/B timing;
    /// results are bit-identical either way)
    #[arg(long, env = "GS_HIP_GENERAL_KERNELS", default_value_t = 0)]
    pub hip_general_kernels: i32,

    /// Full difference sharing in the temporally blocked kernel: 0 = on (form 3) unless measured slower, 1 = within a lane only, 2 = off,
    /// 3 = across lanes too (bit-identical every way)
    #[arg(long, env = "GS_HIP_SHARE_TAPS", default_value_t = 0)]
    pub hip_share_taps: i32,

    /// Row bands a single slab is scheduled as (0 or 1 = off)
    #[arg(long, env = "GS_HIP_SPLIT", default_value_t = 0)]
    pub hip_split: i32,

    /// 1 = replay batches of 16 passes through a hipGraph (single slab)
    #[arg(long, env = "GS_HIP_USE_GRAPH", default_value_t = 0)]
    pub hip_use_graph: i32,

    /// Window of the LDS-resident-window kernel: 1 = 32 x 64, 2 = 16 x 64, 3 = 64 x 64 (0 = 32 x 64)
    #[arg(long, env = "GS_HIP_TILE_SHAPE", default_value_t = 0)]
    pub hip_tile_shape: i32,

    /// Extra f32 of row pitch beyond the round-up to 64
    #[arg(long, env = "GS_HIP_PITCH_PAD", default_value_t = 0)]
    pub hip_pitch_pad: i32,

    /// Placement by measurement (`gs_fields_place`; not a `gs_options` field): every species of
    /// >= 2^26 cells that `make_species` creates on a single device is given blocks of different
    /// physical regions of HBM for its U and V planes, drawing at most N extra blocks.  Worth
    /// 8-12 % of `perform_steps` at 16384^2; 0 = planes as hipMalloc hands them out
    #[arg(long, env = "GS_HIP_PLACE_CANDIDATES", default_value_t = 12)]
    pub hip_place_candidates: i32,
}

/// Owner of the `gs_ctx` (devices, streams); shared by the simulation and its species
pub struct HipContextInner(*mut ffi::gs_ctx);
//
impl Drop for HipContextInner {
    fn drop(&mut self) {
        // SAFETY: created by gs_ctx_create, destroyed exactly once
        unsafe { ffi::gs_ctx_destroy(self.0) };
    }
}
/// `Concentration::Context` of [`HipConcentration`]
pub type HipContext = Rc<HipContextInner>;

/// One species plane in HBM
pub struct HipConcentration {
    context: HipContext,
    field: *mut ffi::gs_field,
    shape: [usize; 2],
    /// Host mirror, lazily allocated by make_scalar_view (cf. simd/mod.rs:330-345)
    scalar: Array2<Precision>,
}
//
impl HipConcentration {
    fn create(context: &mut HipContext, shape: [usize; 2], fill: Option<Precision>) -> Result<Self, HipError> {
        let mut field = ptr::null_mut();
        // SAFETY: plain FFI call, out pointer is valid
        check(unsafe { ffi::gs_field_create(context.0, &mut field, shape[0] as u64, shape[1] as u64) })?;
        if let Some(value) = fill {
            check(unsafe { ffi::gs_field_fill(context.0, field, value) })?;
        }
        Ok(Self { context: context.clone(), field, shape, scalar: Array2::default([0, 0]) })
    }
}
//
impl Drop for HipConcentration {
    fn drop(&mut self) {
        // SAFETY: the context outlives the field through the Rc
        unsafe { ffi::gs_field_destroy(self.context.0, self.field) };
    }
}
//
impl Concentration for HipConcentration {
    type Context = HipContext;
    type Error = HipError;

    fn default(context: &mut HipContext, shape: [usize; 2]) -> Result<Self, HipError> {
        Self::create(context, shape, None) // planes are created zero-filled
    }
    fn zeros(context: &mut HipContext, shape: [usize; 2]) -> Result<Self, HipError> {
        Self::create(context, shape, None)
    }
    fn ones(context: &mut HipContext, shape: [usize; 2]) -> Result<Self, HipError> {
        Self::create(context, shape, Some(1.0))
    }
    fn shape(&self) -> [usize; 2] {
        self.shape
    }
    fn raw_shape(&self) -> [usize; 2] {
        let (mut rows, mut pitch) = (0u64, 0u64);
        // SAFETY: out pointers are valid; the call cannot fail on a live field
        unsafe { ffi::gs_field_raw_shape(self.field, &mut rows, &mut pitch) };
        [rows as usize, pitch as usize]
    }
    fn fill_slice(
        &mut self,
        context: &mut HipContext,
        [rows, cols]: [Range<usize>; 2],
        value: Precision,
    ) -> Result<(), HipError> {
        check(unsafe {
            ffi::gs_field_fill_slice(
                context.0, self.field, rows.start as u64, rows.end as u64, cols.start as u64,
                cols.end as u64, value,
            )
        })
    }
    fn finalize(&mut self, context: &mut HipContext) -> Result<(), HipError> {
        check(unsafe { ffi::gs_field_finalize(context.0, self.field) })
    }

    type ScalarView<'a> = ArrayView2<'a, Precision>;

    fn make_scalar_view(&mut self, context: &mut HipContext) -> Result<ArrayView2<'_, Precision>, HipError> {
        if self.scalar.is_empty() {
            self.scalar = Array2::default(self.shape);
        }
        let ptr = self.scalar.as_mut_ptr();
        check(unsafe { ffi::gs_field_download(context.0, self.field, ptr) })?; // syncs
        Ok(self.scalar.view())
    }
    fn write_scalar_view(
        &mut self,
        context: &mut HipContext,
        mut target: ArrayViewMut2<Precision>,
    ) -> Result<(), HipError> {
        Self::validate_write(self, &target);
        match target.as_slice_mut() {
            // dense row-major target (what `simulate` passes): download straight into it
            Some(slice) => check(unsafe { ffi::gs_field_download(context.0, self.field, slice.as_mut_ptr()) }),
            None => {
                let view = self.make_scalar_view(context)?;
                target.assign(&view);
                Ok(())
            }
        }
    }
}

/// Gray-Scott reaction simulation
pub struct Simulation {
    context: HipContext,
    place_candidates: i32,
}
//
impl SimulateBase for Simulation {
    type CliArgs = HipArgs;
    type Concentration = HipConcentration;
    type Error = HipError;

    fn make_species(&self, shape: [usize; 2]) -> Result<Species<HipConcentration>, HipError> {
        let mut species = Species::new(self.context.clone(), shape)?;
        // planes of >= 256 MiB: below, they largely stay in the last-level cache
        if self.place_candidates > 0 && shape[0] as u64 * shape[1] as u64 >= 1u64 << 26 {
            let (in_u, in_v, out_u, out_v) = species.in_out();
            let planes = [in_u.field, in_v.field, out_u.field, out_v.field];
            // SAFETY: four live planes of this context; the call moves them to other blocks with their contents
            check(unsafe {
                ffi::gs_fields_place(self.context.0, planes.as_ptr(), self.place_candidates, ptr::null_mut(), ptr::null_mut())
            })?;
        }
        Ok(species)
    }
}
//
impl SimulateCreate for Simulation {
    fn new(params: Parameters, args: HipArgs) -> Result<Self, HipError> {
        let weights = params.weights().0;
        let c_params = ffi::gs_params {
            w: weights,
            du: params.diffusion_rate_u,
            dv: params.diffusion_rate_v,
            feed: params.feed_rate,
            kill: params.kill_rate,
            dt: params.time_step,
        };
        let mut opts = ffi::gs_options::default();
        // SAFETY: fills a plain struct
        unsafe { ffi::gs_default_options(&mut opts) };
        opts.math = args.hip_math;
        opts.rows_per_block = args.hip_rows_per_block;
        opts.fuse_steps = args.hip_fuse_steps;
        opts.cols_per_lane = args.hip_cols_per_lane;
        opts.no_tune = args.hip_no_tune;
        opts.kernel = args.hip_kernel;
        opts.boundary = args.hip_boundary;
        opts.general_kernels = args.hip_general_kernels;
        opts.share_taps = args.hip_share_taps;
        opts.split = args.hip_split;
        opts.use_graph = args.hip_use_graph;
        opts.tile_shape = args.hip_tile_shape;
        opts.pitch_pad = args.hip_pitch_pad;
        let mut ctx = ptr::null_mut();
        // one process, `hip_devices.len()` row slabs with ghost-row exchange by peer copies
        check(unsafe {
            ffi::gs_ctx_create(
                &mut ctx, &c_params, &opts, args.hip_devices.as_ptr(), args.hip_devices.len() as i32, 0, 1,
                ptr::null(),
            )
        })?;
        Ok(Self { context: Rc::new(HipContextInner(ctx)), place_candidates: if args.hip_devices.len() == 1 { args.hip_place_candidates } else { 0 } })
    }
}
//
impl Simulate for Simulation {
    /// Synchronous, like every backend of the reference: the Vulkan ones end `perform_steps_impl`
    /// with `.then_signal_fence_and_flush()?.wait(None)?` (compute/shared/src/gpu/mod.rs:77-91).
    fn perform_steps(&self, species: &mut Species<HipConcentration>, steps: usize) -> Result<(), HipError> {
        self.prepare_steps(species, steps)?;
        // SAFETY: plain FFI call on a live context
        check(unsafe { ffi::gs_sync(self.context.0) })
    }
}

impl Simulation {
    /// Asynchronous form, the counterpart of `SimulateGpu::prepare_steps`
    /// (compute/shared/src/gpu/mod.rs:70-75): one call enqueues every step -- the library ping-pongs
    /// between the two slots and fuses up to 4 time steps per pass over HBM -- and returns.  HIP
    /// streams order the work, so there is no future to thread through: whatever is enqueued next
    /// on this context (more steps, `write_scalar_view_after`) runs behind it.
    pub fn prepare_steps(&self, species: &mut Species<HipConcentration>, steps: usize) -> Result<(), HipError> {
        let mut slot = 0i32;
        {
            let (in_u, in_v, out_u, out_v) = species.in_out();
            check(unsafe {
                ffi::gs_run(self.context.0, in_u.field, in_v.field, out_u.field, out_v.field, steps as u64, &mut slot)
            })?;
        }
        // "At the end of the simulation, the input concentrations of `species` will contain the
        // final simulation results" (compute/shared/src/lib.rs:51-52): if the newest state sits
        // in the output slot, swap the Rust-side handles (finalize() is a no-op on stepped planes).
        if slot == 1 {
            species.flip()?;
        }
        Ok(())
    }

    /// `SimulateStep`-style single step (compute/shared/src/cpu.rs:21-28): enqueue, then flip
    pub fn perform_step(&self, species: &mut Species<HipConcentration>) -> Result<(), HipError> {
        let (in_u, in_v, out_u, out_v) = species.in_out();
        check(unsafe { ffi::gs_step(self.context.0, in_u.field, in_v.field, out_u.field, out_v.field) })?;
        species.flip()
    }

    /// Wait for the downloads enqueued by `write_scalar_view_after` (not for later steps)
    pub fn download_wait(&self) -> Result<(), HipError> {
        check(unsafe { ffi::gs_download_wait(self.context.0) })
    }

    /// ... for all but the newest `in_flight` (0 or 1) of them: with two images on their way (the library stages
    /// them in two buffers in turn) the PCIe link never idles between images -- what `simulate`'s image channel
    /// of depth 2 (simulate/src/main.rs:29-43) allows
    pub fn download_wait_but(&self, in_flight: i32) -> Result<(), HipError> {
        check(unsafe { ffi::gs_download_wait_but(self.context.0, in_flight) })
    }
}

impl HipConcentration {
    /// The per-pixel work of data-to-pics (data-to-pics/src/main.rs:139-144) on the device: paints this
    /// plane into dense RGB8 `[rows, cols, 3]` through `palette` (e.g. the 256 colours
    /// `(0..256).map(|i| ui::GRADIENT.eval_rational(i, 256))` flattened to r, g, b bytes) with
    /// `scale` = `ui::AMPLITUDE_SCALE`.
    pub fn colormap(&mut self, context: &mut HipContext, scale: Precision, palette: &[u8], rgb: &mut [u8]) -> Result<(), HipError> {
        assert_eq!(palette.len() % 3, 0);
        assert_eq!(rgb.len(), self.shape[0] * self.shape[1] * 3);
        check(unsafe {
            ffi::gs_field_colormap(context.0, self.field, scale, palette.as_ptr(), (palette.len() / 3) as i32, rgb.as_mut_ptr())
        })
    }

    /// Counterpart of `ImageConcentration::write_scalar_view_after`
    /// (data/src/concentration/gpu/image/mod.rs:196-206), used with `prepare_steps` by a driver loop
    /// like simulate/src/main.rs:99-106: enqueue the download of this plane behind the steps already
    /// enqueued and return at once.
    ///
    /// # Safety
    /// `target` must be dense row-major and must stay alive and untouched until
    /// `Simulation::download_wait` (or any synchronous call on the context) has returned.
    pub unsafe fn write_scalar_view_after(
        &mut self,
        context: &mut HipContext,
        mut target: ArrayViewMut2<Precision>,
    ) -> Result<(), HipError> {
        Self::validate_write(self, &target);
        let slice = target.as_slice_mut().expect("write_scalar_view_after needs a dense row-major target");
        check(ffi::gs_field_download_async(context.0, self.field, slice.as_mut_ptr()))
    }
}
