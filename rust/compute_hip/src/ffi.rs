//! `extern "C"` view of include/gs_hip.h (ABI version 4).  Field order and widths must match
//! the header exactly; `tests/test_capi_cpu.py::test_struct_layouts` pins the C side.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_void};

pub const GS_OK: i32 = 0;
pub const GS_UNIQUE_ID_BYTES: usize = 128;

#[repr(C)]
#[derive(Copy, Clone, Debug)]
pub struct gs_params {
    pub w: [[f32; 3]; 3],
    pub du: f32,
    pub dv: f32,
    pub feed: f32,
    pub kill: f32,
    pub dt: f32,
}

#[repr(C)]
#[derive(Copy, Clone, Debug, Default)]
pub struct gs_options {
    pub math: i32,
    pub kernel: i32,
    pub rows_per_block: i32,
    pub fuse_steps: i32,
    pub use_graph: i32,
    pub pitch_pad: i32,
    pub split: i32,
    pub general_kernels: i32,
    pub cols_per_lane: i32,
    pub boundary: i32,
    pub no_tune: i32,
    pub tile_shape: i32,
    pub share_taps: i32,
    pub reserved: [i32; 3],
}

#[repr(C)]
pub struct gs_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct gs_field {
    _private: [u8; 0],
}

extern "C" {
    pub fn gs_abi_version() -> i32;
    pub fn gs_last_error() -> *const c_char;
    pub fn gs_default_options(out: *mut gs_options);
    pub fn gs_get_unique_id(out128: *mut c_void) -> i32;
    pub fn gs_ctx_create(
        out: *mut *mut gs_ctx,
        params: *const gs_params,
        opts: *const gs_options,
        device_ids: *const i32,
        n_local: i32,
        rank: i32,
        world: i32,
        unique_id: *const c_void,
    ) -> i32;
    pub fn gs_ctx_destroy(ctx: *mut gs_ctx) -> i32;
    pub fn gs_field_create(ctx: *mut gs_ctx, out: *mut *mut gs_field, rows: u64, cols: u64) -> i32;
    pub fn gs_field_destroy(ctx: *mut gs_ctx, f: *mut gs_field) -> i32;
    pub fn gs_field_raw_shape(f: *const gs_field, raw_rows: *mut u64, pitch: *mut u64) -> i32;
    pub fn gs_field_fill(ctx: *mut gs_ctx, f: *mut gs_field, value: f32) -> i32;
    pub fn gs_fields_place(
        ctx: *mut gs_ctx,
        planes: *const *mut gs_field,
        candidates: i32,
        first_ms: *mut f32,
        best_ms: *mut f32,
    ) -> i32;
    pub fn gs_field_fill_slice(
        ctx: *mut gs_ctx,
        f: *mut gs_field,
        r0: u64,
        r1: u64,
        c0: u64,
        c1: u64,
        value: f32,
    ) -> i32;
    pub fn gs_field_finalize(ctx: *mut gs_ctx, f: *mut gs_field) -> i32;
    pub fn gs_field_download(ctx: *mut gs_ctx, f: *mut gs_field, host: *mut f32) -> i32;
    pub fn gs_step(
        ctx: *mut gs_ctx,
        in_u: *mut gs_field,
        in_v: *mut gs_field,
        out_u: *mut gs_field,
        out_v: *mut gs_field,
    ) -> i32;
    pub fn gs_run(
        ctx: *mut gs_ctx,
        u0: *mut gs_field,
        v0: *mut gs_field,
        u1: *mut gs_field,
        v1: *mut gs_field,
        steps: u64,
        result_slot: *mut i32,
    ) -> i32;
    pub fn gs_sync(ctx: *mut gs_ctx) -> i32;
    pub fn gs_field_download_async(ctx: *mut gs_ctx, f: *mut gs_field, host: *mut f32) -> i32;
    pub fn gs_download_wait(ctx: *mut gs_ctx) -> i32;
    pub fn gs_download_wait_but(ctx: *mut gs_ctx, in_flight: i32) -> i32;
    pub fn gs_field_colormap(
        ctx: *mut gs_ctx,
        f: *mut gs_field,
        scale: f32,
        palette_rgb: *const u8,
        n_colors: i32,
        host_rgb: *mut u8,
    ) -> i32;
}
