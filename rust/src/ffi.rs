//! `extern "C"` declarations of every entry point of include/tfhe_hip.h.
//!
//! SOURCE ONLY: no Rust toolchain exists in the image this repository is built in;
//! this file has never been compiled.  It is generated from the header's prototypes
//! (one declaration per exported symbol, same order) and is the one module of the
//! crate that needs `unsafe`: the crate root's `#![forbid(unsafe_code)]`
//! (src/lib.rs:25) becomes `#![deny(unsafe_code)]` with `#[allow(unsafe_code)]` on
//! `mod ffi;` and `mod gpu;`.
#![allow(unsafe_code, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

/// tfhe_hip_params: the integers of Parameters / TGswParams / LweKeySwitchKey the path reads.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct TfheHipParams {
    pub n: i32,
    pub big_n: i32,
    pub k: i32,
    pub l: i32,
    pub bg_bit: i32,
    pub ks_t: i32,
    pub ks_base_bit: i32,
    pub n_ks: i32,
}

/// Opaque context (one GPU, or a list of GPUs).
pub enum TfheHipCtx {}
/// Opaque device-resident batch of LWE samples.
pub enum TfheHipSamples {}

pub const TFHE_HIP_OK: c_int = 0;
pub const TFHE_HIP_ERR_ARG: c_int = -1;
pub const TFHE_HIP_ERR_STATE: c_int = -2;
pub const TFHE_HIP_ERR_RUNTIME: c_int = -3;

/// Gate ids (src/gates.rs:23-268).
pub const NAND: c_int = 0;
pub const OR: c_int = 1;
pub const AND: c_int = 2;
pub const XOR: c_int = 3;
pub const XNOR: c_int = 4;
pub const NOR: c_int = 5;
pub const ANDNY: c_int = 6;
pub const ANDYN: c_int = 7;
pub const ORNY: c_int = 8;
pub const ORYN: c_int = 9;
pub const MUX: c_int = 10;

pub const BR_AUTO: c_int = 0;
pub const BR_BATCH: c_int = 1;
pub const BR_SPLIT: c_int = 2;
pub const BR_BATCH64: c_int = 3;

extern "C" {
    pub fn tfhe_hip_variant() -> *const c_char;
    pub fn tfhe_hip_create(out: *mut *mut TfheHipCtx, device: c_int) -> c_int;
    pub fn tfhe_hip_create_multi(
        out: *mut *mut TfheHipCtx,
        devices: *const c_int,
        ndev: c_int,
    ) -> c_int;
    pub fn tfhe_hip_key_transport(ctx: *mut TfheHipCtx) -> *const c_char;
    pub fn tfhe_hip_destroy(ctx: *mut TfheHipCtx);
    pub fn tfhe_hip_last_error(ctx: *mut TfheHipCtx) -> *const c_char;
    pub fn tfhe_hip_set_stream(ctx: *mut TfheHipCtx, hip_stream: *mut c_void) -> c_int;
    pub fn tfhe_hip_synchronize(ctx: *mut TfheHipCtx) -> c_int;
    pub fn tfhe_hip_set_blind_rotate_mode(ctx: *mut TfheHipCtx, mode: c_int) -> c_int;
    pub fn tfhe_hip_profile_enable(ctx: *mut TfheHipCtx, on: c_int) -> c_int;
    pub fn tfhe_hip_profile_read(
        ctx: *mut TfheHipCtx,
        ms_sum: *mut f64,
        launches: *mut c_int,
    ) -> c_int;
    pub fn tfhe_hip_load_keys(
        ctx: *mut TfheHipCtx,
        params: *const TfheHipParams,
        bk: *const i32,
        ks_a: *const i32,
        ks_b: *const i32,
        ks_var: *const f64,
    ) -> c_int;
    pub fn tfhe_hip_generate_keys(
        ctx: *mut TfheHipCtx,
        params: *const TfheHipParams,
        ks_stdev: f64,
        bk_stdev: f64,
        seed: *const u8,
        lwe_key: *mut i32,
        tlwe_key: *mut i32,
        bk_out: *mut i32,
        ks_a_out: *mut i32,
        ks_b_out: *mut i32,
        ks_var_out: *mut f64,
    ) -> c_int;
    pub fn tfhe_hip_keygen_stream(
        ctx: *mut TfheHipCtx,
        seed: *const u8,
        purpose: u32,
        stream: u64,
        first_block: u32,
        out: *mut u32,
        nblocks: usize,
    ) -> c_int;
    pub fn tfhe_hip_key_file_size(
        params: *const TfheHipParams,
        has_ks_var: c_int,
        bytes: *mut usize,
    ) -> c_int;
    pub fn tfhe_hip_key_file_pack(
        params: *const TfheHipParams,
        bk: *const i32,
        ks_a: *const i32,
        ks_b: *const i32,
        ks_var: *const f64,
        out: *mut c_void,
        capacity: usize,
    ) -> c_int;
    pub fn tfhe_hip_load_keys_blob(
        ctx: *mut TfheHipCtx,
        blob: *const c_void,
        bytes: usize,
    ) -> c_int;
    pub fn tfhe_hip_key_blob_sizes(
        params: *const TfheHipParams,
        bk_bytes: *mut usize,
        ks_bytes: *mut usize,
    ) -> c_int;
    pub fn tfhe_hip_export_keys_dev(
        ctx: *mut TfheHipCtx,
        bk_blob_dev: *mut c_void,
        ks_blob_dev: *mut c_void,
    ) -> c_int;
    pub fn tfhe_hip_import_keys_dev(
        ctx: *mut TfheHipCtx,
        params: *const TfheHipParams,
        bk_blob_dev: *const c_void,
        ks_blob_dev: *const c_void,
    ) -> c_int;
    pub fn tfhe_hip_bootstrap(
        ctx: *mut TfheHipCtx,
        mu: i32,
        a: *const i32,
        b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_bootstrap_dev(
        ctx: *mut TfheHipCtx,
        mu: i32,
        a: *const i32,
        b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_bootstrap_noks(
        ctx: *mut TfheHipCtx,
        mu: i32,
        a: *const i32,
        b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_bootstrap_noks_dev(
        ctx: *mut TfheHipCtx,
        mu: i32,
        a: *const i32,
        b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_keyswitch(
        ctx: *mut TfheHipCtx,
        a: *const i32,
        b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_keyswitch_dev(
        ctx: *mut TfheHipCtx,
        a: *const i32,
        b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_gate(
        ctx: *mut TfheHipCtx,
        gate: c_int,
        ca_a: *const i32,
        ca_b: *const i32,
        cb_a: *const i32,
        cb_b: *const i32,
        cc_a: *const i32,
        cc_b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_gate_dev(
        ctx: *mut TfheHipCtx,
        gate: c_int,
        ca_a: *const i32,
        ca_b: *const i32,
        cb_a: *const i32,
        cb_b: *const i32,
        cc_a: *const i32,
        cc_b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_gate_dev_noalloc(
        ctx: *mut TfheHipCtx,
        gate: c_int,
        ca_a: *const i32,
        ca_b: *const i32,
        cb_a: *const i32,
        cb_b: *const i32,
        cc_a: *const i32,
        cc_b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_not_dev(
        ctx: *mut TfheHipCtx,
        a: *const i32,
        b: *const i32,
        var: *const f64,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_constant_dev(
        ctx: *mut TfheHipCtx,
        value: c_int,
        out_a: *mut i32,
        out_b: *mut i32,
        var_out: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_samples_alloc(
        ctx: *mut TfheHipCtx,
        B: usize,
        out: *mut *mut TfheHipSamples,
    ) -> c_int;
    pub fn tfhe_hip_samples_free(s: *mut TfheHipSamples);
    pub fn tfhe_hip_samples_count(s: *const TfheHipSamples) -> usize;
    pub fn tfhe_hip_samples_upload(
        ctx: *mut TfheHipCtx,
        s: *mut TfheHipSamples,
        a: *const i32,
        b: *const i32,
        var: *const f64,
    ) -> c_int;
    pub fn tfhe_hip_samples_download(
        ctx: *mut TfheHipCtx,
        s: *const TfheHipSamples,
        a: *mut i32,
        b: *mut i32,
        var: *mut f64,
    ) -> c_int;
    pub fn tfhe_hip_samples_gate(
        ctx: *mut TfheHipCtx,
        gate: c_int,
        ca: *const TfheHipSamples,
        cb: *const TfheHipSamples,
        cc: *const TfheHipSamples,
        out: *mut TfheHipSamples,
    ) -> c_int;
    pub fn tfhe_hip_samples_bootstrap(
        ctx: *mut TfheHipCtx,
        mu: i32,
        x: *const TfheHipSamples,
        out: *mut TfheHipSamples,
    ) -> c_int;
    pub fn tfhe_hip_samples_not(
        ctx: *mut TfheHipCtx,
        x: *const TfheHipSamples,
        out: *mut TfheHipSamples,
    ) -> c_int;
    pub fn tfhe_hip_samples_constant(
        ctx: *mut TfheHipCtx,
        value: c_int,
        out: *mut TfheHipSamples,
    ) -> c_int;
    pub fn tfhe_hip_circuit_add(
        ctx: *mut TfheHipCtx,
        a_a: *const i32,
        a_b: *const i32,
        b_a: *const i32,
        b_b: *const i32,
        sum_a: *mut i32,
        sum_b: *mut i32,
        carry_a: *mut i32,
        carry_b: *mut i32,
        sum_var: *mut f64,
        carry_var: *mut f64,
        count: usize,
        bits: c_int,
    ) -> c_int;
    pub fn tfhe_hip_circuit_add_dev(
        ctx: *mut TfheHipCtx,
        a_a: *const i32,
        a_b: *const i32,
        b_a: *const i32,
        b_b: *const i32,
        sum_a: *mut i32,
        sum_b: *mut i32,
        carry_a: *mut i32,
        carry_b: *mut i32,
        sum_var: *mut f64,
        carry_var: *mut f64,
        count: usize,
        bits: c_int,
    ) -> c_int;
    pub fn tfhe_hip_circuit_compare(
        ctx: *mut TfheHipCtx,
        a_a: *const i32,
        a_b: *const i32,
        b_a: *const i32,
        b_b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        out_var: *mut f64,
        count: usize,
        bits: c_int,
    ) -> c_int;
    pub fn tfhe_hip_circuit_compare_dev(
        ctx: *mut TfheHipCtx,
        a_a: *const i32,
        a_b: *const i32,
        b_a: *const i32,
        b_b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        out_var: *mut f64,
        count: usize,
        bits: c_int,
    ) -> c_int;
    pub fn tfhe_hip_circuit_eq(
        ctx: *mut TfheHipCtx,
        a_a: *const i32,
        a_b: *const i32,
        b_a: *const i32,
        b_b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        out_var: *mut f64,
        count: usize,
        bits: c_int,
    ) -> c_int;
    pub fn tfhe_hip_circuit_eq_dev(
        ctx: *mut TfheHipCtx,
        a_a: *const i32,
        a_b: *const i32,
        b_a: *const i32,
        b_b: *const i32,
        out_a: *mut i32,
        out_b: *mut i32,
        out_var: *mut f64,
        count: usize,
        bits: c_int,
    ) -> c_int;
    pub fn tfhe_hip_polymul(
        ctx: *mut TfheHipCtx,
        d: *const i32,
        s: *const i32,
        out: *mut i32,
        accumulate: c_int,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_polymul_dev(
        ctx: *mut TfheHipCtx,
        d: *const i32,
        s: *const i32,
        out: *mut i32,
        accumulate: c_int,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_polymul_margin(
        ctx: *mut TfheHipCtx,
        d: *const i32,
        s: *const i32,
        margin: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_polymul_margin_dev(
        ctx: *mut TfheHipCtx,
        d: *const i32,
        s: *const i32,
        margin: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_extprod(
        ctx: *mut TfheHipCtx,
        t: *const i32,
        bk_index: c_int,
        out: *mut i32,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_extprod_margin(
        ctx: *mut TfheHipCtx,
        t: *const i32,
        bk_index: c_int,
        out: *mut i32,
        margin: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_extprod_dev(
        ctx: *mut TfheHipCtx,
        t: *const i32,
        bk_index: c_int,
        out: *mut i32,
        margin: *mut f64,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_blind_rotate(
        ctx: *mut TfheHipCtx,
        acc: *mut i32,
        bara: *const i32,
        B: usize,
    ) -> c_int;
    pub fn tfhe_hip_blind_rotate_dev(
        ctx: *mut TfheHipCtx,
        acc: *mut i32,
        bara: *const i32,
        B: usize,
    ) -> c_int;
}
