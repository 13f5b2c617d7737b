//! GPU-backed bodies for the crate's bootstrapped path: a context that owns the
//! keys on the GPU(s), the flattening of `LweBootstrappingKey` into the arrays
//! `tfhe_hip_load_keys` takes, and the replacements for `tfhe_bootstrap`, the
//! gates, `mux` and `circuits::add`.
//!
//! SOURCE ONLY: no Rust toolchain exists in the image this repository is built in;
//! this file has never been compiled.  It is written against the crate's types as
//! they are (IsakSundeSingh/tfhe 0.1.0) and contains nothing but the new bodies.
//!
//! What the maintainer changes in the crate itself:
//!   * src/lib.rs:25  `#![forbid(unsafe_code)]` -> `#![deny(unsafe_code)]`, and
//!     `#[allow(unsafe_code)] mod ffi; #[allow(unsafe_code)] pub mod gpu;`
//!   * fields this module reads become `pub(crate)`: `LweKeySwitchKey::{n, t, base_bit, ks}`
//!     (src/lwe.rs:496-513), `TGswSample::all_sample` (src/tgsw.rs:99),
//!     `TGswParams::{l, bg_bit}` (src/tgsw.rs:10-13), `LweParams::n`
//!   * `LweBootstrappingKey` (src/lwe.rs:432-446) gains `#[serde(skip)] pub(crate) gpu: gpu::GpuSlot`
//!     (`GpuSlot::default()` in `LweBootstrappingKey::new`, src/lwe.rs:464-471).  The slot is
//!     empty until the first bootstrapped call, which flattens the key and uploads it; it is
//!     `Clone` (clones share the uploaded keys) and comes back empty from deserialisation.
//!     The handle sits on the bootstrapping key, not on `CloudKey`, because
//!     `tfhe_bootstrap(bk: &LweBootstrappingKey, ..)` (src/bootstrapping.rs:25) only sees that:
//!     its callers pass `&cloud_key.bk` (src/gates.rs:35, benches/bootstrapping.rs:21).
//!   * with `--features bootstrapping`, `tfhe_bootstrap` (src/bootstrapping.rs:25-29) calls
//!     `gpu::bootstrap(&bk.gpu.get(bk), mu, x)` and the gate bodies of src/gates.rs, which
//!     take `bk: &CloudKey`, call `gpu::gate(&bk.bk.gpu.get(&bk.bk), ..)` (table at the end
//!     of this file).
#![allow(unsafe_code)]
use crate::ffi::*;
use crate::lwe::{LweBootstrappingKey, LweSample};
use crate::numerics::Torus32;
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;
use std::sync::{Arc, Mutex};

/// The keys of one `LweBootstrappingKey`, resident on the GPU(s).  A context is not
/// re-entrant (one call at a time): every FFI call below runs with the mutex HELD, the guard
/// is dropped only after the call and its error check have returned.
pub struct GpuKeys {
    ctx: Mutex<*mut TfheHipCtx>,
    n: usize,
    big_n: usize,
}

// The raw context pointer is only dereferenced (by the library) while the mutex is held.
unsafe impl Send for GpuKeys {}
unsafe impl Sync for GpuKeys {}

fn check(ctx: *mut TfheHipCtx, rc: c_int) {
    if rc != TFHE_HIP_OK {
        // the crate reports malformed input by panicking (zip_eq, asserts at src/lwe.rs:476-477)
        let msg = unsafe { CStr::from_ptr(tfhe_hip_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("tfhe_hip: error {}: {}", rc, msg);
    }
}

impl GpuKeys {
    /// `devices`: GPU ordinals; one entry = single-GPU context, several = batches are split
    /// over them and the keys are replicated once (RCCL broadcast inside the library).
    pub fn new(bk: &LweBootstrappingKey, devices: &[i32]) -> Self {
        // decomposition length and log2(base) of the key-switch key (src/lwe.rs:498-502)
        let (ks_t, ks_base_bit) = (bk.ks.t, bk.ks.base_bit);
        let n = bk.in_out_params.n as usize;
        let big_n = bk.accum_params.n as usize;
        let k = bk.accum_params.k as usize;
        let l = bk.bk_params.l as usize;
        let params = TfheHipParams {
            n: n as i32,
            big_n: big_n as i32,
            k: k as i32,
            l: l as i32,
            bg_bit: bk.bk_params.bg_bit,
            ks_t,
            ks_base_bit,
            // the key-switch key is built with the in/out dimension (src/lwe.rs:461)
            n_ks: bk.ks.n,
        };

        // bk [n][l][k+1][k+1][N]: bk[i].all_sample[row][col].a[poly], first N coefficients
        // (row-0 samples carry a (2N-1)-long second polynomial, src/tlwe.rs:112-134; only its
        // first N coefficients can reach a result, src/numerics.rs:152; rows >= 1 hold k+2 zero
        // polynomials of which the first k+1 are read, src/tgsw.rs:229-264)
        let mut flat_bk: Vec<i32> = Vec::with_capacity(n * l * (k + 1) * (k + 1) * big_n);
        for sample in &bk.bk {
            for row in &sample.all_sample {
                for tlwe in row {
                    for poly in tlwe.a.iter().take(k + 1) {
                        flat_bk.extend(poly.coefs.iter().take(big_n).cloned());
                        debug_assert!(poly.coefs.len() >= big_n);
                    }
                }
            }
        }
        assert_eq!(flat_bk.len(), n * l * (k + 1) * (k + 1) * big_n);

        // ks_a [n_ks][t][base][n], ks_b / ks_var [n_ks][t][base] from ks.ks[i][j][h]
        let base = 1usize << ks_base_bit;
        let rows = bk.ks.n as usize * ks_t as usize * base;
        let mut ks_a: Vec<i32> = Vec::with_capacity(rows * n);
        let mut ks_b: Vec<i32> = Vec::with_capacity(rows);
        let mut ks_var: Vec<f64> = Vec::with_capacity(rows);
        for per_coefficient in &bk.ks.ks {
            for per_digit_position in per_coefficient {
                for sample in per_digit_position {
                    ks_a.extend_from_slice(&sample.coefficients);
                    ks_b.push(sample.b);
                    ks_var.push(sample.current_variance);
                }
            }
        }
        assert_eq!(ks_a.len(), rows * n);

        let mut ctx: *mut TfheHipCtx = ptr::null_mut();
        let rc = unsafe {
            if devices.len() == 1 {
                tfhe_hip_create(&mut ctx, devices[0])
            } else {
                tfhe_hip_create_multi(&mut ctx, devices.as_ptr(), devices.len() as c_int)
            }
        };
        assert_eq!(rc, TFHE_HIP_OK, "tfhe_hip_create failed: no usable GPU");
        let rc = unsafe {
            tfhe_hip_load_keys(ctx, &params, flat_bk.as_ptr(), ks_a.as_ptr(), ks_b.as_ptr(), ks_var.as_ptr())
        };
        check(ctx, rc);
        GpuKeys { ctx: Mutex::new(ctx), n, big_n }
    }
}

impl Drop for GpuKeys {
    fn drop(&mut self) {
        // a poisoned mutex still guards a live context: destroy it either way
        let guard = self.ctx.lock().unwrap_or_else(|e| e.into_inner());
        unsafe { tfhe_hip_destroy(*guard) };
        drop(guard);
    }
}

/// GPU ordinals a freshly built `GpuKeys` uses: `TFHE_HIP_DEVICES=0,1,2,3` (default `0`).
fn devices_from_env() -> Vec<i32> {
    std::env::var("TFHE_HIP_DEVICES")
        .ok()
        .map(|v| v.split(',').filter_map(|d| d.trim().parse().ok()).collect::<Vec<i32>>())
        .filter(|v| !v.is_empty())
        .unwrap_or_else(|| vec![0])
}

/// The field `LweBootstrappingKey` gains: the uploaded keys, built on first use.
#[derive(Default)]
pub struct GpuSlot(Mutex<Option<Arc<GpuKeys>>>);

impl GpuSlot {
    /// The uploaded keys of `bk` (the key this slot is a field of); flattens and uploads them
    /// on the first call.
    pub fn get(&self, bk: &LweBootstrappingKey) -> Arc<GpuKeys> {
        let mut slot = self.0.lock().unwrap();
        slot.get_or_insert_with(|| Arc::new(GpuKeys::new(bk, &devices_from_env()))).clone()
    }
    /// As `get`, on an explicit device list (several entries: batches are split over them).
    pub fn get_on(&self, bk: &LweBootstrappingKey, devices: &[i32]) -> Arc<GpuKeys> {
        let mut slot = self.0.lock().unwrap();
        slot.get_or_insert_with(|| Arc::new(GpuKeys::new(bk, devices))).clone()
    }
}

impl Clone for GpuSlot {
    fn clone(&self) -> Self {
        GpuSlot(Mutex::new(self.0.lock().unwrap().clone()))
    }
}

/// Structure-of-arrays view of a slice of samples: masks [B][n], bodies [B].
fn flatten(samples: &[&LweSample], n: usize) -> (Vec<i32>, Vec<i32>) {
    let mut a = Vec::with_capacity(samples.len() * n);
    let mut b = Vec::with_capacity(samples.len());
    for s in samples {
        assert_eq!(s.coefficients.len(), n, "sample dimension does not match the key");
        a.extend_from_slice(&s.coefficients);
        b.push(s.b);
    }
    (a, b)
}

fn unflatten(a: Vec<i32>, b: Vec<i32>, var: Vec<f64>, n: usize) -> Vec<LweSample> {
    a.chunks(n)
        .zip(b.into_iter().zip(var.into_iter()))
        .map(|(coefficients, (b, current_variance))| LweSample {
            coefficients: coefficients.to_vec(),
            b,
            current_variance,
        })
        .collect()
}

/// Replaces the body of `tfhe_bootstrap` (src/bootstrapping.rs:25-29), batched.
pub fn bootstrap_many(keys: &GpuKeys, mu: Torus32, xs: &[&LweSample]) -> Vec<LweSample> {
    let (n, batch) = (keys.n, xs.len());
    let (a, b) = flatten(xs, n);
    let (mut out_a, mut out_b, mut var) = (vec![0i32; batch * n], vec![0i32; batch], vec![0f64; batch]);
    let guard = keys.ctx.lock().unwrap();
    let ctx = *guard;
    let rc = unsafe {
        tfhe_hip_bootstrap(ctx, mu, a.as_ptr(), b.as_ptr(), out_a.as_mut_ptr(), out_b.as_mut_ptr(),
                           var.as_mut_ptr(), batch)
    };
    check(ctx, rc);
    drop(guard);
    unflatten(out_a, out_b, var, n)
}

pub fn bootstrap(keys: &GpuKeys, mu: Torus32, x: LweSample) -> LweSample {
    bootstrap_many(keys, mu, &[&x]).pop().unwrap()
}

/// Replaces `tfhe_bootstrap_without_key_switching` (src/bootstrapping.rs:38-65): the result has
/// N*k coefficients and variance 0 (src/tlwe.rs:165 builds a fresh sample).
pub fn bootstrap_without_key_switching(keys: &GpuKeys, mu: Torus32, x: LweSample) -> LweSample {
    let (a, b) = flatten(&[&x], keys.n);
    let (mut out_a, mut out_b) = (vec![0i32; keys.big_n], vec![0i32; 1]);
    let guard = keys.ctx.lock().unwrap();
    let ctx = *guard;
    let rc = unsafe {
        tfhe_hip_bootstrap_noks(ctx, mu, a.as_ptr(), b.as_ptr(), out_a.as_mut_ptr(), out_b.as_mut_ptr(), 1)
    };
    check(ctx, rc);
    drop(guard);
    LweSample { coefficients: out_a, b: out_b[0], current_variance: 0.0 }
}

/// Replaces `lwe_key_switch` (src/lwe.rs:601-613) on an extracted sample of N*k coefficients.
pub fn key_switch(keys: &GpuKeys, x: LweSample) -> LweSample {
    assert_eq!(x.coefficients.len(), keys.big_n);
    let (mut out_a, mut out_b, mut var) = (vec![0i32; keys.n], vec![0i32; 1], vec![0f64; 1]);
    let guard = keys.ctx.lock().unwrap();
    let ctx = *guard;
    let rc = unsafe {
        tfhe_hip_keyswitch(ctx, x.coefficients.as_ptr(), &x.b, out_a.as_mut_ptr(), out_b.as_mut_ptr(),
                           var.as_mut_ptr(), 1)
    };
    check(ctx, rc);
    drop(guard);
    LweSample { coefficients: out_a, b: out_b[0], current_variance: var[0] }
}

/// One gate on many operand pairs: prologue (the linear combination of src/gates.rs), blind
/// rotation, extraction and key switch in one call.  `cc` is mux's third operand.
pub fn gate_many(
    keys: &GpuKeys,
    gate: c_int,
    ca: &[&LweSample],
    cb: &[&LweSample],
    cc: Option<&[&LweSample]>,
) -> Vec<LweSample> {
    let (n, batch) = (keys.n, ca.len());
    assert_eq!(cb.len(), batch);
    let (ca_a, ca_b) = flatten(ca, n);
    let (cb_a, cb_b) = flatten(cb, n);
    let third = cc.map(|c| {
        assert_eq!(c.len(), batch);
        flatten(c, n)
    });
    let (cc_a, cc_b) = match &third {
        Some((a, b)) => (a.as_ptr(), b.as_ptr()),
        None => (ptr::null(), ptr::null()),
    };
    let (mut out_a, mut out_b, mut var) = (vec![0i32; batch * n], vec![0i32; batch], vec![0f64; batch]);
    let guard = keys.ctx.lock().unwrap();
    let ctx = *guard;
    let rc = unsafe {
        tfhe_hip_gate(ctx, gate, ca_a.as_ptr(), ca_b.as_ptr(), cb_a.as_ptr(), cb_b.as_ptr(), cc_a, cc_b,
                      out_a.as_mut_ptr(), out_b.as_mut_ptr(), var.as_mut_ptr(), batch)
    };
    check(ctx, rc);
    drop(guard);
    unflatten(out_a, out_b, var, n)
}

pub fn gate(keys: &GpuKeys, gate: c_int, ca: &LweSample, cb: &LweSample) -> LweSample {
    gate_many(keys, gate, &[ca], &[cb], None).pop().unwrap()
}

pub fn mux(keys: &GpuKeys, a: &LweSample, b: &LweSample, c: &LweSample) -> LweSample {
    gate_many(keys, MUX, &[a], &[b], Some(&[c])).pop().unwrap()
}

/// `circuits::add` (src/circuits/circuits.rs:105-120) on many pairs of equally wide operands:
/// every dependency level of all pairs is one launch and no intermediate ciphertext returns to
/// the host.  Returns, per pair, the sum bits (least significant first) and the carry.
pub fn add_many(
    keys: &GpuKeys,
    a: &[&[LweSample]],
    b: &[&[LweSample]],
) -> Vec<(Vec<LweSample>, LweSample)> {
    let (n, count) = (keys.n, a.len());
    assert_eq!(b.len(), count);
    let bits = if count == 0 { 0 } else { a[0].len() };
    // bit-sliced operands: [bits][count][n] and [bits][count]
    let slice = |x: &[&[LweSample]]| {
        let (mut xa, mut xb) = (Vec::with_capacity(bits * count * n), Vec::with_capacity(bits * count));
        for bit in 0..bits {
            for operand in x {
                assert_eq!(operand.len(), bits, "Cannot add two numbers with different number of bits!");
                xa.extend_from_slice(&operand[bit].coefficients);
                xb.push(operand[bit].b);
            }
        }
        (xa, xb)
    };
    let ((a_a, a_b), (b_a, b_b)) = (slice(a), slice(b));
    let (mut s_a, mut s_b, mut s_v) =
        (vec![0i32; bits * count * n], vec![0i32; bits * count], vec![0f64; bits * count]);
    let (mut c_a, mut c_b, mut c_v) = (vec![0i32; count * n], vec![0i32; count], vec![0f64; count]);
    let guard = keys.ctx.lock().unwrap();
    let ctx = *guard;
    let rc = unsafe {
        tfhe_hip_circuit_add(ctx, a_a.as_ptr(), a_b.as_ptr(), b_a.as_ptr(), b_b.as_ptr(), s_a.as_mut_ptr(),
                             s_b.as_mut_ptr(), c_a.as_mut_ptr(), c_b.as_mut_ptr(), s_v.as_mut_ptr(),
                             c_v.as_mut_ptr(), count, bits as c_int)
    };
    check(ctx, rc);
    drop(guard);
    (0..count)
        .map(|i| {
            let sum = (0..bits)
                .map(|bit| {
                    let at = bit * count + i;
                    LweSample {
                        coefficients: s_a[at * n..(at + 1) * n].to_vec(),
                        b: s_b[at],
                        current_variance: s_v[at],
                    }
                })
                .collect();
            let carry = LweSample {
                coefficients: c_a[i * n..(i + 1) * n].to_vec(),
                b: c_b[i],
                current_variance: c_v[i],
            };
            (sum, carry)
        })
        .collect()
}

// With `--features bootstrapping` the gate bodies of src/gates.rs become one line each:
//
//   pub fn nand(ca, cb, bk)  -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::NAND,  ca, cb)     // src/gates.rs:23-41
//   pub fn or(ca, cb, bk)    -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::OR,    ca, cb)     // :44-61
//   pub fn and(ca, cb, bk)   -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::AND,   ca, cb)     // :64-80
//   pub fn xor(ca, cb, bk)   -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::XOR,   ca, cb)     // :83-99
//   pub fn xnor(ca, cb, bk)  -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::XNOR,  ca, cb)     // :102-119
//   pub fn nor(ca, cb, bk)   -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::NOR,   ca, cb)     // :127-143
//   pub fn andny(ca, cb, bk) -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::ANDNY, ca, cb)     // :147-162
//   pub fn andyn(ca, cb, bk) -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::ANDYN, ca, cb)     // :166-182
//   pub fn orny(ca, cb, bk)  -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::ORNY,  ca, cb)     // :186-202
//   pub fn oryn(ca, cb, bk)  -> gpu::gate(&bk.bk.gpu.get(&bk.bk), ffi::ORYN,  ca, cb)     // :206-222
//   pub fn mux(a, b, c, bk)  -> gpu::mux(&bk.bk.gpu.get(&bk.bk), a, b, c)                // :239-268
//   pub fn tfhe_bootstrap(bk: &LweBootstrappingKey, mu, x)
//                            -> gpu::bootstrap(&bk.gpu.get(bk), mu, x)     // src/bootstrapping.rs:25-29
//   circuits::add(a, b, key) -> gpu::add_many(&key.bk.gpu.get(&key.bk), &[a], &[b]).pop()  // src/circuits/circuits.rs:105-120
//
// `not`, `constant`, `encrypt`, `decrypt`, `generate_keys` stay as they are (host).
// Chained gates that should not cross PCIe use the sample handles
// (tfhe_hip_samples_alloc / _upload / _gate / _not / _constant / _download in ffi.rs).
