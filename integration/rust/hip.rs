//! `impl Device<HipSlice> for Hip` -- the MI355X backend behind the engine's operator trait
//! (device/device.rs).  Each method is one call into librama_hip.so; a non-zero return panics,
//! which is what the reference's CUDA backend does through `.unwrap()` on every driver call.
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};
use std::ptr;

use rand::{Rng, SeedableRng};
use rand_chacha::ChaCha20Rng;

use super::device::Device;
use super::hip_sys::*;
use crate::transformer::state::{RunState, RunStateView};
use crate::transformer::{Config, MutView, Storage, View};

/// A device allocation of `len` f32 (the counterpart of `CudaSlice<f32>`).
pub struct HipSlice {
    pub(crate) ptr: *mut f32,
    len: usize,
    ctx: *mut rama_ctx,
}
impl Storage for HipSlice {
    fn length(&self) -> usize { self.len }
}
impl Drop for HipSlice {
    fn drop(&mut self) { unsafe { rama_free(self.ctx, self.ptr.cast()); } }
}
// the engine moves weights/state across tokio tasks; the pointers are plain device addresses
unsafe impl Send for HipSlice {}
unsafe impl Sync for HipSlice {}

pub struct Hip {
    pub(crate) ctx: *mut rama_ctx,
}
unsafe impl Send for Hip {}
unsafe impl Sync for Hip {}     // one context = one HIP stream: callers serialise, as with cudarc's default stream

#[track_caller]
fn ck(rc: i32) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(rama_last_error()) }.to_string_lossy().into_owned();
        panic!("rama_hip error {rc}: {msg}");
    }
}

/// Which order does `wide::f32x4::reduce_add` of THIS build sum its lanes in?  [1, 2^-24, -1, 1.5 * 2^-24] gives three different fp32 results:
/// pairwise (l0+l1)+(l2+l3) = 2^-23, strided (l0+l2)+(l1+l3) = 1.25 * 2^-23, sequential ((l0+l1)+l2)+l3 = 1.5 * 2^-24
/// (tests/test_oracle_golden.py::test_lane_reduce_probe_vector pins the three values against the oracle's switch).
fn probe_lane_reduce() -> c_int {
    use wide::f32x4;
    let e = f32::from_bits(0x3380_0000);                       // 2^-24
    let v = f32x4::from(std::hint::black_box([1.0f32, e, -1.0f32, 1.5 * e])).reduce_add();
    match v.to_bits() {
        0x3400_0000 => 0,                                      // 2^-23: pairwise
        0x3420_0000 => 1,                                      // 1.25 * 2^-23: strided
        0x33C0_0000 => 2,                                      // 1.5 * 2^-24: sequential
        other => panic!("wide::f32x4::reduce_add sums its lanes in an order this backend does not know (probe gave {other:#x})"),
    }
}

impl Hip {
    pub fn new() -> Self {
        let mut ctx = ptr::null_mut();
        ck(unsafe { rama_ctx_create(0, ptr::null_mut(), &mut ctx) });
        // parity mode unless RAMA_REF_ORDER says otherwise (0 = fast, 2 = tolerance experiment): the reference CPU path's rounding order in
        // every op, logits bit-identical to cpu.rs (mirrors rama_amd/csrc/host/engine.hpp Hip::Hip)
        let mode: c_int = match std::env::var("RAMA_REF_ORDER") {
            Err(_) => 1,
            Ok(v) => match v.parse::<c_int>() { Ok(k) if (0..=3).contains(&k) => k, _ => panic!("RAMA_REF_ORDER must be 0 (fast), 1 (parity), 2 (tolerance experiment) or 3 (bar), got '{v}'") },
        };
        ck(unsafe { rama_set_tuning(ctx, b"ref_order\0".as_ptr() as *const c_char, mode) });
        // cpu.rs:148 `v.reduce_add()`: the order of wide::f32x4's final 4-lane sum depends on the crate version and on the target features THIS binary was
        // built with.  Ask the crate itself, and tell the library ("lane_reduce"), so that parity mode reproduces the CPU path of the same build.
        let lanes: c_int = match std::env::var("RAMA_LANE_REDUCE") {
            Ok(v) => v.parse().expect("RAMA_LANE_REDUCE must be 0 (pairwise), 1 (strided) or 2 (sequential)"),
            Err(_) => probe_lane_reduce(),
        };
        ck(unsafe { rama_set_tuning(ctx, b"lane_reduce\0".as_ptr() as *const c_char, lanes) });
        Hip { ctx }
    }
    /// htod_sync_copy (hbm.rs:14-16)
    pub fn allocate(&self, data: &[f32]) -> HipSlice {
        let mut p = ptr::null_mut();
        ck(unsafe { rama_upload_f32(self.ctx, data.as_ptr(), data.len(), &mut p) });
        HipSlice { ptr: p, len: data.len(), ctx: self.ctx }
    }
    /// dtoh_sync_copy_into
    pub fn download_into(&self, src: &HipSlice, dst: &mut [f32]) {
        let n = dst.len().min(src.len);
        ck(unsafe { rama_download_f32(self.ctx, src.ptr, n, dst.as_mut_ptr()) });
    }
}
impl Drop for Hip {
    fn drop(&mut self) { unsafe { rama_ctx_destroy(self.ctx); } }
}

// A view's device pointer = storage base + range.start: slice() ranges are ABSOLUTE offsets into
// the backing storage (transformer/mod.rs:44-51), exactly what cudaview() passes to the kernels.
fn dp(v: &View<'_, HipSlice>) -> *const f32 { unsafe { v.data.ptr.add(v.range.start) } }
fn dpm(v: &MutView<'_, HipSlice>) -> *mut f32 { unsafe { v.data.ptr.add(v.range.start) } }

impl Device<HipSlice> for Hip {
    fn array_add(&self, target: &mut MutView<'_, HipSlice>, source: &View<'_, HipSlice>, n: usize) {
        ck(unsafe { rama_array_add(self.ctx, dpm(target), dp(source), n) })
    }
    fn array_mult(&self, target: &mut MutView<'_, HipSlice>, source: &View<'_, HipSlice>, n: usize) {
        ck(unsafe { rama_array_mult(self.ctx, dpm(target), dp(source), n) })
    }
    fn sinu(&self, o: &mut MutView<'_, HipSlice>, n: usize) {
        ck(unsafe { rama_sinu(self.ctx, dpm(o), n) })
    }
    fn multi_head_attention(&self, rsv: &mut RunStateView<'_, HipSlice>, cfg: &Config, layer: usize, pos: usize) {
        let hs = cfg.dim / cfg.n_heads;
        ck(unsafe {
            rama_multi_head_attention(self.ctx, dpm(&rsv.xb), dpm(&rsv.att), dpm(&rsv.q), dpm(&rsv.key_cache),
                                      dpm(&rsv.value_cache), layer as i32, cfg.dim as i32, pos as i32, hs as i32,
                                      cfg.seq_len as i32, cfg.n_heads as i32)
        })
    }
    fn copy_from_slice(&self, target: &mut MutView<'_, HipSlice>, source: &View<'_, HipSlice>, n: usize) {
        ck(unsafe { rama_copy_from_slice(self.ctx, dpm(target), dp(source), n) })
    }
    fn rmsnorm(&self, o: &mut MutView<'_, HipSlice>, x: &View<'_, HipSlice>, weight: &View<'_, HipSlice>, n: usize) {
        ck(unsafe { rama_rmsnorm(self.ctx, dpm(o), dp(x), dp(weight), n) })
    }
    fn apply_position(&self, q: &mut MutView<'_, HipSlice>, k: &mut MutView<'_, HipSlice>,
                      pos_real: &View<'_, HipSlice>, pos_img: &View<'_, HipSlice>, head_size: usize) {
        ck(unsafe { rama_apply_position(self.ctx, dpm(q), dpm(k), dp(pos_real), dp(pos_img), head_size) })
    }
    fn matmul(&self, o: &mut MutView<'_, HipSlice>, a: &View<'_, HipSlice>, b: &View<'_, HipSlice>,
              width: usize, o_rows: usize, o_cols: usize) {
        ck(unsafe { rama_matmul(self.ctx, dpm(o), dp(a), dp(b), width, o_rows, o_cols) })
    }
    fn softmax<'a>(&self, x: &mut MutView<'a, HipSlice>, n: usize) {
        ck(unsafe { rama_softmax(self.ctx, dpm(x), n) })
    }
    fn sample<'a>(&self, cfg: &Config, rsv: &mut RunStateView<'a, HipSlice>, temperature: f32, topp: f32) -> usize {
        let mut next = 0i32;
        if temperature == 0.0 {
            ck(unsafe { rama_sample_argmax(self.ctx, dpm(&rsv.logits), cfg.vocab_size, &mut next) });
        } else {
            // the CPU backend re-seeds on every call (device/cpu.rs:161-162): one constant draw
            let u: f32 = ChaCha20Rng::seed_from_u64(100).gen::<f32>();
            ck(unsafe { rama_sample_topp(self.ctx, dpm(&rsv.logits), cfg.vocab_size, temperature, topp, u, &mut next) });
        }
        next as usize
    }
    fn to_cpu(&self, state: &RunStateView<HipSlice>, cpu_state: &mut RunState<Vec<f32>>) {
        macro_rules! pull { ($($f:ident),*) => { $( self.download_into(&*state.$f.data, &mut cpu_state.$f); )* } }
        pull!(x, xb, xb2, hb, hb2, q, k, v, att, logits, key_cache, value_cache);
    }
}
