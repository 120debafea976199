//! Safe Rust over the C ABI.  Not compiled in the build image (no Rust toolchain there);
//! kept mechanical on purpose - every method is one FFI call plus status handling.
//!
//! Field elements cross as their Montgomery limb: `Fp64<MontBackend<T,1>>` is
//! `Fp(BigInt([u64; 1]), PhantomData)`, whose limb is `x * 2^64 mod p` - exactly what the
//! kernels compute on.  `word` / `from_word` read and rebuild that limb (no reduction).
use std::ffi::CStr;
use std::marker::PhantomData;
use std::ptr;
use std::rc::Rc;

use ark_ff::{BigInt, Fp64, MontBackend, MontConfig};
use ark_poly::univariate::SparsePolynomial;
use sum_check_protocol::SumCheckPolynomial;
use sumcheck_hip_sys as sys;

pub type F64<T> = Fp64<MontBackend<T, 1>>;

#[inline]
pub fn word<T: MontConfig<1>>(x: &F64<T>) -> u64 {
    (x.0).0[0]
}
#[inline]
pub fn from_word<T: MontConfig<1>>(w: u64) -> F64<T> {
    F64::<T>::new_unchecked(BigInt([w]))
}
fn words<T: MontConfig<1>>(xs: &[F64<T>]) -> Vec<u64> {
    xs.iter().map(word::<T>).collect()
}

/// `#[derive(MontConfig)]` constants -> `sc_field`.
pub fn field_of<T: MontConfig<1>>() -> sys::sc_field {
    sys::sc_field {
        p: T::MODULUS.0[0],
        p_inv_neg: T::INV,
        r_mod_p: T::R.0[0],
        r2_mod_p: T::R2.0[0],
    }
}

struct CtxInner(*mut sys::sc_ctx);
impl Drop for CtxInner {
    fn drop(&mut self) {
        unsafe { sys::sc_ctx_destroy(self.0) };
    }
}

/// One GPU + one stream + one field.  `Rc`, not `Arc`: a context is single-threaded, like
/// the reference's `&mut Prover`.
pub struct Context<T: MontConfig<1>> {
    inner: Rc<CtxInner>,
    _t: PhantomData<T>,
}
impl<T: MontConfig<1>> Clone for Context<T> {
    fn clone(&self) -> Self {
        Self { inner: self.inner.clone(), _t: PhantomData }
    }
}

impl<T: MontConfig<1>> Context<T> {
    pub fn new(device: i32) -> Self {
        let f = field_of::<T>();
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_ctx_create(&f, device, &mut h) };
        if rc != sys::SC_OK {
            let msg = unsafe { CStr::from_ptr(sys::sc_last_error(ptr::null())) };
            panic!("sc_ctx_create failed ({rc}): {}", msg.to_string_lossy());
        }
        Self { inner: Rc::new(CtxInner(h)), _t: PhantomData }
    }
    fn raw(&self) -> *mut sys::sc_ctx {
        self.inner.0
    }
    /// The reference's prover methods are infallible by signature and panic on misuse
    /// (`unwrap`/`assert_eq!`); a non-zero status therefore panics here too.
    fn check(&self, rc: i32, what: &str) {
        if rc != sys::SC_OK {
            let msg = unsafe { CStr::from_ptr(sys::sc_last_error(self.raw())) };
            panic!("{what} failed ({rc}): {}", msg.to_string_lossy());
        }
    }
}

/// Device-resident `DenseMultilinearExtension<F>` (LE variable order).
pub struct DeviceMle<T: MontConfig<1>> {
    ctx: Context<T>,
    h: *mut sys::sc_table,
}
impl<T: MontConfig<1>> Drop for DeviceMle<T> {
    fn drop(&mut self) {
        unsafe { sys::sc_table_free(self.ctx.raw(), self.h) };
    }
}
impl<T: MontConfig<1>> DeviceMle<T> {
    /// `DenseMultilinearExtension::from_evaluations_vec`
    pub fn from_evaluations_vec(ctx: &Context<T>, num_vars: usize, evals: Vec<F64<T>>) -> Self {
        assert_eq!(evals.len(), 1usize << num_vars, "The size of evaluations should be 2^num_vars.");
        let w = words::<T>(&evals);
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_table_upload(ctx.raw(), w.as_ptr(), w.len(), &mut h) };
        ctx.check(rc, "sc_table_upload");
        Self { ctx: ctx.clone(), h }
    }
    pub fn num_vars(&self) -> usize {
        unsafe { sys::sc_table_len(self.h) }.trailing_zeros() as usize
    }
    pub fn fix_variables(&self, partial_point: &[F64<T>]) -> Self {
        let r = words::<T>(partial_point);
        let mut h = ptr::null_mut();
        let rc = unsafe {
            sys::sc_table_fix_variables(self.ctx.raw(), self.h, r.as_ptr(), r.len(), sys::SC_ORDER_LE, &mut h)
        };
        self.ctx.check(rc, "sc_table_fix_variables");
        Self { ctx: self.ctx.clone(), h }
    }
    pub fn evaluate(&self, point: &[F64<T>]) -> F64<T> {
        let r = words::<T>(point);
        let mut out = 0u64;
        let rc = unsafe {
            sys::sc_table_evaluate(self.ctx.raw(), self.h, r.as_ptr(), r.len(), sys::SC_ORDER_LE, &mut out)
        };
        self.ctx.check(rc, "sc_table_evaluate");
        from_word::<T>(out)
    }
    pub fn to_evaluations(&self) -> Vec<F64<T>> {
        let len = unsafe { sys::sc_table_len(self.h) };
        let mut w = vec![0u64; len];
        let rc = unsafe { sys::sc_table_download(self.ctx.raw(), self.h, w.as_mut_ptr(), len) };
        self.ctx.check(rc, "sc_table_download");
        w.into_iter().map(from_word::<T>).collect()
    }
}

/// GPU-backed `matrix_multiplication::G`: g(z) = f_A(r1, z) * f_B(z, r2).
/// Tables are immutable once built, so `Clone` shares them (`Prover::new(g.clone())`
/// costs nothing and the verifier's oracle keeps the originals).
pub struct GpuG<T: MontConfig<1>> {
    f_a: Rc<DeviceMle<T>>,
    f_b: Rc<DeviceMle<T>>,
}
impl<T: MontConfig<1>> Clone for GpuG<T> {
    fn clone(&self) -> Self {
        Self { f_a: self.f_a.clone(), f_b: self.f_b.clone() }
    }
}

fn round_poly<T: MontConfig<1>>(e: [u64; 3]) -> SparsePolynomial<F64<T>> {
    let f = field_of::<T>();
    let mut c = [0u64; 3];
    let rc = unsafe { sys::sc_interpolate_quadratic(&f, e.as_ptr(), c.as_mut_ptr()) };
    assert_eq!(rc, sys::SC_OK);
    SparsePolynomial::from_coefficients_vec(
        c.iter().enumerate().map(|(d, w)| (d, from_word::<T>(*w))).collect(),
    )
}

impl<T: MontConfig<1>> GpuG<T> {
    /// `G::new`: a, b are the 2^n x 2^n matrices, row-major.
    pub fn new<M: IntoIterator<Item = F64<T>>>(ctx: &Context<T>, n: usize, a: M, b: M, point: &[F64<T>]) -> Self {
        let a = DeviceMle::from_evaluations_vec(ctx, 2 * n, a.into_iter().collect());
        let b = DeviceMle::from_evaluations_vec(ctx, 2 * n, b.into_iter().collect());
        let pt = words::<T>(point);
        assert_eq!(pt.len(), 2 * n);
        let (mut ha, mut hb) = (ptr::null_mut(), ptr::null_mut());
        let rc = unsafe { sys::sc_matmul_g_new(ctx.raw(), a.h, b.h, n, pt.as_ptr(), &mut ha, &mut hb) };
        ctx.check(rc, "sc_matmul_g_new");
        Self {
            f_a: Rc::new(DeviceMle { ctx: ctx.clone(), h: ha }),
            f_b: Rc::new(DeviceMle { ctx: ctx.clone(), h: hb }),
        }
    }
}

impl<T: MontConfig<1>> SumCheckPolynomial<F64<T>> for GpuG<T> {
    fn evaluate(&self, point: &[F64<T>]) -> Option<F64<T>> {
        if point.len() != self.num_vars() {
            return None;
        }
        let ctx = &self.f_a.ctx;
        let pt = words::<T>(point);
        let mut out = 0u64;
        let rc = unsafe { sys::sc_prod2_evaluate(ctx.raw(), self.f_a.h, self.f_b.h, pt.as_ptr(), pt.len(), &mut out) };
        ctx.check(rc, "sc_prod2_evaluate");
        Some(from_word::<T>(out))
    }

    fn fix_variables(&self, partial_point: &[F64<T>]) -> Self {
        Self {
            f_a: Rc::new(self.f_a.fix_variables(partial_point)),
            f_b: Rc::new(self.f_b.fix_variables(partial_point)),
        }
    }

    fn to_univariate(&self) -> SparsePolynomial<F64<T>> {
        let ctx = &self.f_a.ctx;
        let mut e = [0u64; 3];
        let rc = unsafe { sys::sc_prod2_round_sums(ctx.raw(), self.f_a.h, self.f_b.h, e.as_mut_ptr()) };
        ctx.check(rc, "sc_prod2_round_sums");
        round_poly::<T>(e)
    }

    fn num_vars(&self) -> usize {
        self.f_a.num_vars()
    }

    fn to_evaluations(&self) -> Vec<F64<T>> {
        let ctx = &self.f_a.ctx;
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_prod2_to_evaluations(ctx.raw(), self.f_a.h, self.f_b.h, &mut h) };
        ctx.check(rc, "sc_prod2_to_evaluations");
        DeviceMle { ctx: ctx.clone(), h }.to_evaluations()
    }

    // ---- the two provided methods INTEGRATION.md adds to the trait -------------------
    fn hypercube_sum(&self) -> F64<T> {
        let ctx = &self.f_a.ctx;
        let mut out = 0u64;
        let rc = unsafe { sys::sc_prod2_sum(ctx.raw(), self.f_a.h, self.f_b.h, &mut out) };
        ctx.check(rc, "sc_prod2_sum");
        from_word::<T>(out)
    }

    fn native_engine(&self) -> Option<Box<dyn sum_check_protocol::RoundEngine<F64<T>>>> {
        Some(Box::new(ProverEngine::new(self)))
    }
}

/// `sc_prover`: the fused fold + round-sum state machine behind `Prover::round`.
pub struct ProverEngine<T: MontConfig<1>> {
    g: GpuG<T>, // keeps the borrowed tables alive
    h: *mut sys::sc_prover,
}
impl<T: MontConfig<1>> ProverEngine<T> {
    pub fn new(g: &GpuG<T>) -> Self {
        let ctx = &g.f_a.ctx;
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_prover_create(ctx.raw(), g.f_a.h, g.f_b.h, &mut h) };
        ctx.check(rc, "sc_prover_create");
        Self { g: g.clone(), h }
    }
}
impl<T: MontConfig<1>> Drop for ProverEngine<T> {
    fn drop(&mut self) {
        unsafe { sys::sc_prover_destroy(self.h) };
    }
}
impl<T: MontConfig<1>> sum_check_protocol::RoundEngine<F64<T>> for ProverEngine<T> {
    fn c_1(&self) -> F64<T> {
        let mut out = 0u64;
        let rc = unsafe { sys::sc_prover_c1(self.h, &mut out) };
        self.g.f_a.ctx.check(rc, "sc_prover_c1");
        from_word::<T>(out)
    }
    fn round(&mut self, r_prev: F64<T>, j: usize) -> SparsePolynomial<F64<T>> {
        let mut e = [0u64; 3];
        let rc = unsafe { sys::sc_prover_round(self.h, word::<T>(&r_prev), j, e.as_mut_ptr()) };
        self.g.f_a.ctx.check(rc, "sc_prover_round");
        round_poly::<T>(e)
    }
}
