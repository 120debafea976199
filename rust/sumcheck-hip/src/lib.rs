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

use ark_ff::{BigInt, Fp64, MontBackend, MontConfig, Zero};
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

/// The library must speak the ABI version these bindings were written against (`SC_ABI_VERSION`: structs that cross the
/// boundary grow between versions).
fn check_abi() {
    let v = unsafe { sys::sc_abi_version() };
    assert_eq!(v, sys::SC_ABI_VERSION, "libsumcheck_hip speaks ABI version {v}, these bindings {}", sys::SC_ABI_VERSION);
}

impl<T: MontConfig<1>> Context<T> {
    pub fn new(device: i32) -> Self {
        check_abi();
        let f = field_of::<T>();
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_ctx_create(&f, device, &mut h) };
        if rc != sys::SC_OK {
            let msg = unsafe { CStr::from_ptr(sys::sc_last_error(ptr::null())) };
            panic!("sc_ctx_create failed ({rc}): {}", msg.to_string_lossy());
        }
        Self { inner: Rc::new(CtxInner(h)), _t: PhantomData }
    }
    /// ONE handle over several GPUs of this process (`sc_ctx_create_multi`): `devices.len()` a power of two up to 8.
    /// Everything built on it - `DeviceMle`, `GpuG::new`, `Prover::new(g)` of the unchanged `sum-check-protocol`
    /// callers - works on tables that are split over the devices by their top index bits; the round sums of the
    /// shards are added by the calling thread, so the verifier's `r_j` is drawn once and needs no broadcast
    /// (SURVEY.md section 8e).  Bit-identical to `Context::new`.
    pub fn new_multi(devices: &[i32]) -> Self {
        check_abi();
        let f = field_of::<T>();
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_ctx_create_multi(&f, devices.as_ptr(), devices.len() as i32, &mut h) };
        if rc != sys::SC_OK {
            let msg = unsafe { CStr::from_ptr(sys::sc_last_error(ptr::null())) };
            panic!("sc_ctx_create_multi failed ({rc}): {}", msg.to_string_lossy());
        }
        Self { inner: Rc::new(CtxInner(h)), _t: PhantomData }
    }
    /// Pay the first-use costs of proofs over `num_vars` variables now (option `prewarm`: the library's code object on the
    /// device, the resident-grid queries of the plan's kernels, the pool blocks of every folded table) - for a caller that
    /// proves once, like `matrix-multiplication/benches/mm_benchmark.rs:88-96`.
    pub fn prewarm(&self, num_vars: usize) {
        let rc = unsafe { sys::sc_ctx_set_option(self.raw(), b"prewarm\0".as_ptr() as *const _, num_vars as i64) };
        self.check(rc, "sc_ctx_set_option(prewarm)");
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

fn round_coeffs<T: MontConfig<1>>(e: [u64; 3]) -> [F64<T>; 3] {
    let f = field_of::<T>();
    let mut c = [0u64; 3];
    let rc = unsafe { sys::sc_interpolate_quadratic(&f, e.as_ptr(), c.as_mut_ptr()) };
    assert_eq!(rc, sys::SC_OK);
    [from_word::<T>(c[0]), from_word::<T>(c[1]), from_word::<T>(c[2])]
}

/// The round polynomial as `triangle_counting::G` and `W` hand it out: `Evaluations::interpolate()` then `p.into()`
/// (`From<DensePolynomial>`), i.e. the non-zero coefficients only.
fn round_poly<T: MontConfig<1>>(e: [u64; 3]) -> SparsePolynomial<F64<T>> {
    let c = round_coeffs::<T>(e);
    SparsePolynomial::from_coefficients_vec(
        c.iter().enumerate().filter(|(_, w)| !w.is_zero()).map(|(d, w)| (d, *w)).collect(),
    )
}

/// The round polynomial as `matrix_multiplication::G` hands it out: the reference adds three Lagrange terms, each built
/// by `from_coefficients_vec`, and `SparsePolynomial`'s `+` returns the other operand untouched when one operand is zero.
/// The terms through x = 1 and x = 2 carry an explicit zero constant, so when H(0) = 0 and exactly one of H(1), H(2) is
/// non-zero the sum keeps a `(0, 0)` term; everywhere else only non-zero terms remain.  Same values either way - this
/// keeps `serialize_uncompressed` (fiat-shamir) byte-identical with the CPU `G`.
fn round_poly_lagrange<T: MontConfig<1>>(e: [u64; 3]) -> SparsePolynomial<F64<T>> {
    let c = round_coeffs::<T>(e);
    let keeps_zero_constant = e[0] == 0 && ((e[1] == 0) != (e[2] == 0));
    let mut terms: Vec<(usize, F64<T>)> = Vec::with_capacity(3);
    for (d, w) in c.iter().enumerate() {
        if !w.is_zero() || (d == 0 && keeps_zero_constant) {
            terms.push((d, *w));
        }
    }
    SparsePolynomial::from_coefficients_vec(terms)
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
        round_poly_lagrange::<T>(e)
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
        round_poly_lagrange::<T>(e)
    }
}

// =====================================================================================
// gkr_protocol::round_polynomial::W and triangle_counting::G on the device
// =====================================================================================

impl<T: MontConfig<1>> DeviceMle<T> {
    fn from_raw(ctx: &Context<T>, h: *mut sys::sc_table) -> Self {
        Self { ctx: ctx.clone(), h }
    }
    /// The table at every point of `points` in ONE pass over it (`sc_table_evaluate_many`): a verifier's several oracle
    /// queries, or the k + 1 points of `restrict_to_line`, for one read of the table.
    pub fn evaluate_many(&self, points: &[Vec<F64<T>>]) -> Vec<F64<T>> {
        let n = self.num_vars();
        let flat: Vec<u64> = points.iter().flat_map(|p| { assert_eq!(p.len(), n); words::<T>(p) }).collect();
        let mut out = vec![0u64; points.len()];
        let rc = unsafe {
            sys::sc_table_evaluate_many(self.ctx.raw(), self.h, flat.as_ptr(), points.len(), n, sys::SC_ORDER_LE, out.as_mut_ptr())
        };
        self.ctx.check(rc, "sc_table_evaluate_many");
        out.into_iter().map(from_word::<T>).collect()
    }
    /// A table over device memory the caller owns (`sc_table_from_device`): borrowed, never written or freed by the library.
    /// # Safety
    /// `device_ptr` must point to `1 << num_vars` Montgomery words in device memory that outlive the returned table and
    /// everything made from it, and must not change while they are in use.
    pub unsafe fn from_device(ctx: &Context<T>, device_ptr: *const u64, num_vars: usize) -> Self {
        let mut h = ptr::null_mut();
        let rc = sys::sc_table_from_device(ctx.raw(), device_ptr, 1usize << num_vars, &mut h);
        ctx.check(rc, "sc_table_from_device");
        Self::from_raw(ctx, h)
    }
    /// `restrict_poly(b, c, &mle)` (gkr-protocol/src/lib.rs:291-321): the MLE on the line l(0) = b, l(1) = c.
    pub fn restrict_to_line(&self, b: &[F64<T>], c: &[F64<T>]) -> SparsePolynomial<F64<T>> {
        let k = self.num_vars();
        assert_eq!(b.len(), k);
        assert_eq!(c.len(), k);
        let (bw, cw) = (words::<T>(b), words::<T>(c));
        let mut out = vec![0u64; k + 1];
        let rc = unsafe {
            sys::sc_table_restrict_to_line(self.ctx.raw(), self.h, bw.as_ptr(), cw.as_ptr(), k, out.as_mut_ptr())
        };
        self.ctx.check(rc, "sc_table_restrict_to_line");
        SparsePolynomial::from_coefficients_vec(out.iter().enumerate().map(|(d, w)| (d, from_word::<T>(*w))).collect())
    }
}

/// `line(b, c)` (gkr-protocol/src/lib.rs:278-289) - host arithmetic only.
pub fn line<T: MontConfig<1>>(b: &[F64<T>], c: &[F64<T>]) -> Vec<SparsePolynomial<F64<T>>> {
    std::iter::zip(b, c)
        .map(|(b, c)| SparsePolynomial::from_coefficients_slice(&[(0, *b), (1, *c - b)]))
        .collect()
}

/// One gate of a layered circuit as the wiring predicate sees it (gkr-protocol/src/circuit.rs:17-31).
#[derive(Clone, Copy)]
pub struct WiringGate {
    pub is_mul: bool,
    pub inputs: [u32; 2],
}

/// `add_i(r_i, ., .)` and `mul_i(r_i, ., .)` of `Prover::start_round` / `Circuit::{add,mul}_i_ext`
/// (gkr-protocol/src/lib.rs:388-416, circuit.rs:156-213) straight from layer i's gate list.
pub fn gkr_wiring<T: MontConfig<1>>(
    ctx: &Context<T>,
    gates: &[WiringGate],
    k_next: usize,
    r_i: &[F64<T>],
) -> (DeviceMle<T>, DeviceMle<T>) {
    let k_i = gates.len().trailing_zeros() as usize;
    assert_eq!(gates.len(), 1 << k_i);
    assert_eq!(r_i.len(), k_i);
    let ty: Vec<i32> = gates.iter().map(|g| g.is_mul as i32).collect();
    let i0: Vec<u32> = gates.iter().map(|g| g.inputs[0]).collect();
    let i1: Vec<u32> = gates.iter().map(|g| g.inputs[1]).collect();
    let r = words::<T>(r_i);
    let (mut ha, mut hm) = (ptr::null_mut(), ptr::null_mut());
    let rc = unsafe {
        sys::sc_gkr_wiring(ctx.raw(), ty.as_ptr(), i0.as_ptr(), i1.as_ptr(), k_i, k_next, r.as_ptr(), &mut ha, &mut hm)
    };
    ctx.check(rc, "sc_gkr_wiring");
    (DeviceMle::from_raw(ctx, ha), DeviceMle::from_raw(ctx, hm))
}

/// GPU-backed `gkr_protocol::round_polynomial::W` (round_polynomial.rs:23-44); same field names.
pub struct GpuW<T: MontConfig<1>> {
    add_i: Rc<DeviceMle<T>>,
    mul_i: Rc<DeviceMle<T>>,
    w_b: Rc<DeviceMle<T>>,
    w_c: Rc<DeviceMle<T>>,
}
impl<T: MontConfig<1>> Clone for GpuW<T> {
    fn clone(&self) -> Self {
        Self { add_i: self.add_i.clone(), mul_i: self.mul_i.clone(), w_b: self.w_b.clone(), w_c: self.w_c.clone() }
    }
}
impl<T: MontConfig<1>> GpuW<T> {
    /// `W::new` (round_polynomial.rs:31-43)
    pub fn new(add_i: DeviceMle<T>, mul_i: DeviceMle<T>, w_b: Rc<DeviceMle<T>>, w_c: Rc<DeviceMle<T>>) -> Self {
        Self { add_i: Rc::new(add_i), mul_i: Rc::new(mul_i), w_b, w_c }
    }
    fn ctx(&self) -> &Context<T> {
        &self.add_i.ctx
    }
    fn handles(&self) -> (*mut sys::sc_table, *mut sys::sc_table, *mut sys::sc_table, *mut sys::sc_table) {
        (self.add_i.h, self.mul_i.h, self.w_b.h, self.w_c.h)
    }
}
impl<T: MontConfig<1>> SumCheckPolynomial<F64<T>> for GpuW<T> {
    fn evaluate(&self, point: &[F64<T>]) -> Option<F64<T>> {
        if point.len() != self.num_vars() {
            return None;
        }
        let (a, m, b, c) = self.handles();
        let pt = words::<T>(point);
        let mut out = 0u64;
        let rc = unsafe { sys::sc_gkr_w_evaluate(self.ctx().raw(), a, m, b, c, pt.as_ptr(), pt.len(), &mut out) };
        self.ctx().check(rc, "sc_gkr_w_evaluate");
        Some(from_word::<T>(out))
    }
    fn fix_variables(&self, partial_point: &[F64<T>]) -> Self {
        let (a, m, b, c) = self.handles();
        let r = words::<T>(partial_point);
        let mut h = [ptr::null_mut(); 4];
        let rc = unsafe {
            sys::sc_gkr_w_fix_variables(
                self.ctx().raw(), a, m, b, c, r.as_ptr(), r.len(), &mut h[0], &mut h[1], &mut h[2], &mut h[3],
            )
        };
        self.ctx().check(rc, "sc_gkr_w_fix_variables");
        let t = |x| Rc::new(DeviceMle::from_raw(self.ctx(), x));
        Self { add_i: t(h[0]), mul_i: t(h[1]), w_b: t(h[2]), w_c: t(h[3]) }
    }
    fn to_univariate(&self) -> SparsePolynomial<F64<T>> {
        let (a, m, b, c) = self.handles();
        let mut e = [0u64; 3];
        let rc = unsafe { sys::sc_gkr_w_round_sums(self.ctx().raw(), a, m, b, c, e.as_mut_ptr()) };
        self.ctx().check(rc, "sc_gkr_w_round_sums");
        round_poly::<T>(e)
    }
    fn num_vars(&self) -> usize {
        self.add_i.num_vars()
    }
    fn to_evaluations(&self) -> Vec<F64<T>> {
        let (a, m, b, c) = self.handles();
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_gkr_w_to_evaluations(self.ctx().raw(), a, m, b, c, &mut h) };
        self.ctx().check(rc, "sc_gkr_w_to_evaluations");
        DeviceMle::from_raw(self.ctx(), h).to_evaluations()
    }
    fn native_engine(&self) -> Option<Box<dyn sum_check_protocol::RoundEngine<F64<T>>>> {
        Some(Box::new(WEngine::new(self)))
    }
}

/// `sc_gkr_prover`: the two-phase W prover behind `SumCheckProver<F, W<F>>::round`.
pub struct WEngine<T: MontConfig<1>> {
    w: GpuW<T>,
    h: *mut sys::sc_gkr_prover,
}
impl<T: MontConfig<1>> WEngine<T> {
    pub fn new(w: &GpuW<T>) -> Self {
        let (a, m, b, c) = w.handles();
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_gkr_prover_create(w.ctx().raw(), a, m, b, c, &mut h) };
        w.ctx().check(rc, "sc_gkr_prover_create");
        Self { w: w.clone(), h }
    }
    /// the same prover from layer i's gate list (`sc_gkr_prover_create_sparse`): the dense predicate tables are never built
    pub fn from_gates(ctx: &Context<T>, gates: &[WiringGate], k_next: usize, r_i: &[F64<T>], w_next: Rc<DeviceMle<T>>) -> SparseWEngine<T> {
        let k_i = gates.len().trailing_zeros() as usize;
        let ty: Vec<i32> = gates.iter().map(|g| g.is_mul as i32).collect();
        let i0: Vec<u32> = gates.iter().map(|g| g.inputs[0]).collect();
        let i1: Vec<u32> = gates.iter().map(|g| g.inputs[1]).collect();
        let r = words::<T>(r_i);
        let mut h = ptr::null_mut();
        let rc = unsafe {
            sys::sc_gkr_prover_create_sparse(
                ctx.raw(), ty.as_ptr(), i0.as_ptr(), i1.as_ptr(), k_i, k_next, r.as_ptr(), w_next.h, &mut h,
            )
        };
        ctx.check(rc, "sc_gkr_prover_create_sparse");
        SparseWEngine { w_next, h }
    }
}
impl<T: MontConfig<1>> Drop for WEngine<T> {
    fn drop(&mut self) {
        unsafe { sys::sc_gkr_prover_destroy(self.h) };
    }
}
fn gkr_c1<T: MontConfig<1>>(ctx: &Context<T>, h: *mut sys::sc_gkr_prover) -> F64<T> {
    let mut out = 0u64;
    let rc = unsafe { sys::sc_gkr_prover_c1(h, &mut out) };
    ctx.check(rc, "sc_gkr_prover_c1");
    from_word::<T>(out)
}
fn gkr_round<T: MontConfig<1>>(ctx: &Context<T>, h: *mut sys::sc_gkr_prover, r_prev: F64<T>, j: usize) -> SparsePolynomial<F64<T>> {
    let mut e = [0u64; 3];
    let rc = unsafe { sys::sc_gkr_prover_round(h, word::<T>(&r_prev), j, e.as_mut_ptr()) };
    ctx.check(rc, "sc_gkr_prover_round");
    round_poly::<T>(e)
}
impl<T: MontConfig<1>> sum_check_protocol::RoundEngine<F64<T>> for WEngine<T> {
    fn c_1(&self) -> F64<T> {
        gkr_c1(self.w.ctx(), self.h)
    }
    fn round(&mut self, r_prev: F64<T>, j: usize) -> SparsePolynomial<F64<T>> {
        gkr_round(self.w.ctx(), self.h, r_prev, j)
    }
}
pub struct SparseWEngine<T: MontConfig<1>> {
    w_next: Rc<DeviceMle<T>>,
    h: *mut sys::sc_gkr_prover,
}
impl<T: MontConfig<1>> Drop for SparseWEngine<T> {
    fn drop(&mut self) {
        unsafe { sys::sc_gkr_prover_destroy(self.h) };
    }
}
impl<T: MontConfig<1>> sum_check_protocol::RoundEngine<F64<T>> for SparseWEngine<T> {
    fn c_1(&self) -> F64<T> {
        gkr_c1(&self.w_next.ctx, self.h)
    }
    fn round(&mut self, r_prev: F64<T>, j: usize) -> SparsePolynomial<F64<T>> {
        gkr_round(&self.w_next.ctx, self.h, r_prev, j)
    }
}

/// GPU-backed `triangle_counting::G` (triangle-counting/src/lib.rs:22-27); same field names.
pub struct GpuTriangleG<T: MontConfig<1>> {
    f_a_1: Rc<DeviceMle<T>>,
    f_a_2: Rc<DeviceMle<T>>,
    f_a_3: Rc<DeviceMle<T>>,
    var_len: usize,
}
impl<T: MontConfig<1>> Clone for GpuTriangleG<T> {
    fn clone(&self) -> Self {
        Self { f_a_1: self.f_a_1.clone(), f_a_2: self.f_a_2.clone(), f_a_3: self.f_a_3.clone(), var_len: self.var_len }
    }
}
impl<T: MontConfig<1>> GpuTriangleG<T> {
    /// `G::new_adj_matrix` (:32-51): `matrix` row-major, 2^(num_vars/2) vertices.
    pub fn new_adj_matrix(ctx: &Context<T>, num_vars: usize, matrix: &[bool]) -> Self {
        use ark_ff::{One, Zero};
        let evals = matrix.iter().map(|e| if *e { F64::<T>::one() } else { F64::<T>::zero() }).collect();
        let g = Rc::new(DeviceMle::from_evaluations_vec(ctx, num_vars, evals));
        Self { f_a_1: g.clone(), f_a_2: g.clone(), f_a_3: g, var_len: num_vars / 2 }
    }
    fn ctx(&self) -> &Context<T> {
        &self.f_a_1.ctx
    }
    fn x_vars_num(&self) -> usize {
        self.f_a_1.num_vars().saturating_sub(self.var_len) // :53-55
    }
    fn y_vars_num(&self) -> usize {
        self.f_a_2.num_vars().saturating_sub(self.var_len) // :57-59
    }
    fn z_vars_num(&self) -> usize {
        self.f_a_3.num_vars().min(self.var_len) // :61-67
    }
}
impl<T: MontConfig<1>> SumCheckPolynomial<F64<T>> for GpuTriangleG<T> {
    fn evaluate(&self, point: &[F64<T>]) -> Option<F64<T>> {
        if point.len() != self.num_vars() {
            return None;
        }
        let pt = words::<T>(point);
        let mut out = 0u64;
        let rc = unsafe {
            sys::sc_tri_evaluate(self.ctx().raw(), self.f_a_1.h, self.f_a_2.h, self.f_a_3.h, self.var_len, pt.as_ptr(), pt.len(), &mut out)
        };
        self.ctx().check(rc, "sc_tri_evaluate");
        Some(from_word::<T>(out))
    }
    fn fix_variables(&self, partial_point: &[F64<T>]) -> Self {
        let r = words::<T>(partial_point);
        let mut h = [ptr::null_mut(); 3];
        let rc = unsafe {
            sys::sc_tri_fix_variables(
                self.ctx().raw(), self.f_a_1.h, self.f_a_2.h, self.f_a_3.h, self.var_len, r.as_ptr(), r.len(),
                &mut h[0], &mut h[1], &mut h[2],
            )
        };
        self.ctx().check(rc, "sc_tri_fix_variables");
        let t = |x| Rc::new(DeviceMle::from_raw(self.ctx(), x));
        Self { f_a_1: t(h[0]), f_a_2: t(h[1]), f_a_3: t(h[2]), var_len: self.var_len }
    }
    fn to_univariate(&self) -> SparsePolynomial<F64<T>> {
        let mut e = [0u64; 3];
        let rc = unsafe {
            sys::sc_tri_round_sums(self.ctx().raw(), self.f_a_1.h, self.f_a_2.h, self.f_a_3.h, self.var_len, e.as_mut_ptr())
        };
        self.ctx().check(rc, "sc_tri_round_sums");
        round_poly::<T>(e)
    }
    fn num_vars(&self) -> usize {
        self.x_vars_num() + self.y_vars_num() + self.z_vars_num()
    }
    fn to_evaluations(&self) -> Vec<F64<T>> {
        let mut h = ptr::null_mut();
        let rc = unsafe {
            sys::sc_tri_to_evaluations(self.ctx().raw(), self.f_a_1.h, self.f_a_2.h, self.f_a_3.h, self.var_len, &mut h)
        };
        self.ctx().check(rc, "sc_tri_to_evaluations");
        DeviceMle::from_raw(self.ctx(), h).to_evaluations()
    }
    /// The fast engine applies to the polynomial as `new_adj_matrix` builds it (three views of one table,
    /// nothing fixed yet); any other state goes through the generic path.
    fn native_engine(&self) -> Option<Box<dyn sum_check_protocol::RoundEngine<F64<T>>>> {
        let fresh = Rc::ptr_eq(&self.f_a_1, &self.f_a_2)
            && Rc::ptr_eq(&self.f_a_2, &self.f_a_3)
            && self.var_len >= 1
            && self.f_a_1.num_vars() == 2 * self.var_len;
        if !fresh {
            return None;
        }
        let mut h = ptr::null_mut();
        let rc = unsafe { sys::sc_tri_prover_create(self.ctx().raw(), self.f_a_1.h, self.var_len, &mut h) };
        self.ctx().check(rc, "sc_tri_prover_create");
        Some(Box::new(TriEngine { adj: self.f_a_1.clone(), h }))
    }
}
/// `sc_tri_prover`: one n^3 pass + three product-of-two-tables sumchecks.
pub struct TriEngine<T: MontConfig<1>> {
    adj: Rc<DeviceMle<T>>,
    h: *mut sys::sc_tri_prover,
}
impl<T: MontConfig<1>> Drop for TriEngine<T> {
    fn drop(&mut self) {
        unsafe { sys::sc_tri_prover_destroy(self.h) };
    }
}
impl<T: MontConfig<1>> sum_check_protocol::RoundEngine<F64<T>> for TriEngine<T> {
    fn c_1(&self) -> F64<T> {
        let mut out = 0u64;
        let rc = unsafe { sys::sc_tri_prover_c1(self.h, &mut out) };
        self.adj.ctx.check(rc, "sc_tri_prover_c1");
        from_word::<T>(out)
    }
    fn round(&mut self, r_prev: F64<T>, j: usize) -> SparsePolynomial<F64<T>> {
        let mut e = [0u64; 3];
        let rc = unsafe { sys::sc_tri_prover_round(self.h, word::<T>(&r_prev), j, e.as_mut_ptr()) };
        self.adj.ctx.check(rc, "sc_tri_prover_round");
        round_poly::<T>(e)
    }
}

// =====================================================================================
// multilinear_extensions::{vsbw_, cti_}multilinear_from_evaluations (multilinear-extensions/src/lib.rs:6, :29)
// =====================================================================================

impl<T: MontConfig<1>> Context<T> {
    /// A per-thread default context on device 0 for the free functions below, which keep the reference's
    /// signatures (no context argument).  One per thread and field type, created on first use.
    pub fn thread_default() -> Self {
        use std::any::{Any, TypeId};
        use std::cell::RefCell;
        use std::collections::HashMap;
        thread_local! {
            static CTXS: RefCell<HashMap<TypeId, Box<dyn Any>>> = RefCell::new(HashMap::new());
        }
        CTXS.with(|m| {
            let mut m = m.borrow_mut();
            m.entry(TypeId::of::<T>())
                .or_insert_with(|| Box::new(Context::<T>::new(0)))
                .downcast_ref::<Context<T>>()
                .expect("context of this field type")
                .clone()
        })
    }
}

/// BE evaluation of a table: r[0] pairs with the most significant index bit.
fn evaluate_be<T: MontConfig<1>>(evals: &[F64<T>], r: &[F64<T>]) -> F64<T> {
    assert_eq!(evals.len(), 1usize << r.len());
    let ctx = Context::<T>::thread_default();
    let t = DeviceMle::from_evaluations_vec(&ctx, r.len(), evals.to_vec());
    let pt = words::<T>(r);
    let mut out = 0u64;
    let rc = unsafe { sys::sc_table_evaluate(ctx.raw(), t.h, pt.as_ptr(), pt.len(), sys::SC_ORDER_BE, &mut out) };
    ctx.check(rc, "sc_table_evaluate");
    from_word::<T>(out)
}
/// `multilinear_extensions::vsbw_multilinear_from_evaluations(evals, r)` (:6-24)
pub fn vsbw_multilinear_from_evaluations<T: MontConfig<1>>(evals: &[F64<T>], r: &[F64<T>]) -> F64<T> {
    evaluate_be::<T>(evals, r)
}
/// `multilinear_extensions::cti_multilinear_from_evaluations(evals, r)` (:29-48): the same polynomial,
/// the same value - one streaming pass instead of O(n 2^n) Lagrange products.
pub fn cti_multilinear_from_evaluations<T: MontConfig<1>>(evals: &[F64<T>], r: &[F64<T>]) -> F64<T> {
    evaluate_be::<T>(evals, r)
}

// =====================================================================================
// One process per GPU: the data planes of a sharded context, and whole proofs in one call
// =====================================================================================

impl<T: MontConfig<1>> Context<T> {
    /// `sc_ctx_set_option` (tunables of include/sumcheck_hip.h, e.g. "peer_spin_ms", "grid_sharded")
    pub fn set_option(&self, key: &str, value: i64) {
        let k = std::ffi::CString::new(key).expect("option name");
        let rc = unsafe { sys::sc_ctx_set_option(self.raw(), k.as_ptr(), value) };
        self.check(rc, "sc_ctx_set_option");
    }
    /// Peer transport, step 1: this rank's 64-byte IPC handle.  All-gather the handles of all ranks by any means
    /// (they are plain bytes), then call `comm_peer_connect` with them in rank order.
    pub fn comm_peer_export(&self, rank: i32, world: i32) -> [u8; 64] {
        let mut h = [0u8; 64];
        let rc = unsafe { sys::sc_ctx_comm_peer_export(self.raw(), rank, world, h.as_mut_ptr()) };
        self.check(rc, "sc_ctx_comm_peer_export");
        h
    }
    /// Peer transport, step 2: map the peers, say hello, self-test.  `Err` (the library's message) if this node's
    /// peer memory does not behave as the kernels need - the caller then joins RCCL instead, on every rank.
    pub fn comm_peer_connect(&self, handles: &[[u8; 64]]) -> Result<(), String> {
        let (mut rank, mut world) = (0i32, 0i32);
        let rc = unsafe { sys::sc_ctx_comm_rank(self.raw(), &mut rank, &mut world) };
        self.check(rc, "sc_ctx_comm_rank");
        if handles.len() != world as usize {
            // the C side reads 64 * world bytes
            return Err(format!("comm_peer_connect: expected {} handles, got {}", world, handles.len()));
        }
        let flat: Vec<u8> = handles.iter().flat_map(|h| h.iter().copied()).collect();
        let rc = unsafe { sys::sc_ctx_comm_peer_connect(self.raw(), flat.as_ptr()) };
        if rc == sys::SC_OK {
            Ok(())
        } else {
            Err(unsafe { CStr::from_ptr(sys::sc_last_error(self.raw())) }.to_string_lossy().into_owned())
        }
    }
    /// RCCL: rank 0 makes the id (`Context::rccl_unique_id`), the application broadcasts it, every rank joins.
    pub fn rccl_unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        let rc = unsafe { sys::sc_comm_unique_id(id.as_mut_ptr()) };
        assert_eq!(rc, sys::SC_OK, "sc_comm_unique_id");
        id
    }
    pub fn comm_init_rccl(&self, id: &[u8; 128], rank: i32, world: i32) {
        let rc = unsafe { sys::sc_ctx_comm_init_rccl(self.raw(), id.as_ptr(), rank, world) };
        self.check(rc, "sc_ctx_comm_init_rccl");
    }
    /// (rank, world) of this context's communicator
    pub fn rank_world(&self) -> (i32, i32) {
        let (mut r, mut w) = (0i32, 1i32);
        let rc = unsafe { sys::sc_ctx_comm_rank(self.raw(), &mut r, &mut w) };
        self.check(rc, "sc_ctx_comm_rank");
        (r, w)
    }
}

/// C trampoline for the `draw` callbacks of the whole-proof entry points: `user` points at a
/// `&mut dyn FnMut(usize, [u64; 3]) -> u64` (round, the round's (H(0), H(1), H(2)) -> the next challenge word).
unsafe extern "C" fn draw_trampoline(user: *mut std::ffi::c_void, round: usize, evals: *const u64) -> u64 {
    let f = &mut *(user as *mut &mut dyn FnMut(usize, [u64; 3]) -> u64);
    f(round, [*evals, *evals.add(1), *evals.add(2)])
}

/// (c_1, round polynomials as (H(0), H(1), H(2)) triples, challenges) of a whole proof
pub type Transcript<T> = (F64<T>, Vec<[F64<T>; 3]>, Vec<F64<T>>);

fn transcript_from<T: MontConfig<1>>(c1: u64, evals: &[u64], ch: &[u64]) -> Transcript<T> {
    let ev = evals.chunks(3).map(|e| [from_word::<T>(e[0]), from_word::<T>(e[1]), from_word::<T>(e[2])]).collect();
    (from_word::<T>(c1), ev, ch.iter().map(|&w| from_word::<T>(w)).collect())
}

impl<T: MontConfig<1>> GpuW<T> {
    /// `sc_gkr_prove`: the whole 2k-round sumcheck of one GKR layer in one call; `draw(round, evals)` returns the
    /// verifier's (or Fiat-Shamir's) next challenge as a Montgomery word.
    pub fn prove_with(&self, draw: &mut dyn FnMut(usize, [u64; 3]) -> u64) -> Transcript<T> {
        let n = self.num_vars();
        let (mut c1, mut ev, mut ch) = (0u64, vec![0u64; 3 * n], vec![0u64; n]);
        let mut cb: &mut dyn FnMut(usize, [u64; 3]) -> u64 = draw;
        let rc = unsafe {
            sys::sc_gkr_prove(self.add_i.ctx.raw(), self.add_i.h, self.mul_i.h, self.w_b.h, self.w_c.h, Some(draw_trampoline),
                              &mut cb as *mut _ as *mut std::ffi::c_void, 0, &mut c1, ev.as_mut_ptr(), ch.as_mut_ptr())
        };
        self.add_i.ctx.check(rc, "sc_gkr_prove");
        transcript_from::<T>(c1, &ev, &ch)
    }
}

impl<T: MontConfig<1>> GpuTriangleG<T> {
    /// `sc_tri_prove`: all 3 * var_len rounds of `Prover<F, G>` for `G::new_adj_matrix` in one call.
    pub fn prove_with(&self, draw: &mut dyn FnMut(usize, [u64; 3]) -> u64) -> Transcript<T> {
        let n = 3 * self.var_len;
        let (mut c1, mut ev, mut ch) = (0u64, vec![0u64; 3 * n], vec![0u64; n]);
        let mut cb: &mut dyn FnMut(usize, [u64; 3]) -> u64 = draw;
        let rc = unsafe {
            sys::sc_tri_prove(self.f_a_1.ctx.raw(), self.f_a_1.h, self.var_len, Some(draw_trampoline),
                              &mut cb as *mut _ as *mut std::ffi::c_void, 0, &mut c1, ev.as_mut_ptr(), ch.as_mut_ptr())
        };
        self.f_a_1.ctx.check(rc, "sc_tri_prove");
        transcript_from::<T>(c1, &ev, &ch)
    }
}

