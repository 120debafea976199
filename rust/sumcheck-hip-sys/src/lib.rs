//! Raw declarations of `include/sumcheck_hip.h`.  One item per C entry point, same order.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

pub const SC_OK: c_int = 0;
pub const SC_ERR_ARG: c_int = 1;
pub const SC_ERR_HIP: c_int = 2;
pub const SC_ERR_RCCL: c_int = 3;
pub const SC_ERR_OOM: c_int = 4;
pub const SC_ERR_STATE: c_int = 5;
pub const SC_ERR_UNSUPPORTED: c_int = 6;
pub const SC_ORDER_LE: c_int = 0;
pub const SC_ORDER_BE: c_int = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct sc_field {
    pub p: u64,
    pub p_inv_neg: u64,
    pub r_mod_p: u64,
    pub r2_mod_p: u64,
}

/// One timed launch (`sc_ctx_launch_log`); `kind` is one of the `SC_KIND_*` values of the header.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sc_launch_record {
    pub kind: i32,
    pub kf: i32,
    pub ks: i32,
    pub log_in: i32,
    pub bytes_read: u64,
    pub bytes_written: u64,
    pub ms: f64,
}

#[repr(C)]
pub struct sc_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct sc_table {
    _private: [u8; 0],
}
/// `SC_ABI_VERSION` of the header these declarations were written against; `sc_abi_version()` is the library's.
pub const SC_ABI_VERSION: c_int = 6;
/// The context options the schedule depends on (`sc_plan_proof`).  `struct_size` is set by
/// `sc_plan_options_init(&mut o, size_of::<sc_plan_options>())`; the library touches nothing beyond it.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sc_plan_options {
    pub struct_size: u32,
    pub vars_per_pass: i32,
    pub first_pass_vars: i32,
    pub grid_pass: i32,
    pub grid_log: i32,
    pub grid_max_vars: i32,
    pub grid_sharded: i32,
    pub tail_log: i32,
    pub use_mailbox: i32,
    pub gram_log: i32,
    pub host_tail_log: i32,
    pub wfold_log: i32,
    pub wfold_min_log: i32,
    pub wfold_always: i32,
    pub wfold5_min_log: i32,
}
/// One launch of a planned proof: `action` is one of the `SC_PLAN_*` values of the header.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sc_plan_step {
    pub action: i32,
    pub kf: i32,
    pub ks: i32,
    pub log_in: i32,
    pub sharded: i32,
}
#[repr(C)]
pub struct sc_prover {
    _private: [u8; 0],
}
#[repr(C)]
pub struct sc_gkr_prover {
    _private: [u8; 0],
}
#[repr(C)]
pub struct sc_tri_prover {
    _private: [u8; 0],
}

pub type sc_allreduce_fn = Option<unsafe extern "C" fn(user: *mut c_void, buf: *mut u64, count: usize) -> c_int>;
pub type sc_allgather_fn =
    Option<unsafe extern "C" fn(user: *mut c_void, send: *const u64, recv: *mut u64, count: usize) -> c_int>;
pub type sc_draw_fn = Option<unsafe extern "C" fn(user: *mut c_void, round: usize, evals: *const u64) -> u64>;

extern "C" {
    pub fn sc_field_from_modulus(p: u64, out: *mut sc_field) -> c_int;
    pub fn sc_field_to_mont(f: *const sc_field, canonical: u64) -> u64;
    pub fn sc_field_from_mont(f: *const sc_field, mont: u64) -> u64;
    pub fn sc_interpolate_quadratic(f: *const sc_field, e: *const u64, c: *mut u64) -> c_int;

    pub fn sc_ctx_create(f: *const sc_field, device: c_int, out: *mut *mut sc_ctx) -> c_int;
    /// one handle over `n_devices` GPUs of this process (a power of two, up to 8); every table made on it is split over
    /// the devices by its top index bits and the prover / evaluate / fix_variables calls work on the whole table
    pub fn sc_ctx_create_multi(f: *const sc_field, devices: *const c_int, n_devices: c_int, out: *mut *mut sc_ctx) -> c_int;
    pub fn sc_ctx_destroy(ctx: *mut sc_ctx) -> c_int;
    pub fn sc_last_error(ctx: *const sc_ctx) -> *const c_char;
    pub fn sc_ctx_set_option(ctx: *mut sc_ctx, key: *const c_char, value: i64) -> c_int;
    pub fn sc_ctx_get_option(ctx: *const sc_ctx, key: *const c_char, value: *mut i64) -> c_int;
    pub fn sc_ctx_synchronize(ctx: *mut sc_ctx) -> c_int;
    pub fn sc_ctx_stream(ctx: *const sc_ctx) -> *mut c_void;
    pub fn sc_ctx_kernel_time(ctx: *mut sc_ctx, out: *mut f64, reset: c_int) -> c_int;
    pub fn sc_ctx_launch_log(
        ctx: *mut sc_ctx,
        out: *mut sc_launch_record,
        cap: usize,
        n_out: *mut usize,
        reset: c_int,
    ) -> c_int;

    pub fn sc_comm_unique_id(id: *mut u8) -> c_int;
    pub fn sc_ctx_comm_init_rccl(ctx: *mut sc_ctx, id: *const u8, rank: c_int, world: c_int) -> c_int;
    pub fn sc_ctx_comm_init_host(
        ctx: *mut sc_ctx,
        rank: c_int,
        world: c_int,
        allreduce: sc_allreduce_fn,
        allgather: sc_allgather_fn,
        user: *mut c_void,
    ) -> c_int;
    pub fn sc_ctx_comm_peer_export(ctx: *mut sc_ctx, rank: c_int, world: c_int, handle: *mut u8) -> c_int;
    pub fn sc_ctx_comm_peer_connect(ctx: *mut sc_ctx, handles: *const u8) -> c_int;
    pub fn sc_ctx_comm_peer_connect_local(ctx: *mut sc_ctx, peers: *const *mut sc_ctx) -> c_int;
    pub fn sc_ctx_comm_rank(ctx: *const sc_ctx, rank: *mut c_int, world: *mut c_int) -> c_int;

    pub fn sc_table_upload(ctx: *mut sc_ctx, host: *const u64, len: usize, out: *mut *mut sc_table) -> c_int;
    /// a table over device memory the caller owns (borrowed: never written, never freed by the library)
    pub fn sc_table_from_device(ctx: *mut sc_ctx, device_ptr: *const u64, len: usize, out: *mut *mut sc_table) -> c_int;
    pub fn sc_table_generate(ctx: *mut sc_ctx, seed: u64, start: u64, len: usize, out: *mut *mut sc_table) -> c_int;
    pub fn sc_table_clone(ctx: *mut sc_ctx, t: *const sc_table, out: *mut *mut sc_table) -> c_int;
    pub fn sc_table_download(ctx: *mut sc_ctx, t: *const sc_table, host: *mut u64, len: usize) -> c_int;
    pub fn sc_table_len(t: *const sc_table) -> usize;
    pub fn sc_table_device_ptr(t: *const sc_table) -> *const u64;
    pub fn sc_table_free(ctx: *mut sc_ctx, t: *mut sc_table) -> c_int;
    pub fn sc_table_fix_variables(
        ctx: *mut sc_ctx,
        input: *const sc_table,
        r: *const u64,
        k: usize,
        order: c_int,
        out: *mut *mut sc_table,
    ) -> c_int;
    pub fn sc_table_evaluate(
        ctx: *mut sc_ctx,
        t: *const sc_table,
        r: *const u64,
        n: usize,
        order: c_int,
        out: *mut u64,
    ) -> c_int;
    /// `sc_table_evaluate` at `m` points (rows of `n` words) in one pass over the table
    pub fn sc_table_evaluate_many(ctx: *mut sc_ctx, t: *const sc_table, points: *const u64, m: usize, n: usize, order: c_int, out: *mut u64) -> c_int;
    pub fn sc_table_relabel(
        ctx: *mut sc_ctx,
        input: *const sc_table,
        a: usize,
        b: usize,
        k: usize,
        out: *mut *mut sc_table,
    ) -> c_int;

    pub fn sc_matmul_g_new(
        ctx: *mut sc_ctx,
        a: *const sc_table,
        b: *const sc_table,
        n: usize,
        point: *const u64,
        a_out: *mut *mut sc_table,
        b_out: *mut *mut sc_table,
    ) -> c_int;
    pub fn sc_prod2_to_evaluations(
        ctx: *mut sc_ctx,
        a: *const sc_table,
        b: *const sc_table,
        out: *mut *mut sc_table,
    ) -> c_int;
    pub fn sc_prod2_sum(ctx: *mut sc_ctx, a: *const sc_table, b: *const sc_table, out_c1: *mut u64) -> c_int;
    pub fn sc_prod2_round_sums(ctx: *mut sc_ctx, a: *const sc_table, b: *const sc_table, out_e: *mut u64) -> c_int;
    pub fn sc_prod2_fold_and_sums(
        ctx: *mut sc_ctx,
        a: *const sc_table,
        b: *const sc_table,
        r: *const u64,
        a_out: *mut *mut sc_table,
        b_out: *mut *mut sc_table,
        out_e: *mut u64,
    ) -> c_int;
    pub fn sc_prod2_evaluate(
        ctx: *mut sc_ctx,
        a: *const sc_table,
        b: *const sc_table,
        point: *const u64,
        n: usize,
        out: *mut u64,
    ) -> c_int;

    pub fn sc_abi_version() -> c_int;
    pub fn sc_plan_options_init(o: *mut sc_plan_options, struct_size: usize);
    pub fn sc_plan_proof(
        opt: *const sc_plan_options,
        num_vars: usize,
        world: c_int,
        transport: c_int,
        out: *mut sc_plan_step,
        cap: usize,
        n_out: *mut usize,
    ) -> c_int;
    pub fn sc_prover_create(
        ctx: *mut sc_ctx,
        a: *const sc_table,
        b: *const sc_table,
        out: *mut *mut sc_prover,
    ) -> c_int;
    pub fn sc_prover_c1(pr: *const sc_prover, out: *mut u64) -> c_int;
    pub fn sc_prover_num_vars(pr: *const sc_prover, out: *mut usize) -> c_int;
    pub fn sc_prover_round(pr: *mut sc_prover, r_prev: u64, j: usize, out_e: *mut u64) -> c_int;
    pub fn sc_prover_destroy(pr: *mut sc_prover) -> c_int;
    pub fn sc_prove(
        ctx: *mut sc_ctx,
        a: *const sc_table,
        b: *const sc_table,
        draw: sc_draw_fn,
        user: *mut c_void,
        seed_r: u64,
        c1: *mut u64,
        evals: *mut u64,
        challenges: *mut u64,
    ) -> c_int;

    // ---- gkr_protocol::round_polynomial::W ------------------------------------------------
    pub fn sc_gkr_wiring(
        ctx: *mut sc_ctx,
        gate_type: *const i32,
        in0: *const u32,
        in1: *const u32,
        k_i: usize,
        k_next: usize,
        r_i: *const u64,
        add_out: *mut *mut sc_table,
        mul_out: *mut *mut sc_table,
    ) -> c_int;
    pub fn sc_gkr_w_to_evaluations(
        ctx: *mut sc_ctx,
        add: *const sc_table,
        mul: *const sc_table,
        w_b: *const sc_table,
        w_c: *const sc_table,
        out: *mut *mut sc_table,
    ) -> c_int;
    pub fn sc_gkr_w_round_sums(
        ctx: *mut sc_ctx,
        add: *const sc_table,
        mul: *const sc_table,
        w_b: *const sc_table,
        w_c: *const sc_table,
        out_e: *mut u64,
    ) -> c_int;
    pub fn sc_gkr_w_fix_variables(
        ctx: *mut sc_ctx,
        add: *const sc_table,
        mul: *const sc_table,
        w_b: *const sc_table,
        w_c: *const sc_table,
        r: *const u64,
        k: usize,
        add_out: *mut *mut sc_table,
        mul_out: *mut *mut sc_table,
        w_b_out: *mut *mut sc_table,
        w_c_out: *mut *mut sc_table,
    ) -> c_int;
    pub fn sc_gkr_w_evaluate(
        ctx: *mut sc_ctx,
        add: *const sc_table,
        mul: *const sc_table,
        w_b: *const sc_table,
        w_c: *const sc_table,
        point: *const u64,
        n: usize,
        out: *mut u64,
    ) -> c_int;
    pub fn sc_gkr_prover_create(
        ctx: *mut sc_ctx,
        add: *const sc_table,
        mul: *const sc_table,
        w_b: *const sc_table,
        w_c: *const sc_table,
        out: *mut *mut sc_gkr_prover,
    ) -> c_int;
    pub fn sc_gkr_prover_create_sparse(
        ctx: *mut sc_ctx,
        gate_type: *const i32,
        in0: *const u32,
        in1: *const u32,
        k_i: usize,
        k_next: usize,
        r_i: *const u64,
        w_next: *const sc_table,
        out: *mut *mut sc_gkr_prover,
    ) -> c_int;
    pub fn sc_gkr_prove(
        ctx: *mut sc_ctx,
        add: *const sc_table,
        mul: *const sc_table,
        w_b: *const sc_table,
        w_c: *const sc_table,
        draw: sc_draw_fn,
        user: *mut c_void,
        seed_r: u64,
        c1: *mut u64,
        evals: *mut u64,
        challenges: *mut u64,
    ) -> c_int;
    pub fn sc_gkr_prover_c1(pr: *const sc_gkr_prover, out: *mut u64) -> c_int;
    pub fn sc_gkr_prover_round(pr: *mut sc_gkr_prover, r_prev: u64, j: usize, out_e: *mut u64) -> c_int;
    pub fn sc_gkr_prover_destroy(pr: *mut sc_gkr_prover) -> c_int;
    pub fn sc_table_restrict_to_line(
        ctx: *mut sc_ctx,
        t: *const sc_table,
        b: *const u64,
        c: *const u64,
        k: usize,
        out_coeffs: *mut u64,
    ) -> c_int;

    // ---- triangle_counting::G ----------------------------------------------------------------
    pub fn sc_tri_to_evaluations(
        ctx: *mut sc_ctx,
        f1: *const sc_table,
        f2: *const sc_table,
        f3: *const sc_table,
        var_len: usize,
        out: *mut *mut sc_table,
    ) -> c_int;
    pub fn sc_tri_round_sums(
        ctx: *mut sc_ctx,
        f1: *const sc_table,
        f2: *const sc_table,
        f3: *const sc_table,
        var_len: usize,
        out_e: *mut u64,
    ) -> c_int;
    pub fn sc_tri_fix_variables(
        ctx: *mut sc_ctx,
        f1: *const sc_table,
        f2: *const sc_table,
        f3: *const sc_table,
        var_len: usize,
        r: *const u64,
        k: usize,
        f1_out: *mut *mut sc_table,
        f2_out: *mut *mut sc_table,
        f3_out: *mut *mut sc_table,
    ) -> c_int;
    pub fn sc_tri_evaluate(
        ctx: *mut sc_ctx,
        f1: *const sc_table,
        f2: *const sc_table,
        f3: *const sc_table,
        var_len: usize,
        point: *const u64,
        n: usize,
        out: *mut u64,
    ) -> c_int;
    pub fn sc_tri_prover_create(
        ctx: *mut sc_ctx,
        adj: *const sc_table,
        var_len: usize,
        out: *mut *mut sc_tri_prover,
    ) -> c_int;
    pub fn sc_tri_prove(
        ctx: *mut sc_ctx,
        adj: *const sc_table,
        var_len: usize,
        draw: sc_draw_fn,
        user: *mut c_void,
        seed_r: u64,
        c1: *mut u64,
        evals: *mut u64,
        challenges: *mut u64,
    ) -> c_int;
    pub fn sc_tri_prover_c1(pr: *const sc_tri_prover, out: *mut u64) -> c_int;
    pub fn sc_tri_prover_round(pr: *mut sc_tri_prover, r_prev: u64, j: usize, out_e: *mut u64) -> c_int;
    pub fn sc_tri_prover_destroy(pr: *mut sc_tri_prover) -> c_int;
}
