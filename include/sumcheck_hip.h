/*
 * sumcheck_hip.h - C ABI of libsumcheck_hip.so, the MI355X (gfx950) sumcheck prover hot path.
 *
 * Drop-in boundary.  In the reference (montekki/thaler-study, Rust) this path sits behind
 * the generic trait `SumCheckPolynomial<F>` (sum-check-protocol/src/lib.rs:121-156) driven
 * by `Prover<F,P>` (:73-117) / `Verifier<F,P>` (:227-331); the table arithmetic itself is
 * ark_poly::DenseMultilinearExtension.  There is no FFI in the reference; these entry
 * points are what a thin `-sys` crate would bind (INTEGRATION.md shows the Rust side).
 * Each function cites the reference interface (file:line, relative to the reference
 * root) it replaces.
 *
 * Conventions
 *  - Every uint64_t field element is the Montgomery residue x*2^64 mod p in [0,p): exactly
 *    the memory word of ark-ff's `Fp64<MontBackend<_,1>>`, so `&[F]` crosses zero-copy.
 *  - Tables are multilinear-extension evaluation tables of length 2^k; variable 0 is index
 *    bit 0 (ark-poly order, "LE") unless an `order` argument says otherwise.
 *  - Every function returns an int status (SC_OK = 0).  Nothing throws or aborts across the
 *    ABI; sc_last_error() gives the message for the last non-zero status on that context.
 *    The reference's prover methods are infallible by signature, so the Rust shim panics
 *    on a non-zero status (SURVEY.md section 8b "Errors").
 *  - A context is NOT thread-safe (the reference's `round` takes `&mut self`); use one
 *    context per prover thread.  Different contexts may be used concurrently.
 *  - A context owns one HIP stream on one device.  Calls are synchronous from the caller's
 *    point of view unless stated otherwise.
 *  - There is no CPU fallback: every entry point that computes fails with SC_ERR_HIP when
 *    no gfx950 device is usable.
 */
#ifndef SUMCHECK_HIP_H
#define SUMCHECK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_OK 0
#define SC_ERR_ARG 1         /* bad argument (null, non power-of-two length, wrong round index ...) */
#define SC_ERR_HIP 2         /* HIP runtime / no device */
#define SC_ERR_RCCL 3        /* collective layer */
#define SC_ERR_OOM 4         /* device or host allocation failed */
#define SC_ERR_STATE 5       /* call sequence violates the protocol state machine */
#define SC_ERR_UNSUPPORTED 6

#define SC_ORDER_LE 0 /* variable 0 = index bit 0: ark_poly::DenseMultilinearExtension */
#define SC_ORDER_BE 1 /* variable 0 = index MSB: multilinear-extensions/src/lib.rs:6-60 */

/* Field description.  Mirrors `#[derive(MontConfig)] #[modulus = "..."]`
 * (sum-check-protocol/src/lib.rs:349-354) as runtime constants. */
typedef struct sc_field {
  uint64_t p;         /* odd prime modulus, 2 < p < 2^64 */
  uint64_t p_inv_neg; /* -p^-1 mod 2^64 */
  uint64_t r_mod_p;   /* 2^64 mod p  (Montgomery form of 1) */
  uint64_t r2_mod_p;  /* 2^128 mod p */
} sc_field;

typedef struct sc_ctx sc_ctx;
typedef struct sc_table sc_table;
typedef struct sc_prover sc_prover;

/* ---- field helpers (host, O(1)) ------------------------------------------------------- */

/* Fill an sc_field from a modulus.  p = 2^64-2^32+1 selects the Goldilocks kernels. */
int sc_field_from_modulus(uint64_t p, sc_field* out);
/* canonical integer <-> Montgomery word (F::from_bigint / into_bigint) */
uint64_t sc_field_to_mont(const sc_field* f, uint64_t canonical);
uint64_t sc_field_from_mont(const sc_field* f, uint64_t mont);
/* interpolate_quadratic_poly through (0,e0),(1,e1),(2,e2): matrix-multiplication/src/lib.rs:17-60,
 * :124-130.  c[d] = coefficient of X^d. */
int sc_interpolate_quadratic(const sc_field* f, const uint64_t e[3], uint64_t c[3]);

/* ---- context ------------------------------------------------------------------------- */

int sc_ctx_create(const sc_field* f, int device, sc_ctx** out);
/* ONE handle over n_devices GPUs of this process (sc_ctx_create(field, devices[], n_devices); 8e: "one
 * process drives all 8 devices, so the host-supplied r_j needs no broadcast").  The reference's caller is one process
 * holding one Prover (sum-check-protocol/src/lib.rs:73-117, mm_benchmark.rs:88-96); with this handle that unchanged caller
 * uses every GPU of the node - no launcher, no communicator, no replicated verifier randomness.  n_devices: a power of two,
 * 1..8; devices[] may repeat a device (tests on a one-GPU box).  Every table created on the handle (sc_table_upload /
 * _generate / _clone / _fix_variables, sc_matmul_g_new, sc_prod2_*) is ONE table whose d-th contiguous 1/n_devices (top
 * log2(n_devices) index bits = d) lives on devices[d]; sc_table_evaluate, sc_prod2_*, sc_prover_* and sc_prove work on the
 * whole table.  In one process the exchange step of a pass needs no collective: every device's kernel leaves its own sums
 * in its own pinned mailbox and the calling thread adds them (split limbs / residues mod p) before it answers the round;
 * launches go out from one thread per device.  When a pass leaves <= 2^host_tail_log entries per table and device it writes
 * them to pinned host memory and the host finishes the proof (option "host_tail_log"): every launcher thread folds its
 * device's pending challenges, the calling thread serves the remaining rounds.  Results are bit-identical to a one-device context.
 * Also served: sc_gkr_wiring, sc_gkr_prover_* / sc_gkr_prove (the dense W prover: every device streams its rows of c of add_i /
 * mul_i, the small product proofs run on the first device), sc_gkr_w_evaluate, and sc_tri_prover_* / sc_tri_prove /
 * sc_tri_evaluate (every device squares its rows of the adjacency matrix).
 * Through the handle's FIRST device (round 5: gather, the ordinary call on whole tables, scatter of table results - functional,
 * not fast; SC_ERR_UNSUPPORTED before): tables with fewer entries than devices (they live on the first device), sc_table_fix_variables
 * across the device bits (either order), sc_table_relabel, the product calls on one-entry shards, sc_matmul_g_new on matrices with
 * fewer than two columns per device, and the callers' generic trait calls sc_gkr_w_round_sums / _to_evaluations / _fix_variables
 * and sc_tri_round_sums / _to_evaluations / _fix_variables - the reference's fallback loop fix_variables(&[r]) -> to_univariate
 * (gkr-protocol/src/round_polynomial.rs:59-90, triangle-counting/src/lib.rs:89-132) runs on a handle down to the constant.
 * Not served (SC_ERR_UNSUPPORTED): sc_table_from_device, sc_gkr_prover_create / sc_tri_prover_create on tables with fewer rows than
 * devices (sc_gkr_prover_create_sparse runs on the handle's first device), and the sc_ctx_comm_* calls (the handle is its own
 * communicator; sc_ctx_comm_rank reports rank 0 of 1 - its tables are whole tables; option "n_devices" counts the devices).
 * sc_table_device_ptr returns NULL; sc_ctx_stream, sc_ctx_kernel_time and sc_ctx_launch_log report the first device (one GPU's
 * launches over its own shard).  A handle is validated on ONE physical device only (devices[] naming device 0 N times: this pool's
 * boxes have one GPU); its cross-device copies are hipMemcpyPeerAsync and its tail outputs system-scope write-through stores. */
int sc_ctx_create_multi(const sc_field* f, const int* devices, int n_devices, sc_ctx** out);
int sc_ctx_destroy(sc_ctx* ctx);
/* message of the last failing call on ctx (ctx == NULL: last failing sc_ctx_create) */
const char* sc_last_error(const sc_ctx* ctx);
/* Tunables (all have working defaults; they exist for measurements and tests):
 *   "vars_per_pass"    rounds served by one device pass: 1 | 2 (default 2)
 *   "first_pass_vars"  rounds served by the prover's first pass, which has nothing to fold:
 *                      1 | 2 | 3 | 4, default 0 = four for tables (shards) of >= 2^"gram_log" entries (below), else three
 *                      for tables of >= 2^18 entries, two below (never more than vars_per_pass allows).  4 = the
 *                      matrix-core pass at any size from 2^14 entries (where it does not apply - smaller tables,
 *                      "vars_per_pass" 1, sharded provers with "grid_sharded" 0 - the same as 0)
 *   "fold_dma"         (default 1) pass_kernel<4,2>, the fold behind the matrix-core first pass, brings its sub-steps in by LDS-DMA
 *                      (Goldilocks; 0 and every other modulus: through registers).  Same results; for A/B measurements
 *   "pipe32"           (default 1) pass_kernel<3,2> on whole tiles of tables of >= 2^"pipe32_log" (default 20) entries runs in its
 *                      pipelined form (sub-steps of 64 outputs, loads in flight while the wave multiplies, three waves per
 *                      SIMD); 0 = the staged form everywhere.  Same results; for A/B measurements
 *   "gram_log"         (default 21; 0 = never, else 14..40) a proof on tables (a sharded one: shards) of >= 2^gram_log entries opens
 *                      with gram_pass_kernel: rounds 1..4 from ONE read, as exact integer limb products of the tables'
 *                      bytes on the int8 matrix cores; every block reduces its accumulators to the 81 cells mod p and the
 *                      block that finishes last adds them (ONE launch since round 5); the pass behind it folds four
 *                      variables (wfold_pass_kernel<4,5>, or pass_kernel<4,2>: below).  DESIGN.md section 4
 *   "wfold_log"        (default 40; 0 = never, else 12..40) the pass behind the matrix-core pass - four pending challenges, round 4 -
 *                      on tables (shards) of 2^"wfold_min_log" (default 21) .. 2^wfold_log entries may be wfold_pass_kernel<4,5>:
 *                      it folds the four challenges AND serves FIVE rounds (pass_kernel<4,2>: two).  Taken where the proof then
 *                      needs fewer launches (the planner counts both ways: n = 25, 27, 28, 29 on one device) or where the
 *                      alternative is a grid pass streaming the whole table (n = 21..24); "wfold_always" = 1: wherever it can
 *                      run.  "wfold5_min_log" (default 24): a grid pass with FIVE challenges to fold over tables of >= 2^this
 *                      entries runs in the same kernel's (5, 3..5) form.  Same results; DESIGN.md sections 4, 5
 *   "host_tail_log"    (default 12; 0..12) the host finishes the proof: a pass whose folded tables have <= 2^host_tail_log
 *                      entries (per device on a multi-device handle) writes them to pinned host memory instead of the
 *                      pool, and every later round is served by the host from them - fold the pending challenges
 *                      (<= 2^13 multiply-adds), then one sub-microsecond round at a time - with NO further launch: the
 *                      last launch (or two) of every proof and every shard, ~13 us each, disappears; the planner aims its
 *                      grid passes at this hand-over (fewest launches up to it, fewest rounds per pass among those) - at
 *                      2^11 entries where the larger hand-over saves no launch (the host's first read of what the device
 *                      wrote costs ~0.1 us per KiB).  Not a CPU path
 *                      for the hot loop: what the host touches is what is left when 2^-18 of the work remains.  Applies
 *                      to unsharded provers (incl. a sharded one after its gather) and to multi-device handles; five-round
 *                      passes (and any pass with <= 32 outputs) hand over, 0 = off (a multi-device handle then hands
 *                      over once a shard holds <= 32 entries, as in round 4)
 *   "grid_pass"        the passes whose FOLDED tables have <= 2^"grid_log" (default 20) entries serve up to
 *                      "grid_max_vars" (default 5) rounds each and fold up to five pending challenges at once
 *                      (wgrid_pass_kernel); "grid_blocks" caps the launch (0 = what fits on the chip at once).
 *                      1 (default) | 0 = two rounds per pass all the way down.  Not used with "vars_per_pass" 1
 *                      or, for the first pass, an explicit "first_pass_vars" - those name their own schedules.
 *                      "grid_sharded" (1): the shards of a sharded prover go on with these passes too (cells
 *                      exchanged inside the kernel on the peer transport, summed by one collective per pass on
 *                      the others) until they are down to their pending challenges; on the peer transport one
 *                      small launch then serves the rounds of the rank bits (no gather).  0: two-round sharded
 *                      passes, gather at "tail_log".
 *   "tail_log"         shard log-size at which a sharded prover gathers with "grid_sharded" 0 (default 16)
 *   "max_blocks"       grid cap of the streaming kernels (default 3 per CU = 768); a pass never
 *                      launches more blocks than fit on the chip at once
 *   "use_mailbox"      kernels publish sums to pinned host memory the host spins on (default 1)
 *   "arena_log"        peer transport: a gather arena holds world * 2^arena_log words per table (default 17;
 *                      set before sc_ctx_comm_peer_export; longer gathers go in chunks)
 *   "peer_spin_ms"     peer transport: bound of every in-kernel wait for a peer = the largest skew between the
 *                      ranks' launches of the same pass that is tolerated (default 2000); past it the pass fails
 *                      with SC_ERR_RCCL on every rank that waited.  "peer_connect_ms" (default 120000): how long
 *                      sc_ctx_comm_peer_connect waits for every peer's hello (absorbs the start-up lag of a job)
 *   "rccl_timeout_ms"  RCCL plane: how long the host waits for work queued behind a collective - a sharded pass's sums, a
 *                      gathered table - before it gives the communicator up (default 30000).  A collective whose peer is gone
 *                      (a rank that died or is out of step) never completes on its own; past the bound, or as soon as
 *                      ncclCommGetAsyncError reports an error, the library calls ncclCommAbort, poisons the context and
 *                      returns SC_ERR_RCCL from the call that waited and from every later one
 *   "dbg_delay_ms" / "dbg_skip_tag"   fault injection for tests: delay every sharded launch of this rank on the
 *                      host / make its next sharded launch skip an exchange tag (a rank out of step)
 *   "pool_contiguous"  experiments: 1 = pool blocks of >= 1 MiB are asked for as physically contiguous VRAM
 *                      (hipDeviceMallocContiguous; default 0; experiments/r04_vmm_placement.md, addendum 2)
 *   "dbg_fold_grab"    measurements: tiles per draw of fold_kernel's four-wave launches (0 = the default, 1; 4 = round 3's
 *                      behaviour, profiles/r04_fold_small_grab_ab.txt)
 *   "time_kernels"     HIP-event timing of pass kernels (see sc_ctx_kernel_time)
 *   "prewarm"          an ACTION, not a setting: value = num_vars of the proofs to come.  Pays now what the first proof of that size
 *                      would pay on this context - the library's code object on the device, the resident-grid queries of the plan's
 *                      kernels, the matrix-core pass's workspace, the pool blocks of every folded table - for callers that prove
 *                      once (mm_benchmark.rs:88-96 builds g and proves).  On a multi-device handle every device does its shard's
 *                      share of this (the handle must be healthy: a poisoned handle refuses with SC_ERR_STATE like any other call)
 *   "stat_reset"       (set) zeroes, and "stat_wait_ns" / "stat_launch_ns" (get) read, where a proof's wall time goes on the host:
 *                      ns spent waiting for pass kernels (their run time + launch latency) and ns inside the pass launches (buffers,
 *                      weights, hipLaunchKernelGGL); the rest is host arithmetic between them.  Always on (four clock reads per pass).
 *                      On a multi-device handle both read the FIRST device's context only (the devices run side by side: their
 *                      waits overlap, a sum would count the same wall time N times)
 *   "nt_load_log" / "nt_store_log"  table log-size from which loads / stores are nontemporal
 * Environment: SC_RCCL_LIBRARY = the RCCL build sc_ctx_comm_init_rccl / sc_comm_unique_id dlopen (a path; default librccl.so.1).
 * That library or nothing: a path that cannot be loaded is an error, never a silent second choice. */
int sc_ctx_set_option(sc_ctx* ctx, const char* key, int64_t value);
/* reads any option back; also read-only: "n_devices" (1, or the devices behind a multi-device handle), "transport" (0 none,
 * 1 RCCL, 2 host callbacks, 3 peer, 4 local = a multi-device handle) and "comm_nranks"
 * = the number of ranks the data plane spans as the transport itself reports it (ncclCommCount for RCCL) */
int sc_ctx_get_option(const sc_ctx* ctx, const char* key, int64_t* value);
int sc_ctx_synchronize(sc_ctx* ctx);
/* the context's hipStream_t (for HIP-event timing by a benchmark harness) */
void* sc_ctx_stream(const sc_ctx* ctx);
/* device-time accounting of the pass kernels launched since the last reset:
 * out[0] = number of pass launches, out[1] = sum of their durations in ms (HIP events on
 * the context's stream; enabled by option "time_kernels" = 1). */
int sc_ctx_kernel_time(sc_ctx* ctx, double out[2], int reset);
/* Per-launch records of the same instrumentation (option "time_kernels" = 1): what each timed
 * launch was, the HBM bytes it has to move (every input once, every output once) and its HIP-event
 * duration.  A benchmark derives bytes actually moved and per-kernel GB/s from the launches that
 * really ran instead of from a model.  Copies min(cap, n) records, stores the number available in
 * *n_out, clears the log if reset.  The log keeps at most 65536 records. */
#define SC_KIND_PASS 0       /* pass_kernel<kf,ks>: fold kf variables of both tables + grid sums of ks rounds */
/* 1 was a three-round tail kernel that wgrid_pass_kernel replaced (round 3) */
#define SC_KIND_EVALUATE 2   /* evaluate_kernel: one table, kf = number of variables */
#define SC_KIND_FOLD 3       /* fold_kernel<kf>: LE fix of kf <= 3 variables of one table */
#define SC_KIND_FIX_LOW 4    /* fix_low_kernel: LE fix of 8..17 variables in one pass */
#define SC_KIND_FOLD_BE 5    /* fold_be_kernel: BE fix of one variable */
#define SC_KIND_COLDOT 6     /* coldot_kernel (+ sum_rows_kernel): BE multi-variable fix / f_A of G::new */
#define SC_KIND_GKR 7        /* GKR W pass: fold add/mul/w + round sums */
#define SC_KIND_MATSQ 8      /* triangle counting: square of the adjacency matrix */
/* 9 was the resident prover kernel (removed in round 3: measured equal to launches) */
#define SC_KIND_GRID_PASS 10 /* wgrid_pass_kernel: fold kf <= 5 variables of tables of <= 2^20 folded entries + the 3^ks cells of ks <= 5 rounds */
#define SC_KIND_GRAM_PASS 11   /* gram_pass_kernel: the four-round first pass of a large proof on the int8 matrix cores (ks = 4) */
#define SC_KIND_WFOLD_PASS 13  /* wfold_pass_kernel: a fold of kf = 4 / 5 variables of large tables that serves ks = 5 / 3..5 rounds (behind the matrix-core first pass; the pass behind that) */
/* 12 was gram_finish_kernel (rounds 4: a second launch behind the gram pass; folded into gram_pass_kernel in round 5) */
typedef struct sc_launch_record {
  int32_t kind;           /* SC_KIND_* */
  int32_t kf, ks;         /* variables folded / rounds served (meaning per kind above) */
  int32_t log_in;         /* log2 entries per input table */
  uint64_t bytes_read;    /* HBM bytes the launch must read */
  uint64_t bytes_written; /* HBM bytes the launch must write */
  double ms;              /* duration (HIP events on the context's stream) */
} sc_launch_record;
int sc_ctx_launch_log(sc_ctx* ctx, sc_launch_record* out, size_t cap, size_t* n_out, int reset);

/* ---- sharding (one process per GPU) ------------------------------ */

/* 128-byte RCCL unique id, created on rank 0 and broadcast by the caller. */
int sc_comm_unique_id(uint8_t id[128]);
/* Join an RCCL communicator.  After this call every table of ctx is the rank-th contiguous
 * shard (top log2(world) index bits = rank) of a table world times as long. */
int sc_ctx_comm_init_rccl(sc_ctx* ctx, const uint8_t id[128], int rank, int world);
/* Same sharding semantics with caller-supplied host collectives (used by tests and by
 * transports other than RCCL).  allreduce sums count uint64 words elementwise in place
 * across ranks (plain wrapping u64 adds; the library only ever sends 32-bit limbs);
 * allgather concatenates count words of every rank in rank order into recv. */
typedef int (*sc_allreduce_fn)(void* user, uint64_t* buf, size_t count);
typedef int (*sc_allgather_fn)(void* user, const uint64_t* send, uint64_t* recv, size_t count);
int sc_ctx_comm_init_host(sc_ctx* ctx, int rank, int world, sc_allreduce_fn allreduce,
                          sc_allgather_fn allgather, void* user);
/* Peer transport: the data plane of a one-node run without any collective launch.  Every rank exports a
 * small region of its HBM (an inbox for the per-pass round sums + arenas for gathers), maps every
 * other rank's region (HIP IPC, written over xGMI), and from then on the last workgroup of each sharded
 * pass exchanges its sums with the peers itself and hands the totals to its host (kernels.hpp, PeerX).
 * Up to 8 ranks.  Usage: export on every rank -> all-gather the 64-byte handles by any means (they are
 * plain bytes) -> connect with the world x 64 bytes in rank order.  connect verifies each mapped region's header
 * (same world and arena size, the expected rank), says hello to every peer and waits (up to "peer_connect_ms") until
 * every peer has said hello - by then all ranks have mapped, loaded their code and run a kernel - and then runs a
 * self-test through the real paths (an in-kernel exchange and a gather of known words); a node whose fine-grained
 * peer memory does not behave as the kernels assume fails here with SC_ERR_RCCL and the caller picks another
 * transport.  Afterwards the ranks must call the library in lockstep: an in-kernel wait for a peer is bounded by
 * "peer_spin_ms" (default 2 s).  Sharding semantics are those of sc_ctx_comm_init_rccl.  The challenges every rank
 * feeds to sc_prover_round / returns from `draw` must be identical; the exchange carries a digest of them and a
 * mismatch fails the pass with SC_ERR_STATE. */
int sc_ctx_comm_peer_export(sc_ctx* ctx, int rank, int world, uint8_t handle[64]);
int sc_ctx_comm_peer_connect(sc_ctx* ctx, const uint8_t* handles);
/* the same for ranks that are contexts of ONE process (threads; tests): peers[q] = rank q's context */
int sc_ctx_comm_peer_connect_local(sc_ctx* ctx, sc_ctx* const* peers);
int sc_ctx_comm_rank(const sc_ctx* ctx, int* rank, int* world);

/* ---- tables: DenseMultilinearExtension<F> on the device ------------------------------- */

/* DenseMultilinearExtension::from_evaluations_vec (matrix-multiplication/src/lib.rs:81,85) */
int sc_table_upload(sc_ctx* ctx, const uint64_t* host, size_t len, sc_table** out);
/* The same over device memory the caller already owns (its own kernels' output, another library's buffer, a mapping it made
 * with the HIP virtual-memory API): zero-copy and BORROWED - never written, never freed by the library (sc_table_free drops
 * the handle only); it must stay valid, and unchanged, while any table or prover made from it lives.  16-byte aligned. */
int sc_table_from_device(sc_ctx* ctx, const uint64_t* device_ptr, size_t len, sc_table** out);
/* Synthetic fill on the device:
 * t[i] = to_mont(splitmix64(seed + start + i) mod p), i < len. */
int sc_table_generate(sc_ctx* ctx, uint64_t seed, uint64_t start, size_t len, sc_table** out);
/* `.clone()` (every caller: Prover::new(g.clone()), matrix-multiplication/src/lib.rs:337) */
int sc_table_clone(sc_ctx* ctx, const sc_table* t, sc_table** out);
/* DenseMultilinearExtension::to_evaluations (matrix-multiplication/src/lib.rs:138-139) */
int sc_table_download(sc_ctx* ctx, const sc_table* t, uint64_t* host, size_t len);
size_t sc_table_len(const sc_table* t);
const uint64_t* sc_table_device_ptr(const sc_table* t);
int sc_table_free(sc_ctx* ctx, sc_table* t);

/* DenseMultilinearExtension::fix_variables(&r[..k]) (matrix-multiplication/src/lib.rs:83,86,
 * 104-105): new[b] = t[2b] + r*(t[2b+1]-t[2b]) per variable for SC_ORDER_LE; stride-half
 * pairing for SC_ORDER_BE.  *out has len >> k entries.  Input is not modified.  Asynchronous: the new table is
 * ready in the order of the context's stream (every sc_* call on the context is), sc_ctx_synchronize waits. */
int sc_table_fix_variables(sc_ctx* ctx, const sc_table* in, const uint64_t* r, size_t k, int order,
                           sc_table** out);
/* Polynomial::evaluate (matrix-multiplication/src/lib.rs:97-98) for SC_ORDER_LE;
 * vsbw_/cti_multilinear_from_evaluations (multilinear-extensions/src/lib.rs:6-48) for
 * SC_ORDER_BE.  n must equal log2(len) (times world when sharded: r then holds all n). */
int sc_table_evaluate(sc_ctx* ctx, const sc_table* t, const uint64_t* r, size_t n, int order,
                      uint64_t* out);
/* The same at m points in ONE pass over the table: points = m rows of n words, out = m values.  Callers that evaluate one
 * table several times - restrict_poly's k + 1 points on a line (gkr-protocol/src/lib.rs:291-321), a verifier's oracle
 * queries - pay one read of the table, one launch and one hand-off instead of m (batches of 16 points per launch, 8 on a
 * generic modulus; up to 4
 * points the pass stays memory-bound, beyond it is bound by the 2 m multiply-accumulates per 16 bytes).  Works on sharded
 * contexts and multi-device handles like sc_table_evaluate. */
int sc_table_evaluate_many(sc_ctx* ctx, const sc_table* t, const uint64_t* points, size_t m, size_t n, int order, uint64_t* out);
/* DenseMultilinearExtension::relabel(a, b, k) (matrix-multiplication/src/lib.rs:82) */
int sc_table_relabel(sc_ctx* ctx, const sc_table* in, size_t a, size_t b, size_t k, sc_table** out);

/* ---- product of two tables: matrix_multiplication::G --------------------------------- */

/* G::new (matrix-multiplication/src/lib.rs:77-92): A, B are 2^n x 2^n row-major tables of
 * 2^(2n) entries; point has 2n entries; outputs have 2^n entries. */
int sc_matmul_g_new(sc_ctx* ctx, const sc_table* A, const sc_table* B, size_t n,
                    const uint64_t* point, sc_table** a_out, sc_table** b_out);
/* G::to_evaluations (matrix-multiplication/src/lib.rs:137-146): out[i] = a[i]*b[i] */
int sc_prod2_to_evaluations(sc_ctx* ctx, const sc_table* a, const sc_table* b, sc_table** out);
/* Prover::new's claim: sum_i a[i]*b[i] (sum-check-protocol/src/lib.rs:89) - never
 * materialises the product vector. */
int sc_prod2_sum(sc_ctx* ctx, const sc_table* a, const sc_table* b, uint64_t* out_c1);
/* G::to_univariate's three sums H(0),H(1),H(2) (matrix-multiplication/src/lib.rs:110-122) */
int sc_prod2_round_sums(sc_ctx* ctx, const sc_table* a, const sc_table* b, uint64_t out_e[3]);
/* G::fix_variables(&[r]) followed by to_univariate on the result, fused: one read of the
 * inputs, one write of the folded tables (matrix-multiplication/src/lib.rs:103-131). */
int sc_prod2_fold_and_sums(sc_ctx* ctx, const sc_table* a, const sc_table* b, const uint64_t r[1],
                           sc_table** a_out, sc_table** b_out, uint64_t out_e[3]);
/* G::evaluate (matrix-multiplication/src/lib.rs:96-101) */
int sc_prod2_evaluate(sc_ctx* ctx, const sc_table* a, const sc_table* b, const uint64_t* point,
                      size_t n, uint64_t* out);

/* ---- Prover<F, G> (sum-check-protocol/src/lib.rs:73-117) ------------------------------ */

/* Prover::new(g) (:88-97).  Borrows a and b (they must outlive the prover and are never
 * written, so the verifier's oracle copy needs no clone).  Computes c_1. */
int sc_prover_create(sc_ctx* ctx, const sc_table* a, const sc_table* b, sc_prover** out);
/* Prover::c_1 (:100-102) */
int sc_prover_c1(const sc_prover* pr, uint64_t* out);
/* Prover::num_vars (:114-116) */
int sc_prover_num_vars(const sc_prover* pr, size_t* out);
/* The schedule of a proof WITHOUT a device: the launches sc_prove would issue for a proof of num_vars variables on a
 * communicator of `world` ranks over `transport` (0 none, 1 RCCL, 2 host callbacks, 3 peer, 4 local: the devices of a
 * multi-device handle), computed by the same
 * planner the engine runs at every pass (pure host logic: callable - and tested - on a machine without a GPU).
 * Steps come in launch order; `kf` challenges are folded and `ks` rounds served by each; `log_in` = log2 entries per
 * table on a rank; `sharded` = its sums are exchanged across the ranks.  SC_PLAN_GATHER is the all-gather of both
 * tables (no rounds).  Returns SC_ERR_STATE for option combinations the engine would refuse. */
#define SC_PLAN_PASS 0       /* pass_kernel<kf,ks> */
#define SC_PLAN_GRID_PASS 1  /* wgrid_pass_kernel: up to five rounds */
#define SC_PLAN_RANK_PASS 2  /* rank_pass_kernel (peer transport): the rounds of the rank bits */
#define SC_PLAN_GATHER 3     /* all-gather of the shards; the proof goes on replicated */
#define SC_PLAN_HOST_TAIL 4  /* the host finishes: it folds the kf pending challenges of the <= 2^host_tail_log-entry tables the pass before
                              * wrote to pinned host memory (log_in = log2 entries per table and device) and serves every remaining
                              * round (ks) itself; no launch.  Always the last step */
#define SC_PLAN_GRAM_PASS 5  /* gram_pass_kernel: rounds 1..4 of a proof on tables (shards) of >= 2^gram_log entries */
#define SC_PLAN_WFOLD_PASS 6 /* wfold_pass_kernel: folds kf = 4 challenges (behind the matrix-core pass) and serves five rounds, or kf = 5 and ks = 3..5
                              * on tables of >= 2^wfold5_min_log entries: the streaming fold with a grid pass's cells */
/* ABI version of this header: bumped whenever a struct below grows or an enum is extended.  sc_abi_version()
 * returns the library's; a caller built against another major version must not pass structs.  Version 5 = round 5; version 6 =
 * round 6: the option block is initialised by sc_plan_options_init (a NEW symbol: the one-argument sc_plan_options_default of
 * version 4 and the two-argument one of version 5 are gone, so a caller built against either fails at link / load time instead
 * of handing the library a struct of another size - ADVICE r05). */
#define SC_ABI_VERSION 6
int sc_abi_version(void);
typedef struct sc_plan_options {   /* the context options the schedule depends on (sc_ctx_set_option names) */
  uint32_t struct_size;            /* sizeof(sc_plan_options) as the CALLER compiled it: set by sc_plan_options_init from its argument;
                                    * the library reads and writes no field beyond it (fields a caller's struct lacks take their defaults) */
  int32_t vars_per_pass, first_pass_vars, grid_pass, grid_log, grid_max_vars, grid_sharded, tail_log, use_mailbox, gram_log;
  int32_t host_tail_log;           /* since version 5 */
  int32_t wfold_log;               /* since version 5 */
  int32_t wfold_min_log;           /* since version 5 */
  int32_t wfold_always;            /* since version 5 */
  int32_t wfold5_min_log;          /* since version 5 */
} sc_plan_options;
typedef struct sc_plan_step {
  int32_t action, kf, ks, log_in, sharded;
} sc_plan_step;
/* struct_size = sizeof(sc_plan_options) at the call site */
void sc_plan_options_init(sc_plan_options* o, size_t struct_size);
int sc_plan_proof(const sc_plan_options* opt, size_t num_vars, int world, int transport, sc_plan_step* out, size_t cap,
                  size_t* n_out);

/* Prover::round(r_prev, j) (:105-112).  Rounds must be called in order j = 0,1,...;
 * r_prev is ignored for j == 0 like in the reference.  out_e = (H(0),H(1),H(2)) of the
 * round polynomial; sc_interpolate_quadratic turns it into the coefficients that
 * `univariate::SparsePolynomial` holds. */
int sc_prover_round(sc_prover* pr, uint64_t r_prev, size_t j, uint64_t out_e[3]);
int sc_prover_destroy(sc_prover* pr);

/* Whole interactive run in one call, the loop of mm_benchmark.rs:88-96: Prover::new, then
 * n rounds; after each round the challenge is obtained from `draw` (called on the host
 * with the round's three sums already read back - the verifier's rng.draw() at
 * sum-check-protocol/src/lib.rs:283, or a Fiat-Shamir hash).  draw == NULL uses the
 * synthetic challenger r_j = to_mont(splitmix64(seed_r + j + 1) mod p).
 * evals (3*n words) and challenges (n words) may be NULL.
 * Sharded contexts: every rank runs this call; `draw` is invoked on every rank and MUST return the same
 * challenge everywhere (derive it from the transcript, or draw on one rank and broadcast inside `draw`).
 * The peer transport checks this (digest in the exchange); the other transports do not. */
typedef uint64_t (*sc_draw_fn)(void* user, size_t round, const uint64_t evals[3]);
int sc_prove(sc_ctx* ctx, const sc_table* a, const sc_table* b, sc_draw_fn draw, void* user,
             uint64_t seed_r, uint64_t* c1, uint64_t* evals, uint64_t* challenges);

/* ---- gkr_protocol::round_polynomial::W -----------------------
 *   f(b,c) = add_i(r_i,b,c) (W(b) + W(c)) + mul_i(r_i,b,c) W(b) W(c)
 * (gkr-protocol/src/round_polynomial.rs:13-44).  add/mul: tables of kb+kc variables indexed
 * (c << kb) | b; w_b: kb variables, w_c: kc variables (kb = kc = k_{i+1} when the sumcheck
 * starts; b variables are fixed first).
 * Sharded contexts: add/mul are this rank's rows of c (top log2(world) index bits = rank), w_b and w_c whole on
 * every rank.  sc_gkr_wiring builds exactly those shards (every rank passes the whole gate list and keeps the gates
 * of its rows: no exchange), the prover (sc_gkr_prover_*) and sc_gkr_w_round_sums / _fix_variables / _evaluate work
 * on them (sums exchanged like a pass's; fix_variables while the variables fixed stay shard-local, i.e. all of b
 * and all but the top log2(world) of c).  sc_gkr_w_to_evaluations returns, like every table of a sharded context, this rank's
 * contiguous shard of the result - whose order is b-major (the reference's), so the shard is a range of b and needs every
 * rank's rows of c: add and mul are gathered for the duration of the call (Prover::new only sums the result:
 * sc_gkr_prover_c1 gives that without it). */

/* add_i(r_i,.,.) and mul_i(r_i,.,.) of Prover::start_round (gkr-protocol/src/lib.rs:388-416)
 * straight from the gate list of layer i (2^k_i gates: type 0 = add, 1 = mul; inputs index
 * layer i+1, which has 2^k_next values); r_i has k_i entries.  Equals the reference's dense
 * predicate tables after fix_variables(r_i), without building them. */
int sc_gkr_wiring(sc_ctx* ctx, const int32_t* gate_type, const uint32_t* in0, const uint32_t* in1, size_t k_i,
                  size_t k_next, const uint64_t* r_i, sc_table** add_out, sc_table** mul_out);
/* W::to_evaluations (round_polynomial.rs:96-118): out[b*2^kc + c] = f(b,c) (the reference's order) */
int sc_gkr_w_to_evaluations(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                            const sc_table* w_c, sc_table** out);
/* W::to_univariate (round_polynomial.rs:78-90) as (H(0), H(1), H(2)): the round polynomial has
 * degree <= 2, so sc_interpolate_quadratic gives the coefficient vector the reference obtains
 * from its 4-point roots-of-unity interpolation. */
int sc_gkr_w_round_sums(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                        const sc_table* w_c, uint64_t out_e[3]);
/* W::fix_variables (round_polynomial.rs:59-76): r goes to add/mul and to w_b's variables first,
 * the rest to w_c. */
int sc_gkr_w_fix_variables(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                           const sc_table* w_c, const uint64_t* r, size_t k, sc_table** add_out,
                           sc_table** mul_out, sc_table** w_b_out, sc_table** w_c_out);
/* W::evaluate (round_polynomial.rs:48-57) */
int sc_gkr_w_evaluate(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                      const sc_table* w_c, const uint64_t* point, size_t n, uint64_t* out);
/* SumCheckProver<F, W<F>> (gkr-protocol/src/lib.rs:418-456 drives it): same contract as
 * sc_prover_*: create computes c_1, rounds in order, r_prev ignored at j = 0.  Two-phase linear-time form:
 * add and mul are streamed twice per LAYER (once for P(b) = sum_c add + mul W(c) and L(b) = sum_c add W(c), once
 * to fix b), the 2k rounds themselves run on 2^(k+1)-entry tables; round polynomials identical to the
 * reference's per-round walk. */
typedef struct sc_gkr_prover sc_gkr_prover;
int sc_gkr_prover_create(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                         const sc_table* w_c, sc_gkr_prover** out);
/* The same prover straight from the gate list of layer i (arguments as sc_gkr_wiring) and the
 * 2^k_next values W_{i+1} of the next layer: add_i / mul_i have one non-zero per gate and W is
 * linear in them, so every round visits the 2^k_i gates instead of the 4^k_next table entries,
 * and the dense tables are never built.  Round polynomials are identical to the dense prover's. */
int sc_gkr_prover_create_sparse(sc_ctx* ctx, const int32_t* gate_type, const uint32_t* in0, const uint32_t* in1,
                                size_t k_i, size_t k_next, const uint64_t* r_i, const sc_table* w_next,
                                sc_gkr_prover** out);
/* the whole 2k-round W sumcheck in one call, challenges as in sc_prove (draw == NULL: the synthetic challenger);
 * evals has 3 * (kb + kc) words, challenges kb + kc */
int sc_gkr_prove(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b, const sc_table* w_c,
                 sc_draw_fn draw, void* user, uint64_t seed_r, uint64_t* c1, uint64_t* evals, uint64_t* challenges);
int sc_gkr_prover_c1(const sc_gkr_prover* pr, uint64_t* out);
int sc_gkr_prover_round(sc_gkr_prover* pr, uint64_t r_prev, size_t j, uint64_t out_e[3]);
int sc_gkr_prover_destroy(sc_gkr_prover* pr);

/* restrict_poly (gkr-protocol/src/lib.rs:291-321): the univariate
 * q(t) = W~(l(t)) for the line l(0) = b, l(1) = c (`line`, :278-289), as dense coefficients
 * out_coeffs[0..k] (degree <= k = number of variables).  Computed as the evaluations at t = 0..k - ONE
 * pass over the table for all k + 1 points (sc_table_evaluate_many) - plus exact interpolation on the host, instead
 * of the reference's O(k 2^k) product of linear factors; needs p > k.  Sharded contexts and multi-device handles: t is
 * split by its top index bits like every table there.  The reference returns a SparsePolynomial: drop zero terms. */
int sc_table_restrict_to_line(sc_ctx* ctx, const sc_table* t, const uint64_t* b, const uint64_t* c, size_t k,
                              uint64_t* out_coeffs);

/* ---- triangle_counting::G -------------------------------------
 *   g(X,Y,Z) = f(X,Y) f(Y,Z) f(X,Z)   (triangle-counting/src/lib.rs:10-27)
 * three copies f1, f2, f3 of the adjacency MLE (2*var_len variables each before any fixing,
 * idx(i,j,nv) = (i << nv) | j, :168-172).  Variable counts of a partially fixed G follow
 * :53-67.  Sharded contexts: the prover (sc_tri_prover_*) is available - adj is this rank's rows of the
 * adjacency table; the matrix square (the n^3 work) is split across the ranks, the sumchecks on the 2^(2k)- and
 * 2^k-entry tables run replicated on every rank.  The generic trait methods (sc_tri_*) are single-rank only:
 * they serve arbitrary partially fixed states by walking all 2^(3k) evaluations like the reference, and a caller on a
 * sharded node runs them on an unsharded context of any one rank (the replicated fallback; SC_ERR_UNSUPPORTED
 * otherwise, never a silent wrong answer). */

/* G::to_evaluations (:138-165), order x outer, z inner */
int sc_tri_to_evaluations(sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len,
                          sc_table** out);
/* G::to_univariate (:120-132) as (H(0), H(1), H(2)) of the degree-2 round polynomial, for G in
 * any state, by walking all remaining evaluations like the reference */
int sc_tri_round_sums(sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len,
                      uint64_t out_e[3]);
/* G::fix_variables (:89-118) */
int sc_tri_fix_variables(sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len,
                         const uint64_t* r, size_t k, sc_table** f1_out, sc_table** f2_out, sc_table** f3_out);
/* G::evaluate (:71-87) */
int sc_tri_evaluate(sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len,
                    const uint64_t* point, size_t n, uint64_t* out);
/* Prover<F, G> for G::new_adj_matrix(adj) (:32-51): all 3*var_len rounds.  One n^3 pass
 * (P = adj-matrix squared) turns the x rounds into a product-of-two-tables sumcheck on (P, f3);
 * the y rounds run on (f1(r_x,.), Q) with Q[y] = sum_z f2(y,z) f3(r_x,z); the z rounds on
 * (f2(r_y,.), f3(r_x,.)) scaled by f1(r_x,r_y).  Same contract as sc_prover_*. */
typedef struct sc_tri_prover sc_tri_prover;
int sc_tri_prover_create(sc_ctx* ctx, const sc_table* adj, size_t var_len, sc_tri_prover** out);
/* all 3 * var_len rounds in one call, challenges as in sc_prove */
int sc_tri_prove(sc_ctx* ctx, const sc_table* adj, size_t var_len, sc_draw_fn draw, void* user, uint64_t seed_r,
                 uint64_t* c1, uint64_t* evals, uint64_t* challenges);
int sc_tri_prover_c1(const sc_tri_prover* pr, uint64_t* out);
int sc_tri_prover_round(sc_tri_prover* pr, uint64_t r_prev, size_t j, uint64_t out_e[3]);
int sc_tri_prover_destroy(sc_tri_prover* pr);

#ifdef __cplusplus
}
#endif
#endif /* SUMCHECK_HIP_H */
