/*
 * sc_oracle.c - CPU restatement of the reference's sumcheck hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library, and
 * there only as the checker / the reported CPU baseline.  The product path
 * (thaler-study_amd/csrc) never links, loads or calls it.
 *
 * The reference (montekki/thaler-study) is Rust and cannot be built in this image (no
 * cargo/rustc, arkworks crates not vendored).  Its table arithmetic lives in the
 * un-vendored crates ark-poly = "0.6" / ark-ff = "0.6" (Cargo.toml:23-24 of the
 * reference; exact patch unpinned because the root Cargo.lock is git-ignored).  Those
 * published algorithms are restated here and anchored on the reference's own call sites;
 * every function cites the file:line it follows (paths relative to /root/reference).
 *
 * Pinning: tests/test_oracle_golden.py checks this file against every known-answer
 * vector the reference's tests hold for the path (tests/golden/reference_kats.json) and
 * against an independent big-integer restatement (oracle/pyref.py).  Literal
 * round-polynomial coefficients of a table-backed run are asserted by no reference test
 * ("parity unpinned" for those literals); they are pinned through
 * the verifier identities the reference does assert.
 *
 * Representation: every uint64_t is the Montgomery residue x*2^64 mod p in [0,p) -
 * the memory word of ark-ff's Fp64<MontBackend<_,1>>.  Single-threaded, like the
 * reference.  Shapes (copies, passes) deliberately mirror the reference so that timing
 * sco_prove is a like-for-like CPU baseline ("port").
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

typedef struct {
  u64 p, p_inv_neg, r_mod_p, r2_mod_p;
} sco_field;

/* ---- field: ark-ff Fp64<MontBackend<_,1>> (R = 2^64) ---------------------------- */

int sco_field_init(u64 p, sco_field* f) {
  if (p < 3 || (p & 1) == 0) return 1;
  u64 inv = 1;
  for (int i = 0; i < 7; ++i) inv *= 2 - p * inv;
  f->p = p;
  f->p_inv_neg = (u64)0 - inv;
  u128 r = (((u128)1) << 64) % p;
  f->r_mod_p = (u64)r;
  f->r2_mod_p = (u64)((r * r) % p);
  return 0;
}

static inline u64 f_add(const sco_field* f, u64 a, u64 b) {
  u128 s = (u128)a + b;
  return (u64)(s >= f->p ? s - f->p : s);
}
static inline u64 f_sub(const sco_field* f, u64 a, u64 b) {
  return a >= b ? a - b : a + (f->p - b);
}
static inline u64 f_neg(const sco_field* f, u64 a) { return a ? f->p - a : 0; }
static inline u64 f_mul(const sco_field* f, u64 a, u64 b) {
  u128 t = (u128)a * b;
  u64 m = (u64)t * f->p_inv_neg;
  u128 mp = (u128)m * f->p;
  /* (t + mp) / 2^64 without overflowing 128 bits */
  u128 s = (t >> 64) + (mp >> 64) + ((((u128)(u64)t) + (u64)mp) >> 64);
  return (u64)(s >= f->p ? s - f->p : s);
}
static u64 f_pow(const sco_field* f, u64 a, u64 e) {
  u64 r = f->r_mod_p;
  while (e) {
    if (e & 1) r = f_mul(f, r, a);
    a = f_mul(f, a, a);
    e >>= 1;
  }
  return r;
}
static inline u64 f_inv(const sco_field* f, u64 a) { return f_pow(f, a, f->p - 2); }

u64 sco_add(const sco_field* f, u64 a, u64 b) { return f_add(f, a, b); }
u64 sco_sub(const sco_field* f, u64 a, u64 b) { return f_sub(f, a, b); }
u64 sco_mul(const sco_field* f, u64 a, u64 b) { return f_mul(f, a, b); }
u64 sco_inv(const sco_field* f, u64 a) { return f_inv(f, a); }
u64 sco_to_mont(const sco_field* f, u64 x) { return f_mul(f, x % f->p, f->r2_mod_p); }
u64 sco_from_mont(const sco_field* f, u64 m) { return f_mul(f, m, 1); }

void sco_to_mont_vec(const sco_field* f, const u64* in, size_t n, u64* out) {
  for (size_t i = 0; i < n; ++i) out[i] = sco_to_mont(f, in[i]);
}
void sco_from_mont_vec(const sco_field* f, const u64* in, size_t n, u64* out) {
  for (size_t i = 0; i < n; ++i) out[i] = sco_from_mont(f, in[i]);
}

/* ---- synthetic instance ----------- */

static inline u64 splitmix64(u64 x) {
  u64 z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* out[i] = Montgomery form of (splitmix64(seed + start + i) mod p) */
void sco_generate(const sco_field* f, u64 seed, u64 start, size_t len, u64* out) {
  long i;   /* (entries are independent: large tables are filled by all cores - the n = 28 parity tests build 2 x 2 GiB) */
#pragma omp parallel for schedule(static) if (len >= ((size_t)1 << 20))
  for (i = 0; i < (long)len; ++i) out[i] = sco_to_mont(f, splitmix64(seed + start + (u64)i));
}
u64 sco_challenge(const sco_field* f, u64 seed, u64 j) {
  return sco_to_mont(f, splitmix64(seed + j));
}

/* ---- ark_poly::DenseMultilinearExtension (LE: variable 0 = index bit 0) ---------- */

/* DenseMultilinearExtension::fix_variables as called at
 * matrix-multiplication/src/lib.rs:83,86,104,105: copy the table, for each fixed
 * variable fold adjacent pairs in place  new[b] = t[2b] + r*(t[2b+1]-t[2b]),
 * then copy out the live prefix.  out has 2^(nv-k) entries. */
void sco_mle_fix_variables(const sco_field* f, const u64* t, size_t nv, const u64* r, size_t k,
                           u64* out) {
  size_t len = (size_t)1 << nv;
  u64* poly = (u64*)malloc(len * sizeof(u64));
  memcpy(poly, t, len * sizeof(u64));
  for (size_t i = 1; i <= k; ++i) {
    u64 ri = r[i - 1];
    size_t half = (size_t)1 << (nv - i);
    for (size_t b = 0; b < half; ++b) {
      u64 left = poly[b << 1], right = poly[(b << 1) + 1];
      poly[b] = f_add(f, left, f_mul(f, ri, f_sub(f, right, left)));
    }
  }
  memcpy(out, poly, (((size_t)1) << (nv - k)) * sizeof(u64));
  free(poly);
}

/* Polynomial::evaluate of a DenseMultilinearExtension (matrix-multiplication/src/lib.rs:97-98):
 * fix all nv variables, return the single remaining entry. */
u64 sco_mle_evaluate(const sco_field* f, const u64* t, size_t nv, const u64* point) {
  u64 out;
  sco_mle_fix_variables(f, t, nv, point, nv, &out);
  return out;
}

/* DenseMultilinearExtension::relabel(a, b, k) (matrix-multiplication/src/lib.rs:82):
 * swap variables [a, a+k) with [b, b+k), i.e. swap those index-bit fields. */
void sco_mle_relabel(const u64* t, size_t nv, size_t a, size_t b, size_t k, u64* out) {
  size_t len = (size_t)1 << nv;
  size_t mask = (((size_t)1) << k) - 1;
  if (a > b) { size_t s = a; a = b; b = s; }
  for (size_t i = 0; i < len; ++i) {
    size_t fa = (i >> a) & mask, fb = (i >> b) & mask;
    size_t j = i & ~((mask << a) | (mask << b));
    j |= (fb << a) | (fa << b);
    out[j] = t[i];
  }
}

/* ---- matrix_multiplication::G ----------------------------------------------------- */

/* G::new, matrix-multiplication/src/lib.rs:77-92.  A, B: row-major 2^n x 2^n
 * (2^(2n) entries, column index in the low n bits).  fa, fb: 2^n entries each. */
void sco_g_new(const sco_field* f, size_t n, const u64* A, const u64* B, const u64* point,
               u64* fa, u64* fb) {
  size_t len = (size_t)1 << (2 * n);
  u64* tmp = (u64*)malloc(len * sizeof(u64));
  sco_mle_relabel(A, 2 * n, 0, n, n, tmp);                   /* :82 */
  sco_mle_fix_variables(f, tmp, 2 * n, point, n, fa);        /* :83 */
  sco_mle_fix_variables(f, B, 2 * n, point + n, n, fb);      /* :86 */
  free(tmp);
}

/* G::to_univariate's three running sums, matrix-multiplication/src/lib.rs:110-122:
 * e[0] = H(0), e[1] = H(1), e[2] = H(2). */
void sco_g_round_evals(const sco_field* f, const u64* a, const u64* b, size_t nv, u64 e[3]) {
  u64 two = f_add(f, f->r_mod_p, f->r_mod_p);
  e[0] = e[1] = e[2] = 0;
  size_t len = (size_t)1 << nv;
  for (size_t i = 0; i < len; ++i) {
    if (i & 1) {
      e[1] = f_add(f, e[1], f_mul(f, a[i], b[i]));
      u64 ax = f_sub(f, f_mul(f, two, a[i]), a[i - 1]);
      u64 bx = f_sub(f, f_mul(f, two, b[i]), b[i - 1]);
      e[2] = f_add(f, e[2], f_mul(f, ax, bx));
    } else {
      e[0] = f_add(f, e[0], f_mul(f, a[i], b[i]));
    }
  }
}

/* interpolate_quadratic_poly, matrix-multiplication/src/lib.rs:17-60, for the points
 * (0,e0), (1,e1), (2,e2) of :124-128.  c[d] = coefficient of X^d (Lagrange form summed
 * exactly as the reference does: three scaled basis polynomials, three divisions). */
void sco_interpolate_quadratic(const sco_field* f, const u64 e[3], u64 c[3]) {
  u64 x[3];
  x[0] = 0;
  x[1] = f->r_mod_p;
  x[2] = f_add(f, f->r_mod_p, f->r_mod_p);
  c[0] = c[1] = c[2] = 0;
  for (int i = 0; i < 3; ++i) {
    int j = (i + 1) % 3, k = (i + 2) % 3;
    u64 den = f_mul(f, f_sub(f, x[i], x[j]), f_sub(f, x[i], x[k]));
    u64 dinv = f_inv(f, den);
    u64 b0 = f_mul(f, x[j], x[k]);
    u64 b1 = f_sub(f, f_neg(f, x[j]), x[k]);
    u64 b2 = f->r_mod_p;
    c[0] = f_add(f, c[0], f_mul(f, f_mul(f, b0, e[i]), dinv));
    c[1] = f_add(f, c[1], f_mul(f, f_mul(f, b1, e[i]), dinv));
    c[2] = f_add(f, c[2], f_mul(f, f_mul(f, b2, e[i]), dinv));
  }
}

/* univariate::SparsePolynomial::evaluate for a degree-<=2 polynomial. */
u64 sco_poly2_eval(const sco_field* f, const u64 c[3], u64 x) {
  return f_add(f, c[0], f_mul(f, x, f_add(f, c[1], f_mul(f, x, c[2]))));
}

/* G::to_evaluations, matrix-multiplication/src/lib.rs:137-146: clone both tables,
 * multiply elementwise. */
void sco_g_to_evaluations(const sco_field* f, const u64* a, const u64* b, size_t nv, u64* out) {
  size_t len = (size_t)1 << nv;
  u64* bc = (u64*)malloc(len * sizeof(u64));
  memcpy(out, a, len * sizeof(u64));
  memcpy(bc, b, len * sizeof(u64));
  for (size_t i = 0; i < len; ++i) out[i] = f_mul(f, out[i], bc[i]);
  free(bc);
}

/* G::evaluate, matrix-multiplication/src/lib.rs:96-101. */
u64 sco_g_evaluate(const sco_field* f, const u64* a, const u64* b, size_t nv, const u64* point) {
  return f_mul(f, sco_mle_evaluate(f, a, nv, point), sco_mle_evaluate(f, b, nv, point));
}

/* Prover::new's claim, sum-check-protocol/src/lib.rs:88-90. */
u64 sco_prover_c1(const sco_field* f, const u64* a, const u64* b, size_t nv) {
  size_t len = (size_t)1 << nv;
  u64* ev = (u64*)malloc(len * sizeof(u64));
  sco_g_to_evaluations(f, a, b, nv, ev);
  u64 s = 0;
  for (size_t i = 0; i < len; ++i) s = f_add(f, s, ev[i]);
  free(ev);
  return s;
}

/*
 * One full interactive run: Prover::new + nv calls of Prover::round
 * (sum-check-protocol/src/lib.rs:88-112), with the Verifier's checks of :278-330 applied
 * to every message.  challenges[j] is the r_j the verifier "draws" in round j
 * (j = 0..nv-1); the prover folds with challenges[j-1] at round j >= 1, exactly like
 * the loop at matrix-multiplication/src/lib.rs:356-370.
 *
 * Outputs (any may be NULL): c1[1]; evals[3*nv] = (H(0),H(1),H(2)) per round;
 * coeffs[3*nv] = (c0,c1,c2) per round; final_eval[1] = g(r_0..r_{nv-1}) as the verifier's
 * oracle computes it from the ORIGINAL tables (:302-307).
 * Returns 0 if every verifier check passes, else 1 + index of the failing round.
 */
int sco_prove(const sco_field* f, const u64* a, const u64* b, size_t nv, const u64* challenges,
              u64* c1_out, u64* evals, u64* coeffs, u64* final_eval) {
  size_t len = (size_t)1 << nv;
  u64 c1 = sco_prover_c1(f, a, b, nv);                       /* Prover::new :89 */
  if (c1_out) *c1_out = c1;
  /* Prover owns g (moved in); the bench clones it first (mm_benchmark.rs:90) */
  u64* ga = (u64*)malloc(len * sizeof(u64));
  u64* gb = (u64*)malloc(len * sizeof(u64));
  memcpy(ga, a, len * sizeof(u64));
  memcpy(gb, b, len * sizeof(u64));
  int status = 0;
  u64 prev_c[3] = {0, 0, 0};
  size_t cur = nv;
  for (size_t j = 0; j < nv; ++j) {
    if (j != 0) {                                            /* :106-109 */
      u64 r_prev = challenges[j - 1];
      u64* na = (u64*)malloc((len >> j) * sizeof(u64));
      u64* nb = (u64*)malloc((len >> j) * sizeof(u64));
      sco_mle_fix_variables(f, ga, cur, &r_prev, 1, na);    /* G::fix_variables :103-108 */
      sco_mle_fix_variables(f, gb, cur, &r_prev, 1, nb);
      free(ga);
      free(gb);
      ga = na;
      gb = nb;
      cur -= 1;
    }
    u64 e[3], c[3];
    sco_g_round_evals(f, ga, gb, cur, e);                    /* to_univariate :111 */
    sco_interpolate_quadratic(f, e, c);
    if (evals) memcpy(evals + 3 * j, e, sizeof(e));
    if (coeffs) memcpy(coeffs + 3 * j, c, sizeof(c));
    /* Verifier::round :278-330 */
    u64 s01 = f_add(f, sco_poly2_eval(f, c, 0), sco_poly2_eval(f, c, f->r_mod_p));
    if (j == 0) {
      if (s01 != c1 && !status) status = 1 + (int)j;        /* :286-291 */
    } else {
      u64 prev = sco_poly2_eval(f, prev_c, challenges[j - 1]);
      if (prev != s01 && !status) status = 1 + (int)j;      /* :313-323 */
    }
    if (j == nv - 1) {                                       /* :298-307 */
      u64 oracle = sco_g_evaluate(f, a, b, nv, challenges);
      if (final_eval) *final_eval = oracle;
      if (sco_poly2_eval(f, c, challenges[j]) != oracle && !status) status = 1 + (int)j;
    }
    memcpy(prev_c, c, sizeof(c));
  }
  free(ga);
  free(gb);
  return status;
}

/*
 * The timed region of the reference's criterion bench, benches/mm_benchmark.rs:88-96:
 * Prover::new(g.clone()) then num_vars calls of prover.round(r_j, j) - no verifier.
 * challenges[j] is what the bench draws after round j (r_j for round j+1).
 * This is the function bench.py times as the single-core CPU baseline ("port").
 */
void sco_prover_run(const sco_field* f, const u64* a, const u64* b, size_t nv, const u64* challenges,
                    u64* c1_out, u64* evals) {
  size_t len = (size_t)1 << nv;
  u64* ga = (u64*)malloc(len * sizeof(u64));                 /* g.clone() :90 */
  u64* gb = (u64*)malloc(len * sizeof(u64));
  memcpy(ga, a, len * sizeof(u64));
  memcpy(gb, b, len * sizeof(u64));
  u64 c1 = sco_prover_c1(f, ga, gb, nv);                     /* Prover::new */
  if (c1_out) *c1_out = c1;
  size_t cur = nv;
  for (size_t j = 0; j < nv; ++j) {
    if (j != 0) {
      u64 r_prev = challenges[j - 1];
      u64* na = (u64*)malloc((len >> j) * sizeof(u64));
      u64* nb = (u64*)malloc((len >> j) * sizeof(u64));
      sco_mle_fix_variables(f, ga, cur, &r_prev, 1, na);
      sco_mle_fix_variables(f, gb, cur, &r_prev, 1, nb);
      free(ga);
      free(gb);
      ga = na;
      gb = nb;
      cur -= 1;
    }
    u64 e[3], c[3];
    sco_g_round_evals(f, ga, gb, cur, e);
    sco_interpolate_quadratic(f, e, c);
    if (evals) memcpy(evals + 3 * j, e, sizeof(e));
  }
  free(ga);
  free(gb);
}

/*
 * All-cores variant of sco_prover_run for the second, clearly labelled CPU row of BASELINE.md
 * section 4 ("CPU-ref-allT").  Same passes and copies, the element loops split over OpenMP
 * threads; modular sums are order-independent, so the outputs are identical.  Built with
 * -fopenmp; without it the pragmas are ignored and this is sco_prover_run.
 */
static void mt_fix_one(const sco_field* f, const u64* t, size_t nv, u64 r, u64* out) {
  size_t half = (size_t)1 << (nv - 1);
  long b;
#pragma omp parallel for schedule(static)
  for (b = 0; b < (long)half; ++b) {
    u64 left = t[(size_t)b << 1], right = t[((size_t)b << 1) + 1];
    out[b] = f_add(f, left, f_mul(f, r, f_sub(f, right, left)));
  }
}
static void mt_round_evals(const sco_field* f, const u64* a, const u64* b, size_t nv, u64 e[3]) {
  size_t pairs = (size_t)1 << (nv - 1);
  u64 e0 = 0, e1 = 0, e2 = 0;
#pragma omp parallel
  {
    u64 l0 = 0, l1 = 0, l2 = 0;
    long q;
#pragma omp for schedule(static) nowait
    for (q = 0; q < (long)pairs; ++q) {
      size_t i = ((size_t)q << 1) + 1;
      l0 = f_add(f, l0, f_mul(f, a[i - 1], b[i - 1]));
      l1 = f_add(f, l1, f_mul(f, a[i], b[i]));
      u64 ax = f_sub(f, f_add(f, a[i], a[i]), a[i - 1]), bx = f_sub(f, f_add(f, b[i], b[i]), b[i - 1]);
      l2 = f_add(f, l2, f_mul(f, ax, bx));
    }
#pragma omp critical
    {
      e0 = f_add(f, e0, l0);
      e1 = f_add(f, e1, l1);
      e2 = f_add(f, e2, l2);
    }
  }
  e[0] = e0; e[1] = e1; e[2] = e2;
}
void sco_prover_run_mt(const sco_field* f, const u64* a, const u64* b, size_t nv, const u64* challenges,
                       u64* c1_out, u64* evals) {
  size_t len = (size_t)1 << nv;
  u64* ga = (u64*)malloc(len * sizeof(u64));
  u64* gb = (u64*)malloc(len * sizeof(u64));
  long i;
  u64 c1 = 0;
#pragma omp parallel
  {
    u64 l = 0;
#pragma omp for schedule(static) nowait
    for (i = 0; i < (long)len; ++i) {
      ga[i] = a[i];
      gb[i] = b[i];
      l = f_add(f, l, f_mul(f, a[i], b[i]));
    }
#pragma omp critical
    c1 = f_add(f, c1, l);
  }
  if (c1_out) *c1_out = c1;
  size_t cur = nv;
  for (size_t j = 0; j < nv; ++j) {
    if (j != 0) {
      u64* na = (u64*)malloc((len >> j) * sizeof(u64));
      u64* nb = (u64*)malloc((len >> j) * sizeof(u64));
      mt_fix_one(f, ga, cur, challenges[j - 1], na);
      mt_fix_one(f, gb, cur, challenges[j - 1], nb);
      free(ga);
      free(gb);
      ga = na;
      gb = nb;
      cur -= 1;
    }
    u64 e[3], c[3];
    mt_round_evals(f, ga, gb, cur, e);
    sco_interpolate_quadratic(f, e, c);
    if (evals) memcpy(evals + 3 * j, e, sizeof(e));
  }
  free(ga);
  free(gb);
}

/* ---- multilinear-extensions crate (BE: r[0] <-> index MSB) ------------------------ */

/* vsbw_multilinear_from_evaluations, multilinear-extensions/src/lib.rs:6-24. */
u64 sco_vsbw(const sco_field* f, const u64* evals, const u64* r, size_t n) {
  size_t len = 1;
  u64* tab = (u64*)malloc(sizeof(u64));
  tab[0] = f->r_mod_p;
  for (size_t j = 0; j < n; ++j) {
    u64* nt = (u64*)malloc(2 * len * sizeof(u64));           /* :10 re-allocates */
    u64 one_minus = f_sub(f, f->r_mod_p, r[j]);
    for (size_t i = 0; i < len; ++i) {
      nt[2 * i] = f_mul(f, tab[i], one_minus);               /* :13 */
      nt[2 * i + 1] = f_mul(f, tab[i], r[j]);                /* :14 */
    }
    free(tab);
    tab = nt;
    len *= 2;
  }
  u64 acc = 0;
  for (size_t i = 0; i < len; ++i) acc = f_add(f, acc, f_mul(f, tab[i], evals[i]));  /* :20-23 */
  free(tab);
  return acc;
}

/* cti_multilinear_from_evaluations + lagrange_basis_poly_at,
 * multilinear-extensions/src/lib.rs:29-60. */
u64 sco_cti(const sco_field* f, const u64* evals, const u64* r, size_t n) {
  size_t len = (size_t)1 << n;
  u64 one = f->r_mod_p;
  u64 res = 0;
  for (size_t i = 0; i < len; ++i) {
    u64 basis = one;
    for (size_t idx = 0; idx < n; ++idx) {
      size_t j = n - 1 - idx;                                /* :37 (0..len).rev() */
      u64 w = (i & ((size_t)1 << j)) ? one : 0;              /* :38-40 */
      u64 term = f_add(f, f_mul(f, r[idx], w),
                       f_mul(f, f_sub(f, one, r[idx]), f_sub(f, one, w)));  /* :55 */
      basis = f_mul(f, basis, term);
    }
    res = f_add(f, res, f_mul(f, evals[i], basis));          /* :44 */
  }
  return res;
}

/* BE partial fix (stride-half pairing): fixes the k leading variables of the
 * multilinear-extensions ordering, i.e. index bits nv-1, nv-2, ...; the streaming
 * equivalent of truncating vsbw's table.  out has 2^(nv-k) entries. */
void sco_mle_fix_variables_be(const sco_field* f, const u64* t, size_t nv, const u64* r, size_t k,
                              u64* out) {
  size_t len = (size_t)1 << nv;
  u64* poly = (u64*)malloc(len * sizeof(u64));
  memcpy(poly, t, len * sizeof(u64));
  for (size_t i = 0; i < k; ++i) {
    size_t half = len >> (i + 1);
    for (size_t b = 0; b < half; ++b) {
      u64 lo = poly[b], hi = poly[b + half];
      poly[b] = f_add(f, lo, f_mul(f, r[i], f_sub(f, hi, lo)));
    }
  }
  memcpy(out, poly, (len >> k) * sizeof(u64));
  free(poly);
}

/* ---- sharded partial sums (checker for the multi-GPU path) ------------------------ */

/* 3x3 grid of sums used by the two-variables-per-pass schedule, in the evaluation basis
 * {0, 1, inf} per variable (inf = leading coefficient, t1 - t0):
 * S[3*u+v] = sum over quads q of a(u,v)*b(u,v), where a(u,v) is the bilinear extension
 * of (t[4q], t[4q+1], t[4q+2], t[4q+3]) in (index bit 0, index bit 1).
 * Round j:   H(u) = S[u][0] + S[u][1], u in {0,1,inf};  H(2) = 2H(1) - H(0) + 2H(inf).
 * Round j+1 after challenge r: column v is the quadratic
 *   S[0][v] + r (S[1][v] - S[0][v] - S[inf][v]) + r^2 S[inf][v]. */
void sco_g_grid_sums(const sco_field* f, const u64* a, const u64* b, size_t nv, u64 S[9]) {
  for (int i = 0; i < 9; ++i) S[i] = 0;
  size_t quads = (size_t)1 << (nv - 2);
  for (size_t q = 0; q < quads; ++q) {
    u64 av[3][3], bv[3][3];
    for (int tsel = 0; tsel < 2; ++tsel) {
      const u64* t = tsel ? b : a;
      u64(*o)[3] = tsel ? bv : av;
      u64 g[3][2]; /* g[u][bit1] */
      for (int h = 0; h < 2; ++h) {
        u64 t0 = t[4 * q + 2 * h], t1 = t[4 * q + 2 * h + 1];
        g[0][h] = t0;
        g[1][h] = t1;
        g[2][h] = f_sub(f, t1, t0);
      }
      for (int u = 0; u < 3; ++u) {
        o[u][0] = g[u][0];
        o[u][1] = g[u][1];
        o[u][2] = f_sub(f, g[u][1], g[u][0]);
      }
    }
    for (int u = 0; u < 3; ++u)
      for (int v = 0; v < 3; ++v)
        S[3 * u + v] = f_add(f, S[3 * u + v], f_mul(f, av[u][v], bv[u][v]));
  }
}

/* The same grid for k = 1, 2 or 3 rounds per pass: S has 3^k cells, the first variable (index
 * bit 0) on the slowest axis - S[(3u + v)*3 + w] for k = 3.  Cell c = (c_0..c_{k-1}), c_d in
 * {0,1,inf}, holds the sum over blocks of 2^k entries of ext_a(c) * ext_b(c), where ext takes
 * the entry with bit d = c_d for c_d in {0,1} and the difference (bit d = 1) - (bit d = 0) for
 * c_d = inf.  Written for clarity, not speed. */
static u64 grid_ext(const sco_field* f, const u64* t, int k, const int* c, int d, size_t idx) {
  if (d == k) return t[idx];
  if (c[d] < 2) return grid_ext(f, t, k, c, d + 1, idx | ((size_t)c[d] << d));
  return f_sub(f, grid_ext(f, t, k, c, d + 1, idx | ((size_t)1 << d)), grid_ext(f, t, k, c, d + 1, idx));
}
void sco_g_gridk_sums(const sco_field* f, const u64* a, const u64* b, size_t nv, int k, u64* S) {
  int cells = 1;
  for (int i = 0; i < k; ++i) cells *= 3;
  for (int i = 0; i < cells; ++i) S[i] = 0;
  size_t blocks = (size_t)1 << (nv - (size_t)k);
  for (size_t q = 0; q < blocks; ++q) {
    for (int cell = 0; cell < cells; ++cell) {
      int c[8], rem = cell;   /* k <= 5 in practice (243 cells) */
      for (int d = k - 1; d >= 0; --d) {
        c[d] = rem % 3;
        rem /= 3;
      }
      u64 ea = grid_ext(f, a + (q << k), k, c, 0, 0), eb = grid_ext(f, b + (q << k), k, c, 0, 0);
      S[cell] = f_add(f, S[cell], f_mul(f, ea, eb));
    }
  }
}

/* ---- gkr_protocol::round_polynomial::W ----------------
 * f(b,c) = add(b,c) (W(b) + W(c)) + mul(b,c) W(b) W(c); add/mul have 2k variables and are
 * indexed (c << k) | b; w_b, w_c have k variables.  kb = current number of variables of
 * w_b, kc of w_c (they shrink as the sumcheck fixes variables, b first). */

/* W::to_evaluations, gkr-protocol/src/round_polynomial.rs:96-118 (push order b outer, c inner) */
void sco_w_to_evaluations(const sco_field* f, const u64* add, const u64* mul, const u64* w_b,
                          size_t kb, const u64* w_c, size_t kc, u64* out) {
  size_t nb = (size_t)1 << kb, nc = (size_t)1 << kc, o = 0;
  for (size_t b = 0; b < nb; ++b)
    for (size_t c = 0; c < nc; ++c) {
      size_t bc = (c << kb) | b;
      u64 s = f_add(f, w_b[b], w_c[c]), pr = f_mul(f, w_b[b], w_c[c]);
      out[o++] = f_add(f, f_mul(f, add[bc], s), f_mul(f, mul[bc], pr));
    }
}

/* W::fix_variables for ONE variable, gkr-protocol/src/round_polynomial.rs:59-76: the point goes
 * to add/mul and to w_b while it still has variables, else to w_c.  Outputs are malloc'ed. */
static void w_fix_one(const sco_field* f, u64** add, u64** mul, u64** w_b, size_t* kb, u64** w_c,
                      size_t* kc, u64 r) {
  size_t nv = *kb + *kc;
  u64* na = (u64*)malloc((((size_t)1) << (nv - 1)) * sizeof(u64));
  u64* nm = (u64*)malloc((((size_t)1) << (nv - 1)) * sizeof(u64));
  sco_mle_fix_variables(f, *add, nv, &r, 1, na);
  sco_mle_fix_variables(f, *mul, nv, &r, 1, nm);
  free(*add);
  free(*mul);
  *add = na;
  *mul = nm;
  if (*kb > 0) {
    u64* nw = (u64*)malloc((((size_t)1) << (*kb - 1)) * sizeof(u64));
    sco_mle_fix_variables(f, *w_b, *kb, &r, 1, nw);
    free(*w_b);
    *w_b = nw;
    *kb -= 1;
  } else {
    u64* nw = (u64*)malloc((((size_t)1) << (*kc - 1)) * sizeof(u64));
    sco_mle_fix_variables(f, *w_c, *kc, &r, 1, nw);
    free(*w_c);
    *w_c = nw;
    *kc -= 1;
  }
}

static u64* dup_words(const u64* src, size_t n) {
  u64* d = (u64*)malloc(n * sizeof(u64));
  memcpy(d, src, n * sizeof(u64));
  return d;
}

/* H(0), H(1), H(2) of W::to_univariate (gkr-protocol/src/round_polynomial.rs:78-90).  The
 * reference evaluates on the size-4 roots-of-unity domain and interpolates; the round
 * polynomial has degree <= 2, so the integer points 0, 1, 2 determine the same coefficient
 * vector (oracle/pyref.py w_to_univariate_domain restates the domain form and checks this). */
void sco_w_round_evals(const sco_field* f, const u64* add, const u64* mul, const u64* w_b, size_t kb,
                       const u64* w_c, size_t kc, u64 e[3]) {
  u64 xs[3];
  xs[0] = 0;
  xs[1] = f->r_mod_p;
  xs[2] = f_add(f, f->r_mod_p, f->r_mod_p);
  size_t nv = kb + kc;
  for (int i = 0; i < 3; ++i) {
    u64 *a = dup_words(add, (size_t)1 << nv), *m = dup_words(mul, (size_t)1 << nv);
    u64 *wb = dup_words(w_b, (size_t)1 << kb), *wc = dup_words(w_c, (size_t)1 << kc);
    size_t b = kb, c = kc;
    w_fix_one(f, &a, &m, &wb, &b, &wc, &c, xs[i]);
    size_t len = (size_t)1 << (b + c);
    u64* ev = (u64*)malloc(len * sizeof(u64));
    sco_w_to_evaluations(f, a, m, wb, b, wc, c, ev);
    u64 s = 0;
    for (size_t t = 0; t < len; ++t) s = f_add(f, s, ev[t]);
    e[i] = s;
    free(ev); free(a); free(m); free(wb); free(wc);
  }
}

/* W::evaluate, gkr-protocol/src/round_polynomial.rs:48-57 */
u64 sco_w_evaluate(const sco_field* f, const u64* add, const u64* mul, const u64* w_b, size_t kb,
                   const u64* w_c, size_t kc, const u64* point) {
  u64 ae = sco_mle_evaluate(f, add, kb + kc, point), me = sco_mle_evaluate(f, mul, kb + kc, point);
  u64 wb = sco_mle_evaluate(f, w_b, kb, point), wc = sco_mle_evaluate(f, w_c, kc, point + kb);
  return f_add(f, f_mul(f, ae, f_add(f, wb, wc)), f_mul(f, me, f_mul(f, wb, wc)));
}

/* Prover::new + all rounds on W (sum-check-protocol/src/lib.rs:88-112) with the verifier's
 * identities; k = variables of w_b = w_c at the start.  evals: 3*(2k) words. */
int sco_w_prove(const sco_field* f, const u64* add, const u64* mul, const u64* w_b, const u64* w_c,
                size_t k, const u64* challenges, u64* c1_out, u64* evals, u64* final_eval) {
  size_t nv = 2 * k, len = (size_t)1 << nv;
  u64* ev = (u64*)malloc(len * sizeof(u64));
  sco_w_to_evaluations(f, add, mul, w_b, k, w_c, k, ev);
  u64 c1 = 0;
  for (size_t t = 0; t < len; ++t) c1 = f_add(f, c1, ev[t]);
  free(ev);
  if (c1_out) *c1_out = c1;
  u64 *a = dup_words(add, len), *m = dup_words(mul, len);
  u64 *wb = dup_words(w_b, (size_t)1 << k), *wc = dup_words(w_c, (size_t)1 << k);
  size_t kb = k, kc = k;
  int status = 0;
  u64 claim = c1;
  for (size_t j = 0; j < nv; ++j) {
    if (j) w_fix_one(f, &a, &m, &wb, &kb, &wc, &kc, challenges[j - 1]);
    u64 e[3], c[3];
    sco_w_round_evals(f, a, m, wb, kb, wc, kc, e);
    sco_interpolate_quadratic(f, e, c);
    if (evals) memcpy(evals + 3 * j, e, sizeof(e));
    if (f_add(f, e[0], e[1]) != claim && !status) status = 1 + (int)j;
    claim = sco_poly2_eval(f, c, challenges[j]);
  }
  u64 fin = sco_w_evaluate(f, add, mul, w_b, k, w_c, k, challenges);
  if (final_eval) *final_eval = fin;
  if (claim != fin && !status) status = 1 + (int)nv;
  free(a); free(m); free(wb); free(wc);
  return status;
}

/* add_i(r_i, b, c), mul_i(r_i, b, c) of Prover::start_round, gkr-protocol/src/lib.rs:388-416:
 * dense 0/1 tables over (a, b, c) with index ((c << k_next) | b) << k_i | a, then
 * fix_variables(r_i) over the k_i low variables.  gate_type[a]: 0 add, 1 mul. */
void sco_wiring_fixed(const sco_field* f, const int* gate_type, const uint32_t* in0, const uint32_t* in1,
                      size_t k_i, size_t k_next, const u64* r_i, u64* add_out, u64* mul_out) {
  size_t na = (size_t)1 << k_i, nn = (size_t)1 << k_next, len = na * nn * nn;
  u64* add_t = (u64*)calloc(len, sizeof(u64));
  u64* mul_t = (u64*)calloc(len, sizeof(u64));
  for (size_t c = 0; c < nn; ++c)
    for (size_t b = 0; b < nn; ++b)
      for (size_t a = 0; a < na; ++a) {
        size_t idx = (((c << k_next) | b) << k_i) | a;
        if (in0[a] == b && in1[a] == c) {
          if (gate_type[a] == 0) add_t[idx] = f->r_mod_p;
          else mul_t[idx] = f->r_mod_p;
        }
      }
  sco_mle_fix_variables(f, add_t, k_i + 2 * k_next, r_i, k_i, add_out);
  sco_mle_fix_variables(f, mul_t, k_i + 2 * k_next, r_i, k_i, mul_out);
  free(add_t);
  free(mul_t);
}

/* ---- triangle_counting::G ------------------------------------
 * g(X,Y,Z) = f(X,Y) f(Y,Z) f(X,Z) as three copies of the adjacency MLE; idx(i,j,nv) = (i<<nv)|j
 * (triangle-counting/src/lib.rs:168-172).  n1, n2, n3 = current variable counts of the copies,
 * var_len = k. */
static void tri_counts(size_t n1, size_t n2, size_t n3, size_t k, size_t* xv, size_t* yv, size_t* zv) {
  *xv = n1 > k ? n1 - k : 0;                                  /* :53-55 */
  *yv = n2 > k ? n2 - k : 0;                                  /* :57-59 */
  *zv = n3 < k ? n3 : k;                                      /* :61-67 */
}

/* G::to_evaluations, triangle-counting/src/lib.rs:138-165 */
void sco_tri_to_evaluations(const sco_field* f, const u64* f1, size_t n1, const u64* f2, size_t n2,
                            const u64* f3, size_t n3, size_t k, u64* out) {
  size_t xv, yv, zv, o = 0;
  tri_counts(n1, n2, n3, k, &xv, &yv, &zv);
  for (size_t x = 0; x < ((size_t)1 << xv); ++x)
    for (size_t y = 0; y < ((size_t)1 << yv); ++y)
      for (size_t z = 0; z < ((size_t)1 << zv); ++z)
        out[o++] = f_mul(f, f_mul(f, f1[(y << xv) | x], f2[(z << yv) | y]), f3[(z << xv) | x]);
}

/* G::fix_variables for ONE variable, triangle-counting/src/lib.rs:89-118 */
static void tri_fix_one(const sco_field* f, u64** f1, size_t* n1, u64** f2, size_t* n2, u64** f3, size_t* n3,
                        size_t k, u64 r) {
  size_t xv, yv, zv;
  tri_counts(*n1, *n2, *n3, k, &xv, &yv, &zv);
  /* x_y_point = pp[..min(xv+yv,1)], y_z_point = pp[xv..], x_z_point = pp[..min(xv,1)] ++ pp[xv+yv..] */
  int to1 = (xv + yv) >= 1, to2 = xv == 0, to3 = (xv >= 1) || (xv + yv == 0);
  u64** tabs[3] = {f1, f2, f3};
  size_t* nvs[3] = {n1, n2, n3};
  int use[3] = {to1, to2, to3};
  for (int t = 0; t < 3; ++t) {
    if (!use[t]) continue;
    u64* nt = (u64*)malloc((((size_t)1) << (*nvs[t] - 1)) * sizeof(u64));
    sco_mle_fix_variables(f, *tabs[t], *nvs[t], &r, 1, nt);
    free(*tabs[t]);
    *tabs[t] = nt;
    *nvs[t] -= 1;
  }
}

void sco_tri_round_evals(const sco_field* f, const u64* f1, size_t n1, const u64* f2, size_t n2, const u64* f3,
                         size_t n3, size_t k, u64 e[3]) {
  u64 xs[3];
  xs[0] = 0;
  xs[1] = f->r_mod_p;
  xs[2] = f_add(f, f->r_mod_p, f->r_mod_p);
  for (int i = 0; i < 3; ++i) {
    u64 *a = dup_words(f1, (size_t)1 << n1), *b = dup_words(f2, (size_t)1 << n2), *c = dup_words(f3, (size_t)1 << n3);
    size_t m1 = n1, m2 = n2, m3 = n3, xv, yv, zv;
    tri_fix_one(f, &a, &m1, &b, &m2, &c, &m3, k, xs[i]);
    tri_counts(m1, m2, m3, k, &xv, &yv, &zv);
    size_t len = (size_t)1 << (xv + yv + zv);
    u64* ev = (u64*)malloc(len * sizeof(u64));
    sco_tri_to_evaluations(f, a, m1, b, m2, c, m3, k, ev);
    u64 s = 0;
    for (size_t t = 0; t < len; ++t) s = f_add(f, s, ev[t]);
    e[i] = s;
    free(ev); free(a); free(b); free(c);
  }
}

/* G::evaluate, triangle-counting/src/lib.rs:71-87, on the ORIGINAL polynomial (three 2k-variable copies) */
u64 sco_tri_evaluate(const sco_field* f, const u64* adj, size_t k, const u64* point) {
  u64* xz = (u64*)malloc(2 * k * sizeof(u64));
  memcpy(xz, point, k * sizeof(u64));
  memcpy(xz + k, point + 2 * k, k * sizeof(u64));
  u64 v = f_mul(f, f_mul(f, sco_mle_evaluate(f, adj, 2 * k, point), sco_mle_evaluate(f, adj, 2 * k, point + k)),
                sco_mle_evaluate(f, adj, 2 * k, xz));
  free(xz);
  return v;
}

/* Prover::new + all 3k rounds on G::new_adj_matrix (triangle-counting/src/lib.rs:32-51) */
int sco_tri_prove(const sco_field* f, const u64* adj, size_t k, const u64* challenges, u64* c1_out, u64* evals,
                  u64* final_eval) {
  size_t len = (size_t)1 << (2 * k), nv = 3 * k;
  u64 *a = dup_words(adj, len), *b = dup_words(adj, len), *c = dup_words(adj, len);
  size_t n1 = 2 * k, n2 = 2 * k, n3 = 2 * k;
  u64* ev = (u64*)malloc((((size_t)1) << nv) * sizeof(u64));
  sco_tri_to_evaluations(f, a, n1, b, n2, c, n3, k, ev);
  u64 c1 = 0;
  for (size_t t = 0; t < ((size_t)1 << nv); ++t) c1 = f_add(f, c1, ev[t]);
  free(ev);
  if (c1_out) *c1_out = c1;
  int status = 0;
  u64 claim = c1;
  for (size_t j = 0; j < nv; ++j) {
    if (j) tri_fix_one(f, &a, &n1, &b, &n2, &c, &n3, k, challenges[j - 1]);
    u64 e[3], co[3];
    sco_tri_round_evals(f, a, n1, b, n2, c, n3, k, e);
    sco_interpolate_quadratic(f, e, co);
    if (evals) memcpy(evals + 3 * j, e, sizeof(e));
    if (f_add(f, e[0], e[1]) != claim && !status) status = 1 + (int)j;
    claim = sco_poly2_eval(f, co, challenges[j]);
  }
  u64 fin = sco_tri_evaluate(f, adj, k, challenges);
  if (final_eval) *final_eval = fin;
  if (claim != fin && !status) status = 1 + (int)nv;
  free(a); free(b); free(c);
  return status;
}
