// Part of kernels.hpp (included there, in order): gkr_protocol::round_polynomial::W.
#pragma once

namespace sc {

// ------------------------------------------------------------------------------------
// gkr_protocol::round_polynomial::W (gkr-protocol/src/round_polynomial.rs:23-119):
//   f(b,c) = add(b,c) (W(b) + W(c)) + mul(b,c) W(b) W(c),  add/mul indexed (c << kb) | b.
// The variable being summed lives in ONE of the two small tables: `V` (w_b while it still
// has variables, else w_c); the other contributes one value `y` per pair, taken from `Fx`
// (w_c indexed by the high bits, or the single remaining entry of w_b).  The formula is
// symmetric in the two small tables, so one kernel serves both phases.

// Round sums H(0), H(1), H(inf) over pairs (2q, 2q+1) of add/mul.  Streams add and mul
// (coalesced 16-byte pieces), gathers the matching piece of V (index = low bits: also
// coalesced) and one word of Fx per pair (broadcast within a row).  Wiring tables are mostly
// zero: pairs whose four add/mul words are all zero are skipped.
template <class F>
__global__ void __launch_bounds__(kBlock)
gkr_sums_kernel(F f, const u64* __restrict__ add, const u64* __restrict__ mul, const u64* __restrict__ V, int logV,
                const u64* __restrict__ Fx, size_t n_pieces, PassOut out) {
  __shared__ u64 lds[(kBlock / kWave) * 3];
  __shared__ int lds_flag;
  const ull2* __restrict__ Ap = reinterpret_cast<const ull2*>(add);
  const ull2* __restrict__ Mp = reinterpret_cast<const ull2*>(mul);
  const ull2* __restrict__ Vp = reinterpret_cast<const ull2*>(V);
  const size_t vmask = (((size_t)1 << logV) >> 1) - 1;  // pieces of V minus one
  typename F::Acc acc[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) f.acc_zero(acc[s]);
  for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < n_pieces; q += (size_t)gridDim.x * kBlock) {
    const ull2 a = Ap[q], m = Mp[q];
    if ((a.x | a.y | m.x | m.y) == 0) continue;
    const ull2 x = Vp[q & vmask];
    const u64 y = Fx[(2 * q) >> logV];
    const u64 dx = f.sub(x.y, x.x);
    f.acc_mac(acc[0], a.x, f.add(x.x, y));
    f.acc_mac(acc[0], m.x, f.mul(x.x, y));
    f.acc_mac(acc[1], a.y, f.add(x.y, y));
    f.acc_mac(acc[1], m.y, f.mul(x.y, y));
    f.acc_mac(acc[2], f.sub(a.y, a.x), dx);
    f.acc_mac(acc[2], f.sub(m.y, m.x), f.mul(dx, y));
  }
  u64 res[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) res[s] = f.acc_get(acc[s]);
  block_reduce<F, 3>(f, res, lds);
  finish_pass<F, 3>(f, out, res[0], &lds_flag);
}

// W::to_evaluations (round_polynomial.rs:96-118): out[b * 2^kc + c] = f(b, c) - the
// reference pushes with b outer and c inner while it READS the tables at (c << kb) | b.
template <class F>
__global__ void __launch_bounds__(kBlock)
gkr_to_evaluations_kernel(F f, const u64* __restrict__ add, const u64* __restrict__ mul, const u64* __restrict__ w_b,
                          int kb, const u64* __restrict__ w_c, int kc, size_t o_begin, size_t o_count, u64* __restrict__ out) {
  // entries [o_begin, o_begin + o_count) of the b-major result (everything unsharded; a rank's shard of the OUTPUT otherwise)
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < o_count; i += (size_t)gridDim.x * kBlock) {
    const size_t o = o_begin + i;
    const size_t b = o >> kc, c = o & (((size_t)1 << kc) - 1);
    const size_t bc = (c << kb) | b;
    const u64 wb = w_b[b], wc = w_c[c];
    out[i] = f.add(f.mul(add[bc], f.add(wb, wc)), f.mul(mul[bc], f.mul(wb, wc)));
  }
}

// add_i(r_i, b, c) / mul_i(r_i, b, c) without the dense 2^(k_i + 2 k_next) predicate table of
// Prover::start_round (gkr-protocol/src/lib.rs:388-416): gate a contributes eq(r_i, a) at
// (in1[a] << k_next) | in0[a] of the table of its type.  Gates sharing a target are summed
// with a compare-and-swap loop (there is no modular atomic add).  Outputs start zeroed.
template <class F>
__global__ void __launch_bounds__(kBlock)
gkr_wiring_scatter_kernel(F f, const u64* __restrict__ eq, const int* __restrict__ gate_type,
                          const unsigned* __restrict__ in0, const unsigned* __restrict__ in1, size_t n_gates, int k_next,
                          unsigned row_lo, unsigned rows, u64* __restrict__ add_out, u64* __restrict__ mul_out) {
  // the outputs hold rows [row_lo, row_lo + rows) of c (all of them unsharded; a rank's shard otherwise)
  for (size_t a = (size_t)blockIdx.x * kBlock + threadIdx.x; a < n_gates; a += (size_t)gridDim.x * kBlock) {
    const unsigned c = in1[a] - row_lo;
    if (c >= rows) continue;
    u64* slot = (gate_type[a] == 0 ? add_out : mul_out) + (((size_t)c << k_next) | in0[a]);
    const u64 w = eq[a];
    unsigned long long old = __hip_atomic_load((unsigned long long*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
      const unsigned long long want = f.add((u64)old, w);
      if (__hip_atomic_compare_exchange_strong((unsigned long long*)slot, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT))
        break;
    }
  }
}

// ------------------------------------------------------------------------------------
// Two-phase form of the W sumcheck (the linear-time GKR prover of Thaler's book, section 4.6.5 / "Libra"):
// with the c variables summed out,
//     sum_c f(b, c) = W(b) * P(b) + L(b),   P(b) = sum_c add(b,c) + mul(b,c) W(c),   L(b) = sum_c add(b,c) W(c)
// so the rounds over the b variables are a product sumcheck on the 2^kb-entry tables (P, W_b) plus a linear
// one on L, and once b is fixed at r_b, with w* = W(r_b),
//     f(r_b, c) = W(c) * Q(c) + w* add(r_b, c),   Q(c) = add(r_b, c) + w* mul(r_b, c)
// - the same shape on 2^kc-entry tables.  Both are exact identities of the polynomial the reference sums
// (gkr-protocol/src/round_polynomial.rs:78-90 walks all 4^k evaluations four times per round), so every
// round polynomial is the reference's.  add and mul are read twice per LAYER (once for P and L, once to
// fix b) instead of twice per round.

// P[b], L[b] as above for one chunk of rows (= values of c) per blockIdx.y; index of add/mul = c * M + b.
// Lanes own 16-byte pieces of b: coalesced.  w[c] = W(c).
template <class F, bool NT, int PW>
__global__ void __launch_bounds__(kBlock)
gkr_phase1_kernel(F f, const u64* __restrict__ add, const u64* __restrict__ mul, const u64* __restrict__ w, size_t rows,
                  size_t rows_per_chunk, size_t M, u64* __restrict__ partialP, u64* __restrict__ partialL) {
  constexpr int RIF = 2;   // rows in flight per thread: 2 rows x 2 tables x PW 16-byte loads (access pattern: see coldot_kernel)
  const ull2* __restrict__ Ap = reinterpret_cast<const ull2*>(add);
  const ull2* __restrict__ Mp = reinterpret_cast<const ull2*>(mul);
  ull2* __restrict__ Pp = reinterpret_cast<ull2*>(partialP);
  ull2* __restrict__ Lp = reinterpret_cast<ull2*>(partialL);
  const size_t mp = M / 2;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const size_t i0 = (size_t)blockIdx.y * rows_per_chunk;
  const size_t i1 = (i0 + rows_per_chunk < rows) ? i0 + rows_per_chunk : rows;
  const size_t n_spans = (mp + (size_t)kWave * PW - 1) / ((size_t)kWave * PW);
  const ull2 zero = {0, 0};
  for (size_t span = (size_t)blockIdx.x * (kBlock / kWave) + wave; span < n_spans; span += (size_t)gridDim.x * (kBlock / kWave)) {
    const size_t pc0 = span * kWave * PW + lane;
    typename F::Acc p0[PW], p1[PW], l0[PW], l1[PW];
    u64 s0[PW], s1[PW];   // sum_c add: plain modular adds
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      f.acc_zero(p0[j]); f.acc_zero(p1[j]); f.acc_zero(l0[j]); f.acc_zero(l1[j]);
      s0[j] = s1[j] = 0;
    }
    // wiring tables are mostly zero (one non-zero per gate in 4^k entries), and a piece whose four words are all zero
    // adds nothing: where a whole wave sees zeros the ~70 instructions of a piece are skipped (gkr_sums_kernel does
    // the same)
    auto take = [&](int j, const ull2& a, const ull2& m, u64 wi) {
      if ((a.x | a.y | m.x | m.y) != 0) {
        s0[j] = f.add(s0[j], a.x); s1[j] = f.add(s1[j], a.y);
        f.acc_mac(p0[j], m.x, wi); f.acc_mac(p1[j], m.y, wi);
        f.acc_mac(l0[j], a.x, wi); f.acc_mac(l1[j], a.y, wi);
      }
    };
    size_t i = i0;
    for (; i + RIF <= i1; i += RIF) {   // fixed-count inner loop (acc_mac is inline asm: no runtime unrolling)
      ull2 a[RIF][PW], m[RIF][PW];
#pragma unroll
      for (int k = 0; k < RIF; ++k)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
          const size_t pc = pc0 + (size_t)j * kWave;
          const bool in = PW == 1 || pc < mp;
          a[k][j] = in ? ld16<NT>(Ap + (i + k) * mp + (pc < mp ? pc : 0)) : zero;
          m[k][j] = in ? ld16<NT>(Mp + (i + k) * mp + (pc < mp ? pc : 0)) : zero;
        }
#pragma unroll
      for (int k = 0; k < RIF; ++k) {
        const u64 wi = w[i + k];
#pragma unroll
        for (int j = 0; j < PW; ++j) take(j, a[k][j], m[k][j], wi);
      }
    }
    for (; i < i1; ++i) {
      const u64 wi = w[i];
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        const size_t pc = pc0 + (size_t)j * kWave;
        if (pc < mp) take(j, ld16<NT>(Ap + i * mp + pc), ld16<NT>(Mp + i * mp + pc), wi);
      }
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const size_t pc = pc0 + (size_t)j * kWave;
      if (pc < mp) {
        Pp[(size_t)blockIdx.y * mp + pc] = ull2{f.add(s0[j], f.acc_get(p0[j])), f.add(s1[j], f.acc_get(p1[j]))};
        Lp[(size_t)blockIdx.y * mp + pc] = ull2{f.acc_get(l0[j]), f.acc_get(l1[j])};
      }
    }
  }
}

// The phase's pair of tables for the product prover, with the linear term riding on one more variable s
// (the highest index bit, never reached by the k rounds that are run):
//   TA = [ X + sY * Y | sZ * Z ],  TB = [ V | 1 ]       (n entries each half)
// phase b: X = P, sY = 0, Z = L, sZ = 1, V = W_b;  phase c: X = add_r, Y = mul_r, sY = w*, Z = add_r, sZ = w*, V = W_c.
template <class F>
__global__ void __launch_bounds__(kBlock)
gkr_combine_kernel(F f, const u64* __restrict__ X, const u64* __restrict__ Y, u64 sY, const u64* __restrict__ Z, u64 sZ,
                   const u64* __restrict__ V, size_t n, u64* __restrict__ TA, u64* __restrict__ TB) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    TA[i] = f.add(X[i], f.mul(sY, Y[i]));
    TA[n + i] = f.mul(sZ, Z[i]);
    TB[i] = V[i];
    TB[n + i] = f.one();
  }
}

// slot += w (mod p): there is no modular atomic add, so a compare-and-swap loop
template <class F>
__device__ __forceinline__ void atomic_add_mod(const F& f, u64* slot, u64 w) {
  unsigned long long old = __hip_atomic_load((unsigned long long*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    const unsigned long long want = f.add((u64)old, w);
    if (__hip_atomic_compare_exchange_strong((unsigned long long*)slot, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT))
      break;
  }
}
// The same P and L straight from the gate list (add_i / mul_i have one non-zero per gate): gate a with
// inputs (b, c) = (in0, in1) and weight v = eq(r_i, a) adds v to P[b] and v W(c) to L[b] if it is an add gate,
// v W(c) to P[b] if it is a mul gate.  Outputs start zeroed.
template <class F>
__global__ void __launch_bounds__(kBlock)
gkr_sparse_phase1_kernel(F f, const u64* __restrict__ val, const int* __restrict__ gate_type, const unsigned* __restrict__ in0,
                         const unsigned* __restrict__ in1, size_t n_gates, const u64* __restrict__ w, u64* __restrict__ P,
                         u64* __restrict__ L) {
  for (size_t a = (size_t)blockIdx.x * kBlock + threadIdx.x; a < n_gates; a += (size_t)gridDim.x * kBlock) {
    const u64 v = val[a], vw = f.mul(v, w[in1[a]]);
    if (gate_type[a] == 0) {
      atomic_add_mod(f, P + in0[a], v);
      atomic_add_mod(f, L + in0[a], vw);
    } else {
      atomic_add_mod(f, P + in0[a], vw);
    }
  }
}
// add(r_b, c) and mul(r_b, c) from the gate list: gate a adds eq(r_i, a) eq(r_b, in0) at c = in1 of its type's table
template <class F>
__global__ void __launch_bounds__(kBlock)
gkr_sparse_phase2_kernel(F f, const u64* __restrict__ val, const int* __restrict__ gate_type, const unsigned* __restrict__ in0,
                         const unsigned* __restrict__ in1, size_t n_gates, const u64* __restrict__ eqb, u64* __restrict__ add_r,
                         u64* __restrict__ mul_r) {
  for (size_t a = (size_t)blockIdx.x * kBlock + threadIdx.x; a < n_gates; a += (size_t)gridDim.x * kBlock) {
    const u64 v = f.mul(val[a], eqb[in0[a]]);
    atomic_add_mod(f, (gate_type[a] == 0 ? add_r : mul_r) + in1[a], v);
  }
}

}  // namespace sc
