// Part of kernels.hpp (included there, in order): the single-table kernels (fold, evaluate, fix_low, coldot) and the elementwise ones.
#pragma once

namespace sc {

// ------------------------------------------------------------------------------------
// Single-table kernels (DenseMultilinearExtension::fix_variables / evaluate on their own,
// and the two GEMV-shaped halves of matrix_multiplication::G::new).

// LE fold of KF in {1,2,3} variables in one pass: coalesced 16-byte loads, wave-private LDS
// transposition (a lane needs 2^(KF+1) consecutive entries), one coalesced 16-byte store
// per lane.  n_units = number of output pieces (pairs of output entries).
constexpr int kFoldBlock = 1024;   // fold_kernel: four-wave blocks for small tables, all sixteen waves of a CU beyond (host)
constexpr int kFoldGrab = 4;       // consecutive tiles per draw from the block's counter
constexpr size_t fold_kernel_lds_bytes(int kf, int threads) { return (size_t)(threads / kWave) * kWave * (size_t)(1 << kf) * sizeof(ull2); }
template <class F, int KF, bool NT>
__global__ void __launch_bounds__(kFoldBlock)
fold_kernel(F f, const u64* __restrict__ T, u64* __restrict__ T2, FoldW fw, size_t n_units, int grab) {
  constexpr int IN = 2 << KF, NP = IN / 2;
  extern __shared__ ull2 fold_lds[];   // [waves of the block][kWave * NP]: sized by the launch (fold_kernel_lds_bytes)
  __shared__ unsigned lds_next;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  ull2* const my_lds = fold_lds + wave * kWave * NP;
  const ull2* __restrict__ Tp = reinterpret_cast<const ull2*>(T);
  ull2* __restrict__ T2p = reinterpret_cast<ull2*>(T2);
  const size_t n_tiles = (n_units + kWave - 1) / kWave, in_pieces = n_units * NP;
  if (threadIdx.x == 0) lds_next = 0;
  __syncthreads();
  // the waves of a block draw runs of `grab` tiles from a counter in LDS (evaluate_kernel; block b owns the runs c * grid + b).
  // grab = kFoldGrab for the one-block-per-CU launches of large tables; 1 for the four-wave blocks of small ones, whose grid
  // is sized for one tile per wave (with runs of four there, one wave of each block folded its four tiles serially while
  // the other three drew a run past the end and left: a quarter of the memory-level parallelism - ADVICE r03)
  auto next_run = [&]() -> size_t {
    unsigned c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(&lds_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return ((size_t)__builtin_amdgcn_readfirstlane(c) * gridDim.x + blockIdx.x) * (size_t)grab;
  };
  for (size_t run = next_run(); run < n_tiles; run = next_run())
  for (size_t tile = run; tile < run + (size_t)grab && tile < n_tiles; ++tile) {
    ull2 pv[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const size_t q = tile * kWave * NP + (size_t)k * kWave + lane;
      const ull2 zero = {0, 0};
      pv[k] = zero;
      if (q < in_pieces) pv[k] = ld16<NT>(Tp + q);
    }
    transpose_to_runs<NP>(my_lds, pv, lane);
    u64 v[IN];
#pragma unroll
    for (int k = 0; k < NP; ++k) { v[2 * k] = pv[k].x; v[2 * k + 1] = pv[k].y; }
    fold_run<F, KF, IN>(f, v, fw);
    const size_t qo = tile * kWave + lane;
    if (qo < n_units) {
      ull2 o = {v[0], v[1]};
      T2p[qo] = o;
    }
    // (bulk write-back of the L2, one wave per XCD: kernels/pass.hpp, bulk_writeback)
    if (grab > 1 && tile + 1 == run + (size_t)grab && wave == 0 && blockIdx.x < 8) asm volatile("buffer_wbl2 sc1" ::: "memory");
  }
}
// LE, scalar tail: outputs that do not fill a 16-byte piece (n_out == 1).
template <class F>
__global__ void fold_le_small_kernel(F f, const u64* __restrict__ T, u64* __restrict__ T2, u64 r,
                                     size_t n_out) {
  size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n_out) T2[b] = f.add(T[2 * b], f.mul(r, f.sub(T[2 * b + 1], T[2 * b])));
}
// BE (variable = current MSB): out[b] = t[b] + r*(t[b+half] - t[b]); two entries per thread
// when half is even, scalar otherwise.
template <class F>
__global__ void __launch_bounds__(kBlock)
fold_be_kernel(F f, const u64* __restrict__ T, u64* __restrict__ T2, u64 r, size_t half) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  if ((half & 1) == 0) {
    const ull2* __restrict__ Tp = reinterpret_cast<const ull2*>(T);
    ull2* __restrict__ T2p = reinterpret_cast<ull2*>(T2);
    for (size_t u = (size_t)blockIdx.x * kBlock + threadIdx.x; u < half / 2; u += stride) {
      const ull2 lo = Tp[u], hi = Tp[half / 2 + u];
      ull2 o = {f.add(lo.x, f.mul(r, f.sub(hi.x, lo.x))), f.add(lo.y, f.mul(r, f.sub(hi.y, lo.y)))};
      T2p[u] = o;
    }
  } else {
    for (size_t b = (size_t)blockIdx.x * kBlock + threadIdx.x; b < half; b += stride)
      T2[b] = f.add(T[b], f.mul(r, f.sub(T[b + half], T[b])));
  }
}

// Up to 64 challenges by value (kernel argument).
struct RVec {
  u64 v[64];
};

// out[i] = prod_j ( bit_j(i) ? r[off+j] : 1 - r[off+j] ),  i < 2^nbits   (LE bit order)
template <class F>
__global__ void __launch_bounds__(kBlock)
eq_table_kernel(F f, RVec rv, int off, int nbits, u64* __restrict__ out) {
  const size_t n = (size_t)1 << nbits;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    u64 w = f.one();
    for (int j = 0; j < nbits; ++j) {
      const u64 rj = rv.v[off + j];
      w = f.mul(w, ((i >> j) & 1) ? rj : f.sub(f.one(), rj));
    }
    out[i] = w;
  }
}

// eqA[i] = prod_{j < ta} (bit_j(i) ? r[j] : 1 - r[j]), i < 2^ta <= 1024, by the whole block: products of two half
// tables (<= 32 entries of <= 5 factors each, then one product per weight).  2^ta weights of ta factors each were ~1000
// instructions per thread on waves issuing alone - 4 us of a 35 us launch on a 2^24-entry table.  Ends with a barrier.
template <class F>
__device__ __forceinline__ void build_eq_weights(const F& f, const u64* r, int ta, u64* eqA /* [1 << ta] */) {
  __shared__ u64 eqH[2][32];
  const int lo_bits = ta < 5 ? ta : 5, hi_bits = ta - lo_bits;
  if (threadIdx.x < 64) {
    const int half = threadIdx.x >> 5, i = threadIdx.x & 31;
    const int nb = half ? hi_bits : lo_bits, off = half ? lo_bits : 0;
    u64 w = f.one();
    for (int j = 0; j < nb; ++j) {
      const u64 rj = r[off + j];
      w = f.mul(w, ((i >> j) & 1) ? rj : f.sub(f.one(), rj));
    }
    eqH[half][i] = w;   // entries with bits above nb repeat lower ones and are never read
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (1 << ta); i += blockDim.x) eqA[i] = f.mul(eqH[0][i & ((1 << lo_bits) - 1)], eqH[1][i >> lo_bits]);
  __syncthreads();
}

// Polynomial::evaluate of a 2^n-entry table (n >= 8) in ONE streaming pass:
//   sum_i t[i] * eq(r, i),  eq factored over the index bits as
//   bit 0 (inside a 16-byte piece) | bits 1..6 (lane) | ta bits (tile within a segment,
//   weights eqA) | tb bits (segment, weights eqB).
// Inner sums are unreduced (lazy) accumulations of t * eqA (two per lane, for bit 0 = 0/1);
// they are reduced once per chunk of tiles and folded into the outer accumulators with
// eqB; the bit-0 and lane weights are applied once per thread at the end.  This is the
// streaming form of vsbw_multilinear_from_evaluations' "eq table, then dot product"
// (multilinear-extensions/src/lib.rs:6-24) without materialising the 2^n eq table.
// Launched with kBlock threads while every wave gets at most one chunk, with stream_block<F>::evaluate threads = all twelve
// waves a CU holds of it (three per SIMD) beyond that: the waves of a block then draw their chunks from a counter in LDS.  With three
// 256-thread blocks per CU and a fixed share per wave the three wave slots of a SIMD left the loop of a 2^28-entry
// table after 202 / 270 / 336 us - a SIMD issues its oldest wave first (pass_kernel, "Tiles are not dealt out in advance").
template <class F> struct stream_block {            // threads of the one-block-per-CU launches of the two streaming readers
  static constexpr int evaluate = 768, fix_low = 1024;
};
// (until round 4 the generic modulus reduced every product and needed 174-186 VGPRs here: 512-thread blocks.  With its lazy
// sums it is at the Goldilocks kernels' 122-126 and takes the same launch shapes.)
template <class F, bool NT>
__global__ void __launch_bounds__(stream_block<F>::evaluate)
evaluate_kernel(F f, const u64* __restrict__ T, int n, RVec rv, int ta, int chunk_log, u64 w_extra, PassOut out) {
  __shared__ u64 eqA[1024];  // ta <= 10
  __shared__ u64 lds[stream_block<F>::evaluate / kWave];
  __shared__ int lds_flag;
  __shared__ unsigned lds_next;
  const int lane = threadIdx.x & (kWave - 1);
  const int tb = n - 7 - ta;
  if (threadIdx.x == 0) lds_next = 0;
  build_eq_weights(f, rv.v + 7, ta, eqA);   // the tile-in-segment weights (ends with a barrier)
  auto next_chunk = [&]() -> size_t {
    unsigned c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(&lds_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (size_t)__builtin_amdgcn_readfirstlane(c) * gridDim.x + blockIdx.x;
  };
  const ull2* __restrict__ Tp = reinterpret_cast<const ull2*>(T);
  const size_t n_tiles = (size_t)1 << (n - 7);
  const size_t n_chunks = n_tiles >> chunk_log;
  const int C = 1 << chunk_log;
  typename F::Acc o0, o1;
  f.acc_zero(o0);
  f.acc_zero(o1);
  for (size_t chunk = next_chunk(); chunk < n_chunks; chunk = next_chunk()) {
    const size_t tile0 = chunk << chunk_log;
    const size_t seg = tile0 >> ta;
    const int in_seg = (int)(tile0 & (((size_t)1 << ta) - 1));
    typename F::Acc a0, a1;
    f.acc_zero(a0);
    f.acc_zero(a1);
    // Batches of eight 16-byte loads per lane, DOUBLE-BUFFERED: the next batch is requested before the products of the
    // current one.  On a 2^24-entry table the launch has one wave per SIMD, and a wave that waits for its loads (~0.7 us)
    // and then multiplies (240 instructions issued alone, ~0.9 us) in turn reads at 5.2 TB/s whatever the grid shape
    // (profiles/r03_mle24_sweep.txt); with the next batch in flight during the products the two overlap.  Written as
    // fixed-count loops because the runtime unroller does not touch loops that contain inline assembly (acc_mac).
    auto load8 = [&](ull2 (&p8)[8], size_t tile) {
#pragma unroll
      for (int k = 0; k < 8; ++k) p8[k] = ld16<NT>(Tp + (tile + k) * kWave + lane);
    };
    auto mac8 = [&](const ull2 (&p8)[8], int w0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const u64 w = eqA[w0 + k];
        f.acc_mac(a0, p8[k].x, w);
        f.acc_mac(a1, p8[k].y, w);
      }
    };
    int i = 0;
    if (C >= 16) {
      ull2 pa[8], pb[8];
      load8(pa, tile0);
      for (; i + 16 <= C; i += 16) {
        load8(pb, tile0 + i + 8);
        mac8(pa, in_seg + i);
        if (i + 32 <= C) load8(pa, tile0 + i + 16);
        mac8(pb, in_seg + i + 8);
      }
    }
    for (; i + 8 <= C; i += 8) {
      ull2 pc[8];
      load8(pc, tile0 + i);
      mac8(pc, in_seg + i);
    }
    for (; i < C; ++i) {
      const size_t q = (tile0 + i) * kWave + lane;
      const ull2 pc = ld16<NT>(Tp + q);
      const u64 w = eqA[in_seg + i];
      f.acc_mac(a0, pc.x, w);
      f.acc_mac(a1, pc.y, w);
    }
    u64 wB = f.one();  // segment weight, wave-uniform: tb factors per chunk of 2*C products
    for (int j = 0; j < tb; ++j) {
      const u64 rj = rv.v[7 + ta + j];
      wB = f.mul(wB, ((seg >> j) & 1) ? rj : f.sub(f.one(), rj));
    }
    f.acc_mac(o0, f.acc_get(a0), wB);
    f.acc_mac(o1, f.acc_get(a1), wB);
  }
  // bit 0, lane and (sharded evaluate) rank weights
  const u64 r0 = rv.v[0];
  u64 v = f.add(f.mul(f.sub(f.one(), r0), f.acc_get(o0)), f.mul(r0, f.acc_get(o1)));
  u64 wl = w_extra;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const u64 rj = rv.v[1 + j];
    wl = f.mul(wl, ((lane >> j) & 1) ? rj : f.sub(f.one(), rj));
  }
  u64 res[1] = {f.mul(v, wl)};
  block_reduce<F, 1>(f, res, lds);
  finish_pass<F, 1>(f, out, res[0], &lds_flag);
}

// Polynomial::evaluate of ONE table at M points in ONE streaming pass (round 4): m separate evaluations read the table m
// times and pay m launches and m hand-offs - restrict_poly's k + 1 points on a line (gkr-protocol/src/lib.rs:291-321), the
// verifier's oracle queries.  Same factoring of eq(r, i) as evaluate_kernel, per point j:
//   E[j][tile bits | bit 0] (LDS, built per block from half tables), lane weights L[j][lane] (LDS), segment weight (per chunk).
// A lane keeps ONE lazy accumulator per point - both entries of its 16-byte piece go into it, each with its own weight
// (bit 0 is an index of E) - and one outer residue per point; a piece costs 2 M multiply-accumulates, so the pass is
// bound by instruction issue, not by the table's bytes: what is amortised is the read, the launch and the hand-off.  pts: M points of n words each in device memory (point j at pts + 64 j); w_extra[j]: the weight
// of this rank's bits at point j (1 unless sharded).  Sums leave through finish_pass as M values.
constexpr int kEvalManyTa = 7;   // at most 2^7 tiles per segment: E is M x 256 words
struct EvalManyW {
  u64 w[16];
};
template <class F, int M, bool NT>
__global__ void __launch_bounds__(kBlock)
evaluate_many_kernel(F f, const u64* __restrict__ T, int n, const u64* __restrict__ pts, int ta, int chunk_log, EvalManyW wx, PassOut out) {
  constexpr int kWaves = kBlock / kWave;
  __shared__ __attribute__((aligned(16))) u64 E[M][2 << kEvalManyTa];   // E[j][2 t + b]: weight of entry b of the piece in tile t of a segment
  __shared__ u64 L[M][kWave];
  __shared__ u64 H[M][3][16];              // half tables: [0] bits 0..3 of E's index, [1] bits 4..7, [2] unused / lane halves
  __shared__ u64 R[M][64];                 // the points
  __shared__ u64 S[M][3][32];              // segment weights by 5-bit digit of the segment index: 2 products per chunk and point
  __shared__ u64 lds[kWaves * M];
  __shared__ int lds_flag;
  __shared__ unsigned lds_next;
  const int lane = threadIdx.x & (kWave - 1);
  const int tb = n - 7 - ta, ebits = ta + 1;
  if (threadIdx.x == 0) lds_next = 0;
  for (int i = threadIdx.x; i < M * 64; i += kBlock) R[i >> 6][i & 63] = ((i & 63) < n) ? pts[i] : 0;
  __syncthreads();
  // E's index bits, low to high: bit 0 of the table index (variable 0), then the tile-in-segment bits (variables 7 .. 7+ta-1)
  auto evar = [&](int j, int q) -> u64 { return q == 0 ? R[j][0] : R[j][7 + q - 1]; };
  const int lo_bits = ebits < 4 ? ebits : 4, hi_bits = ebits - lo_bits;
  for (int i = threadIdx.x; i < M * 32; i += kBlock) {
    const int j = i >> 5, half = (i >> 4) & 1, e = i & 15;
    const int nb = half ? hi_bits : lo_bits, off = half ? lo_bits : 0;
    u64 w = f.one();
    for (int q = 0; q < nb; ++q) {
      const u64 rq = evar(j, off + q);
      w = f.mul(w, ((e >> q) & 1) ? rq : f.sub(f.one(), rq));
    }
    H[j][half][e] = w;
  }
  // lane weights: variables 1..6, as products of two 8-entry halves
  for (int i = threadIdx.x; i < M * 16; i += kBlock) {
    const int j = i >> 4, half = (i >> 3) & 1, e = i & 7;
    u64 w = f.one();
    for (int q = 0; q < 3; ++q) {
      const u64 rq = R[j][1 + 3 * half + q];
      w = f.mul(w, ((e >> q) & 1) ? rq : f.sub(f.one(), rq));
    }
    H[j][2][8 * half + e] = w;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (M << ebits); i += kBlock) {
    const int j = i >> ebits, e = i & ((1 << ebits) - 1);
    E[j][e] = f.mul(H[j][0][e & ((1 << lo_bits) - 1)], H[j][1][e >> lo_bits]);
  }
  // (the tb factors of a segment's weight, recomputed per chunk and point, were most of a small chunk's instructions)
  for (int i = threadIdx.x; i < M * 96; i += kBlock) {
    const int j = i / 96, lvl = (i % 96) >> 5, e = i & 31;
    u64 w = f.one();
    for (int q = 0; q < 5; ++q) {
      const int bit = 5 * lvl + q;
      if (bit < tb) {
        const u64 rq = R[j][7 + ta + bit];
        w = f.mul(w, ((e >> q) & 1) ? rq : f.sub(f.one(), rq));
      }
    }
    S[j][lvl][e] = w;
  }
  for (int i = threadIdx.x; i < M * kWave; i += kBlock) {
    const int j = i >> 6, l = i & 63;
    L[j][l] = f.mul(f.mul(H[j][2][l & 7], H[j][2][8 + (l >> 3)]), wx.w[j]);
  }
  __syncthreads();
  auto next_chunk = [&]() -> size_t {
    unsigned c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(&lds_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (size_t)__builtin_amdgcn_readfirstlane(c) * gridDim.x + blockIdx.x;
  };
  const ull2* __restrict__ Tp = reinterpret_cast<const ull2*>(T);
  const size_t n_tiles = (size_t)1 << (n - 7), n_chunks = n_tiles >> chunk_log;
  const int C = 1 << chunk_log;
  u64 o[M];
#pragma unroll
  for (int j = 0; j < M; ++j) o[j] = 0;
  for (size_t chunk = next_chunk(); chunk < n_chunks; chunk = next_chunk()) {
    const size_t tile0 = chunk << chunk_log, seg = tile0 >> ta;
    const int in_seg = (int)(tile0 & (((size_t)1 << ta) - 1));
    // the three-class accumulator (eight instructions per multiply-accumulate, nine registers) instead of the four-register
    // lazy sum (fifteen): with 2 M products per piece the pass is bound by instruction issue, and M accumulators fit
    typename F::Acc3 a[M];
#pragma unroll
    for (int j = 0; j < M; ++j) f.acc3_zero(a[j]);
    constexpr int B = (M == 16) ? 2 : 4;   // pieces in flight per lane (sixteen nine-register accumulators leave room for two)
    int i = 0;
    for (; i + B <= C; i += B) {   // fixed-count inner loops: see evaluate_kernel
      ull2 p[B];
#pragma unroll
      for (int k = 0; k < B; ++k) p[k] = ld16<NT>(Tp + (tile0 + i + k) * kWave + lane);
#pragma unroll
      for (int k = 0; k < B; ++k) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const ull2 w = *reinterpret_cast<const ull2*>(&E[j][2 * (in_seg + i + k)]);
          f.acc3_mac(a[j], p[k].x, w.x);
          f.acc3_mac(a[j], p[k].y, w.y);
        }
      }
    }
    for (; i < C; ++i) {
      const ull2 p = ld16<NT>(Tp + (tile0 + i) * kWave + lane);
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const ull2 w = *reinterpret_cast<const ull2*>(&E[j][2 * (in_seg + i)]);
        f.acc3_mac(a[j], p.x, w.x);
        f.acc3_mac(a[j], p.y, w.y);
      }
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      u64 wB = f.mul(S[j][0][seg & 31], S[j][1][(seg >> 5) & 31]);   // segment weight, wave-uniform
      if (tb > 10) wB = f.mul(wB, S[j][2][(seg >> 10) & 31]);
      for (int q = 15; q < tb; ++q) {   // (tables beyond 2^29 entries)
        const u64 rq = R[j][7 + ta + q];
        wB = f.mul(wB, ((seg >> q) & 1) ? rq : f.sub(f.one(), rq));
      }
      o[j] = f.add(o[j], f.mul(f.acc3_get(a[j]), wB));
    }
  }
  u64 res[M];
#pragma unroll
  for (int j = 0; j < M; ++j) res[j] = f.mul(o[j], L[j][lane]);
  block_reduce<F, M>(f, res, lds);
  finish_pass<F, M>(f, out, res[0], &lds_flag);
}
// the same for a table of fewer than 2^8 entries: one block, a wave per point (lane = entry)
template <class F, int M>
__global__ void __launch_bounds__(kBlock)
evaluate_many_small_kernel(F f, const u64* __restrict__ T, int n, const u64* __restrict__ pts, EvalManyW wx, PassOut out) {
  __shared__ u64 val[M];
  __shared__ int lds_flag;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const size_t len = (size_t)1 << n;
  for (int j = wave; j < M; j += kBlock / kWave) {
    u64 v = 0;
    for (size_t i = lane; i < len; i += kWave) {
      u64 w = T[i];
      for (int q = 0; q < n; ++q) {
        const u64 rq = pts[64 * j + q];
        w = f.mul(w, ((i >> q) & 1) ? rq : f.sub(f.one(), rq));
      }
      v = f.add(v, w);
    }
#pragma unroll
    for (int off = kWave / 2; off >= 1; off >>= 1) v = f.add(v, shfl_down_u64(v, off));
    if (lane == 0) val[j] = f.mul(v, wx.w[j]);
  }
  __syncthreads();
  finish_pass<F, M>(f, out, threadIdx.x < M ? val[threadIdx.x] : 0, &lds_flag);
}

// LE fix of the LOW k variables (8 <= k <= 17) in ONE pass: out[b] = sum_c eq(r, c) * t[b*2^k + c],
// i.e. evaluate_kernel's inner product on every contiguous segment of 2^k entries, one wave per
// segment (coalesced 1 KiB wave loads, tile weights eqA in LDS, bit-0 and lane weights applied
// once per segment, a shuffle reduction, one 8-byte store).  A chain of three-variable folds
// reads the table 1.14 times and writes an eighth of it; this reads it once.
// (DenseMultilinearExtension::fix_variables with many variables; the f_B half of G::new.)
template <class F, bool NT>
__global__ void __launch_bounds__(stream_block<F>::fix_low)
fix_low_kernel(F f, const u64* __restrict__ T, u64* __restrict__ out, int k, RVec rv, size_t n_out) {
  __shared__ u64 eqA[1024];  // k - 7 <= 10
  __shared__ unsigned lds_next;   // the block's segment counter (see evaluate_kernel)
  const int lane = threadIdx.x & (kWave - 1);
  const int ta = k - 7;
  if (threadIdx.x == 0) lds_next = 0;
  build_eq_weights(f, rv.v + 7, ta, eqA);
  auto next_seg = [&]() -> size_t {
    unsigned c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(&lds_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (size_t)__builtin_amdgcn_readfirstlane(c) * gridDim.x + blockIdx.x;
  };
  u64 wl = f.one();
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const u64 rj = rv.v[1 + j];
    wl = f.mul(wl, ((lane >> j) & 1) ? rj : f.sub(f.one(), rj));
  }
  const u64 r0 = rv.v[0], one_minus_r0 = f.sub(f.one(), r0);
  __syncthreads();
  const ull2* __restrict__ Tp = reinterpret_cast<const ull2*>(T);
  const int tiles = 1 << ta;
  for (size_t seg = next_seg(); seg < n_out; seg = next_seg()) {
    const ull2* __restrict__ Sp = Tp + (seg << (k - 1)) + lane;
    typename F::Acc a0, a1;
    f.acc_zero(a0);
    f.acc_zero(a1);
    // double-buffered batches of eight loads (see evaluate_kernel)
    auto load8 = [&](ull2 (&p8)[8], int tile) {
#pragma unroll
      for (int q = 0; q < 8; ++q) p8[q] = ld16<NT>(Sp + (size_t)(tile + q) * kWave);
    };
    auto mac8 = [&](const ull2 (&p8)[8], int w0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const u64 w = eqA[w0 + q];
        f.acc_mac(a0, p8[q].x, w);
        f.acc_mac(a1, p8[q].y, w);
      }
    };
    int i = 0;
    if (tiles >= 16) {
      ull2 pa[8], pb[8];
      load8(pa, 0);
      for (; i + 16 <= tiles; i += 16) {
        load8(pb, i + 8);
        mac8(pa, i);
        if (i + 32 <= tiles) load8(pa, i + 16);
        mac8(pb, i + 8);
      }
    }
    for (; i + 8 <= tiles; i += 8) {
      ull2 pc[8];
      load8(pc, i);
      mac8(pc, i);
    }
    for (; i < tiles; ++i) {
      const ull2 pc = Sp[(size_t)i * kWave];
      const u64 w = eqA[i];
      f.acc_mac(a0, pc.x, w);
      f.acc_mac(a1, pc.y, w);
    }
    u64 v = f.add(f.mul(one_minus_r0, f.acc_get(a0)), f.mul(r0, f.acc_get(a1)));
    v = f.mul(v, wl);
#pragma unroll
    for (int off = kWave / 2; off >= 1; off >>= 1) v = f.add(v, shfl_down_u64(v, off));
    if (lane == 0) out[seg] = v;
  }
}

// "Column dot": out[c] = sum_{i in [i0, i1)} w[i] * t[i*M + c]  for one chunk of rows per
// blockIdx.y; partial[y][c] holds chunk y (reduced by sum_rows_kernel when there are
// several).  This is fix_variables of the TOP k index bits (BE order), and the f_A half of
// G::new: f_A[col] = sum_row eq(r1)[row] * A[row][col] (matrix-multiplication/src/lib.rs:81-83,
// relabel + fold collapsed into one pass).  Lanes own 16-byte pieces of c: coalesced.
// Row-walking access pattern (this kernel and gkr_phase1_kernel): a WAVE owns PW consecutive 1 KiB spans of every row
// of its chunk (lane l: pieces span*64*PW + 64 j + l, j < PW).  Measured on this chip (tools/rowwalk.hip,
// profiles/r03_rowwalk.txt, two 2^13 x 2^13 tables): one 1 KiB span per wave and row reads at 6.3 TB/s, four
// contiguous KiB at 6.9; FOUR waves per SIMD are slower than one (5.1-5.7 TB/s: more rows open at once than the
// DRAM pages like) - so the launch is sized for one wave per SIMD and the memory pipe is fed by the loads in flight
// per lane (rows in flight x PW), not by occupancy.
template <class F, bool NT, int PW>
__global__ void __launch_bounds__(kBlock)
coldot_kernel(F f, const u64* __restrict__ T, const u64* __restrict__ w, size_t rows, size_t rows_per_chunk,
              size_t M, u64* __restrict__ partial) {
  constexpr int RIF = 4;   // rows in flight per thread (tools/rowwalk.hip: 4 x 4 KiB reads fastest on one table)
  const ull2* __restrict__ Tp = reinterpret_cast<const ull2*>(T);
  ull2* __restrict__ Pp = reinterpret_cast<ull2*>(partial);
  const size_t mp = M / 2;  // pieces per row
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const size_t i0 = (size_t)blockIdx.y * rows_per_chunk;
  const size_t i1 = (i0 + rows_per_chunk < rows) ? i0 + rows_per_chunk : rows;
  const size_t n_spans = (mp + (size_t)kWave * PW - 1) / ((size_t)kWave * PW);
  for (size_t span = (size_t)blockIdx.x * (kBlock / kWave) + wave; span < n_spans; span += (size_t)gridDim.x * (kBlock / kWave)) {
    const size_t pc0 = span * kWave * PW + lane;
    typename F::Acc a0[PW], a1[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      f.acc_zero(a0[j]);
      f.acc_zero(a1[j]);
    }
    const ull2 zero = {0, 0};
    size_t i = i0;
    for (; i + RIF <= i1; i += RIF) {  // fixed-count inner loop: see evaluate_kernel
      ull2 v[RIF][PW];
#pragma unroll
      for (int k = 0; k < RIF; ++k)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
          const size_t pc = pc0 + (size_t)j * kWave;
          v[k][j] = (PW == 1 || pc < mp) ? ld16<NT>(Tp + (i + k) * mp + (pc < mp ? pc : 0)) : zero;
        }
#pragma unroll
      for (int k = 0; k < RIF; ++k) {
        const u64 wi = w[i + k];
#pragma unroll
        for (int j = 0; j < PW; ++j) {
          f.acc_mac(a0[j], v[k][j].x, wi);
          f.acc_mac(a1[j], v[k][j].y, wi);
        }
      }
    }
    for (; i < i1; ++i) {
      const u64 wi = w[i];
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        const size_t pc = pc0 + (size_t)j * kWave;
        const ull2 v = pc < mp ? ld16<NT>(Tp + i * mp + pc) : zero;
        f.acc_mac(a0[j], v.x, wi);
        f.acc_mac(a1[j], v.y, wi);
      }
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const size_t pc = pc0 + (size_t)j * kWave;
      if (pc < mp) Pp[(size_t)blockIdx.y * mp + pc] = ull2{f.acc_get(a0[j]), f.acc_get(a1[j])};
    }
  }
}
// out[c] = sum_y partial[y][c], for one or two arrays of partial rows (blockIdx.y selects; the second is the L of a GKR
// phase).  Eight rows in flight per thread: the first version walked the rows one dependent load at a time with 2^13
// threads - 20 us per array for 64 x 2^13 words, a tenth of the streaming pass it follows.
template <class F>
__global__ void __launch_bounds__(kBlock)
sum_rows_kernel(F f, const u64* __restrict__ partial0, const u64* __restrict__ partial1, size_t chunks, size_t M,
                u64* __restrict__ out0, u64* __restrict__ out1) {
  const u64* __restrict__ partial = blockIdx.y ? partial1 : partial0;
  u64* __restrict__ out = blockIdx.y ? out1 : out0;
  for (size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x; c < M; c += (size_t)gridDim.x * kBlock) {
    u64 t = 0;
    size_t y = 0;
    for (; y + 8 <= chunks; y += 8) {
      u64 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = partial[(y + q) * M + c];
      t = f.add(t, f.add(f.add(f.add(v[0], v[1]), f.add(v[2], v[3])), f.add(f.add(v[4], v[5]), f.add(v[6], v[7]))));
    }
    for (; y < chunks; ++y) t = f.add(t, partial[y * M + c]);
    out[c] = t;
  }
}

// ------------------------------------------------------------------------------------
// Elementwise / utility kernels.

// t[i] = to_mont(splitmix64(seed + start + i) mod p)   
template <class F>
__global__ void __launch_bounds__(kBlock)
generate_kernel(F f, u64 seed, u64 start, size_t len, u64* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += stride)
    out[i] = f.to_mont(f.reduce_word(splitmix64(seed + start + i)));
}

// G::to_evaluations: out[i] = a[i]*b[i]   (matrix-multiplication/src/lib.rs:137-146)
template <class F>
__global__ void __launch_bounds__(kBlock)
mul_kernel(F f, const u64* __restrict__ A, const u64* __restrict__ B, u64* __restrict__ out, size_t len) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += stride)
    out[i] = f.mul(A[i], B[i]);
}

// DenseMultilinearExtension::relabel: out[swap_fields(i)] = t[i]; the swap is an involution
// so it is applied to the (coalesced) output index.
__global__ void __launch_bounds__(kBlock)
relabel_kernel(const u64* __restrict__ T, u64* __restrict__ out, size_t len, unsigned a, unsigned b,
               unsigned k) {
  const size_t mask = ((size_t)1 << k) - 1;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < len; j += stride) {
    size_t fa = (j >> a) & mask, fb = (j >> b) & mask;
    size_t i = (j & ~((mask << a) | (mask << b))) | (fb << a) | (fa << b);
    out[j] = T[i];
  }
}

// Sharded evaluate helper: out_split = split limbs of w * v (one thread).
template <class F>
__global__ void scale_split_kernel(F f, const u64* __restrict__ v, u64 w, u64* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) write_split(out, 0, f.mul(w, v[0]));
}

}  // namespace sc
