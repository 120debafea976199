// gfx950 device kernels of the sumcheck prover hot path.
//
// All kernels stream evaluation tables of 64-bit field words; they are bound by HBM bandwidth
// (MFMA does not apply to exact 64-bit modular arithmetic) - except the 27-cell first pass,
// whose VALU issue rate is the limit (DESIGN.md "Kernels").  Wavefront = 64 lanes.  Every global
// access is "lane i <-> 16-byte piece base+i" (dwordx4, 1 KiB contiguous per wave
// instruction); where the arithmetic needs a lane to own a longer run of consecutive
// entries (both halves of every LE fold pair, entries 2b and 2b+1 -
// matrix-multiplication/src/lib.rs:114-121 of the reference) the wave's tile is transposed
// through a wave-private LDS region.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "field.hpp"

namespace sc {

constexpr int kBlock = 256;        // 4 waves per workgroup
constexpr int kWave = 64;
constexpr int kMaxSums = 27;      // 3^3 grid cells of a three-round pass

typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

// 16-byte global accesses with a COMPILE-TIME streaming hint.  Tables that are read once (or
// written for a much later reader) go around the caches with `nt` accesses: on this chip a plain
// read stream reaches 6.2-6.3 TB/s, a nontemporal one 6.9-7.0 TB/s, a copy 5.3 against 5.5 TB/s
// with nontemporal stores (tools/streambench.hip).  The hint has to be a template parameter:
// written as `flag ? __builtin_nontemporal_load(p) : *p` the two loads are merged into one
// plain load and the hint is lost (that is what the first version of these kernels did).
template <bool NT>
__device__ __forceinline__ ull2 ld16(const ull2* __restrict__ p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st16(ull2* __restrict__ p, ull2 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// Fold weights of one pass: w[c] = eq((r0,..,r_{KF-1}), c) = prod_j (bit_j(c) ? r_j : 1 - r_j),
// computed exactly on the host (Montgomery words).  KF = 1 uses w[1] = r0.
struct FoldW {
  u64 w[16];   // (KF <= 3 uses the first eight)
};

// Fold KF variables (LE) of a run of IN entries in registers; the first IN >> KF entries
// of v hold the result.
//  KF = 1: new[b] = t[2b] + r*(t[2b+1] - t[2b])                    (ark-poly fix_variables)
//  KF >= 2: new[b] = sum_c w[c] * t[2^KF b + c] - the same value (folding is linear), but as
//  2^KF unreduced multiply-accumulates and ONE Montgomery reduction per output instead of
//  2^KF - 1 dependent (sub, mul, reduce, add) steps: on gfx950 v_mad_u64_u32 issues as fast as
//  a 64-bit add or compare (tools/instbench.hip), so trading modular adds for multiplies
//  cuts the VALU work of a two-variable fold by ~40 %.
template <class F, int KF, int IN>
__device__ __forceinline__ void fold_run(const F& f, u64 (&v)[IN], const FoldW& fw) {
  if constexpr (KF == 1) {
    const u64 r0 = fw.w[1];
#pragma unroll
    for (int b = 0; b < IN / 2; ++b) v[b] = f.add(v[2 * b], f.mul(r0, f.sub(v[2 * b + 1], v[2 * b])));
  } else if constexpr (KF >= 2) {
    constexpr int G = 1 << KF;
#pragma unroll
    for (int b = 0; b < IN / G; ++b) {
      typename F::Acc3 acc;
      f.acc3_zero(acc);
#pragma unroll
      for (int c = 0; c < G; ++c) f.acc3_mac(acc, v[G * b + c], fw.w[c]);
      v[b] = f.acc3_get(acc);
    }
  }
}

// Round sums of the product of two tables over one run of OUT = 2^KS entries, in the
// evaluation basis {0, 1, inf} per variable ("inf" = leading coefficient = t1 - t0): one
// subtraction per extension value instead of the double-and-subtract of the point 2.  The
// host converts exactly: H(2) = 2 H(1) - H(0) + 2 H(inf) for a quadratic H
// (matrix-multiplication/src/lib.rs:116-120 evaluates at 0, 1, 2 directly).
//  KS = 1: acc[0..2] = H(0), H(1), H(inf)
//  KS = 2: acc[3u+v] = sum a(u,v)*b(u,v), (u,v) in {0,1,inf}^2, u on index bit 0, v on bit 1
// one 2x2 slice (index bits u, v) -> its nine extension values in {0,1,inf}^2
// both tables' quads at once: e[u][v], u on index bit 0, v on bit 1.  Four of the five
// differences per table depend on the inputs only (one sub4 each); the two (inf,inf) corners go
// through one sub2.
template <class F>
__device__ __forceinline__ void extend_quads(const F& f, const X64 (&ta)[4], const X64 (&tb)[4], X64 (&ea)[3][3],
                                             X64 (&eb)[3][3]) {
  const X64 ha[4] = {ta[1], ta[3], ta[2], ta[3]}, la[4] = {ta[0], ta[2], ta[0], ta[1]};
  const X64 hb[4] = {tb[1], tb[3], tb[2], tb[3]}, lb[4] = {tb[0], tb[2], tb[0], tb[1]};
  X64 da[4], db[4];
  f.sub4(da, ha, la);
  f.sub4(db, hb, lb);
  const X64 ch[2] = {da[1], db[1]}, cl[2] = {da[0], db[0]};
  X64 corner[2];
  f.sub2(corner, ch, cl);
  ea[0][0] = ta[0]; ea[1][0] = ta[1]; ea[2][0] = da[0];
  ea[0][1] = ta[2]; ea[1][1] = ta[3]; ea[2][1] = da[1];
  ea[0][2] = da[2]; ea[1][2] = da[3]; ea[2][2] = corner[0];
  eb[0][0] = tb[0]; eb[1][0] = tb[1]; eb[2][0] = db[0];
  eb[0][1] = tb[2]; eb[1][1] = tb[3]; eb[2][1] = db[1];
  eb[0][2] = db[2]; eb[1][2] = db[3]; eb[2][2] = corner[1];
}

template <class F, int KS>
__device__ __forceinline__ void accumulate_run(const F& f, typename F::Acc* acc, const u64* a,
                                               const u64* b) {
  if constexpr (KS == 1) {
    f.acc_mac(acc[0], a[0], b[0]);
    f.acc_mac(acc[1], a[1], b[1]);
    f.acc_mac(acc[2], f.sub(a[1], a[0]), f.sub(b[1], b[0]));
  } else if constexpr (KS == 2) {
    X64 ea[3][3], eb[3][3];
    const X64 ta[4] = {split64(a[0]), split64(a[1]), split64(a[2]), split64(a[3])};
    const X64 tb[4] = {split64(b[0]), split64(b[1]), split64(b[2]), split64(b[3])};
    extend_quads(f, ta, tb, ea, eb);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) f.acc_mac(acc[3 * u + v], ea[u][v], eb[u][v]);
  }
}

// KS = 3: acc[(3u + v)*3 + w] over octets, index bits (u, v, w) = (0, 1, 2), one w-slice at a
// time: w = 0 (entries 0..3), w = inf (entries 4..7 minus 0..3), w = 1 (entries 4..7), so that
// only two 3x3 extension blocks are live beside the 27 accumulators.
template <class F>
__device__ __forceinline__ void accumulate_octet(const F& f, typename F::Acc* acc, const u64* a, const u64* b) {
#pragma unroll
  for (int step = 0; step < 3; ++step) {
    const int w = (step == 0) ? 0 : (step == 1) ? 2 : 1;
    X64 sa[4], sb[4];
    if (w == 2) {
      const X64 ah[4] = {split64(a[4]), split64(a[5]), split64(a[6]), split64(a[7])};
      const X64 al[4] = {split64(a[0]), split64(a[1]), split64(a[2]), split64(a[3])};
      const X64 bh[4] = {split64(b[4]), split64(b[5]), split64(b[6]), split64(b[7])};
      const X64 bl[4] = {split64(b[0]), split64(b[1]), split64(b[2]), split64(b[3])};
      f.sub4(sa, ah, al);
      f.sub4(sb, bh, bl);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sa[i] = split64(a[4 * w + i]);
        sb[i] = split64(b[4 * w + i]);
      }
    }
    X64 ea[3][3], eb[3][3];
    extend_quads(f, sa, sb, ea, eb);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) f.acc_mac(acc[(3 * u + v) * 3 + w], ea[u][v], eb[u][v]);
  }
}

__device__ __forceinline__ u64 shfl_down_u64(u64 v, int off) {
  return (u64)__shfl_down((unsigned long long)v, off, kWave);
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
  return (u64)__shfl((unsigned long long)v, src, kWave);
}

// Block-wide modular sum of NS per-thread residues; result valid in threads [0, NS).
template <class F, int NS>
__device__ __forceinline__ void block_reduce(const F& f, u64 (&res)[NS], u64* lds /*[waves of the block][NS]*/) {
#pragma unroll
  for (int off = kWave / 2; off >= 1; off >>= 1) {
#pragma unroll
    for (int s = 0; s < NS; ++s) res[s] = f.add(res[s], shfl_down_u64(res[s], off));
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) lds[wave * NS + s] = res[s];
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    u64 t = lds[threadIdx.x];
    const int n_waves = (int)blockDim.x / kWave;
    for (int w = 1; w < n_waves; ++w) t = f.add(t, lds[w * NS + threadIdx.x]);
    res[0] = t;  // thread s holds sum s in res[0]
  }
}

// Split a residue into 32-bit limbs so that a plain u64 sum across <= 2^32 ranks cannot
// wrap ("modular all-reduce").
__device__ __forceinline__ void write_split(u64* out, int s, u64 v) {
  out[2 * s] = v & 0xFFFFFFFFull;
  out[2 * s + 1] = v >> 32;
}

// ------------------------------------------------------------------------------------
// Wave-private LDS transposition.
//
// Global accesses are only ever "lane i <-> 16-byte piece base+i" (one dwordx4 per lane,
// 1 KiB contiguous per wave-instruction): that pattern streams at ~6.4 TB/s on MI355X,
// while letting each lane walk its own 32..128-byte run costs 15-75 % of the bandwidth
// (profiles/r01_*).  The arithmetic, however, wants every lane to own a run of NP
// consecutive pieces.  Each wave therefore bounces its tile (64*NP pieces) through a
// private LDS region: written in load order, read back in run order.  The slot of piece
// q = NP*l + m is NP*l + (m ^ ((l >> log2(16/NP)) & (NP-1))): with that XOR both the
// ds_write_b128 (8 consecutive lanes = 128 contiguous bytes, permuted) and the
// ds_read_b128 (lane stride NP*16 bytes) are bank-conflict free for every 16-lane group
// the hardware forms (MI355X_MICROARCH.md, LDS table).  No workgroup barrier is needed:
// the region is private to the wave and LDS operations of one wave execute in order.
template <int NP>
__device__ __forceinline__ int swz_slot(int q) {
  if constexpr (NP == 1) return q;
  static_assert(NP == 2 || NP == 4 || NP == 8 || NP == 16, "pieces per lane");
  constexpr int SH = (NP == 2) ? 3 : (NP == 4) ? 2 : (NP == 8) ? 1 : 0;  // log2(16 / NP)
  const int l = q / NP;
  return (q & ~(NP - 1)) | ((q ^ (l >> SH)) & (NP - 1));
}
// LDS hand-off between the lanes of ONE wave: a wave's LDS operations execute in order, so all that is needed is
// that the earlier ones have been issued and returned and that the compiler keeps the order.  (A workgroup-scope
// fence - what the transposes used until round 4 - also waits for vmcnt(0): for the wave's global stores, the folded
// entries on their way out, which nobody in these kernels waits for, and for any load issued ahead.)
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
// in: v[k] = piece 64k + lane of the wave tile; out: v[m] = piece NP*lane + m
template <int NP>
__device__ __forceinline__ void transpose_to_runs(ull2* __restrict__ lds, ull2 (&v)[NP], int lane) {
  if constexpr (NP > 1) {
#pragma unroll
    for (int k = 0; k < NP; ++k) lds[swz_slot<NP>(64 * k + lane)] = v[k];
    wave_lds_sync();
#pragma unroll
    for (int m = 0; m < NP; ++m) v[m] = lds[swz_slot<NP>(NP * lane + m)];
    wave_lds_sync();
  }
}
// in: v[m] = piece NP*lane + m; out: v[k] = piece 64k + lane
template <int NP>
__device__ __forceinline__ void transpose_to_pieces(ull2* __restrict__ lds, ull2 (&v)[NP], int lane) {
  if constexpr (NP > 1) {
#pragma unroll
    for (int m = 0; m < NP; ++m) lds[swz_slot<NP>(NP * lane + m)] = v[m];
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < NP; ++k) v[k] = lds[swz_slot<NP>(64 * k + lane)];
    wave_lds_sync();
  }
}

}  // namespace sc

// the kernels by subject (each part reopens namespace sc; they build on each other in this order)
#include "kernels/pass.hpp"
#include "kernels/grid_pass.hpp"
#include "kernels/gram.hpp"
#include "kernels/mle.hpp"
#include "kernels/gkr.hpp"
#include "kernels/triangle.hpp"
#include "kernels/peer.hpp"
