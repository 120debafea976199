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
  u64 w[8];
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
__device__ __forceinline__ void wave_lds_fence() {
  // orders this wave's LDS writes before its later LDS reads (s_waitcnt lgkmcnt(0)) and
  // stops the compiler from moving them across
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// in: v[k] = piece 64k + lane of the wave tile; out: v[m] = piece NP*lane + m
template <int NP>
__device__ __forceinline__ void transpose_to_runs(ull2* __restrict__ lds, ull2 (&v)[NP], int lane) {
  if constexpr (NP > 1) {
#pragma unroll
    for (int k = 0; k < NP; ++k) lds[swz_slot<NP>(64 * k + lane)] = v[k];
    wave_lds_fence();
#pragma unroll
    for (int m = 0; m < NP; ++m) v[m] = lds[swz_slot<NP>(NP * lane + m)];
    wave_lds_fence();
  }
}
// in: v[m] = piece NP*lane + m; out: v[k] = piece 64k + lane
template <int NP>
__device__ __forceinline__ void transpose_to_pieces(ull2* __restrict__ lds, ull2 (&v)[NP], int lane) {
  if constexpr (NP > 1) {
#pragma unroll
    for (int m = 0; m < NP; ++m) lds[swz_slot<NP>(NP * lane + m)] = v[m];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < NP; ++k) v[k] = lds[swz_slot<NP>(64 * k + lane)];
    wave_lds_fence();
  }
}

// ------------------------------------------------------------------------------------
// The pass kernel: fold KF pending variables of both tables, write the folded tables,
// and accumulate the round sums of the FOLDED tables for the next KS rounds - one read of
// the inputs, one write of the outputs (reference: Prover::round =
// fix_variables + to_univariate, sum-check-protocol/src/lib.rs:105-112).
//
// unit = one run of IN = 2^(KF+KS) input entries per table -> OUT = 2^KS output entries,
// owned by one lane; a wave tile is 64 units.  KF in 0..3, KS in 1..3 (KS = 3, the 27-cell grid
// of a three-round first pass, is only instantiated with KF = 0).  Sums leave through
// finish_pass (PassOut).
// Where a pass leaves its sums.
//  * grid of one block: that block publishes directly.
//  * larger grids: every block stores its partial residues (sum-major rows), takes a ticket,
//    and the block that draws the last ticket reduces all partials and publishes - one
//    launch per pass instead of pass + reduce (+ copy).  Hand-off follows
//    cdna_hip_programming.md Guideline 16 (form R1): partials are stored write-through
//    (sc1), the storing wave drains them (s_waitcnt vmcnt(0)), then lane 0 adds to the
//    ticket; the block whose add returns the last ticket acquires at agent scope behind a
//    workgroup barrier and reads the partials with sc1 loads.  The ticket counter only grows
//    (base = value before this launch), so nothing has to be re-zeroed between launches.
//  * publish target: `mailbox` (pinned host memory the host spins on: 2*NS split limbs, then
//    the sequence word at index kMailboxSeq) or, for the sharded transports that still have
//    to all-reduce on the device, `sums_dev`.
constexpr int kMailboxSeq = 60;   // 2*27 limbs first, the sequence word after them
constexpr int kMailboxErr = 62;   // 0, or why the pass's cross-rank exchange failed (kXchg*)

// In-kernel exchange of the round sums between the ranks of a sharded prover (one process per GPU,
// SURVEY.md section 8e).  Each rank owns an INBOX in its own HBM that every peer maps (HIP IPC) and writes
// over xGMI: inbox[parity][source rank][kInboxWords] 8-byte granules {tag : 32 | value : 32}.  The values
// are the 32-bit limbs of the pass's sums (a u64 sum of residues would wrap mod 2^64, not mod p), so the
// data IS the flag (cdna_hip_programming.md Guideline 16, R2): the last block of a pass stores its 2*NS
// limbs into every rank's inbox with one store each, sweeps its own inbox until every source's tag is
// this pass's, adds the limbs and publishes the totals to its host - no collective launch, no separate
// flag, no ordering requirement between the stores.  Two parities: a rank can be at most one pass ahead
// of a peer that has not read the previous pass yet.  One more granule carries a digest of the
// challenges the pass folds; ranks that were fed different challenges fail loudly instead of proving
// different statements.
constexpr int kMaxPeers = 8;
constexpr int kInboxWords = 64 + 512; // 64 for the passes with up to 27 cells (+ digest, gather flag), then the wide part
constexpr int kInboxWide = 64;       // first granule of the wide part: 2 x 243 limbs of a five-round pass
constexpr int kInboxDigest = 56;     // granule index of the challenge digest
constexpr int kInboxGather = 57;     // granule index of the table-gather flag
// failure codes of an exchange (mailbox word kMailboxErr): a timeout carries the source rank it waited for in bits 8..15
// and outranks a digest mismatch wherever codes are combined with max()
constexpr int kXchgDigest = 2, kXchgTimeout = 0x40000000;
struct PeerX {
  u64* inbox[kMaxPeers] = {};   // inbox[q]: rank q's inbox as this process maps it (q == rank: the local one)
  int world = 0;                // 0: no in-kernel exchange
  int rank = 0;
  unsigned tag = 0;             // this pass's exchange tag: the same on every rank, never 0
  unsigned digest = 0;
  u64 spin_ticks = 0;           // bound of the sweep (wall clock, 100 MHz)
};
struct PassOut {
  u64* partials;
  int n_rows;
  unsigned* ticket;
  unsigned ticket_base;
  u64* sums_dev;
  u64* mailbox;
  u64 seq;
  PeerX px;
};

__device__ __forceinline__ void publish_value(const PassOut& o, int s, u64 v) {
  if (o.mailbox) {
    __hip_atomic_store(o.mailbox + 2 * s, v & 0xFFFFFFFFull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(o.mailbox + 2 * s + 1, v >> 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    write_split(o.sums_dev, s, v);
  }
}
// after a workgroup barrier that follows every publish_value of the block
__device__ __forceinline__ void publish_seq(const PassOut& o) {
  if (o.mailbox && threadIdx.x == 0)
    __hip_atomic_store(o.mailbox + kMailboxSeq, o.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Cross-rank exchange by ONE workgroup (the pass's last block): xl[0 .. 2*NS) are this rank's limbs.
// Leaves the limb totals in the host mailbox; every thread of the block must call it.
template <int NS>
__device__ __forceinline__ void exchange_and_publish(const PassOut& o, u64* xl) {
  const PeerX& px = o.px;
  const int lane = threadIdx.x;
  __syncthreads();   // xl is complete
  if (threadIdx.x < kWave) {
    const bool mine = lane < 2 * NS || lane == kInboxDigest;
    const u64 val = (lane < 2 * NS) ? xl[lane] : (u64)px.digest;
    const u64 granule = ((u64)px.tag << 32) | (val & 0xFFFFFFFFull);
    const size_t slot = ((size_t)(px.tag & 1u) * kMaxPeers + (size_t)px.rank) * kInboxWords + (size_t)lane;
    if (mine) {
      for (int q = 0; q < px.world; ++q)
        __hip_atomic_store(px.inbox[q] + slot, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // sweep the local inbox: one granule per source rank and lane
    u64 total = 0;
    int err = 0;
    const unsigned long long t0 = wall_clock64();
    if (mine) {
      const u64* base = px.inbox[px.rank] + (size_t)(px.tag & 1u) * kMaxPeers * kInboxWords + (size_t)lane;
      for (int r = 0; r < px.world && !err; ++r) {
        unsigned spins = 0;
        while (true) {
          const u64 g = __hip_atomic_load(base + (size_t)r * kInboxWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if ((unsigned)(g >> 32) == px.tag) {
            if (lane == kInboxDigest) err = ((unsigned)g != px.digest) ? kXchgDigest : 0;
            else total += g & 0xFFFFFFFFull;
            break;
          }
          if ((++spins & 31) == 0 && wall_clock64() - t0 > px.spin_ticks) {
            // diagnosis for the host's message: which source, and the tag its slot still held
            err = kXchgTimeout | (r << 8) | ((int)((g >> 32) & 0x3FFF) << 16);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    if (lane < 2 * NS) __hip_atomic_store(o.mailbox + lane, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // any lane's failure reaches the host before the sequence word does
    const int any = __any(err != 0) ? 1 : 0;
    int code = err;
#pragma unroll
    for (int off = kWave / 2; off >= 1; off >>= 1) code = max(code, __shfl_down(code, off, kWave));
    if (lane == 0) __hip_atomic_store(o.mailbox + kMailboxErr, (u64)(any ? code : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  publish_seq(o);
}

// Tail of every pass: res[0] of thread s < NS holds the block's residue of sum s.
template <class F, int NS, int BS = kBlock>
__device__ __forceinline__ void finish_pass(const F& f, const PassOut& o, u64 my_res, int* lds_flag) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  __shared__ u64 xl[2 * NS + 2];   // this rank's limbs on their way to the peers (sharded passes only)
  const bool xchg = o.px.world > 0;
  if (gridDim.x == 1) {
    if (xchg) {
      if (threadIdx.x < NS) write_split(xl, threadIdx.x, my_res);
      exchange_and_publish<NS>(o, xl);
      return;
    }
    if (threadIdx.x < NS) publish_value(o, threadIdx.x, my_res);
    __syncthreads();
    publish_seq(o);
    return;
  }
  if (threadIdx.x < NS)
    __hip_atomic_store(o.partials + (size_t)threadIdx.x * o.n_rows + blockIdx.x, my_res, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    // wave 0 holds every storing lane: drain its write-through stores, then signal.  No
    // agent-scope release fence: that is a whole-L2 write-back per block (~2-6 us each and
    // 2048 of them per launch); sc1 stores + drain is Guideline 16's R1 form.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(o.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t - o.ticket_base == gridDim.x - 1) ? 1 : 0;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *lds_flag = last;
  }
  __syncthreads();
  if (!*lds_flag) return;
  const int n_blocks = gridDim.x;
  if constexpr (NS >= 9) {
    // thread = (cell, slice), the sixteen slices of a cell in adjacent lanes: a load instruction of a wave reads four
    // 128-byte lines (cell-major partials: slices = consecutive blocks), not 64 scattered words - with the cells in
    // adjacent lanes the same loads took 4.6 us for 256 x 27 partials (profiles/r03_pass_block_stamps.txt), two thirds
    // of the last block's work.  Every load of a thread is in flight at once; the slices are summed through LDS.
    constexpr int K = 16, U = 16;
    static_assert(BS >= K * NS, "sixteen slices per cell");
    __shared__ u64 fin[K * NS];
    const int row = threadIdx.x / K, slice = threadIdx.x % K;
    if (row < NS) {
      const u64* src = o.partials + (size_t)row * o.n_rows;
      u64 part = 0;
      for (int b0 = slice; b0 < n_blocks; b0 += K * U) {
        u64 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int b = b0 + u * K;
          x[u] = (b < n_blocks) ? __hip_atomic_load(src + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
#pragma unroll
        for (int u = 0; u < U / 2; ++u) x[u] = f.add(x[u], x[u + U / 2]);
#pragma unroll
        for (int u = 0; u < U / 4; ++u) x[u] = f.add(x[u], x[u + U / 4]);
#pragma unroll
        for (int u = 0; u < U / 4; ++u) part = f.add(part, x[u]);
      }
      fin[slice * NS + row] = part;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
      u64 t = 0;
#pragma unroll
      for (int q = 0; q < K; ++q) t = f.add(t, fin[q * NS + threadIdx.x]);
      if (xchg) write_split(xl, threadIdx.x, t);
      else publish_value(o, threadIdx.x, t);
    }
  } else {
    // each wave takes the rows wave, wave+4, ... two at a time: the loads of one row are a chain of
    // dependent rounds (~1 us each from L2), so two rows in flight halve the serial tail
    constexpr int kWavesPerBlock = kBlock / kWave;
    for (int s = wave; s < NS; s += 2 * kWavesPerBlock) {
      const int s2 = s + kWavesPerBlock;
      const bool two = s2 < NS;
      const u64* row0 = o.partials + (size_t)s * o.n_rows;
      const u64* row1 = o.partials + (size_t)(two ? s2 : s) * o.n_rows;
      u64 a0 = 0, a1 = 0, c0 = 0, c1 = 0;
      int b = lane;
      for (; b + 3 * kWave < n_blocks; b += 4 * kWave) {  // eight loads in flight per lane
        u64 x[4], y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          x[q] = __hip_atomic_load(row0 + b + q * kWave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          y[q] = __hip_atomic_load(row1 + b + q * kWave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a0 = f.add(a0, f.add(x[0], x[2])); a1 = f.add(a1, f.add(x[1], x[3]));
        c0 = f.add(c0, f.add(y[0], y[2])); c1 = f.add(c1, f.add(y[1], y[3]));
      }
      for (; b < n_blocks; b += kWave) {
        a0 = f.add(a0, __hip_atomic_load(row0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        c0 = f.add(c0, __hip_atomic_load(row1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      u64 t = f.add(a0, a1), u = f.add(c0, c1);
#pragma unroll
      for (int off = kWave / 2; off >= 1; off >>= 1) {
        t = f.add(t, shfl_down_u64(t, off));
        u = f.add(u, shfl_down_u64(u, off));
      }
      if (lane == 0) {
        if (xchg) {
          write_split(xl, s, t);
          if (two) write_split(xl, s2, u);
        } else {
          publish_value(o, s, t);
          if (two) publish_value(o, s2, u);
        }
      }
    }
  }
  if (xchg) {
    exchange_and_publish<NS>(o, xl);
    return;
  }
  __syncthreads();
  publish_seq(o);
}

// Per-thread accumulators of NS cells -> block sums: thread s ends up with the block's sum of cell s.
// Nine sums at a time (a 27-cell grid would otherwise hold 27 residues next to the accumulators they come
// from).  lds: kWaves * min(NS, 9) words of scratch.
template <class F, int NS>
__device__ __forceinline__ u64 reduce_cells(const F& f, const typename F::Acc (&acc)[NS], u64* lds) {
  constexpr int CH = (NS < 9) ? NS : 9;
  u64 mine = 0;
#pragma unroll
  for (int c0 = 0; c0 < NS; c0 += CH) {
    u64 res[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) res[s] = f.acc_get(acc[c0 + s]);
    if (c0 > 0) __syncthreads();  // the previous chunk's scratch has been read
    block_reduce<F, CH>(f, res, lds);
    if ((int)threadIdx.x >= c0 && (int)threadIdx.x < c0 + CH) mine = res[0];
    if constexpr (NS > CH) {
      // block_reduce leaves chunk sum s in thread s; hand it to thread c0 + s
      __syncthreads();
      if (threadIdx.x < CH) lds[threadIdx.x] = res[0];
      __syncthreads();
      if ((int)threadIdx.x >= c0 && (int)threadIdx.x < c0 + CH) mine = lds[threadIdx.x - c0];
    }
  }
  return mine;
}

// The same through LDS, on the RAW accumulators: a thread that turns 27 lazy sums into residues and then takes
// part in 27 x 6 shuffle rounds spends ~2800 instructions on it, on a wave that by then issues alone (~10 us of
// every launch of the 27-cell pass, ~3.5 us of a 9-cell one: nothing at 2^28 entries, 10 % of a pass on a 2^25-entry
// shard).  Here the accumulators of eight cells at a time go to LDS, thread (cell, part) adds eight of them as
// integers, 32 lanes finish with five shuffle rounds and ONE lane per cell reduces to a residue: ~170 instructions
// per chunk of BS / 32 cells.  scratch: (BS / 32) * BS accumulators (BS = threads of the block); out: NS words.
template <class A>
__device__ __forceinline__ A shfl_down_acc(const A& a, int off) {
  static_assert(sizeof(A) % 4 == 0, "accumulator words");
  A r;
  const unsigned* src = reinterpret_cast<const unsigned*>(&a);
  unsigned* dst = reinterpret_cast<unsigned*>(&r);
#pragma unroll
  for (int w = 0; w < (int)(sizeof(A) / 4); ++w) dst[w] = (unsigned)__shfl_down((int)src[w], off, kWave);
  return r;
}
template <class F, int NS, int BS = kBlock>
__device__ __forceinline__ u64 reduce_cells_lds(const F& f, const typename F::Acc (&acc)[NS], typename F::Acc* scratch, u64* out) {
  typedef typename F::Acc Acc;
  constexpr int CH = BS / 32;   // cells per chunk: 32 threads sum one cell
  const int tid = threadIdx.x, cell = tid >> 5, part = tid & 31;
#pragma unroll
  for (int c0 = 0; c0 < NS; c0 += CH) {
    constexpr int kRest = NS % CH;
    const int n = (c0 + CH <= NS) ? CH : kRest;
    if (c0 > 0) __syncthreads();   // the previous chunk's accumulators have been read
#pragma unroll
    for (int s = 0; s < CH; ++s)
      if (s < n) scratch[s * BS + tid] = acc[(c0 + s < NS) ? c0 + s : 0];
    __syncthreads();
    if (cell < n) {
      Acc t = scratch[cell * BS + part];
#pragma unroll
      for (int k = 1; k < BS / 32; ++k) f.acc_add(t, scratch[cell * BS + part + 32 * k]);
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) {
        const Acc o = shfl_down_acc(t, off);
        f.acc_add(t, o);
      }
      if (part == 0) out[c0 + cell] = f.acc_get(t);
    }
  }
  __syncthreads();
  return tid < NS ? out[tid] : 0;
}

// Threads per block of pass_kernel<., KF, KS, .>.  The arithmetic-heavy instantiations hold two or three waves per SIMD
// (their registers allow no more) and get ALL of a CU's waves into ONE block, so that the waves of a SIMD can share
// their work through LDS (see the tile loop); the light ones keep 256 threads and several blocks per CU.
__host__ __device__ constexpr int pass_block_threads(int kf, int ks) {
  return (ks == 3 || (kf == 3 && ks == 2)) ? 512 : (kf == 2 && ks == 2) ? 768 : kBlock;
}

// NT: bit 0 = nontemporal loads, bit 1 = nontemporal stores (see ld16 / st16)
template <class F, int KF, int KS, int NT>
__global__ void __launch_bounds__(pass_block_threads(KF, KS))
pass_kernel(F f, const u64* __restrict__ A, const u64* __restrict__ B, u64* __restrict__ A2,
            u64* __restrict__ B2, FoldW fw, size_t n_units, PassOut out) {
  constexpr bool kNtLoad = (NT & 1) != 0, kNtStore = (NT & 2) != 0;
  constexpr int IN = 1 << (KF + KS), OUT = 1 << KS, NS = (KS == 1) ? 3 : (KS == 2) ? 9 : 27;
  constexpr int NP = IN / 2, NPO = OUT / 2;  // 16-byte pieces per lane, in and out
  constexpr int BS = pass_block_threads(KF, KS), kWaves = BS / kWave;
  static_assert(BS == kBlock || NS >= 9, "reduce_cells (the KS = 1 passes) is written for 256 threads");
  // the tile transposes; after the loop the same bytes hold a chunk of every thread's accumulators (reduce_cells_lds)
  constexpr int kTransposeSlots = (NP > 1 || NPO > 1) ? kWaves * kWave * NP : 1;
  constexpr int kReduceSlots = (NS >= 9) ? (int)((NS < BS / 32 ? NS : BS / 32) * BS * sizeof(typename F::Acc) / sizeof(ull2)) : 1;
  __shared__ ull2 lds_t[kTransposeSlots > kReduceSlots ? kTransposeSlots : kReduceSlots];
  __shared__ u64 lds[kWaves * NS];
  __shared__ int lds_flag;
  __shared__ unsigned lds_next;   // the block's tile counter
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  ull2* const my_lds = lds_t + ((NP > 1 || NPO > 1) ? wave * kWave * NP : 0);

  typename F::Acc acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) f.acc_zero(acc[s]);

  const size_t n_tiles = (n_units + kWave - 1) / kWave;
  const size_t in_pieces = n_units * NP, out_pieces = n_units * NPO;
  const ull2* __restrict__ Ap = reinterpret_cast<const ull2*>(A);
  const ull2* __restrict__ Bp = reinterpret_cast<const ull2*>(B);
  ull2* __restrict__ A2p = reinterpret_cast<ull2*>(A2);
  ull2* __restrict__ B2p = reinterpret_cast<ull2*>(B2);

  // inactive lanes carry zeros: they add nothing to the sums and store nothing.  Tables far
  // larger than the 256 MiB Infinity Cache are read once: stream them (nontemporal).
  auto load_tile = [&](size_t tile, ull2 (&pa)[NP], ull2 (&pb)[NP]) {
    const size_t q0 = tile * kWave * NP;
    if (q0 + (size_t)kWave * NP <= in_pieces) {  // full tile (wave-uniform): no per-piece test
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const size_t q = q0 + (size_t)k * kWave + lane;
        pa[k] = ld16<kNtLoad>(Ap + q);
        pb[k] = ld16<kNtLoad>(Bp + q);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const size_t q = q0 + (size_t)k * kWave + lane;
        const ull2 zero = {0, 0};
        pa[k] = (q < in_pieces) ? Ap[q] : zero;
        pb[k] = (q < in_pieces) ? Bp[q] : zero;
      }
    }
  };
  // KF = 3: a run is 2^(3+KS) entries; read it back from LDS one output (8 entries) at a time so
  // that only the staged pieces and OUT folded values are live, not the whole run twice.
  auto stage_and_fold3 = [&](ull2 (&p)[NP], u64 (&t)[IN]) {
    if constexpr (KF == 3) {  // (the body only instantiates for run lengths swz_slot supports)
#pragma unroll
      for (int k = 0; k < NP; ++k) my_lds[swz_slot<NP>(64 * k + lane)] = p[k];
      wave_lds_fence();
#pragma unroll
      for (int o = 0; o < OUT; ++o) {
        u64 v[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const ull2 x = my_lds[swz_slot<NP>(NP * lane + 4 * o + m)];
          v[2 * m] = x.x; v[2 * m + 1] = x.y;
        }
        fold_run<F, 3, 8>(f, v, fw);
        t[o] = v[0];
      }
      wave_lds_fence();
    }
  };
  // The 27-cell grid runs at two waves per SIMD and is ALU-heavy: it cannot count on other
  // waves to cover its loads, so it fetches the wave's next tile before it starts on the
  // arithmetic of the current one.  (The three-variable fold has no registers for a second tile; asking for one table
  // of the next tile at a time, while the other table is folded out of LDS, was measured and gave nothing: that pass is
  // not waiting for its own loads.)
  constexpr bool kPrefetch = (KS == 3);
  auto process_tile = [&](size_t tile, size_t next, ull2 (&pa)[NP], ull2 (&pb)[NP]) {
    u64 a[IN], b[IN];
    if constexpr (KF == 3) {
      stage_and_fold3(pa, a);
      stage_and_fold3(pb, b);
    } else {
      transpose_to_runs<NP>(my_lds, pa, lane);
      transpose_to_runs<NP>(my_lds, pb, lane);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        a[2 * k] = pa[k].x; a[2 * k + 1] = pa[k].y;
        b[2 * k] = pb[k].x; b[2 * k + 1] = pb[k].y;
      }
      if constexpr (kPrefetch) {
        if (next < n_tiles) load_tile(next, pa, pb);
      }
      fold_run<F, KF, IN>(f, a, fw);
      fold_run<F, KF, IN>(f, b, fw);
    }
    if constexpr (KF > 0) {
      ull2 oa[NPO], ob[NPO];
#pragma unroll
      for (int m = 0; m < NPO; ++m) {
        oa[m].x = a[2 * m]; oa[m].y = a[2 * m + 1];
        ob[m].x = b[2 * m]; ob[m].y = b[2 * m + 1];
      }
      transpose_to_pieces<NPO>(my_lds, oa, lane);
      transpose_to_pieces<NPO>(my_lds, ob, lane);
      const size_t o0 = tile * kWave * NPO;
#pragma unroll
      for (int k = 0; k < NPO; ++k) {
        const size_t q = o0 + (size_t)k * kWave + lane;
        if (q < out_pieces) {
          st16<kNtStore>(A2p + q, oa[k]);
          st16<kNtStore>(B2p + q, ob[k]);
        }
      }
    }
    if constexpr (KS == 3) accumulate_octet<F>(f, acc, a, b);
    else accumulate_run<F, KS>(f, acc, a, b);
  };

  // Tiles are not dealt out in advance.  The waves that share a SIMD are issued oldest-first: with a fixed share each,
  // the older wave runs at the pace of its arithmetic, the younger one gets the memory bandwidth that is left and
  // then finishes its share ALONE, at half the SIMD's issue rate (per-block stamps of an n = 28 first pass with
  // two 256-thread blocks per CU: blocks 0..255 left the loop after 459 us, blocks 256..511 - the second block of
  // every CU - after 707 us; profiles/r03_pass_block_stamps.txt).  So a block's waves draw their tiles from a
  // counter in LDS (block b takes the tiles c * gridDim + b, c = 0, 1, ...): whoever is faster takes more, and the
  // waves of a SIMD finish together.  An LDS atomic is ~100 cycles and not in the way of the global loads (a
  // global counter per CU was tried: its returns queue behind the tile loads and cost more than the balance gave).
  if (threadIdx.x == 0) lds_next = 0;
  __syncthreads();
  auto next_tile = [&]() -> size_t {
    unsigned c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(&lds_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (size_t)__builtin_amdgcn_readfirstlane(c) * gridDim.x + blockIdx.x;
  };
  if constexpr (kPrefetch) {
    ull2 pa[NP], pb[NP];
    size_t tile = next_tile();
    if (tile < n_tiles) load_tile(tile, pa, pb);
    while (tile < n_tiles) {
      const size_t next = next_tile();
      process_tile(tile, next, pa, pb);
      tile = next;
    }
  } else {
    for (size_t tile = next_tile(); tile < n_tiles; tile = next_tile()) {
      ull2 pa[NP], pb[NP];
      load_tile(tile, pa, pb);
      process_tile(tile, 0, pa, pb);
    }
  }

  u64 mine;
  if constexpr (NS >= 9) {
    __syncthreads();   // every wave is done with its transposes
    mine = reduce_cells_lds<F, NS, BS>(f, acc, reinterpret_cast<typename F::Acc*>(lds_t), lds);
  } else {
    mine = reduce_cells<F, NS>(f, acc, lds);
  }
  finish_pass<F, NS, BS>(f, out, mine, &lds_flag);
}

// ------------------------------------------------------------------------------------
// Up to FIVE rounds per pass on the smaller tables of a proof (folded size <= 2^20 entries).
//
// Below ~2^21 entries a pass is latency - launch, one dependent chain of work, hand-off to the host - and not
// bytes; at 8 GPUs (2^25-entry shards) that is more than a third of the proof.  Two things shorten it: fewer
// passes (a pass that serves KS rounds accumulates the 3^KS-cell grid in the {0,1,inf} basis; for KS = 4, 5 that is
// 81 / 243 cells over groups of 16 / 32 folded entries - 5 to 7.6 products per entry, nothing at these sizes) and a
// shorter dependent chain inside a pass (round 1's tail kernel let one thread in eight walk all 27 cells of its
// octet, ~600 instructions on a wave that issues alone at half rate, tools/valu_rate.hip).
//
// Every WAVE works alone.  One wave iteration takes 32 consecutive folded entries of both tables:
//  1. fold: lane = table x entry - all 64 lanes fold one entry, sum_c w[c] * in[2^kf i + c] (kf = 0..5 pending
//     challenges, run-time; all loads of an entry in flight, one lazy sum, one reduction), and store it to the
//     folded table and to its place in the wave's extension arrays
//  2. extend: ext[table][group][cell], cell = sum_j d_j 3^(KS-1-j), d_j in {0,1,inf} the evaluation point of the
//     group's variable j (variable 0 = index bit 0, the round served first: the slowest axis, as in pass_kernel);
//     the 32 entries are 2^(5-KS) groups of 2^KS.  Level j fills the cells with d_j = inf from d_j = 1 minus
//     d_j = 0: 2 * groups * 3^j * 2^(KS-1-j) subtractions, at most three per lane, whose LDS addresses are the same
//     in every iteration and are decoded once; levels are separated by wave-level LDS ordering only (the arrays
//     are private to the wave: no barrier)
//  3. multiply: the (group, cell) pairs - at most 243 - by lane p, p + 64, p + 128, p + 192 into four lazy
//     accumulators per lane that live across the wave's iterations
// ~120 VGPRs and 4 KiB of LDS per wave: four waves per SIMD cover each other's latencies.  KS is a template
// parameter (constant strides), kf a run-time switch.  End: accumulators -> residues, waves and groups added through
// LDS, thread c < 3^KS holds cell c of the block.
// wgrid_pass_kernel: rows of 256 words per block, two ticket levels (groups of 32 blocks, then the groups;
// Guideline 16 R1 as in finish_pass), each one round of up to 32 loads per thread; the block that finishes last resets
// the counters and hands the cells on: as whole residues in the wide part of the host mailbox and then the sequence
// word (unsharded passes, and sharded ones on a host transport - the host splits and sums the limbs); through the
// in-kernel exchange (sharded passes on the peer transport, exchange_wide); or as split limbs in device memory for the
// collective that follows on the stream (sharded passes on RCCL, WgOut::limbs_dev).
constexpr int kGridChunk = 256;          // words per row of partials (>= 243 cells)
constexpr int kGridMaxVars = 5;
constexpr int kGridMaxCells = 243;
constexpr int kMailboxWide = 64;         // first word of the wide area (kGridMaxCells words)
constexpr int kMailboxWords = kMailboxWide + 512;   // 243 residues, or 486 limb totals of a sharded pass
constexpr int kWgEntries = 32;           // folded entries per table and wave iteration
constexpr int kWgGroupBlocks = 32;       // blocks per first-level ticket
struct GridW {
  u64 w[1 << kGridMaxVars];   // w[c] = eq((r_0 .. r_{kf-1}), c); w[0] = 1 for kf = 0
};
struct WgOut {
  u64* partials;     // [blocks][kGridChunk]
  u64* group_rows;   // [groups of 32 blocks][kGridChunk]
  unsigned* tickets; // [0]: groups done; [1 + g]: blocks of group g done; all zero between launches
  u64* mailbox;
  u64 seq;
  u64* limbs_dev;    // non-null: leave the cells as 2 x 3^KS split limbs here (device memory) and publish nothing
  PeerX px;          // world > 0: a sharded pass - the cells are exchanged with the peers before they are published
};
// LDS hand-off between the lanes of ONE wave: a wave's LDS operations execute in order, so all that is needed is
// that the earlier ones have been issued and returned and that the compiler keeps the order.  (wave_lds_fence()
// is a workgroup-scope fence: it would also wait for the wave's global stores - here the folded entries on
// their way out, which nobody in this kernel waits for.)
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
// folded entry i of one table: sum_c w[c] * in[2^KF i + c], stored to the folded table
template <class F, int KF>
__device__ __forceinline__ u64 grid_fold1(const F& f, const u64* __restrict__ T, u64* __restrict__ T2, const GridW& gw, size_t i) {
  constexpr int FAN = 1 << KF, NPIECE = FAN / 2;
  const ull2* __restrict__ pt = reinterpret_cast<const ull2*>(T + i * FAN);
  ull2 x[NPIECE];
#pragma unroll
  for (int m = 0; m < NPIECE; ++m) x[m] = pt[m];
  typename F::Acc3 s;
  f.acc3_zero(s);
#pragma unroll
  for (int m = 0; m < NPIECE; ++m) {
    f.acc3_mac(s, x[m].x, gw.w[2 * m]);
    f.acc3_mac(s, x[m].y, gw.w[2 * m + 1]);
  }
  const u64 v = f.acc3_get(s);
  T2[i] = v;
  return v;
}
// fix_variables of four or five variables of one SMALL table in one launch (thread = output entry, its 2^kf inputs a
// contiguous run: fine for tables that sit in the caches, where a chain of <= 3-variable folds is two launches)
template <class F>
__global__ void __launch_bounds__(kBlock)
fold_wide_kernel(F f, const u64* __restrict__ T, u64* __restrict__ T2, GridW gw, int kf, size_t n_out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_out; i += (size_t)gridDim.x * kBlock) {
    if (kf == 4) (void)grid_fold1<F, 4>(f, T, T2, gw, i);
    else (void)grid_fold1<F, 5>(f, T, T2, gw, i);
  }
}

// the block's sums: thread c < 3^KS returns cell c
template <class F, int KS, bool PF>
__device__ __forceinline__ u64 wgrid_body(const F& f, const u64* __restrict__ A, const u64* __restrict__ B, u64* __restrict__ A2,
                                          u64* __restrict__ B2, const GridW& gw, int kf, size_t n_out) {
  constexpr int kWaves = kBlock / kWave;
  constexpr int kPow3[6] = {1, 3, 9, 27, 81, 243};
  constexpr int cells = kPow3[KS], G = 1 << KS, gpi = kWgEntries >> KS, pairs = gpi * cells;
  __shared__ u64 ext[kWaves][2][kGridChunk];   // wave-private: [table][group][cell]
  __shared__ u64 red[kWaves][kGridChunk];
  __shared__ int cell_of[kWgEntries], suffix_of[kWgEntries];
  typedef __attribute__((address_space(3))) u64 lds_u64;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  if (tid < kWgEntries) {
    // cell of a group's entry e (its bits are the points of the group's variables, variable 0 = bit 0) and the
    // cell offset of a suffix s whose bit m is the point of variable KS-1-m
    int c = 0, u = 0, p3 = 1;
    for (int m = 0; m < KS; ++m) {
      c += ((tid >> (KS - 1 - m)) & 1) * p3;
      u += ((tid >> m) & 1) * p3;
      p3 *= 3;
    }
    cell_of[tid] = c;
    suffix_of[tid] = u;
  }
  __syncthreads();
  u64* const ef = &ext[wave][0][0];   // [table][256], flat
  // fold role of the lane: table and entry of the iteration; its place in the extension array
  const int tbl = lane >> 5, ent = lane & (kWgEntries - 1);
  const int slot = tbl * kGridChunk + (ent >> KS) * cells + cell_of[ent & (G - 1)];
  const u64* __restrict__ src = tbl ? B : A;
  u64* __restrict__ dst = tbl ? B2 : A2;
  // step[j][q] = bit 31 | LDS byte address of the d_j = 0 cell of the lane's q-th subtraction of level j, or 0
  unsigned step[KS][3];
#pragma clang loop unroll(full)
  for (int j = 0; j < KS; ++j) {
    const int low = KS - 1 - j, pj = kPow3[j], stride = kPow3[low], items = (gpi * pj) << low;   // per table
    const unsigned inv = (1u << 20) / (unsigned)pj + 1u;   // t / pj for t < 4096, pj in {1,3,9,27,81}: exact
#pragma clang loop unroll(full)
    for (int q = 0; q < 3; ++q) {
      const int idx = lane + kWave * q;
      unsigned d = 0;
      if (idx < 2 * items) {
        const int tb = idx >= items ? 1 : 0, id = idx - tb * items;
        const int sfx = id & ((1 << low) - 1), t = id >> low;
        const int g = (int)(((unsigned)t * inv) >> 20), p = t - g * pj;
        d = 0x80000000u | (unsigned)(size_t)(lds_u64*)(ef + tb * kGridChunk + g * cells + p * 3 * stride + suffix_of[sfx]);
      }
      step[j][q] = d;
    }
  }
  typename F::Acc acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) f.acc_zero(acc[k]);

  if constexpr (PF) {
    const size_t n_iter = (n_out + kWgEntries - 1) / kWgEntries;
    const size_t it0 = (size_t)blockIdx.x * kWaves + wave, it_stride = (size_t)gridDim.x * kWaves;
    // kf = 0 and kf = 2 are the fan-ins that occur on tables large enough for a wave to take several iterations (first
    // passes of small proofs; the pass behind a (., 2) pass_kernel launch).  For those the NEXT iteration's global loads
    // are requested before this iteration's fold, so that their ~0.8 us latency runs under the ~300 instructions and
    // nine LDS round trips of an iteration instead of in front of them.  A separate instantiation (PF; the host picks it
    // for kf = 0 / 2 on tables with more iterations than waves): next to the 64 load registers of the kf = 5 fold the
    // prefetch registers cost a wave per SIMD (139-165 VGPRs instead of 99-120).
    ull2 nx0 = {0, 0}, nx1 = {0, 0};
    auto request = [&](size_t i) {
      if (kf == 0) {
        nx0.x = src[i];
      } else {
        const ull2* __restrict__ pt = reinterpret_cast<const ull2*>(src + i * 4);
        nx0 = pt[0];
        nx1 = pt[1];
      }
    };
    bool have = it0 < n_iter && it0 * kWgEntries + ent < n_out;
    if (have) request(it0 * kWgEntries + ent);
    for (size_t it = it0; it < n_iter; it += it_stride) {
      const size_t i = it * kWgEntries + ent;
      u64 v = 0;   // entries past the end of a short table are zeros: they add nothing to any cell
      {
        const ull2 x0 = nx0, x1 = nx1;
        const bool mine = have;
        const size_t ni = (it + it_stride) * kWgEntries + ent;
        have = it + it_stride < n_iter && ni < n_out;
        if (have) request(ni);
        if (mine) {
          if (kf == 0) {
            v = x0.x;
          } else {
            typename F::Acc3 s;
            f.acc3_zero(s);
            f.acc3_mac(s, x0.x, gw.w[0]);
            f.acc3_mac(s, x0.y, gw.w[1]);
            f.acc3_mac(s, x1.x, gw.w[2]);
            f.acc3_mac(s, x1.y, gw.w[3]);
            v = f.acc3_get(s);
            dst[i] = v;
          }
        }
      }
      ef[slot] = v;
      wave_lds_sync();
#pragma clang loop unroll(full)
      for (int j = 0; j < KS; ++j) {
        const int st = kPow3[KS - 1 - j];
#pragma clang loop unroll(full)
        for (int q = 0; q < 3; ++q) {
          if (2 * ((gpi * kPow3[j]) << (KS - 1 - j)) > kWave * q) {   // does any lane have a q-th step at this level?
            const unsigned d = step[j][q];
            if (d != 0) {
              lds_u64* const x = (lds_u64*)(size_t)(d & 0x7FFFFFFFu);
              x[2 * st] = f.sub(x[st], x[0]);
            }
          }
        }
        wave_lds_sync();
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int p = lane + kWave * k;
        if (p < pairs) f.acc_mac(acc[k], ef[p], ef[kGridChunk + p]);
      }
      wave_lds_sync();    // the next iteration overwrites the arrays
    }
  } else {
    const size_t n_iter = (n_out + kWgEntries - 1) / kWgEntries;
    for (size_t it = (size_t)blockIdx.x * kWaves + wave; it < n_iter; it += (size_t)gridDim.x * kWaves) {
      const size_t i = it * kWgEntries + ent;
      u64 v = 0;   // entries past the end of a short table are zeros: they add nothing to any cell
      if (i < n_out) {
        switch (kf) {   // compile-time fan-in: all loads of an entry are in flight together
          case 0: v = src[i]; break;
          case 1: v = grid_fold1<F, 1>(f, src, dst, gw, i); break;
          case 2: v = grid_fold1<F, 2>(f, src, dst, gw, i); break;
          case 3: v = grid_fold1<F, 3>(f, src, dst, gw, i); break;
          case 4: v = grid_fold1<F, 4>(f, src, dst, gw, i); break;
          default: v = grid_fold1<F, 5>(f, src, dst, gw, i); break;
        }
      }
      ef[slot] = v;
      wave_lds_sync();
#pragma clang loop unroll(full)
      for (int j = 0; j < KS; ++j) {
        const int st = kPow3[KS - 1 - j];
#pragma clang loop unroll(full)
        for (int q = 0; q < 3; ++q) {
          if (2 * ((gpi * kPow3[j]) << (KS - 1 - j)) > kWave * q) {   // does any lane have a q-th step at this level?
            const unsigned d = step[j][q];
            if (d != 0) {
              lds_u64* const x = (lds_u64*)(size_t)(d & 0x7FFFFFFFu);
              x[2 * st] = f.sub(x[st], x[0]);
            }
          }
        }
        wave_lds_sync();
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int p = lane + kWave * k;
        if (p < pairs) f.acc_mac(acc[k], ef[p], ef[kGridChunk + p]);
      }
      wave_lds_sync();    // the next iteration overwrites the arrays
    }
  }

#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = lane + kWave * k;
    red[wave][p] = (p < pairs) ? f.acc_get(acc[k]) : 0;
  }
  __syncthreads();
  u64 total = 0;
  if (tid < cells) {
    for (int w = 0; w < kWaves; ++w)
      for (int g = 0; g < gpi; ++g) total = f.add(total, red[w][g * cells + tid]);
  }
  return total;
}

// The in-kernel exchange of finish_pass (PeerX) for up to 243 cells, by the whole last block: thread c owns cell c,
// i.e. the granule PAIR kInboxWide + 2c (low limb), + 2c + 1 (high limb) of every inbox, written and polled as ONE
// 16-byte access - a wave then moves whole 64-byte lines.  (Two 8-byte stores per thread at a 16-byte stride leave
// every line of the uncached inbox half written: measured 50 us per pass for the 486 granules, against ~1 us.)
// Each half still carries its own tag, so a torn pair is just a pair that has not arrived yet.  Leaves the limb
// TOTALS in the wide mailbox (the host recombines them mod p), the error word and then the sequence word.
__device__ __forceinline__ void st16_system(u64* p, ull2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ ull2 ld16_system(const u64* p) {
  ull2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int CELLS>
__device__ __forceinline__ void exchange_wide(const WgOut& o, u64 total) {
  const PeerX& px = o.px;
  const int tid = threadIdx.x;
  const size_t par = (size_t)(px.tag & 1u) * kMaxPeers * kInboxWords;
  int err = 0;
  u64 lo = 0, hi = 0;
  if (tid < CELLS) {
    const size_t mine = par + (size_t)px.rank * kInboxWords + kInboxWide + 2 * (size_t)tid;
    const ull2 pair = {((u64)px.tag << 32) | (total & 0xFFFFFFFFull), ((u64)px.tag << 32) | (total >> 32)};
    for (int q = 0; q < px.world; ++q) st16_system(px.inbox[q] + mine, pair);
  }
  if (tid == CELLS) {   // one more thread carries the digest of the challenges
    const u64 g = ((u64)px.tag << 32) | (u64)px.digest;
    for (int q = 0; q < px.world; ++q)
      __hip_atomic_store(px.inbox[q] + par + (size_t)px.rank * kInboxWords + kInboxDigest, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const unsigned long long t0 = wall_clock64();
  if (tid < CELLS) {
    const u64* base = px.inbox[px.rank] + par + kInboxWide + 2 * (size_t)tid;
    for (int r = 0; r < px.world && !err; ++r) {
      unsigned spins = 0;
      while (true) {
        const ull2 g = ld16_system(base + (size_t)r * kInboxWords);
        if ((unsigned)(g.x >> 32) == px.tag && (unsigned)(g.y >> 32) == px.tag) {
          lo += g.x & 0xFFFFFFFFull;
          hi += g.y & 0xFFFFFFFFull;
          break;
        }
        if ((++spins & 31) == 0 && wall_clock64() - t0 > px.spin_ticks) {
          err = kXchgTimeout | (r << 8);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    st16_system(o.mailbox + kMailboxWide + 2 * tid, ull2{lo, hi});
  } else if (tid == CELLS) {
    const u64* base = px.inbox[px.rank] + par + kInboxDigest;
    for (int r = 0; r < px.world && !err; ++r) {
      unsigned spins = 0;
      while (true) {
        const u64 g = __hip_atomic_load(base + (size_t)r * kInboxWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(g >> 32) == px.tag) {
          err = ((unsigned)g != px.digest) ? kXchgDigest : 0;
          break;
        }
        if ((++spins & 31) == 0 && wall_clock64() - t0 > px.spin_ticks) {
          err = kXchgTimeout | (r << 8);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  // any thread's failure reaches the host before the sequence word does (a timeout outranks a digest mismatch)
  __shared__ int worst;
  if (tid == 0) worst = 0;
  __syncthreads();
  if (err) atomicMax(&worst, err);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's mailbox store has left
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(o.mailbox + kMailboxErr, (u64)worst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(o.mailbox + kMailboxSeq, o.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <class F, int KS, bool PF>
__global__ void __launch_bounds__(kBlock)
wgrid_pass_kernel(F f, const u64* __restrict__ A, const u64* __restrict__ B, u64* __restrict__ A2, u64* __restrict__ B2,
                  GridW gw, int kf, size_t n_out, WgOut out) {
  constexpr int kPow3[6] = {1, 3, 9, 27, 81, 243};
  constexpr int cells = kPow3[KS];
  __shared__ int lds_flag;
  const int tid = threadIdx.x;
  u64 total = wgrid_body<F, KS, PF>(f, A, B, A2, B2, gw, kf, n_out);
  if (gridDim.x > 1) {
    // level 1: the blocks of a group of 32
    const int n_blocks = gridDim.x, group = blockIdx.x / kWgGroupBlocks, n_groups = (n_blocks + kWgGroupBlocks - 1) / kWgGroupBlocks;
    const int group_size = min(kWgGroupBlocks, n_blocks - group * kWgGroupBlocks);
    if (tid < cells)
      __hip_atomic_store(out.partials + (size_t)blockIdx.x * kGridChunk + tid, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(out.tickets + 1 + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t == (unsigned)group_size - 1) ? 1 : 0;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      lds_flag = last;
    }
    __syncthreads();
    if (!lds_flag) return;
    total = 0;
    if (tid < cells) {   // every load of the column in flight at once: a round of dependent loads costs ~0.6 us from L2
      u64 x[kWgGroupBlocks];
#pragma unroll
      for (int q = 0; q < kWgGroupBlocks; ++q)
        x[q] = (q < group_size) ? __hip_atomic_load(out.partials + (size_t)(group * kWgGroupBlocks + q) * kGridChunk + tid, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT)
                                : 0;
#pragma unroll
      for (int q = 0; q < kWgGroupBlocks; ++q) total = f.add(total, x[q]);
    }
    if (n_groups > 1) {
      // level 2: the groups
      __syncthreads();   // lds_flag is reused
      if (tid < cells)
        __hip_atomic_store(out.group_rows + (size_t)group * kGridChunk + tid, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(out.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)n_groups - 1) ? 1 : 0;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_flag = last;
      }
      __syncthreads();
      if (!lds_flag) return;
      total = 0;
      if (tid < cells) {
        u64 x[kWgGroupBlocks];
#pragma unroll
        for (int q = 0; q < kWgGroupBlocks; ++q)
          x[q] = (q < n_groups) ? __hip_atomic_load(out.group_rows + (size_t)q * kGridChunk + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
        for (int q = 0; q < kWgGroupBlocks; ++q) total = f.add(total, x[q]);
      }
    }
    // everything of this launch has been counted: leave the counters at zero for the next one
    if (tid <= n_groups) __hip_atomic_store(out.tickets + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (out.px.world > 0) {
    exchange_wide<cells>(out, total);
    return;
  }
  if (out.limbs_dev) {   // the stream's next operation (an all-reduce) reads them: kernel-boundary ordering
    if (tid < cells) write_split(out.limbs_dev, tid, total);
    return;
  }
  if (tid < cells) __hip_atomic_store(out.mailbox + kMailboxWide + tid, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (tid == 0) __hip_atomic_store(out.mailbox + kMailboxSeq, out.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The last pass of a sharded prover on the peer transport: the shard is down to its 2^kf pending entries (kf <= 5),
// the rounds left are those of the rank bits.  One workgroup per rank folds the pending challenges (one entry per
// table is left), hands that entry to every peer through the wide part of the inboxes - the gather and the exchange
// in one - and computes the 3^g cells of the g = log2(world) <= 3 remaining rounds on the world-entry tables itself:
// thread c forms its two extension values as signed sums of at most eight entries.  Every rank ends up with the same
// tables (written to A2 / B2, world entries each) and the same cells (whole residues in the wide mailbox).
template <class F>
__global__ void __launch_bounds__(kBlock)
rank_pass_kernel(F f, const u64* __restrict__ A, const u64* __restrict__ B, u64* __restrict__ A2, u64* __restrict__ B2, GridW gw, int kf,
                 WgOut out) {
  const PeerX& px = out.px;
  __shared__ u64 ta[kMaxPeers], tb[kMaxPeers];
  __shared__ int worst;
  const int tid = threadIdx.x;
  const size_t par = (size_t)(px.tag & 1u) * kMaxPeers * kInboxWords;
  int err = 0;
  if (tid == 0) worst = 0;
  if (tid < 2) {   // thread 0: table a, thread 1: table b - fold the 2^kf entries, publish the result to every inbox
    const u64* __restrict__ src = tid ? B : A;
    const int fan = 1 << kf;
    u64 v = 0;
    for (int c = 0; c < fan; ++c) v = f.add(v, f.mul(src[c], gw.w[c]));
    const ull2 pair = {((u64)px.tag << 32) | (v & 0xFFFFFFFFull), ((u64)px.tag << 32) | (v >> 32)};
    const size_t mine = par + (size_t)px.rank * kInboxWords + kInboxWide + 2 * (size_t)tid;
    for (int q = 0; q < px.world; ++q) st16_system(px.inbox[q] + mine, pair);
  }
  if (tid == 2) {   // the digest of the challenges
    const u64 g = ((u64)px.tag << 32) | (u64)px.digest;
    for (int q = 0; q < px.world; ++q)
      __hip_atomic_store(px.inbox[q] + par + (size_t)px.rank * kInboxWords + kInboxDigest, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // sweep: thread (source r, what): what = 0 / 1 the entries of a / b, 2 the digest
  const unsigned long long t0 = wall_clock64();
  if (tid < 3 * px.world) {
    const int r = tid / 3, what = tid % 3;
    const u64* base = px.inbox[px.rank] + par + (size_t)r * kInboxWords;
    unsigned spins = 0;
    while (true) {
      if (what < 2) {
        const ull2 g = ld16_system(base + kInboxWide + 2 * what);
        if ((unsigned)(g.x >> 32) == px.tag && (unsigned)(g.y >> 32) == px.tag) {
          const u64 v = (g.x & 0xFFFFFFFFull) | (g.y << 32);
          (what ? tb : ta)[r] = v;
          break;
        }
      } else {
        const u64 g = __hip_atomic_load(base + kInboxDigest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(g >> 32) == px.tag) {
          err = ((unsigned)g != px.digest) ? kXchgDigest : 0;
          break;
        }
      }
      if ((++spins & 31) == 0 && wall_clock64() - t0 > px.spin_ticks) {
        err = kXchgTimeout | (r << 8);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  if (err) atomicMax(&worst, err);
  __syncthreads();
  if (worst == 0) {
    if (tid < px.world) {
      A2[tid] = ta[tid];
      B2[tid] = tb[tid];
    }
    // cell c = sum_j d_j 3^(g-1-j), d_j the point of rank bit j (bit 0 = the next variable); its extension values are
    // sum_e coef(c, e) t[e], coef = prod_j k(d_j, bit_j(e)), k(0, b) = [b = 0], k(1, b) = [b = 1], k(inf, b) = b ? +1 : -1
    int g = 0;
    while ((1 << g) < px.world) ++g;
    int cells = 1;
    for (int j = 0; j < g; ++j) cells *= 3;
    if (tid < cells) {
      int d[3] = {0, 0, 0};
      int c = tid;
      for (int j = g - 1; j >= 0; --j) {
        d[j] = c % 3;
        c /= 3;
      }
      u64 ea = 0, eb = 0;
      for (int e = 0; e < px.world; ++e) {
        int sign = 1;
        for (int j = 0; j < g; ++j) {
          const int b = (e >> j) & 1;
          if (d[j] == 2) sign = b ? sign : -sign;
          else if (d[j] != b) sign = 0;
        }
        if (sign > 0) {
          ea = f.add(ea, ta[e]);
          eb = f.add(eb, tb[e]);
        } else if (sign < 0) {
          ea = f.sub(ea, ta[e]);
          eb = f.sub(eb, tb[e]);
        }
      }
      __hip_atomic_store(out.mailbox + kMailboxWide + tid, f.mul(ea, eb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(out.mailbox + kMailboxErr, (u64)worst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(out.mailbox + kMailboxSeq, out.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


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
fold_kernel(F f, const u64* __restrict__ T, u64* __restrict__ T2, FoldW fw, size_t n_units) {
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
  // the waves of a block draw runs of kFoldGrab tiles from a counter in LDS (evaluate_kernel; block b owns the runs c * grid + b)
  auto next_run = [&]() -> size_t {
    unsigned c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(&lds_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return ((size_t)__builtin_amdgcn_readfirstlane(c) * gridDim.x + blockIdx.x) * kFoldGrab;
  };
  for (size_t run = next_run(); run < n_tiles; run = next_run())
  for (size_t tile = run; tile < run + kFoldGrab && tile < n_tiles; ++tile) {
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
template <> struct stream_block<MontGeneric> {      // the generic-modulus arithmetic needs more registers per wave
  static constexpr int evaluate = 512, fix_low = 512;
};
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
                          int kb, const u64* __restrict__ w_c, int kc, u64* __restrict__ out) {
  const size_t n = (size_t)1 << (kb + kc);
  for (size_t o = (size_t)blockIdx.x * kBlock + threadIdx.x; o < n; o += (size_t)gridDim.x * kBlock) {
    const size_t b = o >> kc, c = o & (((size_t)1 << kc) - 1);
    const size_t bc = (c << kb) | b;
    const u64 wb = w_b[b], wc = w_c[c];
    out[o] = f.add(f.mul(add[bc], f.add(wb, wc)), f.mul(mul[bc], f.mul(wb, wc)));
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

// ------------------------------------------------------------------------------------
// triangle_counting::G (triangle-counting/src/lib.rs:22-166): g(X,Y,Z) = f(X,Y) f(Y,Z) f(X,Z).

// P[(z << k) | x] = sum_y f[(y << k) | x] * f[(z << k) | y]: the square of the adjacency MLE's
// matrix.  sum_{y} f1(x,y) f2(y,z) is multilinear in x and in z, so the k x-rounds of the
// sumcheck are a product-of-two-tables sumcheck on (P, f3) - one n^3 pass here instead of an
// n^3 pass per round (the reference's to_univariate walks all 2^(3k) evaluations, :138-165).
// Consecutive lanes own consecutive x: the column read is coalesced, the row read a broadcast.
template <class F>
__global__ void __launch_bounds__(kBlock)
matsq_kernel(F f, const u64* __restrict__ T, int k, u64* __restrict__ P, size_t z_begin, size_t z_rows) {
  const size_t n = (size_t)1 << k, first = z_begin * n, total = (z_begin + z_rows) * n;   // rows z_begin .. of P
  for (size_t o = first + (size_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (size_t)gridDim.x * kBlock) {
    const size_t z = o >> k, x = o & (n - 1);
    typename F::Acc acc;
    f.acc_zero(acc);
    for (size_t y = 0; y < n; ++y) f.acc_mac(acc, T[(y << k) | x], T[(z << k) | y]);
    P[o] = f.acc_get(acc);
  }
}

// The same square, LDS-tiled, for n >= 64: a block of 256 threads owns a 64 x 64 tile of P and walks y in
// steps of 32; per step it stages A[y][x0..x0+64) and, transposed, Bt[y][z0..z0+64) = f[(z << k) | y] in LDS
// (16 KiB each), and every thread accumulates a 4 x 4 patch: 4 ds_read_b128 per 16 lazy multiply-adds
// instead of 2 global loads per multiply-add, so the kernel runs at the VALU rate of the products
// (15 instructions each) rather than at the L1 rate of the naive form.
template <class F>
__global__ void __launch_bounds__(kBlock)
matsq_tiled_kernel(F f, const u64* __restrict__ T, int k, u64* __restrict__ P, size_t z_begin, size_t z_rows,
                   const unsigned* __restrict__ only_if /* null, or: run only if this word is non-zero */) {
  constexpr int TS = 64, KT = 32;
  if (only_if && *only_if == 0) return;   // the table was 0/1: matsq_mfma_kernel has done the work
  __shared__ ull2 lds_a[KT * TS / 2];   // A[yy][xx], 16 KiB
  __shared__ u64 lds_b[KT * TS];        // Bt[yy][zz], 16 KiB
  const size_t n = (size_t)1 << k;
  const int tiles = (int)(n / TS), tiles_z = (int)(z_rows / TS);   // rows z_begin .. z_begin + z_rows of P (a rank's share)
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  for (int tile = blockIdx.x; tile < tiles * tiles_z; tile += gridDim.x) {
    const size_t x0 = (size_t)(tile % tiles) * TS, z0 = z_begin + (size_t)(tile / tiles) * TS;
    typename F::Acc acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) f.acc_zero(acc[j][i]);
    for (size_t y0 = 0; y0 < n; y0 += KT) {
      __syncthreads();   // the previous step's reads are done
      // A tile: 32 rows of 64 entries; thread t loads 16-byte pieces (row t/32 + 8i, piece t%32)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (threadIdx.x >> 5) + 8 * i, pc = threadIdx.x & 31;
        lds_a[row * (TS / 2) + pc] = reinterpret_cast<const ull2*>(T + ((y0 + row) << k) + x0)[pc];
      }
      // B tile: rows z0 + zz hold 32 consecutive y; thread t loads piece t%16 of row t/16 + 16i and
      // stores it transposed
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int zz = (threadIdx.x >> 4) + 16 * i, pc = threadIdx.x & 15;
        const ull2 v = reinterpret_cast<const ull2*>(T + ((z0 + zz) << k) + y0)[pc];
        lds_b[(2 * pc) * TS + zz] = v.x;
        lds_b[(2 * pc + 1) * TS + zz] = v.y;
      }
      __syncthreads();
      for (int yy = 0; yy < KT; ++yy) {
        const ull2 a01 = lds_a[yy * (TS / 2) + 2 * tx], a23 = lds_a[yy * (TS / 2) + 2 * tx + 1];
        const ull2 b01 = reinterpret_cast<const ull2*>(lds_b + yy * TS)[2 * ty],
                   b23 = reinterpret_cast<const ull2*>(lds_b + yy * TS)[2 * ty + 1];
        const u64 a[4] = {a01.x, a01.y, a23.x, a23.y}, b[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) f.acc_mac(acc[j][i], a[i], b[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ull2 o0 = {f.acc_get(acc[j][0]), f.acc_get(acc[j][1])}, o1 = {f.acc_get(acc[j][2]), f.acc_get(acc[j][3])};
      ull2* dst = reinterpret_cast<ull2*>(P + ((z0 + 4 * ty + j) << k) + x0 + 4 * tx);
      dst[0] = o0;
      dst[1] = o1;
    }
  }
}

// The square of a 0/1 matrix on the matrix cores.  G::new_adj_matrix (triangle-counting/src/lib.rs:32-51) builds the
// three tables from a Vec<bool>: every entry is 0 or 1, so P[z][x] = sum_y T[z][y] T[y][x] is a COUNT (<= n <= 2^15)
// and an int8 x int8 -> int32 MFMA computes it exactly - this is a matrix product by nature, not a reshaped stream.
//  1. matsq_bytes_kernel: the table as bytes, row-major (T8[z][y]) and transposed (T8t[x][y] = T[y][x]) so that both
//     MFMA operands are 16 contiguous bytes per lane; any entry that is neither 0 nor 1 raises `flag`.
//  2. matsq_mfma_kernel (if the flag stayed down): one wave per 32 x 32 tile of P, v_mfma_i32_32x32x32_i8 over y in
//     steps of 32, operands straight from the (L2-resident) byte tables; count -> Montgomery word (count * R^2 * R^-1).
//  3. matsq_tiled_kernel (if the flag went up; launched behind the other two either way, no host round trip): the
//     generic field-valued square.
// The hardware pairs element e of lane (r, h)'s A fragment with element e of lane (r', h)'s B fragment; both are loaded
// with the same y = y0 + 16 h + e, so whatever k order the instruction uses inside a step the sum is over the same y.
// C/D layout (cdna_hip_programming.md section 3): col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));
template <class F>
__global__ void __launch_bounds__(kBlock)
matsq_bytes_kernel(F f, const u64* __restrict__ T, int k, unsigned char* __restrict__ T8, unsigned char* __restrict__ T8t,
                   unsigned* __restrict__ flag) {
  constexpr int TS = 64;
  __shared__ unsigned char tile[TS][TS + 16];   // rows 16-byte aligned (80 bytes)
  const size_t n = (size_t)1 << k;
  const int tps = (int)(n / TS);
  const u64 one = f.one();
  int bad = 0;
  for (int tid = blockIdx.x; tid < tps * tps; tid += gridDim.x) {
    const size_t r0 = (size_t)(tid / tps) * TS, c0 = (size_t)(tid % tps) * TS;
    __syncthreads();   // the previous tile has been written out
#pragma unroll
    for (int i = 0; i < TS * TS / kBlock; ++i) {
      const int row = (threadIdx.x >> 6) + 4 * i, col = threadIdx.x & 63;
      const u64 v = T[((r0 + row) << k) | (c0 + col)];
      bad |= (v != 0 && v != one) ? 1 : 0;
      tile[row][col] = (v == one) ? 1 : 0;
    }
    __syncthreads();
    const int rr = threadIdx.x >> 2, q = threadIdx.x & 3;   // 64 rows x 4 chunks of 16 bytes
    *reinterpret_cast<uint4*>(T8 + (r0 + rr) * n + c0 + 16 * q) = *reinterpret_cast<const uint4*>(&tile[rr][16 * q]);
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = (unsigned)tile[16 * q + 4 * j][rr] | ((unsigned)tile[16 * q + 4 * j + 1][rr] << 8) | ((unsigned)tile[16 * q + 4 * j + 2][rr] << 16) |
             ((unsigned)tile[16 * q + 4 * j + 3][rr] << 24);
    *reinterpret_cast<uint4*>(T8t + (c0 + rr) * n + r0 + 16 * q) = uint4{w[0], w[1], w[2], w[3]};
  }
  if (bad) atomicOr(flag, 1u);
}
template <class F>
__global__ void __launch_bounds__(kBlock)
matsq_mfma_kernel(F f, const unsigned char* __restrict__ T8, const unsigned char* __restrict__ T8t, int k, u64* __restrict__ P,
                  size_t z_begin, size_t z_rows, const unsigned* __restrict__ flag) {
  if (*flag != 0) return;   // not a 0/1 table: the generic kernel behind this launch does the work
  const size_t n = (size_t)1 << k;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, r = lane & 31, h = lane >> 5;
  const size_t tiles_x = n / 32, n_tiles = (z_rows / 32) * tiles_x;
  const u64 r2 = f.r_squared();
  for (size_t tid = (size_t)blockIdx.x * (kBlock / kWave) + wave; tid < n_tiles; tid += (size_t)gridDim.x * (kBlock / kWave)) {
    const size_t z0 = z_begin + (tid / tiles_x) * 32, x0 = (tid % tiles_x) * 32;
    const unsigned char* ap = T8 + (z0 + r) * n + 16 * h;    // row z0 + r of T:  T[z][y0 + 16 h + e]
    const unsigned char* bp = T8t + (x0 + r) * n + 16 * h;   // column x0 + r of T: T[y0 + 16 h + e][x]
    mfma_v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t y0 = 0;
    for (; y0 + 128 <= n; y0 += 128) {   // four steps of loads in flight
      mfma_v4i a[4], b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = *reinterpret_cast<const mfma_v4i*>(ap + y0 + 32 * s);
        b[s] = *reinterpret_cast<const mfma_v4i*>(bp + y0 + 32 * s);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], acc, 0, 0, 0);
    }
    for (; y0 < n; y0 += 32)
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const mfma_v4i*>(ap + y0), *reinterpret_cast<const mfma_v4i*>(bp + y0), acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      P[((z0 + row) << k) | (x0 + r)] = f.mul((u64)(unsigned)acc[reg], r2);   // count -> Montgomery word
    }
  }
}

// Round sums H(0), H(1), H(inf) of G in ANY state (xv, yv, zv variables left), by walking every
// remaining (x, y, z) like the reference does: the generic SumCheckPolynomial::to_univariate.
// Two of the three copies hold the current variable (pairs p, q), the third a constant c.
template <class F>
__global__ void __launch_bounds__(kBlock)
tri_sums_kernel(F f, const u64* __restrict__ f1, const u64* __restrict__ f2, const u64* __restrict__ f3, int xv, int yv,
                int zv, PassOut out) {
  __shared__ u64 lds[(kBlock / kWave) * 3];
  __shared__ int lds_flag;
  const size_t total = (size_t)1 << (xv + yv + zv - 1);
  typename F::Acc acc[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) f.acc_zero(acc[s]);
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
    u64 p0, p1, q0, q1, c;
    if (xv > 0) {
      const size_t xh = t & (((size_t)1 << (xv - 1)) - 1), y = (t >> (xv - 1)) & (((size_t)1 << yv) - 1),
                   z = t >> (xv - 1 + yv);
      const size_t i1 = (y << xv) | (2 * xh), i3 = (z << xv) | (2 * xh);
      p0 = f1[i1]; p1 = f1[i1 + 1]; q0 = f3[i3]; q1 = f3[i3 + 1]; c = f2[(z << yv) | y];
    } else if (yv > 0) {
      const size_t yh = t & (((size_t)1 << (yv - 1)) - 1), z = t >> (yv - 1);
      const size_t i2 = (z << yv) | (2 * yh);
      p0 = f1[2 * yh]; p1 = f1[2 * yh + 1]; q0 = f2[i2]; q1 = f2[i2 + 1]; c = f3[z];
    } else {
      p0 = f2[2 * t]; p1 = f2[2 * t + 1]; q0 = f3[2 * t]; q1 = f3[2 * t + 1]; c = f1[0];
    }
    f.acc_mac(acc[0], f.mul(p0, q0), c);
    f.acc_mac(acc[1], f.mul(p1, q1), c);
    f.acc_mac(acc[2], f.mul(f.sub(p1, p0), f.sub(q1, q0)), c);
  }
  u64 res[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) res[s] = f.acc_get(acc[s]);
  block_reduce<F, 3>(f, res, lds);
  finish_pass<F, 3>(f, out, res[0], &lds_flag);
}

// G::to_evaluations (:138-165): out[((x << yv) | y) << zv | z] = f1[(y<<xv)|x] f2[(z<<yv)|y] f3[(z<<xv)|x]
template <class F>
__global__ void __launch_bounds__(kBlock)
tri_to_evaluations_kernel(F f, const u64* __restrict__ f1, const u64* __restrict__ f2, const u64* __restrict__ f3, int xv,
                          int yv, int zv, u64* __restrict__ out) {
  const size_t total = (size_t)1 << (xv + yv + zv);
  for (size_t o = (size_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (size_t)gridDim.x * kBlock) {
    const size_t z = o & (((size_t)1 << zv) - 1), y = (o >> zv) & (((size_t)1 << yv) - 1), x = o >> (zv + yv);
    out[o] = f.mul(f.mul(f1[(y << xv) | x], f2[(z << yv) | y]), f3[(z << xv) | x]);
  }
}

// Vector form of the split-limb exchange (sharded G::new: the f_A half is a sum over the
// row blocks the ranks own).  limbs[2i], limbs[2i+1] = low / high 32 bits of v[i].
__global__ void __launch_bounds__(kBlock)
split_limbs_kernel(const u64* __restrict__ v, size_t n, u64* __restrict__ limbs) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    ull2 o = {v[i] & 0xFFFFFFFFull, v[i] >> 32};
    reinterpret_cast<ull2*>(limbs)[i] = o;
  }
}
// out[i] = (LO + 2^32 * HI) mod p for the limb sums LO, HI (< 2^63) of word i.  The words are
// plain integers here (sums of Montgomery words), so the product with 2^32 is an ordinary
// modular product: mont_mul(mont_mul(x, y), R^2) = x*y mod p.
template <class F>
__global__ void __launch_bounds__(kBlock)
recombine_limbs_kernel(F f, const u64* __restrict__ limbs, size_t n, u64* __restrict__ out) {
  const u64 c32 = f.reduce_word((u64)1 << 32);
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const ull2 l = reinterpret_cast<const ull2*>(limbs)[i];
    const u64 lo = f.reduce_word(l.x), hi = f.reduce_word(l.y);
    out[i] = f.add(lo, f.mul(f.mul(hi, c32), f.r_squared()));
  }
}

// All-gather of both tables of a sharded prover over the peer mapping (the tail gather of SURVEY.md
// section 8e): every rank copies its `len` words of A and B into slot `rank` of EVERY rank's arena with
// system-scope write-through stores, drains them, and the last block then tells every peer (a tagged
// granule in the peer's inbox) and waits until every peer has told it.  arena layout: [table][rank][len].
struct PeerG {
  u64* arena[kMaxPeers] = {};
  size_t table_stride = 0;   // words between the two tables' regions
};
__device__ __forceinline__ void st16_sys(ull2* p, ull2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__global__ void __launch_bounds__(kBlock)
peer_gather_kernel(const u64* __restrict__ A, const u64* __restrict__ B, size_t len, PeerG pg, PassOut out) {
  __shared__ int lds_flag;
  const PeerX& px = out.px;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (int q = 0; q < px.world; ++q) {
    u64* dstA = pg.arena[q] + (size_t)px.rank * len;
    u64* dstB = dstA + pg.table_stride;
    if ((len & 1) == 0) {
      const ull2* Ap = reinterpret_cast<const ull2*>(A);
      const ull2* Bp = reinterpret_cast<const ull2*>(B);
      for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len / 2; i += stride) {
        st16_sys(reinterpret_cast<ull2*>(dstA) + i, Ap[i]);
        st16_sys(reinterpret_cast<ull2*>(dstB) + i, Bp[i]);
      }
    } else {
      for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += stride) {
        __hip_atomic_store(dstA + i, A[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dstB + i, B[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the block signals
  __syncthreads();
  if (threadIdx.x == 0) {
    int last = 1;
    if (gridDim.x > 1) {
      const unsigned t = __hip_atomic_fetch_add(out.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (t - out.ticket_base == gridDim.x - 1) ? 1 : 0;
    }
    lds_flag = last;
  }
  __syncthreads();
  if (!lds_flag) return;
  if (threadIdx.x < kWave) {
    const int lane = threadIdx.x;
    const size_t par = (size_t)(px.tag & 1u) * kMaxPeers * kInboxWords;
    const u64 granule = ((u64)px.tag << 32) | 1u;
    if (lane < px.world)
      __hip_atomic_store(px.inbox[lane] + par + (size_t)px.rank * kInboxWords + kInboxGather, granule, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    int err = 0;
    if (lane < px.world) {
      const u64* w = px.inbox[px.rank] + par + (size_t)lane * kInboxWords + kInboxGather;
      const unsigned long long t0 = wall_clock64();
      unsigned spins = 0;
      while ((unsigned)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >> 32) != px.tag) {
        if ((++spins & 31) == 0 && wall_clock64() - t0 > px.spin_ticks) { err = kXchgTimeout; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const int any = __any(err != 0) ? 1 : 0;
    if (lane == 0 && out.mailbox)
      __hip_atomic_store(out.mailbox + kMailboxErr, (u64)(any ? kXchgTimeout : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  publish_seq(out);
}

// Connect-time hello of the peer transport: one granule {kHelloTag | rank + 1} into every peer's inbox (parity 0,
// slot kInboxHello).  The host of each rank polls its own inbox until every peer's hello is there: by then every
// peer has mapped this rank's region, loaded its code object and run a kernel, so the cold-start lag of a freshly
// started job (seconds) is absorbed here and the per-pass waits can be bounded tightly (peer_spin_ms).
constexpr int kInboxHello = 58;
constexpr unsigned kHelloTag = 0x48454c4fu;
__global__ void peer_hello_kernel(PeerX px) {
  const int lane = threadIdx.x;
  if (lane < px.world)
    __hip_atomic_store(px.inbox[lane] + (size_t)px.rank * kInboxWords + kInboxHello, ((u64)kHelloTag << 32) | (u64)(px.rank + 1),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Cross-rank sum of limbs that a small kernel left in device memory (the degenerate paths that do not end
// in finish_pass): one workgroup of one wave.
template <int NS>
__global__ void peer_exchange_kernel(const u64* __restrict__ limbs, PassOut out) {
  __shared__ u64 xl[2 * NS + 2];
  if (threadIdx.x < 2 * NS) xl[threadIdx.x] = limbs[threadIdx.x];
  exchange_and_publish<NS>(out, xl);
}
// out[i] = sum over `rows` rows of in[r * n + i]: plain u64 adds (the words are 32-bit limbs)
__global__ void __launch_bounds__(kBlock)
sum_limb_rows_kernel(const u64* __restrict__ in, int rows, size_t n, u64* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    u64 t = 0;
    for (int r = 0; r < rows; ++r) t += in[(size_t)r * n + i];
    out[i] = t;
  }
}

// The same for the up to 486 limb totals of a five-round pass: into the wide part of the mailbox (one workgroup).
__global__ void __launch_bounds__(kBlock)
mailbox_copy_wide_kernel(const u64* __restrict__ sums, int count, u64* __restrict__ mailbox, u64 seq) {
  for (int i = threadIdx.x; i < count; i += kBlock)
    __hip_atomic_store(mailbox + kMailboxWide + i, sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every thread's stores have left before the barrier
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(mailbox + kMailboxSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// After a device-side all-reduce: hand the summed limbs to the host mailbox (one wave).
__global__ void mailbox_copy_kernel(const u64* __restrict__ sums, int count, u64* __restrict__ mailbox, u64 seq) {
  if (blockIdx.x == 0 && threadIdx.x < kWave) {
    if ((int)threadIdx.x < count)
      __hip_atomic_store(mailbox + threadIdx.x, sums[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0)  // same wave: the release orders it behind the data stores above
      __hip_atomic_store(mailbox + kMailboxSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace sc
