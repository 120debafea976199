// Part of kernels.hpp (included there, in order): the hand-off of a pass to the host and the peers (PassOut, finish_pass), and pass_kernel.
#pragma once

namespace sc {

// ------------------------------------------------------------------------------------
// The pass kernel: fold KF pending variables of both tables, write the folded tables,
// and accumulate the round sums of the FOLDED tables for the next KS rounds - one read of
// the inputs, one write of the outputs (reference: Prover::round =
// fix_variables + to_univariate, sum-check-protocol/src/lib.rs:105-112).
//
// unit = one run of IN = 2^(KF+KS) input entries per table -> OUT = 2^KS output entries,
// owned by one lane; a wave tile is 64 units.  KF in 0..4, KS in 1..3 (KS = 3, the 27-cell grid
// of a three-round first pass, is only instantiated with KF = 0; KF = 4, the pass behind the four-round first pass of
// kernels/gram.hpp, only with KS = 2).  Sums leave through finish_pass (PassOut).
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
  // the NS < 9 branch publishes from lane 0 of EVERY wave and a workgroup barrier does not drain vmcnt: each wave waits
  // for its own mailbox stores here, so thread 0's release of the sequence word cannot pass a late value of waves 1..3
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
// cells per chunk: 32 threads sum one cell, and a chunk's accumulators (BS of them per cell) stay within 256 bytes of LDS per
// thread - sixteen 16-byte Goldilocks accumulators, twelve 20-byte ones of the generic field
template <class Acc, int BS>
__host__ __device__ constexpr int reduce_chunk_cells() {
  return (BS / 32 < 256 / (int)sizeof(Acc)) ? BS / 32 : 256 / (int)sizeof(Acc);
}
template <class F, int NS, int BS = kBlock>
__device__ __forceinline__ u64 reduce_cells_lds(const F& f, const typename F::Acc (&acc)[NS], typename F::Acc* scratch, u64* out) {
  typedef typename F::Acc Acc;
  constexpr int CH = reduce_chunk_cells<Acc, BS>();
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
      for (int k = 1; k < BS / 32; ++k) f.acc_add(t, scratch[cell * BS + part + 32 * k]);   // (BS / 32 slices of 32 threads each)
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
// (nt bit 2: the pipelined whole-tile form of the three-variable fold, see the tile loop)
constexpr int kPipe32Threads = 768;
__host__ __device__ constexpr int pass_block_threads(int kf, int ks, int nt = 0) {
  return (kf == 3 && ks == 2 && (nt & 4)) ? kPipe32Threads : (ks == 3 || (kf >= 3 && ks == 2)) ? 512 : (kf == 2 && ks == 2) ? 768 : kBlock;
}

// NT: bit 0 = nontemporal loads, bit 1 = nontemporal stores (see ld16 / st16)
template <class F, int KF, int KS, int NT>
__global__ void __launch_bounds__(pass_block_threads(KF, KS, NT))
pass_kernel(F f, const u64* __restrict__ A, const u64* __restrict__ B, u64* __restrict__ A2,
            u64* __restrict__ B2, FoldW fw, size_t n_units, PassOut out) {
  constexpr bool kNtLoad = (NT & 1) != 0, kNtStore = (NT & 2) != 0;
  constexpr int IN = 1 << (KF + KS), OUT = 1 << KS, NS = (KS == 1) ? 3 : (KS == 2) ? 9 : 27;
  constexpr int NP = IN / 2, NPO = OUT / 2;  // 16-byte pieces per lane, in and out
  constexpr int BS = pass_block_threads(KF, KS, NT), kWaves = BS / kWave;
  constexpr bool kPipe = (KF == 4) || (KF == 3 && KS == 2 && (NT & 4) != 0);   // sub-step pipeline (whole tiles only)
  constexpr bool kDma = (KF == 4) && (NT & 8) != 0;   // ... with its sub-steps brought in by LDS-DMA (no staging registers)
  static_assert(BS == kBlock || NS >= 9, "reduce_cells (the KS = 1 passes) is written for 256 threads");
  // the tile transposes; after the loop the same bytes hold a chunk of every thread's accumulators (reduce_cells_lds)
  // KF = 4: a unit is 64 entries = 32 pieces per table - too many to stage at once; its four outputs (16 entries = 8 pieces
  // each) are produced one after the other, 8 pieces per lane and table in flight (NPS)
  constexpr int NPS = (1 << KF) / 2 > 0 ? (1 << KF) / 2 : 1;   // pieces of one output
  constexpr int NPL = kDma ? 2 * NPS + 2 : kPipe ? NPS + 4 : NP;   // pieces per lane the wave's LDS region is laid out for (pipelined: a sub-step + the 4-KiB exchange area; DMA: two stages + a 2-KiB exchange area)
  constexpr int kTransposeSlots = (NP > 1 || NPO > 1) ? kWaves * kWave * NPL : 1;
  constexpr int kChunkCells = reduce_chunk_cells<typename F::Acc, BS>();
  constexpr int kReduceSlots = (NS >= 9) ? (int)(((NS < kChunkCells ? NS : kChunkCells) * BS * sizeof(typename F::Acc) + sizeof(ull2) - 1) / sizeof(ull2)) : 1;
  __shared__ ull2 lds_t[kTransposeSlots > kReduceSlots ? kTransposeSlots : kReduceSlots];
  __shared__ u64 lds[kWaves * NS];
  __shared__ int lds_flag;
  __shared__ unsigned lds_next;   // the block's tile counter
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  ull2* const my_lds = lds_t + ((NP > 1 || NPO > 1) ? wave * kWave * NPL : 0);

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
  // (generic lambdas: their bodies are only instantiated where they are called - not for KF = 4, whose unit has no NP-piece form)
  auto load_tile = [&](size_t tile, auto& pa, auto& pb) {
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
  auto stage_and_fold3 = [&](auto& p, auto& t) {
    if constexpr (KF == 3) {  // (the body only instantiates for run lengths swz_slot supports)
#pragma unroll
      for (int k = 0; k < NP; ++k) my_lds[swz_slot<NP>(64 * k + lane)] = p[k];
      wave_lds_sync();
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
      wave_lds_sync();
    }
  };
  // The 27-cell grid runs at two waves per SIMD and is ALU-heavy: it cannot count on other
  // waves to cover its loads, so it fetches the wave's next tile before it starts on the
  // arithmetic of the current one.  (The three-variable fold has no registers for a second tile; asking for one table
  // of the next tile at a time, while the other table is folded out of LDS, was measured and gave nothing: that pass is
  // not waiting for its own loads.)
  constexpr bool kPrefetch = (KS == 3);
  auto process_tile = [&](size_t tile, size_t next, auto& pa, auto& pb) {
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
  // Output stores of a folding pass and the L2: the folded entries leave the CUs as a trickle of 1-KiB stores, stay in the
  // (write-back) L2 until something evicts them line by line, and reach HBM as a trickle too - which costs the READ stream far
  // more than their bytes (the four-variable fold on 2^28-entry tables: 624-656 us without its stores, 772-822 with them:
  // 6 % of the bytes, +150 us; nontemporal stores change nothing).  Written back in bulk they cost less: one wave per XCD
  // (the first eight blocks: workgroups are dealt to the XCDs round-robin) asks its L2 to write its dirty lines back after
  // every tile it finishes (~1 MB per request at n = 28): 777-784 us where the pass took 820-822 on the same box; every two
  // tiles 795-801, every four 806-813, four waves per XCD 802-809 (profiles/r04_fold_writeback.txt).  A hint, not a fence:
  // nothing waits for it.  The pipelined forms only (tiles of 32-64 KiB per table): in the staged two-variable fold, whose
  // tiles are 8 KiB, a request per tile made the pass on 2^24-entry tables 3 us slower.
  auto bulk_writeback = [&]() {
    if constexpr (KF > 0) {
      if (wave == 0 && blockIdx.x < 8) asm volatile("buffer_wbl2 sc1" ::: "memory");
    }
  };
  if constexpr (kDma) {
    // The four-variable fold with its sub-steps brought in by LDS-DMA: no staging registers, the next stage always in
    // flight.  A stage = one table's sub-step (8 KiB, 64 outputs); a wave owns two stages (table a's and table b's) and a
    // 2-KiB exchange area; stage j of a tile (j = 0..7: a, b, a, b, ...) is refilled with stage j + 2 the moment its
    // sixteen entries per lane are in registers.  The destination of an LDS-DMA load is lane-linear, so the swizzle of
    // transpose_to_runs is applied to the SOURCE piece (swz_slot is an involution inside aligned groups of NPS pieces).
    // Every LDS access of the loop is inline assembly: behind an LDS-DMA load the compiler orders LDS accesses it cannot
    // disambiguate with s_waitcnt vmcnt(0), i.e. behind the prefetch.  vmcnt is counted by hand: loads, stores and DMA
    // complete in issue order, and when stage j is read the only younger operations allowed to be outstanding are the
    // eight DMA instructions of stage j + 1.
    static_assert(KS == 2 && NPS == 8, "written for the four-variable fold");
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glob_ptr_t;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    ull2* const slot0 = my_lds;                          // table a's stage
    ull2* const slot1 = my_lds + kWave * NPS;            // table b's
    const unsigned xbase = (unsigned)(size_t)(lds_ptr_t)(my_lds + 2 * kWave * NPS);   // 2 KiB: exchange / output transposes, one table at a time
    const int rot = wave & 3;
    // (swz_slot<8>(64 k + lane) = 64 k + (lane ^ ((4 k + lane / 16) & 7)): two lane offsets, for even and odd k;
    //  swz_slot<8>(8 lane + m) = 8 lane + (m ^ ((lane / 2) & 7)): one base and one XOR mask - registers are what this kernel is short of)
    const int src_even = lane ^ ((lane >> 4) & 7), src_odd = lane ^ (((lane >> 4) + 4) & 7);
    const unsigned rd_base = (unsigned)(size_t)(lds_ptr_t)slot0 + 16u * (unsigned)(NPS * lane), rd_mask = (unsigned)((lane >> 1) & 7);
    auto issue = [&](const ull2* __restrict__ T, size_t tile, int o, int slot) {
      const ull2* src = T + tile * kWave * NP + (size_t)o * NPS * kWave;
      ull2* const dst = slot ? slot1 : slot0;
#pragma unroll
      for (int k = 0; k < NPS; ++k)
        __builtin_amdgcn_global_load_lds((glob_ptr_t)(src + kWave * k + ((k & 1) ? src_odd : src_even)), (lds_ptr_t)(dst + kWave * k), 16, 0,
                                         kNtLoad ? 2 : 0);
    };
    // one stage -> the lane's output: its eight pieces in two batches of four reads (sixteen registers live, not thirty-two)
#define SC_DMA_READ4(X, R0, OFF)                                                                                      \
  asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%8\n\t" \
               "ds_read_b128 %3, %7 offset:%8\n\ts_waitcnt lgkmcnt(0)"                                                 \
               : "=&v"(X[0]), "=&v"(X[1]), "=&v"(X[2]), "=&v"(X[3])                                                  \
               : "v"(rd_base + 16u * ((unsigned)(R0) ^ rd_mask)), "v"(rd_base + 16u * ((unsigned)(R0 + 1) ^ rd_mask)),  \
                 "v"(rd_base + 16u * ((unsigned)(R0 + 2) ^ rd_mask)), "v"(rd_base + 16u * ((unsigned)(R0 + 3) ^ rd_mask)), "n"(OFF) \
               : "memory")
    auto mac4 = [&](typename F::Acc3& s, const u32x4 (&x)[4], int m0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f.acc3_mac(s, ((u64)x[m].y << 32) | x[m].x, fw.w[2 * (m0 + m)]);
        f.acc3_mac(s, ((u64)x[m].w << 32) | x[m].z, fw.w[2 * (m0 + m) + 1]);
      }
    };
    // the exchange area, in assembly as well (wave-private; LDS operations of a wave execute in order)
    auto x_write64 = [&](unsigned byte_off, u64 v) {
      const u32x2 w = {(unsigned)v, (unsigned)(v >> 32)};
      asm volatile("ds_write_b64 %0, %1" ::"v"(xbase + byte_off), "v"(w) : "memory");
    };
    auto x_write128 = [&](unsigned byte_off, u64 lo, u64 hi) {
      const u32x4 w = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
      asm volatile("ds_write_b128 %0, %1" ::"v"(xbase + byte_off), "v"(w) : "memory");
    };
    auto x_read128 = [&](unsigned byte_off) -> u32x4 {
      u32x4 r;
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(xbase + byte_off) : "memory");
      return r;
    };
    size_t tile = next_tile();
    if (tile < n_tiles) {
      issue(Ap, tile, rot, 0);
      issue(Bp, tile, rot ^ 2, 1);
    }
    while (tile < n_tiles) {
      const size_t next = next_tile();
      u64 va[OUT], vb[OUT];
#pragma unroll
      for (int i = 0; i < OUT; ++i) {
        const int na = (i + 1 + rot) & 3;
        u32x4 x[4], y[4];
        typename F::Acc3 s;
        // table a: stage 2 i (slot 0); behind it in the queue: stage 2 i + 1
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        SC_DMA_READ4(x, 0, 0);
        SC_DMA_READ4(y, 4, 0);
        if (i + 1 < OUT) issue(Ap, tile, na, 0);
        else if (next < n_tiles) issue(Ap, next, na, 0);
        f.acc3_zero(s);
        mac4(s, x, 0);
        mac4(s, y, 4);
        va[i] = f.acc3_get(s);
        // table b: stage 2 i + 1 (slot 1); behind it: stage 2 i + 2 - unless this was the wave's last tile
        if (i + 1 < OUT || next < n_tiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SC_DMA_READ4(x, 0, NPS * kWave * 16);
        SC_DMA_READ4(y, 4, NPS * kWave * 16);
        if (i + 1 < OUT) issue(Bp, tile, na ^ 2, 1);
        else if (next < n_tiles) issue(Bp, next, na ^ 2, 1);
        f.acc3_zero(s);
        mac4(s, x, 0);
        mac4(s, y, 4);
        vb[i] = f.acc3_get(s);
      }
#undef SC_DMA_READ4
      // outputs 64 o + lane -> quads 4 l .. 4 l + 3 per lane, one table at a time through the 2-KiB area; then the quad as
      // two 16-byte pieces 2 l, 2 l + 1 -> pieces 64 k + lane for the stores
      u64 a[OUT], b[OUT];
      ull2 oa[NPO], ob[NPO];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < OUT; ++i) {
          const int o = t ? (((i + rot) & 3) ^ 2) : ((i + rot) & 3);
          x_write64((unsigned)(kWave * o + lane) * 8u, t ? vb[i] : va[i]);
        }
        const u32x4 q0 = x_read128((unsigned)(OUT * lane) * 8u), q1 = x_read128((unsigned)(OUT * lane + 2) * 8u);
        u64* const q = t ? b : a;
        q[0] = ((u64)q0.y << 32) | q0.x; q[1] = ((u64)q0.w << 32) | q0.z;
        q[2] = ((u64)q1.y << 32) | q1.x; q[3] = ((u64)q1.w << 32) | q1.z;
#pragma unroll
        for (int m = 0; m < NPO; ++m) x_write128((unsigned)swz_slot<NPO>(NPO * lane + m) * 16u, q[2 * m], q[2 * m + 1]);
#pragma unroll
        for (int k = 0; k < NPO; ++k) {
          const u32x4 r = x_read128((unsigned)swz_slot<NPO>(64 * k + lane) * 16u);
          ull2& dstp = t ? ob[k] : oa[k];
          dstp.x = ((u64)r.y << 32) | r.x;
          dstp.y = ((u64)r.w << 32) | r.z;
        }
      }
      const size_t o0 = tile * kWave * NPO;
#pragma unroll
      for (int k = 0; k < NPO; ++k) {
        const size_t q = o0 + (size_t)k * kWave + lane;
        st16<kNtStore>(A2p + q, oa[k]);
        st16<kNtStore>(B2p + q, ob[k]);
      }
      accumulate_run<F, KS>(f, acc, a, b);
      bulk_writeback();
      tile = next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if constexpr (kPipe) {
    static_assert(KS == 2, "the pipelined folds are instantiated with KS = 2 only");
    // A tile (64 units = 256 outputs per table) is folded in four sub-steps of 64 outputs: the wave loads 8 KiB of each
    // table CONTIGUOUSLY (1 KiB per instruction, as everywhere), stores them to its LDS region, every lane reads its own
    // eight pieces = sixteen entries back and folds them with ONE lazy sum of sixteen products - output 64 s + lane of the
    // tile.  A last bounce through LDS hands lane l the quad 4 l .. 4 l + 3 that the two-round grid and the 16-byte
    // output pieces want.
    // Software pipeline: the moment a sub-step's registers have been written to LDS they are refilled with the next
    // sub-step (of this tile or of the wave's next tile), so loads are in flight WHILE the wave multiplies.  (With one batch of loads per two sub-steps and nothing in flight during the arithmetic the pass
    // ran at 5.5 TB/s: 8 waves x 32 KiB per CU take ~9.5 us to arrive and each wave then computed ~2 us with nothing
    // requested.)  LDS ordering inside the wave by wave_lds_sync, NOT wave_lds_fence: the fence would wait for vmcnt(0),
    // i.e. for the prefetched loads.
    // Which sub-step a wave is on decides which HBM channels it reads: a sub-step is 8 KiB of each table at tile * 32 KiB +
    // o * 8 KiB, the tiles of a block's eight waves lie a multiple of 8 MiB apart (the same channels), and the waves of a
    // block - and, through the memory system's back-pressure, the blocks of the grid - fall into step.  Taken in the same
    // order everywhere, a quarter of the channels would serve the whole chip at any moment (with the multiplications removed
    // - perfect lockstep - the pass took 1 530 us instead of 800).  So wave w walks the sub-steps in the order
    // (i + w) mod 4 for table a and (i + w + 2) mod 4 for table b.
    ull2* const reg = my_lds;                                   // 8 KiB: one table's sub-step
    u64* const xchg = reinterpret_cast<u64*>(my_lds + kWave * NPS);   // 4 KiB: the tile's 256 + 256 outputs
    const int rot = wave & 3;
    auto load_sub = [&](const ull2* __restrict__ T, size_t tile, int o, ull2 (&p)[NPS]) {
      // (whole tiles only: the host sends this pass tables of >= 2^12 entries)
      const ull2* src = T + tile * kWave * NP + (size_t)o * NPS * kWave + lane;
#pragma unroll
      for (int k = 0; k < NPS; ++k) p[k] = ld16<kNtLoad>(src + k * kWave);
    };
    auto stash = [&](const ull2 (&p)[NPS]) {
#pragma unroll
      for (int k = 0; k < NPS; ++k) reg[swz_slot<NPS>(64 * k + lane)] = p[k];
    };
    auto fold16 = [&]() -> u64 {
      typename F::Acc3 s;
      f.acc3_zero(s);
#pragma unroll
      for (int m = 0; m < NPS; ++m) {
        const ull2 x = reg[swz_slot<NPS>(NPS * lane + m)];
        f.acc3_mac(s, x.x, fw.w[2 * m]);
        f.acc3_mac(s, x.y, fw.w[2 * m + 1]);
      }
      return f.acc3_get(s);
    };
    ull2 pa[NPS], pb[NPS];
    size_t tile = next_tile();
    if (tile < n_tiles) {
      load_sub(Ap, tile, rot, pa);
      load_sub(Bp, tile, rot ^ 2, pb);
    }
    while (tile < n_tiles) {
      const size_t next = next_tile();
#pragma unroll
      for (int i = 0; i < OUT; ++i) {
        const int oa = (i + rot) & 3, ob = oa ^ 2, na = (i + 1 + rot) & 3;
        // table a, then table b, through the same 8 KiB: the moment a table's registers are in LDS they are refilled with
        // the wave's next sub-step (of this tile, or the first of its next tile), so loads are in flight while it multiplies
        stash(pa);
        wave_lds_sync();
        if (i + 1 < OUT) load_sub(Ap, tile, na, pa);
        else if (next < n_tiles) load_sub(Ap, next, na, pa);
        const u64 xa = fold16();
        wave_lds_sync();
        stash(pb);
        wave_lds_sync();
        if (i + 1 < OUT) load_sub(Bp, tile, na ^ 2, pb);
        else if (next < n_tiles) load_sub(Bp, next, na ^ 2, pb);
        const u64 xb = fold16();
        xchg[kWave * oa + lane] = xa;
        xchg[kWave * OUT + kWave * ob + lane] = xb;
        wave_lds_sync();
      }
      u64 a[OUT], b[OUT];
#pragma unroll
      for (int i = 0; i < OUT; ++i) {
        a[i] = xchg[OUT * lane + i];
        b[i] = xchg[kWave * OUT + OUT * lane + i];
      }
      wave_lds_sync();
      ull2* const reg_a = reg;
      // outputs: lane l holds pieces 2 l, 2 l + 1 of the tile's 128 output pieces per table; store as pieces 64 k + lane
      ull2 oa[NPO], ob[NPO];
#pragma unroll
      for (int m = 0; m < NPO; ++m) {
        reg_a[swz_slot<NPO>(NPO * lane + m)] = ull2{a[2 * m], a[2 * m + 1]};
        reg_a[kWave * NPO + swz_slot<NPO>(NPO * lane + m)] = ull2{b[2 * m], b[2 * m + 1]};
      }
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < NPO; ++k) {
        oa[k] = reg_a[swz_slot<NPO>(64 * k + lane)];
        ob[k] = reg_a[kWave * NPO + swz_slot<NPO>(64 * k + lane)];
      }
      wave_lds_sync();
      const size_t o0 = tile * kWave * NPO;
#pragma unroll
      for (int k = 0; k < NPO; ++k) {
        const size_t q = o0 + (size_t)k * kWave + lane;
        st16<kNtStore>(A2p + q, oa[k]);
        st16<kNtStore>(B2p + q, ob[k]);
      }
      accumulate_run<F, KS>(f, acc, a, b);
      bulk_writeback();
      tile = next;
    }
  } else if constexpr (kPrefetch) {
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

}  // namespace sc
