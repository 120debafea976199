// Part of kernels.hpp (included there, in order): the passes that serve up to five rounds (wgrid_pass_kernel, rank_pass_kernel, fold_wide_kernel).
#pragma once

namespace sc {

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
  int host_out = 0;  // the folded tables go to pinned HOST memory and the host reads them as soon as it has seen this launch's
                     // sequence word (option "host_tail_log"): they are stored write-through at system scope.  A plain store sits
                     // in the L2 of the storing block's XCD until that L2 is written back - the release of the block that
                     // publishes the sequence word writes back ITS XCD's only, the others' at the end of the kernel: a host that
                     // is quicker than that reads the previous proof's entries (seen once in 8 processes x 11 proofs, round 5)
};
// folded entry i of one table: sum_c w[c] * in[2^KF i + c], stored to the folded table
template <class F, int KF>
__device__ __forceinline__ u64 grid_fold1(const F& f, const u64* __restrict__ T, u64* __restrict__ T2, const GridW& gw, size_t i, bool sys_out = false) {
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
  if (sys_out) __hip_atomic_store(T2 + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else T2[i] = v;
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
                                          u64* __restrict__ B2, const GridW& gw, int kf, size_t n_out, bool sys_out) {
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
            if (sys_out) __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else dst[i] = v;
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
          case 1: v = grid_fold1<F, 1>(f, src, dst, gw, i, sys_out); break;
          case 2: v = grid_fold1<F, 2>(f, src, dst, gw, i, sys_out); break;
          case 3: v = grid_fold1<F, 3>(f, src, dst, gw, i, sys_out); break;
          case 4: v = grid_fold1<F, 4>(f, src, dst, gw, i, sys_out); break;
          default: v = grid_fold1<F, 5>(f, src, dst, gw, i, sys_out); break;
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

// Where the cells of a pass leave the device (every thread of the block calls it; `total` = cell tid of this rank for
// tid < CELLS): through the in-kernel exchange with the peers (sharded passes on the peer transport), as split limbs in
// device memory for the collective that follows on the stream (RCCL), or as whole residues in the wide mailbox + the
// sequence word (unsharded passes; sharded ones on a host transport or a multi-device handle - the host sums).
template <int CELLS>
__device__ __forceinline__ void publish_cells(const WgOut& out, u64 total) {
  const int tid = threadIdx.x;
  if (out.px.world > 0) {
    exchange_wide<CELLS>(out, total);
    return;
  }
  if (out.limbs_dev) {   // the stream's next operation (an all-reduce) reads them: kernel-boundary ordering
    if (tid < CELLS) write_split(out.limbs_dev, tid, total);
    return;
  }
  if (tid < CELLS) __hip_atomic_store(out.mailbox + kMailboxWide + tid, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // a workgroup barrier does not drain vmcnt: every storing wave waits for its own cell stores before the barrier, so
  // that thread 0's release store of the sequence word cannot overtake a late cell of waves 1..3 (as exchange_wide does)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(out.mailbox + kMailboxSeq, out.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// What every block does with its cells (thread c < 3^KS holds cell c in `total`): two ticket levels, the last block publishes.
// Every thread of the block calls it (blocks of 256 threads or more).
// Ordering (ADVICE r05 asked for it in the memory model rather than in asm): a block's row leaves as agent-scope atomic stores
// (write-through: they are in memory, not in this XCD's L2, once vmcnt has drained), EVERY storing wave drains (s_waitcnt vmcnt(0)),
// the workgroup barrier orders the waves, and only then does thread 0 draw the ticket; the block that draws the last one issues an
// agent-scope acquire before it loads the rows with agent-scope atomic loads.  The ticket itself is RELAXED on purpose: a RELEASE
// fetch_add makes the compiler put a buffer_wbl2 in front of it - a write-back of everything this XCD's L2 holds dirty, the folded
// tables included - which is what the stores' write-through and the drain already did for the only data the reader needs, and it
// costs 1.2-1.4 us per launch (tools/wfbench.hip with -DSC_TICKET_ORDER=__ATOMIC_RELEASE in round 6: wgrid(5,5)@21 23.3 against
// 21.8 us, wfold(4,5)@21 30.4 against 29.2; profiles/r06_wfold_lds_ab.txt).  What holds the relaxed form in place is measurement:
// tools/stress_grid.py, tools/stress_handover.py and the 8-process tests draw ~10^5 tickets per run with alternating instances, so
// that a stale row is a wrong transcript.
template <class F, int KS>
__device__ __forceinline__ void wgrid_finish(const F& f, u64 total, const WgOut& out) {
  constexpr int kPow3[6] = {1, 3, 9, 27, 81, 243};
  constexpr int cells = kPow3[KS];
  __shared__ int lds_flag;
  const int tid = threadIdx.x;
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
  publish_cells<cells>(out, total);
}

template <class F, int KS, bool PF>
__global__ void __launch_bounds__(kBlock)
wgrid_pass_kernel(F f, const u64* __restrict__ A, const u64* __restrict__ B, u64* __restrict__ A2, u64* __restrict__ B2,
                  GridW gw, int kf, size_t n_out, WgOut out) {
  const u64 total = wgrid_body<F, KS, PF>(f, A, B, A2, B2, gw, kf, n_out, out.host_out != 0);
  wgrid_finish<F, KS>(f, total, out);
}

// ------------------------------------------------------------------------------------
// wfold_pass_kernel (round 5): a fold of four or five challenges over a LARGE table that serves up to five rounds.
//
// On a 2^25-entry shard the fold pass behind the matrix-core pass (pass_kernel<4,2>, 100 us) and the five-round pass behind it
// (26 us + a launch) are a third of the proof, and what separates them is not work but a launch: the fold is bound by memory
// (28 % VALU-busy), the five-round pass by instruction issue per folded entry.  This kernel is pass_kernel<4,.>'s front end - a
// wave streams 8 KiB sub-steps of both tables (lane <-> 16-byte piece, 1 KiB contiguous per instruction), bounces them through
// its private LDS region so that every lane folds ITS sixteen entries with one lazy sum, the order of the sub-steps rotated from
// wave to wave so that the chip's requests cover all HBM channels - with wgrid_pass_kernel<F, KS>'s back end: every sub-step
// leaves 64 (KF = 4) or 32 (KF = 5: the halves of an output sit in neighbouring lanes, out = h0 + r_5 (h1 - h0)) folded entries
// per table, i.e. two or one iterations of wgrid_body - the 64 lanes drop 32 entries per table into the wave's extension arrays,
// fill the {0,1,inf}^KS grids level by level (LDS addresses decoded once) and multiply the cell pairs into four lazy
// accumulators per lane - done right behind the sub-step, while the next sub-step's loads are in flight (behind the whole tile
// instead: +3 us at n = 25).  (4, 5): ~2.3x the instructions of the two-round fold, still under its memory time - 112 us against
// 101 + 26 + a trip to the host at n = 25; (5, ks) on 2^23..2^25-entry tables: what wgrid_pass_kernel - made for tables that sit
// in the caches - streams at 4.5 TB/s.  Whole tiles only (tables of >= 2^12 entries); cells leave exactly as
// wgrid_pass_kernel's (wgrid_finish).  r_top: the fifth challenge (KF = 5), in the tables' representation.
constexpr int kWfThreads = 512;
template <class F, int KF, int KS, bool NT>
__global__ void __launch_bounds__(kWfThreads)
wfold_pass_kernel(F f, const u64* __restrict__ A, const u64* __restrict__ B, u64* __restrict__ A2, u64* __restrict__ B2, FoldW fw,
                  u64 r_top, size_t n_tiles, WgOut out) {
  static_assert(KF == 4 || KF == 5, "four or five pending challenges");
  static_assert(KS >= 1 && KS <= 5, "one to five rounds");
  constexpr int OUT = 4, NPS = 8, NP = 32;          // a tile: 4 sub-steps of 8 KiB (1024 entries) per table
  constexpr int OUTS = kWave >> (KF - 4);           // folded entries per table and sub-step
  constexpr int TILE_OUT = OUT * OUTS;              // ... and tile (256 / 128)
  constexpr int NPO = TILE_OUT / (2 * kWave);       // 16-byte stores per lane, table and tile (2 / 1)
  constexpr int kWaves = kWfThreads / kWave;
  constexpr int kPow3[6] = {1, 3, 9, 27, 81, 243};
  constexpr int cells = kPow3[KS], G = 1 << KS, gpi = kWgEntries >> KS, pairs = gpi * cells;
  // per wave: 8 KiB sub-step | 4 KiB tile outputs [table][256] | 4 KiB extension arrays [table][256]
  constexpr int kRegionWords = 1024 + 512 + 512;
  __shared__ u64 lds_all[kWaves * kRegionWords];
  __shared__ int cell_of[kWgEntries], suffix_of[kWgEntries];
  __shared__ unsigned lds_next;
  typedef __attribute__((address_space(3))) u64 lds_u64;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  if (tid < kWgEntries) {   // (wgrid_body's tables)
    int c = 0, u = 0, p3 = 1;
    for (int m = 0; m < KS; ++m) {
      c += ((tid >> (KS - 1 - m)) & 1) * p3;
      u += ((tid >> m) & 1) * p3;
      p3 *= 3;
    }
    cell_of[tid] = c;
    suffix_of[tid] = u;
  }
  if (tid == 0) lds_next = 0;
  __syncthreads();
  ull2* const reg = reinterpret_cast<ull2*>(lds_all + (size_t)wave * kRegionWords);
  u64* const xchg = lds_all + (size_t)wave * kRegionWords + 1024;
  u64* const ef = xchg + 512;
  const int tbl = lane >> 5, ent = lane & (kWgEntries - 1);
  const int slot = tbl * kGridChunk + (ent >> KS) * cells + cell_of[ent & (G - 1)];
  unsigned step[KS][3];
#pragma clang loop unroll(full)
  for (int j = 0; j < KS; ++j) {
    const int low = KS - 1 - j, pj = kPow3[j], stride = kPow3[low], items = (gpi * pj) << low;   // per table (32 entries)
    const unsigned inv = (1u << 20) / (unsigned)pj + 1u;
#pragma clang loop unroll(full)
    for (int q = 0; q < 3; ++q) {
      const int idx = lane + kWave * q;
      unsigned d = 0;
      if (idx < 2 * items) {
        const int tb = idx >= items ? 1 : 0, id = idx - tb * items;
        const int sfx = id & ((1 << low) - 1), t = id >> low;
        const int g = (int)(((unsigned)t * inv) >> 20), pp = t - g * pj;
        d = 0x80000000u | (unsigned)(size_t)(lds_u64*)(ef + tb * kGridChunk + g * cells + pp * 3 * stride + suffix_of[sfx]);
      }
      step[j][q] = d;
    }
  }
  typename F::Acc acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) f.acc_zero(acc[k]);

  const ull2* __restrict__ Ap = reinterpret_cast<const ull2*>(A);
  const ull2* __restrict__ Bp = reinterpret_cast<const ull2*>(B);
  ull2* __restrict__ A2p = reinterpret_cast<ull2*>(A2);
  ull2* __restrict__ B2p = reinterpret_cast<ull2*>(B2);
  const int rot = wave & 3;
  auto next_tile = [&]() -> size_t {
    unsigned c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(&lds_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (size_t)__builtin_amdgcn_readfirstlane(c) * gridDim.x + blockIdx.x;
  };
  auto load_sub = [&](const ull2* __restrict__ T, size_t tile, int o, ull2 (&p)[NPS]) {
    const ull2* src = T + tile * kWave * NP + (size_t)o * NPS * kWave + lane;
#pragma unroll
    for (int k = 0; k < NPS; ++k) p[k] = ld16<NT>(src + k * kWave);
  };
  auto stash = [&](const ull2 (&p)[NPS]) {
#pragma unroll
    for (int k = 0; k < NPS; ++k) reg[swz_slot<NPS>(64 * k + lane)] = p[k];
  };
  // the lane's sixteen entries under the first four challenges; KF = 5: the pair of lanes (2o, 2o + 1) holds the halves of output o
  auto fold16 = [&]() -> u64 {
    typename F::Acc3 sacc;
    f.acc3_zero(sacc);
#pragma unroll
    for (int m = 0; m < NPS; ++m) {
      const ull2 x = reg[swz_slot<NPS>(NPS * lane + m)];
      f.acc3_mac(sacc, x.x, fw.w[2 * m]);
      f.acc3_mac(sacc, x.y, fw.w[2 * m + 1]);
    }
    const u64 h = f.acc3_get(sacc);
    if constexpr (KF == 5) {
      const u64 hn = __shfl_xor(h, 1);
      return f.add(h, f.mul(r_top, f.sub(hn, h)));   // (what the even lane holds is the output)
    } else {
      return h;
    }
  };
  auto put = [&](u64* dst, int o, u64 x) {   // sub-step o's outputs into the tile's exchange area
    if constexpr (KF == 5) {
      if ((lane & 1) == 0) dst[OUTS * o + (lane >> 1)] = x;
    } else {
      dst[OUTS * o + lane] = x;
    }
  };
  // the cells over 32 folded entries per table (wgrid_body's iteration)
  auto grid_group = [&](int grp) {
    ef[slot] = xchg[kWave * OUT * tbl + kWgEntries * grp + ent];
    wave_lds_sync();
#pragma clang loop unroll(full)
    for (int j = 0; j < KS; ++j) {
      const int st = kPow3[KS - 1 - j];
#pragma clang loop unroll(full)
      for (int q = 0; q < 3; ++q) {
        if (2 * ((gpi * kPow3[j]) << (KS - 1 - j)) > kWave * q) {
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
    wave_lds_sync();
  };
  ull2 pa[NPS], pb[NPS];
  size_t tile = next_tile();
  if (tile < n_tiles) {
    load_sub(Ap, tile, rot, pa);
    load_sub(Bp, tile, rot, pb);
  }
  while (tile < n_tiles) {
    const size_t next = next_tile();
#pragma unroll
    for (int i = 0; i < OUT; ++i) {
      const int o = (i + rot) & 3, no = (i + 1 + rot) & 3;
      stash(pa);
      wave_lds_sync();
      if (i + 1 < OUT) load_sub(Ap, tile, no, pa);
      else if (next < n_tiles) load_sub(Ap, next, no, pa);
      const u64 xa = fold16();
      wave_lds_sync();
      stash(pb);
      wave_lds_sync();
      if (i + 1 < OUT) load_sub(Bp, tile, no, pb);
      else if (next < n_tiles) load_sub(Bp, next, no, pb);
      const u64 xb = fold16();
      put(xchg, o, xa);
      put(xchg + kWave * OUT, o, xb);
      wave_lds_sync();
      // the groups this sub-step completed, while the next sub-step's loads are in flight
#pragma unroll 1
      for (int g = 0; g < OUTS / kWgEntries; ++g) grid_group(o * (OUTS / kWgEntries) + g);
    }
    // the folded tables: the exchange area holds the tile's outputs in index order - stored straight out of it
    {
      const ull2* xo = reinterpret_cast<const ull2*>(xchg);
      const size_t o0 = tile * kWave * NPO;
#pragma unroll
      for (int k = 0; k < NPO; ++k) {
        const size_t q = o0 + (size_t)k * kWave + lane;
        st16<false>(A2p + q, xo[kWave * k + lane]);
        st16<false>(B2p + q, xo[kWave * OUT / 2 + kWave * k + lane]);
      }
    }
    wave_lds_sync();   // (the stores' LDS reads before the next tile's outputs)
    if (wave == 0 && blockIdx.x < 8) asm volatile("buffer_wbl2 sc1" ::: "memory");   // (the bulk write-back hint of pass_kernel's pipelined forms)
    tile = next;
  }
  // the block's cells: accumulators -> residues, the waves added through LDS (the wave regions are free now)
  __syncthreads();
  u64* const red = lds_all;   // [kWaves][kGridChunk]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = lane + kWave * k;
    red[wave * kGridChunk + p] = (p < pairs) ? f.acc_get(acc[k]) : 0;
  }
  __syncthreads();
  u64 total = 0;
  if (tid < cells) {
    for (int w = 0; w < kWaves; ++w)
      for (int g = 0; g < gpi; ++g) total = f.add(total, red[w * kGridChunk + g * cells + tid]);
  }
  wgrid_finish<F, KS>(f, total, out);
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

}  // namespace sc
