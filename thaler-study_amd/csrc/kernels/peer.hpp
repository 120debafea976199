// Part of kernels.hpp (included there, in order): limb splitting for collectives, the peer transport's gather / hello / exchange kernels, the mailbox copies.
#pragma once

namespace sc {

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
