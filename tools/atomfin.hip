// How the cells of a pass should leave the blocks (round 6): the two-level ticket hand-off of wgrid_finish (store row, drain,
// ticket, read 32 rows, store group row, drain, ticket, read, publish, drain, sequence word: ~7 dependent trips to memory) against
// ONE returning atomic per limb: every block adds (1 << 48 | limb) to the cell's accumulator, the thread that sees count ==
// blocks - 1 come back holds the total and publishes the cell itself, tagged with the launch's sequence number (the host waits
// for 243 tagged pairs instead of one sequence word).  Not part of the product.   atomfin [blocks ...]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../thaler-study_amd/csrc/kernels.hpp"
using namespace sc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ u64 fake_total(int block, int tid, u64 salt) {
  u64 z = (u64)block * 0x9E3779B97F4A7C15ull + (u64)tid * 0xBF58476D1CE4E5B9ull + salt;
  z ^= z >> 29;
  return z % 0xFFFFFFFF00000001ull;
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS) ticket_kernel(GoldilocksMont f, WgOut out, u64 salt) {
  const u64 total = threadIdx.x < 243 ? fake_total(blockIdx.x, threadIdx.x, salt) : 0;
  wgrid_finish<GoldilocksMont, 5>(f, total, out);
}

// acc[2 c], acc[2 c + 1]: count << 48 | sum of the low / high 32-bit limbs of cell c; all zero between launches
template <int THREADS>
__global__ void __launch_bounds__(THREADS) atomic_kernel(GoldilocksMont f, u64* acc, u64* mailbox, u64 seq, u64 salt) {
  const int tid = threadIdx.x;
  if (tid >= 243) return;
  const u64 total = fake_total(blockIdx.x, tid, salt);
  const u64 one = 1ull << 48;
  const u64 lo = __hip_atomic_fetch_add(acc + 2 * tid, one | (total & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 hi = __hip_atomic_fetch_add(acc + 2 * tid + 1, one | (total >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 last = (u64)(gridDim.x - 1);
  // (the two counters of a cell may be completed by different blocks: each limb is published by whoever completes it)
  if ((lo >> 48) == last) {
    const u64 s = (lo & (one - 1)) + (total & 0xFFFFFFFFull);
    __hip_atomic_store(mailbox + kMailboxWide + 2 * tid, (seq << 48) | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(acc + 2 * tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if ((hi >> 48) == last) {
    const u64 s = (hi & (one - 1)) + (total >> 32);
    __hip_atomic_store(mailbox + kMailboxWide + 2 * tid + 1, (seq << 48) | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(acc + 2 * tid + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void empty_kernel(u64* mailbox, u64 seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(mailbox + kMailboxSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main(int argc, char** argv) {
  std::vector<int> grids;
  for (int i = 1; i < argc; ++i) grids.push_back(atoi(argv[i]));
  if (grids.empty()) grids = {16, 256, 512, 1024};
  u64 *P, *GR, *mb, *acc;
  unsigned* T;
  CK(hipMalloc(&P, (size_t)1024 * kGridChunk * 8)); CK(hipMalloc(&GR, (size_t)32 * kGridChunk * 8));
  CK(hipMalloc(&T, 64 * sizeof(unsigned))); CK(hipMemset(T, 0, 64 * sizeof(unsigned)));
  CK(hipMalloc(&acc, 512 * 8)); CK(hipMemset(acc, 0, 512 * 8));
  CK(hipHostMalloc(&mb, kMailboxWords * 8 + 64, hipHostMallocMapped)); memset(mb, 0, kMailboxWords * 8 + 64);
  GoldilocksMont f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  u64 seq = 0;
  const int reps = 200;
  for (int grid : grids) {
    // expected totals on the host
    const u64 p = 0xFFFFFFFF00000001ull;
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<float> ts, host_us;
      bool ok = true;
      for (int r = 0; r < reps; ++r) {
        ++seq;
        const u64 salt = seq * 77;
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, mb, seq);
        else if (mode == 1) {
          WgOut wo; wo.partials = P; wo.group_rows = GR; wo.tickets = T; wo.mailbox = mb; wo.seq = seq; wo.limbs_dev = nullptr;
          hipLaunchKernelGGL(ticket_kernel<256>, dim3(grid), dim3(256), 0, 0, f, wo, salt);
        } else hipLaunchKernelGGL(atomic_kernel<256>, dim3(grid), dim3(256), 0, 0, f, acc, mb, seq & 0xFFFF, salt);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
        if (mode && (r % 50) == 0) {   // check the totals
          for (int c = 0; c < 243; ++c) {
            unsigned __int128 want = 0;
            for (int b = 0; b < grid; ++b) {
              u64 z = (u64)b * 0x9E3779B97F4A7C15ull + (u64)c * 0xBF58476D1CE4E5B9ull + salt;
              z ^= z >> 29;
              want += z % p;
            }
            u64 got;
            if (mode == 1) got = mb[kMailboxWide + c];
            else {
              const u64 lo = mb[kMailboxWide + 2 * c], hi = mb[kMailboxWide + 2 * c + 1];
              if ((lo >> 48) != (seq & 0xFFFF) || (hi >> 48) != (seq & 0xFFFF)) ok = false;
              const unsigned __int128 v = (unsigned __int128)(lo & ((1ull << 48) - 1)) + ((unsigned __int128)(hi & ((1ull << 48) - 1)) << 32);
              got = (u64)(v % p);
            }
            if (got != (u64)(want % p)) ok = false;
          }
        }
      }
      std::sort(ts.begin(), ts.end());
      printf("grid %4d  %-7s median %.2f us  min %.2f  p90 %.2f  %s\n", grid, mode == 0 ? "empty" : mode == 1 ? "ticket" : "atomic", ts[reps / 2] * 1e3, ts[0] * 1e3,
             ts[reps * 9 / 10] * 1e3, ok ? "totals ok" : "TOTALS WRONG");
    }
  }
  return 0;
}
