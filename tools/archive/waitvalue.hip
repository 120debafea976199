// How long from "the host knows the challenge" to "the next pass's result is back" - three ways of getting a small kernel going
// (not part of the product; round 5, the launch gap of the latency-bound passes):
//   A. the product's way: the host calls hipLaunchKernelGGL when it has the value (kernel arguments carry it)
//   B. pre-enqueued behind hipStreamWaitValue64: the launch call is made BEFORE the value is known, the kernel reads the value
//      from pinned host memory; the host releases the stream with one store
//   C. pre-launched kernel that spins on pinned host memory for the value (it occupies a CU while the previous kernel runs -
//      here there is no previous kernel, so this is the floor: the pingpong round trip)
// Each prints the median wall time from the host's store/launch to the mailbox word of the kernel, over 2000 rounds.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
__global__ void by_arg(u64* mailbox, u64 v) { if (threadIdx.x == 0) __hip_atomic_store(mailbox, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void by_mem(u64* mailbox, const u64* box) {
  if (threadIdx.x == 0) __hip_atomic_store(mailbox, __hip_atomic_load(box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void by_spin(u64* mailbox, const u64* box, u64 want) {
  if (threadIdx.x != 0) return;
  u64 t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(box, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want)
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) return;
  __hip_atomic_store(mailbox, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  u64 *h_mail, *d_mail, *h_box, *d_box, *h_gate, *d_gate;
  CK(hipHostMalloc(&h_mail, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc(&h_box, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&d_mail, h_mail, 0));
  CK(hipHostGetDevicePointer((void**)&d_box, h_box, 0));
  int can = 0;
  (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  const int iters = 2000;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  std::vector<double> t;
  *h_mail = 0;
  for (int i = 1; i <= iters; ++i) {   // A
    auto t0 = now();
    hipLaunchKernelGGL(by_arg, dim3(1), dim3(64), 0, s, d_mail, (u64)i);
    while (__atomic_load_n(h_mail, __ATOMIC_ACQUIRE) != (u64)i) __builtin_ia32_pause();
    t.push_back(us(t0, now()));
    CK(hipStreamSynchronize(s));
  }
  printf("A  launch when the value is known:            median %.2f us\n", med(t));
  // the gate: signal memory if the runtime has it, else pinned host memory
  bool signal_mem = hipExtMallocWithFlags((void**)&d_gate, 8, hipMallocSignalMemory) == hipSuccess;
  if (signal_mem) { h_gate = d_gate; printf("gate: hipMallocSignalMemory %p\n", (void*)d_gate); }
  else { (void)hipGetLastError(); CK(hipHostMalloc(&h_gate, 64, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&d_gate, h_gate, 0)); printf("gate: pinned host memory\n"); }
  if (can) {
    for (int variant = 0; variant < 2; ++variant) {
      t.clear();
      *h_mail = 0; *h_gate = 0;
      bool ok = true;
      for (int i = 1; i <= iters && ok; ++i) {
        hipError_t e = hipStreamWaitValue64(s, d_gate, (uint64_t)i, variant ? hipStreamWaitValueGte : hipStreamWaitValueEq, ~0ull);
        if (e != hipSuccess) { printf("hipStreamWaitValue64: %s\n", hipGetErrorString(e)); ok = false; break; }
        hipLaunchKernelGGL(by_mem, dim3(1), dim3(64), 0, s, d_mail, (const u64*)d_box);
        // ... the previous kernel would run here; the host learns the value:
        for (volatile int k = 0; k < 2000; ++k) {}
        auto t0 = now();
        __atomic_store_n(h_box, (u64)i, __ATOMIC_RELAXED);
        __atomic_store_n(h_gate, (u64)i, __ATOMIC_RELEASE);
        while (__atomic_load_n(h_mail, __ATOMIC_ACQUIRE) != (u64)i) __builtin_ia32_pause();
        t.push_back(us(t0, now()));
        CK(hipStreamSynchronize(s));
      }
      if (ok) printf("B%d pre-enqueued behind hipStreamWaitValue64 (%s): median %.2f us\n", variant, variant ? "Gte" : "Eq", med(t));
    }
  }
  t.clear();
  *h_mail = 0; *h_box = 0;
  for (int i = 1; i <= iters; ++i) {   // C
    hipLaunchKernelGGL(by_spin, dim3(1), dim3(64), 0, s, d_mail, (const u64*)d_box, (u64)i);
    for (volatile int k = 0; k < 20000; ++k) {}   // the kernel is resident and spinning by now
    auto t0 = now();
    __atomic_store_n(h_box, (u64)i, __ATOMIC_RELEASE);
    while (__atomic_load_n(h_mail, __ATOMIC_ACQUIRE) != (u64)i) __builtin_ia32_pause();
    t.push_back(us(t0, now()));
    CK(hipStreamSynchronize(s));
  }
  printf("C  resident kernel spinning on pinned memory:  median %.2f us\n", med(t));
  return 0;
}
