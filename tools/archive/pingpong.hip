// host <-> resident-kernel round-trip latency through pinned host memory (not part of the product)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
template <int MODE>
__global__ void pong(u64* mailbox, const u64* inbox, int iters, int sleep_n) {
  if (threadIdx.x != 0) return;
  for (int i = 1; i <= iters; ++i) {
    u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (true) {
      u64 got = MODE == 0 ? __hip_atomic_load(inbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                          : __hip_atomic_load(inbox, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (got == (u64)i) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) return;  // 2 s
      if (sleep_n) __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(mailbox, (u64)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main() {
  u64 *h_mail, *h_in, *d_mail, *d_in;
  CK(hipHostMalloc(&h_mail, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc(&h_in, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&d_mail, h_mail, 0));
  CK(hipHostGetDevicePointer((void**)&d_in, h_in, 0));
  // device-resident inbox written by the host through a mapped pointer, if the runtime allows it
  u64* dev_in = nullptr;
  bool fine = hipExtMallocWithFlags((void**)&dev_in, 64, hipDeviceMallocFinegrained) == hipSuccess;
  const int iters = 2000;
  for (int variant = 0; variant < 4; ++variant) {
    *h_mail = 0; *h_in = 0;
    int sleep_n = variant & 1;
    if (variant < 2) hipLaunchKernelGGL(pong<0>, dim3(1), dim3(64), 0, 0, d_mail, (const u64*)d_in, iters, sleep_n);
    else hipLaunchKernelGGL(pong<1>, dim3(1), dim3(64), 0, 0, d_mail, (const u64*)d_in, iters, sleep_n);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= iters; ++i) {
      __atomic_store_n(h_in, (u64)i, __ATOMIC_RELEASE);
      while (__atomic_load_n(h_mail, __ATOMIC_ACQUIRE) != (u64)i) { __builtin_ia32_pause(); }
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipDeviceSynchronize());
    printf("pinned-host inbox, %s poll, sleep=%d: %.2f us per round trip\n", variant < 2 ? "relaxed" : "acquire", sleep_n, us / iters);
  }
  printf("fine-grained device alloc %s\n", fine ? "ok" : "unavailable");
  if (fine) {
    // can the host write it directly?
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dev_in) == hipSuccess) printf("  hostPointer=%p devicePointer=%p\n", attr.hostPointer, attr.devicePointer);
  }
  return 0;
}
