// launch -> mailbox round trip of a trivial kernel (the floor of a launch-per-pass design):
//   hipcc -O2 --offload-arch=gfx950 -o launchlat launchlat.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void ping(uint64_t* mailbox, uint64_t seq) {
  if (threadIdx.x == 0) __hip_atomic_store(mailbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint64_t *h = nullptr, *d = nullptr;
  CK(hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&d, h, 0));
  *h = 0;
  const int iters = 20000;
  double launch_us = 0, wait_us = 0;
  for (int it = 1; it <= iters + 100; ++it) {
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(ping, dim3(1), dim3(64), 0, s, d, (uint64_t)it);
    auto t1 = std::chrono::steady_clock::now();
    while (__atomic_load_n(h, __ATOMIC_ACQUIRE) != (uint64_t)it) __builtin_ia32_pause();
    auto t2 = std::chrono::steady_clock::now();
    if (it > 100) {
      launch_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
      wait_us += std::chrono::duration<double, std::micro>(t2 - t1).count();
    }
  }
  printf("trivial kernel: hipLaunchKernelGGL returns after %.2f us, mailbox word seen %.2f us later (%.2f us per round trip)\n",
         launch_us / iters, wait_us / iters, (launch_us + wait_us) / iters);
  return 0;
}
