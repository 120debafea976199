// Can the host write a command word straight into (fine-grained) device memory that a running kernel polls?
// Measures the host -> kernel -> host round trip for (a) the command in pinned host memory polled over PCIe,
// (b) the command in fine-grained device memory written by the host through the BAR.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void pong(volatile unsigned long long* cmd, unsigned long long* reply, int iters) {
  for (int i = 1; i <= iters; ++i) {
    while (__hip_atomic_load((unsigned long long*)cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)i) __builtin_amdgcn_s_sleep(1);
    __hip_atomic_store(reply, (unsigned long long)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double run(unsigned long long* cmd_host_view, unsigned long long* cmd_dev_view, const char* what) {
  unsigned long long *reply_h, *reply_d;
  CK(hipHostMalloc(&reply_h, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&reply_d, reply_h, 0));
  *reply_h = 0;
  *cmd_host_view = 0;
  const int iters = 2000;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipLaunchKernelGGL(pong, dim3(1), dim3(1), 0, s, cmd_dev_view, reply_d, iters);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 1; i <= iters; ++i) {
    __atomic_store_n(cmd_host_view, (unsigned long long)i, __ATOMIC_RELEASE);
    while (__atomic_load_n(reply_h, __ATOMIC_ACQUIRE) != (unsigned long long)i) __builtin_ia32_pause();
  }
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  CK(hipStreamSynchronize(s));
  printf("%-58s round trip %.2f us\n", what, us);
  return us;
}
int main() {
  unsigned long long *h, *d;
  CK(hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&d, h, 0));
  run(h, d, "command in pinned host memory (kernel polls over PCIe):");
  unsigned long long* fg = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&fg, 4096, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { printf("fine-grained alloc failed: %s\n", hipGetErrorString(e)); return 0; }
  hipPointerAttribute_t at;
  CK(hipPointerGetAttributes(&at, fg));
  printf("fine-grained device allocation: hostPointer=%p devicePointer=%p\n", at.hostPointer, at.devicePointer);
  fflush(stdout);
  run(fg, fg, "command in fine-grained device memory (host writes through the BAR):");
  return 0;
}
