// How should the one written byte per eight read bytes of the fold pass leave the chip?  hipcc -O3 --offload-arch=gfx950 -o writemix writemix.hip
// A wave reads tiles of 8 KiB (eight 1 KiB wave loads, nontemporal) and writes 1 KiB per tile:
//   R = 1        after every tile (what pass_kernel does)
//   R = 2, 4, 8  the wave walks R ADJACENT tiles, keeps their outputs in registers and writes R contiguous KiB at the end
// with nontemporal, plain and write-through (sc1) stores, for 256 blocks of 512 threads and 512 blocks of 256.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));

template <int R, int ST, int BS>
__global__ void __launch_bounds__(BS)
mix_kernel(const ull2* __restrict__ A, ull2* __restrict__ O, size_t n_tiles, u64* sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * BS + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * BS) >> 6;
  const size_t n_runs = n_tiles / R;
  u64 acc = 0;
  for (size_t run = wave; run < n_runs; run += n_waves) {
    ull2 o[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const size_t t = run * R + j;
      ull2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(A + t * 512 + k * 64 + lane);
      u64 x = 0, y = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { x += v[k].x * 3; y ^= v[k].y + x; }
      o[j].x = x; o[j].y = y;
      acc ^= x;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      ull2* p = O + (run * R + j) * 64 + lane;
      if (ST == 0) __builtin_nontemporal_store(o[j], p);
      else if (ST == 1) *p = o[j];
      else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(o[j]) : "memory");
    }
  }
  if (acc == 0x1234567) sink[0] = acc;
}

template <int R, int ST, int BS>
double run(int grid, const ull2* buf, ull2* out, size_t n_tiles, u64* sink, hipEvent_t e0, hipEvent_t e1) {
  std::vector<float> ms;
  for (int it = 0; it < 7; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<R, ST, BS>), dim3(grid), dim3(BS), 0, 0, buf, out, n_tiles, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[3];
}

int main() {
  const size_t bytes = (size_t)4 << 30, n_tiles = bytes / 8192;
  ull2 *buf = nullptr, *out = nullptr;
  u64* sink = nullptr;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, bytes / 8));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, bytes));
  CK(hipMemset(out, 0, bytes / 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double moved = bytes + bytes / 8.0;
  const char* st[] = {"nt", "plain", "sc1"};
#define ROW(R, ST)                                                                                                       \
  {                                                                                                                      \
    const double a = run<R, ST, 512>(256, buf, out, n_tiles, sink, e0, e1), b = run<R, ST, 256>(512, buf, out, n_tiles, sink, e0, e1), \
                 c = run<R, ST, 256>(768, buf, out, n_tiles, sink, e0, e1);                                              \
    printf("R = %d, %-5s stores: 256 x 512 %.1f us %.2f TB/s | 512 x 256 %.1f us %.2f TB/s | 768 x 256 %.1f us %.2f TB/s\n", R, st[ST], a * 1e3,           \
           moved / (a * 1e-3) / 1e12, b * 1e3, moved / (b * 1e-3) / 1e12, c * 1e3, moved / (c * 1e-3) / 1e12);          \
  }
  for (int rep = 0; rep < 2; ++rep) {
    ROW(1, 0) ROW(2, 0) ROW(4, 0) ROW(8, 0)
    ROW(1, 1) ROW(2, 1) ROW(4, 1) ROW(8, 1)
    ROW(1, 2) ROW(4, 2)
  }
  return 0;
}
