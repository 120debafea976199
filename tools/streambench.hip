// What the memory system of this box gives a plain streaming kernel (the ceiling the pass kernels
// are compared with in DESIGN.md):  hipcc -O3 --offload-arch=gfx950 -o streambench streambench.hip
//   read1   one 4 GiB stream, 16-byte loads, 8 in flight per lane, xor-reduced
//   read2   two 2 GiB streams read together (the access pattern of a round-sum pass)
//   copy    read 2 GiB, write 2 GiB
//   r8w1    read 4 GiB, write 0.5 GiB (the read/write mix of the three-variable fold pass)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256;

template <int MODE>
__global__ void __launch_bounds__(kBlock)
stream_kernel(const ull2* __restrict__ A, const ull2* __restrict__ B, ull2* __restrict__ O, size_t n_pieces, u64* sink) {
  // tiles of 8 x 1 KiB per wave, interleaved over the waves of the grid
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * kBlock) >> 6;
  const size_t n_tiles = n_pieces / 512;
  u64 acc = 0;
  for (size_t t = wave; t < n_tiles; t += n_waves) {
    ull2 v[8], w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = __builtin_nontemporal_load(A + t * 512 + k * 64 + lane);
      if (MODE == 1) w[k] = __builtin_nontemporal_load(B + t * 512 + k * 64 + lane);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc ^= v[k].x + v[k].y;
      if (MODE == 1) acc ^= w[k].x * 3 + w[k].y;
      if (MODE == 2) O[t * 512 + k * 64 + lane] = v[k];
    }
    if (MODE == 3) {  // one output piece per eight input pieces
      ull2 o = {acc, acc + 1};
      O[t * 64 + lane] = o;
    }
  }
  if (acc == 0x1234567) sink[0] = acc;  // keep the loads alive
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)4 << 30, pieces = bytes / 16;
  ull2 *buf = nullptr, *out = nullptr;
  u64* sink = nullptr;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&out, bytes / 2));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, bytes));
  CK(hipMemset(out, 0, bytes / 2));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int grid : {512, 768, 1024, 2048}) {
    for (int mode = 0; mode < 4; ++mode) {
      std::vector<float> ms;
      for (int it = 0; it < 7; ++it) {
        CK(hipEventRecord(e0));
        switch (mode) {
          case 0: hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 1: hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(kBlock), 0, 0, buf, buf + pieces / 2, out, pieces / 2, sink); break;
          case 2: hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces / 2, sink); break;
          default: hipLaunchKernelGGL(stream_kernel<3>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      const double moved = mode == 0 ? bytes : mode == 1 ? bytes : mode == 2 ? bytes : bytes + bytes / 8.0;
      const char* names[] = {"read1", "read2", "copy ", "r8w1 "};
      printf("grid %4d  %s  %.1f us  %.2f TB/s\n", grid, names[mode], ms[3] * 1e3, moved / (ms[3] * 1e-3) / 1e12);
    }
  }
  return 0;
}
