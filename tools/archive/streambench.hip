// What the memory system of this box gives a plain streaming kernel (the ceiling the pass kernels
// are compared with in DESIGN.md):  hipcc -O3 --offload-arch=gfx950 -o streambench streambench.hip
//   read1   one 4 GiB stream, 16-byte loads, 8 in flight per lane, xor-reduced
//   read2   two 2 GiB streams read together (the access pattern of a round-sum pass)
//   copy    read 2 GiB, write 2 GiB
//   r8w1    read 4 GiB, write 0.5 GiB (the read/write mix of the three-variable fold pass)
#include <hip/hip_runtime.h>
#include "../../thaler-study_amd/csrc/field.hpp"
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
  __shared__ u64 lw[1024];
  for (int i = threadIdx.x; i < 1024; i += kBlock) lw[i] = 0x9E3779B97F4A7C15ull * (i + 1);
  __syncthreads();
  sc::GoldilocksMont F;
  sc::GoldilocksMont::Acc a0, a1;
  F.acc_zero(a0);
  F.acc_zero(a1);
  const size_t n_iter = (n_tiles + n_waves - 1) / n_waves;
  for (size_t it = 0; it < n_iter; ++it) {
    // MODE 9: the wave walks contiguous segments of 16 tiles (128 KiB), segments interleaved over waves
    const size_t t = (MODE == 9) ? (((it >> 4) * n_waves + wave) << 4 | (it & 15)) : wave + it * n_waves;
    if (t >= n_tiles) continue;
    ull2 v[8], w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 10 || MODE == 11) {  // plain (temporal) loads
        v[k] = A[t * 512 + k * 64 + lane];
        if (MODE == 11) w[k] = B[t * 512 + k * 64 + lane];
      } else {
        v[k] = __builtin_nontemporal_load(A + t * 512 + k * 64 + lane);
        if (MODE == 1) w[k] = __builtin_nontemporal_load(B + t * 512 + k * 64 + lane);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc ^= v[k].x + v[k].y;
      if (MODE == 1 || MODE == 11) acc ^= w[k].x * 3 + w[k].y;
      if (MODE == 2) O[t * 512 + k * 64 + lane] = v[k];
      if (MODE == 12) __builtin_nontemporal_store(v[k], O + t * 512 + k * 64 + lane);
      if (MODE == 7 || MODE == 8 || MODE == 9) {  // the library's lazy multiply-accumulate, two per piece
        const u64 w = (MODE == 7) ? lw[(t * 8 + k) & 1023] : 0x123456789ABCDEFull;
        F.acc_mac(a0, v[k].x, w);
        F.acc_mac(a1, v[k].y, w);
      }
      if (MODE >= 4 && MODE < 7) {  // read1 plus (MODE - 3) * 8 dependent 64-bit multiply-adds per piece
        u64 x = v[k].x, y = v[k].y;
#pragma unroll
        for (int r = 0; r < (MODE - 3) * 8; ++r) { x = x * y + (u64)r; y = y * x + 1; }
        acc ^= x ^ y;
      }
    }
    if (MODE == 3 || MODE == 13) {  // one output piece per eight input pieces
      ull2 o = {acc, acc + 1};
      if (MODE == 13) __builtin_nontemporal_store(o, O + t * 64 + lane);
      else O[t * 64 + lane] = o;
    }
  }
  if (MODE >= 7) acc ^= F.acc_get(a0) ^ F.acc_get(a1);
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
  for (int grid : {768, 2048}) {
    for (int mode : {2, 12, 3, 13}) {
      std::vector<float> ms;
      for (int it = 0; it < 7; ++it) {
        CK(hipEventRecord(e0));
        switch (mode) {
          case 0: hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 1: hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(kBlock), 0, 0, buf, buf + pieces / 2, out, pieces / 2, sink); break;
          case 2: hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces / 2, sink); break;
          case 3: hipLaunchKernelGGL(stream_kernel<3>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 4: hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 5: hipLaunchKernelGGL(stream_kernel<5>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 6: hipLaunchKernelGGL(stream_kernel<6>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 7: hipLaunchKernelGGL(stream_kernel<7>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 8: hipLaunchKernelGGL(stream_kernel<8>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 9: hipLaunchKernelGGL(stream_kernel<9>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 10: hipLaunchKernelGGL(stream_kernel<10>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
          case 11: hipLaunchKernelGGL(stream_kernel<11>, dim3(grid), dim3(kBlock), 0, 0, buf, buf + pieces / 2, out, pieces / 2, sink); break;
          case 12: hipLaunchKernelGGL(stream_kernel<12>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces / 2, sink); break;
          default: hipLaunchKernelGGL(stream_kernel<13>, dim3(grid), dim3(kBlock), 0, 0, buf, buf, out, pieces, sink); break;
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      const double moved = (mode == 3 || mode == 13) ? bytes + bytes / 8.0 : bytes;
      const char* names[] = {"read1", "read2", "copy ", "r8w1 ", "rd+16mul", "rd+32mul", "rd+48mul", "rd+mac(lds w)", "rd+mac(const w)", "rd+mac, 128 KiB segments per wave", "read1, plain loads", "read2, plain loads", "copy, nt stores", "r8w1, nt stores"};
      printf("grid %4d  %s  %.1f us  %.2f TB/s\n", grid, names[mode], ms[3] * 1e3, moved / (ms[3] * 1e-3) / 1e12);
    }
  }
  return 0;
}
