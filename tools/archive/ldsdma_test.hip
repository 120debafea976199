// semantics check of __builtin_amdgcn_global_load_lds (16-byte form) on gfx950 (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glob_ptr_t;

__global__ void k(const ull2* __restrict__ in, ull2* __restrict__ out, int perm) {
  __shared__ __attribute__((aligned(16))) ull2 buf[4 * 64 * 4];  // 4 waves x 4 pieces x 64 lanes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ull2* my = buf + wave * 256;
  const size_t base = ((size_t)blockIdx.x * 4 + wave) * 256;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    // lane fetches piece (64*kk + (lane ^ perm)) of the wave tile; it lands at slot 64*kk + lane
    const ull2* src = in + base + 64 * kk + (lane ^ perm);
    __builtin_amdgcn_global_load_lds((glob_ptr_t)src, (lds_ptr_t)(my + 64 * kk), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // read back "run order": lane reads slots 4*lane .. 4*lane+3
#pragma unroll
  for (int m = 0; m < 4; ++m) out[base + 4 * lane + m] = my[4 * lane + m];
}
int main() {
  const int blocks = 64; const size_t n = (size_t)blocks * 4 * 256;
  ull2 *h = (ull2*)malloc(n * 16), *r = (ull2*)malloc(n * 16), *d_in, *d_out;
  for (size_t i = 0; i < n; ++i) { h[i].x = 2 * i; h[i].y = 2 * i + 1; }
  CK(hipMalloc(&d_in, n * 16)); CK(hipMalloc(&d_out, n * 16));
  CK(hipMemcpy(d_in, h, n * 16, hipMemcpyHostToDevice));
  for (int perm : {0, 5}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_in, d_out, perm);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r, d_out, n * 16, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t t = 0; t < n / 256; ++t)
      for (int s = 0; s < 256; ++s) {
        // slot s of tile t holds piece 64*(s/64) + ((s%64) ^ perm); out[base + s] = slot s
        size_t piece = t * 256 + 64 * (s / 64) + ((s % 64) ^ perm);
        if (r[t * 256 + s].x != 2 * piece || r[t * 256 + s].y != 2 * piece + 1) ++bad;
      }
    printf("perm=%d: %zu mismatches of %zu\n", perm, bad, n);
  }
  return 0;
}
