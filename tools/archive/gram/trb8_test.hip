// semantics check of ds_read_b64_tr_b8 on gfx950 (not part of the product): which byte of an LDS image a lane receives
// build: hipcc --offload-arch=gfx950 -O2 -o trb8_test trb8_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2i* lds_v2i;
// image: 16 rows of 256 bytes; byte (row, col) holds a 16-bit id split over two images (hi / lo) so that every byte is identifiable
__global__ void k(unsigned long long* out_lo, unsigned long long* out_hi, int variant) {
  __shared__ __attribute__((aligned(16))) unsigned char lo[16 * 256], hi[16 * 256];
  for (int i = threadIdx.x; i < 16 * 256; i += 64) { lo[i] = (unsigned char)(i & 255); hi[i] = (unsigned char)(i >> 8); }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  int q, p;
  if (variant == 0) { q = l >> 1; p = l & 1; }   // lane 2q+p: row q, columns 8p .. 8p+7
  else { q = l & 7; p = l >> 3; }                // lane q + 8p
  const unsigned addr = (unsigned)(q * 256 + 16 * g + 8 * p);
  v2i a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i)(lo + addr));
  v2i b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i)(hi + addr));
  out_lo[lane] = ((unsigned long long)(unsigned)a.y << 32) | (unsigned)a.x;
  out_hi[lane] = ((unsigned long long)(unsigned)b.y << 32) | (unsigned)b.x;
}
int main() {
  unsigned long long *d_lo, *d_hi, h_lo[64], h_hi[64];
  CK(hipMalloc(&d_lo, 512)); CK(hipMalloc(&d_hi, 512));
  for (int variant = 0; variant < 2; ++variant) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_lo, d_hi, variant);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_lo, d_lo, 512, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_hi, d_hi, 512, hipMemcpyDeviceToHost));
    printf("variant %d (lane -> address: %s)\n", variant, variant == 0 ? "l = 2q+p: row q, cols 16g+8p" : "l = q+8p: row q, cols 16g+8p");
    for (int lane = 0; lane < 64; ++lane) {
      printf("  lane %2d:", lane);
      for (int e = 0; e < 8; ++e) {
        const int id = (int)((h_lo[lane] >> (8 * e)) & 255) | ((int)((h_hi[lane] >> (8 * e)) & 255) << 8);
        printf(" (r%d,c%d)", id >> 8, id & 255);
      }
      printf("\n");
      if (lane == 17 && variant == 1) break;
    }
  }
  return 0;
}
