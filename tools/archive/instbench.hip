// VALU instruction throughput microbenchmark for gfx950 (not part of the product).
// Each kernel runs ITER iterations of 16 independent copies of one instruction per lane;
// with 8 waves/SIMD resident the SIMD pipe is saturated, so time/instr = issue cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 2000;
#define REP16(X) X X X X X X X X X X X X X X X X

#define DEFKERNEL(NAME, ASMSTR, ...)                                            \
  __global__ void NAME(unsigned long long* out, unsigned long long seed) {      \
    unsigned long long a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55; \
    unsigned int x = (unsigned int)a, y = (unsigned int)b, z = (unsigned int)c;  \
    for (int i = 0; i < ITER; ++i) {                                            \
      REP16(asm volatile(ASMSTR : __VA_ARGS__);)                                \
    }                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + x + y + z;         \
  }

DEFKERNEL(k_add_u32, "v_add_u32 %0, %1, %2", "+v"(x) : "v"(y), "v"(z))
DEFKERNEL(k_add_co, "v_add_co_u32 %0, vcc, %1, %2", "+v"(x) : "v"(y), "v"(z) : "vcc")
DEFKERNEL(k_addc_co, "v_addc_co_u32 %0, vcc, %1, %2, vcc", "+v"(x) : "v"(y), "v"(z) : "vcc")
DEFKERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc", "+v"(x) : "v"(y), "v"(z) : "vcc")
DEFKERNEL(k_lshl_add_u64, "v_lshl_add_u64 %0, %1, 0, %2", "+v"(a) : "v"(b), "v"(c))
DEFKERNEL(k_cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %1", "+v"(a) : "v"(b) : "vcc")
DEFKERNEL(k_cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %1", "+v"(x) : "v"(y) : "vcc")
DEFKERNEL(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "+v"(a) : "v"(y), "v"(z) : "vcc")
DEFKERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %2", "+v"(x) : "v"(y), "v"(z))
DEFKERNEL(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %2", "+v"(x) : "v"(y), "v"(z))
DEFKERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0", "+v"(x) : "v"(y), "v"(z))
DEFKERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "+v"(x) : "v"(y))
DEFKERNEL(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %1", "+v"(a) : "v"(b))
DEFKERNEL(k_sub_co, "v_sub_co_u32 %0, vcc, %1, %2", "+v"(x) : "v"(y), "v"(z) : "vcc")
DEFKERNEL(k_add3, "v_add3_u32 %0, %1, %2, %0", "+v"(x) : "v"(y), "v"(z))
DEFKERNEL(k_fma_f64, "v_fma_f64 %0, %1, %2, %0", "+v"(a) : "v"(b), "v"(c))
DEFKERNEL(k_mul_f64, "v_mul_f64 %0, %1, %2", "+v"(a) : "v"(b), "v"(c))
DEFKERNEL(k_fma_f32, "v_fma_f32 %0, %1, %2, %0", "+v"(x) : "v"(y), "v"(z))
DEFKERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %1, %2", "+v"(x) : "v"(y), "v"(z))
DEFKERNEL(k_mul_u32_u24, "v_mul_u32_u24 %0, %1, %2", "+v"(x) : "v"(y), "v"(z))
DEFKERNEL(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %2", "+v"(x) : "v"(y), "v"(z))

template <class K> void run(const char* name, K kern, unsigned long long* d, int waves_per_simd) {
  int blocks = 256 * waves_per_simd;  // 256-thread blocks: 4 waves = 1 per SIMD each
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345ull);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345ull + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double wave_instrs_per_simd = (double)ITER * 16 * waves_per_simd;  // per SIMD
  double ns_per_instr = best * 1e6 / wave_instrs_per_simd;
  printf("%-18s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD  (= %.1f cycles @2.4GHz)\n", name, waves_per_simd, best,
         ns_per_instr, ns_per_instr * 2.4);
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 256 * 8 * 256 * 8));
  for (int w : {1, 4, 8}) {
    run("v_add_u32", k_add_u32, d, w);
    run("v_add_co_u32", k_add_co, d, w);
    run("v_addc_co_u32", k_addc_co, d, w);
    run("v_sub_co_u32", k_sub_co, d, w);
    run("v_add3_u32", k_add3, d, w);
    run("v_cndmask_b32", k_cndmask, d, w);
    run("v_cmp_lt_u32", k_cmp_lt_u32, d, w);
    run("v_cmp_lt_u64", k_cmp_lt_u64, d, w);
    run("v_lshl_add_u64", k_lshl_add_u64, d, w);
    run("v_lshlrev_b64", k_lshlrev_b64, d, w);
    run("v_mad_u64_u32", k_mad_u64_u32, d, w);
    run("v_mul_lo_u32", k_mul_lo_u32, d, w);
    run("v_mul_hi_u32", k_mul_hi_u32, d, w);
    run("v_mad_u32_u24", k_mad_u32_u24, d, w);
    run("v_mul_u32_u24", k_mul_u32_u24, d, w);
    run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, d, w);
    run("v_mov_b32_dpp", k_mov_dpp, d, w);
    run("v_fma_f32", k_fma_f32, d, w);
    run("v_fma_f64", k_fma_f64, d, w);
    run("v_mul_f64", k_mul_f64, d, w);
    run("v_pk_add_u16", k_pk_add_u16, d, w);
  }
  return 0;
}
