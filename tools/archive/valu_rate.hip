// Issue rate of the VALU instructions the accumulators are built from, on gfx950, relative to v_add_u32.
// One kernel per instruction: 32 independent copies in a loop body (no dependency inside a body, each
// copy depends on its own result of the previous iteration), run with one and with two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

constexpr int kCopies = 32;
constexpr int kIters = 4096;

#define R8(X, b) X(b + 0) X(b + 1) X(b + 2) X(b + 3) X(b + 4) X(b + 5) X(b + 6) X(b + 7)
#define R32(X) R8(X, 0) R8(X, 8) R8(X, 16) R8(X, 24)

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(unsigned* out, unsigned seed) {
  unsigned a[kCopies], c[kCopies];
  unsigned long long w[kCopies];
  double d[kCopies];
  const unsigned x = seed * 2654435761u + threadIdx.x, y = x ^ 0x9e3779b9u, zero = seed >> 31;
  const unsigned long long sm = __builtin_amdgcn_read_exec();
  const double dx = (double)(x & 1023), dy = 1.0 + (double)(y & 3);
#pragma unroll
  for (int i = 0; i < kCopies; ++i) {
    a[i] = x + i;
    c[i] = y + i;
    w[i] = ((unsigned long long)x << 20) + i;
    d[i] = (double)i;
  }
  for (int it = 0; it < kIters; ++it) {
    if constexpr (OP == 0) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      R32(X)
#undef X
    } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(x), "v"(y) : "vcc");
      R32(X)
#undef X
    } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
      R32(X)
#undef X
    } else if constexpr (OP == 3) {
#define X(i)                                                                                                     \
  asm volatile("v_addc_co_u32_sdwa %0, vcc, %0, %1, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 " \
               "src1_sel:DWORD"                                                                                  \
               : "+v"(a[i])                                                                                      \
               : "v"(zero)                                                                                       \
               : "vcc");
      R32(X)
#undef X
    } else if constexpr (OP == 4) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(dx), "v"(dy));
      R32(X)
#undef X
    } else if constexpr (OP == 5) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      R32(X)
#undef X
    } else if constexpr (OP == 6) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      R32(X)
#undef X
    } else if constexpr (OP == 7) {
#define X(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
      R32(X)
#undef X
    } else if constexpr (OP == 8) {
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      R32(X)
#undef X
    } else if constexpr (OP == 9) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      R32(X)
#undef X
    } else if constexpr (OP == 10) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dx));
      R32(X)
#undef X
    } else if constexpr (OP == 11) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 31]));
      R32(X)
#undef X
    } else if constexpr (OP == 12) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(c[i]));
      R32(X)
#undef X
    } else if constexpr (OP == 13) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(dx), "v"(dy));
      R32(X)
#undef X
    } else if constexpr (OP == 14) {
      // the product-accumulate of field.hpp (4 multiply-adds + 11 carry ops), one per copy pair
#define X(i) asm volatile("v_subb_co_u32_e64 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
      R32(X)
#undef X
    } else if constexpr (OP == 15) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      R32(X)
#undef X
    } else if constexpr (OP == 16) {
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      R32(X)
#undef X
    } else if constexpr (OP == 17) {
      // v_add_u32 with an s_mov_b64 vcc between every two: do scalar moves cost VALU issue slots?
#define X(i) asm volatile("v_add_u32 %0, %0, %1\n\ts_mov_b64 vcc, %2" : "+v"(a[i]) : "v"(x), "s"(sm) : "vcc");
      R32(X)
#undef X
    } else if constexpr (OP == 18) {
      // the 15-instruction product-accumulate of field.hpp (acc = a[4i..4i+3]), 8 per body
#define X(i)                                                                                                    \
  {                                                                                                             \
    unsigned long long t, m, q, sC, sB, sD;                                                                     \
    asm volatile("v_mad_u64_u32 %1, vcc, %4, %7, 0\n\tv_mad_u64_u32 %1, %3, %5, %6, %1\n\t"                     \
                 "v_mad_u64_u32 %0, vcc, %4, %6, 0\n\tv_mad_u64_u32 %2, vcc, %5, %7, 0"                          \
                 : "=&v"(t), "=&v"(m), "=&v"(q), "=&s"(sC)                                                      \
                 : "v"(x), "v"(y), "v"(c[i]), "v"(c[i + 8])                                                     \
                 : "vcc");                                                                                      \
    const unsigned t0 = (unsigned)t, t1 = (unsigned)(t >> 32), m0 = (unsigned)m, m1 = (unsigned)(m >> 32),      \
                   q0 = (unsigned)q, q1 = (unsigned)(q >> 32);                                                  \
    asm volatile("v_subb_co_u32_e64 %0, %5, %0, %11, %12\n\tv_add_co_u32_e32 %0, vcc, %0, %6\n\t"               \
                 "v_add_co_u32_e64 %1, %4, %1, %8\n\tv_subb_co_u32_e64 %1, %5, %1, 0, %5\n\t"                   \
                 "v_addc_co_u32_e32 %1, vcc, %1, %7, vcc\n\tv_addc_co_u32_e64 %2, %4, %2, %9, %4\n\t"           \
                 "v_subb_co_u32_e64 %2, %5, %2, 0, %5\n\tv_addc_co_u32_e32 %2, vcc, %2, %10, vcc\n\t"           \
                 "v_addc_co_u32_e64 %3, %4, %3, 0, %4\n\tv_subb_co_u32_e64 %3, %5, %3, 0, %5\n\t"               \
                 "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc"                                                        \
                 : "+v"(a[4 * i]), "+v"(a[4 * i + 1]), "+v"(a[4 * i + 2]), "+v"(a[4 * i + 3]), "=&s"(sB), "=&s"(sD) \
                 : "v"(t0), "v"(t1), "v"(m0), "v"(m1), "v"(q0), "v"(q1), "s"(sC)                                \
                 : "vcc");                                                                                      \
  }
      R8(X, 0)
#undef X
    } else if constexpr (OP == 19) {
      // three 64-bit classes + byte carry counters packed in one register (SDWA add-with-carry reads VCC only:
      // the other carries travel through s_mov_b64), two products interleaved, 8 products per body
#define X(i)                                                                                                    \
  {                                                                                                             \
    unsigned long long s1, s2, s3;                                                                              \
    asm volatile(                                                                                               \
        "v_mad_u64_u32 %0, vcc, %7, %9, %0\n\t"                                                                 \
        "v_mad_u64_u32 %1, %4, %7, %10, %1\n\t"                                                                 \
        "v_mad_u64_u32 %2, %5, %8, %10, %2\n\t"                                                                 \
        "v_addc_co_u32_sdwa %3, vcc, %3, %11, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD\n\t" \
        "v_mad_u64_u32 %1, %6, %8, %9, %1\n\t"                                                                  \
        "s_mov_b64 vcc, %4\n\t"                                                                                 \
        "v_addc_co_u32_sdwa %3, vcc, %3, %11, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:DWORD\n\t" \
        "s_mov_b64 vcc, %5\n\t"                                                                                 \
        "v_addc_co_u32_sdwa %3, vcc, %3, %11, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
        "s_mov_b64 vcc, %6\n\t"                                                                                 \
        "v_addc_co_u32_sdwa %3, vcc, %3, %11, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:DWORD" \
        : "+v"(w[3 * i]), "+v"(w[3 * i + 1]), "+v"(w[3 * i + 2]), "+v"(a[i]), "=&s"(s1), "=&s"(s2), "=&s"(s3)      \
        : "v"(x), "v"(y), "v"(c[i]), "v"(c[i + 8]), "v"(zero)                                                   \
        : "vcc");                                                                                               \
  }
      R8(X, 0)
#undef X
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < kCopies; ++i) s += a[i] + c[i] + (unsigned)w[i] + (unsigned)(w[i] >> 32) + (unsigned)d[i];
  if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int OP>
static double run(unsigned* out, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  constexpr int kReps = 20;
  for (int i = 0; i < kReps; ++i) rate_kernel<OP><<<blocks, 256>>>(out, 1);  // clocks up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < kReps; ++i) rate_kernel<OP><<<blocks, 256>>>(out, 1);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e6 / ((double)kIters * kCopies * kReps);  // ns per wave instruction and SIMD slot
}

int main() {
  unsigned* out;
  CK(hipMalloc(&out, 4096));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  const char* names[] = {"v_add_u32", "v_mad_u64_u32", "v_addc_co_u32", "v_addc_co_u32_sdwa", "v_fma_f64", "v_mul_lo_u32",
                         "v_mul_hi_u32", "v_cvt_f64_u32", "v_dot4_u32_u8", "v_mad_u32_u24", "v_add_f64", "v_lshl_add_u64",
                         "v_perm_b32", "v_pk_fma_f32", "v_subb_co_u32_e64", "v_mul_u32_u24", "v_mul_hi_u32_u24",
                         "v_add_u32 + s_mov vcc", "mac15 (x8: /4)", "mac8 sdwa (x8: /4)"};
  double base1 = 0;
  for (int w = 1; w <= 4; w *= 2) {  // waves per SIMD
    const int blocks = cus * w;
    double t[20];
    t[0] = run<0>(out, blocks);  t[1] = run<1>(out, blocks);  t[2] = run<2>(out, blocks);  t[3] = run<3>(out, blocks);
    t[4] = run<4>(out, blocks);  t[5] = run<5>(out, blocks);  t[6] = run<6>(out, blocks);  t[7] = run<7>(out, blocks);
    t[8] = run<8>(out, blocks);  t[9] = run<9>(out, blocks);  t[10] = run<10>(out, blocks); t[11] = run<11>(out, blocks);
    t[12] = run<12>(out, blocks); t[13] = run<13>(out, blocks); t[14] = run<14>(out, blocks); t[15] = run<15>(out, blocks);
    t[16] = run<16>(out, blocks); t[17] = run<17>(out, blocks); t[18] = run<18>(out, blocks); t[19] = run<19>(out, blocks);
    if (w == 1) base1 = t[0];
    printf("--- %d wave(s) per SIMD: ns per wave instruction (x = relative to v_add_u32 at 1 wave/SIMD)\n", w);
    for (int i = 0; i < 20; ++i) printf("%-22s %8.3f ns  x%.2f\n", names[i], t[i], t[i] / base1);
  }
  return 0;
}
