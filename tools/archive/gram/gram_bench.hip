// Prototype of a FIVE-round first pass on the int8 matrix cores (not part of the product; see DESIGN_HISTORY.md section 11).
//
// The round polynomials of rounds 1..5 of the product sumcheck are functions of the 32 x 32 Gram matrix
//   M[x][x'] = sum_rows a[32 row + x] * b[32 row + x'],   x, x' in {0,1}^5 (the five lowest index bits),
// i.e. of A^T B for the tables seen as matrices of 2^(n-5) rows by 32 entries.  With the entries' eight bytes as separate
// columns that is a 256 x 256 x 2^(n-5) int8 GEMM whose operands are the tables' bytes exactly as they lie in HBM; the
// exact integer sums are reduced mod p afterwards, so the kernel is the same for every modulus.
//
// build: hipcc --offload-arch=gfx950 -O3 -o gram_bench gram_bench.hip      usage: gram_bench [n = 28] [blocks = 256] [stages = 5] [reps = 10]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) v2i* lds_v2i;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glob_ptr_t;

constexpr int kRowBytes = 256;           // 32 entries of 8 bytes
constexpr int kStepRows = 32;            // rows per k-step (one MFMA 32x32x32 deep)
constexpr int kStageBytes = 2 * kStepRows * kRowBytes;   // A rows then B rows: 16 KiB
constexpr int kThreads = 512;            // 8 waves: (M half) x (N quarter)
constexpr int kPartialWords = 65536 + 512;   // the 256 x 256 limb products in accumulator order, then the 256 + 256 byte sums

__device__ __forceinline__ int swz(int row, int chunk) { return (((chunk >> 1) ^ (row & 7)) << 1) | (chunk & 1); }

template <int NS, int HALF, int NB>
__global__ void __launch_bounds__(NB == 2 ? 512 : 256) gram5_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, size_t n_steps,
                                                          int* __restrict__ partials, int mode, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int kNW = 8 / NB;            // waves along N
  constexpr int kWavesK = 2 * kNW, kDma = 16 / kWavesK;   // waves of the block; DMA instructions per wave and stage
  const int mh = wave / kNW, nq = wave % kNW;
  const int h = lane >> 5, g2 = (lane >> 4) & 1, ll = lane & 15, q = ll >> 1, p = ll & 1;
  // DMA role: instruction i of a stage moves rows 4 i' .. 4 i' + 3 of A (i < 8) or B; this wave issues i = 2 wave, 2 wave + 1
  // lane L lands at row r0 + L / 16, position L % 16 and fetches the chunk that belongs there
  size_t src_off[kDma];
  int dst_off[kDma];
  const unsigned char* tab_of[kDma];
#pragma unroll
  for (int u = 0; u < kDma; ++u) {
    const int i = kDma * wave + u, tab = i >> 3, r0 = 4 * (i & 7);
    const int row = r0 + (lane >> 4), pos = lane & 15;
    src_off[u] = (size_t)row * kRowBytes + 16 * ((mode & 1) ? pos : swz(row, pos));
    dst_off[u] = tab * kStepRows * kRowBytes + r0 * kRowBytes;   // wave-uniform base; the hardware adds lane * 16
    tab_of[u] = tab ? B : A;
  }
  auto issue = [&](size_t step, int stage) {
#pragma unroll
    for (int u = 0; u < kDma; ++u)
      __builtin_amdgcn_global_load_lds((glob_ptr_t)(tab_of[u] + step * (size_t)(kStepRows * kRowBytes) + src_off[u]),
                                       (lds_ptr_t)(lds + stage * kStageBytes + dst_off[u]), 16, 0, 0);
  };
  // transposed reads: block of 32 columns starting at m0, rows 16 h + 8 t + q
  auto tr_addr = [&](int tab, int m0, int t) -> int {
    const int row = 16 * h + 8 * t + q, chunk = (m0 >> 4) + g2;
    return tab * kStepRows * kRowBytes + row * kRowBytes + 16 * swz(row, chunk) + 8 * p;
  };
  int a_addr[4][2], b_addr[NB][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int t = 0; t < 2; ++t) a_addr[a][t] = tr_addr(0, 128 * mh + 32 * a, t);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int t = 0; t < 2; ++t) b_addr[b][t] = tr_addr(1, 32 * NB * nq + 32 * b, t);

  v16i acc[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
  unsigned su_a[4] = {0, 0, 0, 0}, su_b[NB] = {};

  // steps of this block: blockIdx, blockIdx + grid, ...
  const size_t first = blockIdx.x, stride = gridDim.x;
  const size_t my_steps = (n_steps > first) ? (n_steps - first + stride - 1) / stride : 0;
  // Software pipeline: while the matrix cores work on step s (operands in registers), the transposed reads of step s + 1 are
  // in flight and the DMA of steps s + 2 .. s + NS is on its way.  One barrier per step: behind it every wave's part of
  // step s + 1 has landed and every wave has the operands of step s in registers, so stage s % NS can be refilled.
  unsigned st = (unsigned)(size_t)(lds_ptr_t)lds;   // stage of the NEXT read
  const unsigned st_end = st + NS * kStageBytes;
  int fill = 0;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if ((size_t)s < my_steps) issue(first + (size_t)s * stride, s);
  constexpr int NR = 8 + 2 * NB;   // transposed reads per step: 8 for the four A blocks, 2 per B block
  v2i r0[NR], r1[NR];
  auto read_async = [&](v2i (&r)[NR]) {
    // (inline assembly: behind an LDS-DMA load the compiler would make every LDS read wait for vmcnt(0))
    asm volatile("ds_read_b64_tr_b8 %0, %8\n\tds_read_b64_tr_b8 %1, %9\n\tds_read_b64_tr_b8 %2, %10\n\tds_read_b64_tr_b8 %3, %11\n\t"
                 "ds_read_b64_tr_b8 %4, %12\n\tds_read_b64_tr_b8 %5, %13\n\tds_read_b64_tr_b8 %6, %14\n\tds_read_b64_tr_b8 %7, %15"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(st + a_addr[0][0]), "v"(st + a_addr[0][1]), "v"(st + a_addr[1][0]), "v"(st + a_addr[1][1]), "v"(st + a_addr[2][0]),
                   "v"(st + a_addr[2][1]), "v"(st + a_addr[3][0]), "v"(st + a_addr[3][1])
                 : "memory");
    asm volatile("ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %1, %5\n\tds_read_b64_tr_b8 %2, %6\n\tds_read_b64_tr_b8 %3, %7"
                 : "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11])
                 : "v"(st + b_addr[0][0]), "v"(st + b_addr[0][1]), "v"(st + b_addr[1][0]), "v"(st + b_addr[1][1])
                 : "memory");
    if constexpr (NB == 4)
      asm volatile("ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %1, %5\n\tds_read_b64_tr_b8 %2, %6\n\tds_read_b64_tr_b8 %3, %7"
                   : "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
                   : "v"(st + b_addr[NB - 2][0]), "v"(st + b_addr[NB - 2][1]), "v"(st + b_addr[NB - 1][0]), "v"(st + b_addr[NB - 1][1])
                   : "memory");
    st += kStageBytes;
    if (st == st_end) st -= NS * kStageBytes;
  };
  auto read_wait = [&](v2i (&r)[NR]) {   // the registers are valid behind this wait (the operands tie the uses to it)
    if constexpr (NB == 4)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]),
                     "+v"(r[11]), "+v"(r[NR - 4]), "+v"(r[NR - 3]), "+v"(r[NR - 2]), "+v"(r[NR - 1])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]),
                     "+v"(r[11])
                   :
                   : "memory");
  };
  auto landed_and_refill = [&](size_t s) {   // top of step s: make step s + 1 readable, refill the stage of step s
    if (s + NS - 1 < my_steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + NS < my_steps) issue(first + (s + NS) * stride, fill);
    fill = (fill + 1 == NS) ? 0 : fill + 1;
  };
  auto compute = [&](v2i (&r)[NR]) {
    v4i fa[4], fb[NB];
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = v4i{r[2 * a].x, r[2 * a].y, r[2 * a + 1].x, r[2 * a + 1].y};
#pragma unroll
    for (int b = 0; b < NB; ++b) fb[b] = v4i{r[8 + 2 * b].x, r[8 + 2 * b].y, r[8 + 2 * b + 1].x, r[8 + 2 * b + 1].y};
    // unsigned byte sums (for the signed-byte correction), then bytes -> signed (u - 128)
    if (nq == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) su_a[a] = __builtin_amdgcn_sad_u8((unsigned)fa[a][e], 0u, su_a[a]);
    }
    if (mh == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) su_b[b] = __builtin_amdgcn_sad_u8((unsigned)fb[b][e], 0u, su_b[b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < 4; ++e) fa[a][e] ^= 0x80808080;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) fb[b][e] ^= 0x80808080;
    if (mode & 4) {
      acc[0][0][0] += fa[0][0] ^ fa[1][1] ^ fa[2][2] ^ fa[3][3] ^ fb[0][0] ^ fb[1][1];
      return;
    }
    if constexpr (HALF != 0) {   // half of the matrix work (what a four-round pass would need per byte): timing only
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[0], acc[a][0], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  if (my_steps > 0) {
    // step 0's operands
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma * (NS - 1)) : "memory");
    if (my_steps < (size_t)NS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_async(r0);
    read_wait(r0);
    size_t s = 0;
    for (; s + 2 <= my_steps; s += 2) {
      landed_and_refill(s);
      if (!(mode & 2)) { read_async(r1); compute(r0); read_wait(r1); }
      landed_and_refill(s + 1);
      if (!(mode & 2)) {
        if (s + 2 < my_steps) read_async(r0);
        compute(r1);
        if (s + 2 < my_steps) read_wait(r0);
      }
    }
    if (s < my_steps) {   // an odd last step: its operands are in r0
      landed_and_refill(s);
      if (!(mode & 2)) compute(r0);
    }
  }
  if (stamps && tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - w0;
  }
  // partials in accumulator order: word ((wave * 8 + a * 2 + b) * 16 + reg) * 64 + lane
  int* const out = partials + (size_t)blockIdx.x * kPartialWords;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) out[((wave * (4 * NB) + a * NB + b) * 16 + e) * 64 + lane] = acc[a][b][e];
  // byte sums: the two k-halves of a column live in lanes l and l + 32
  if (nq == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const unsigned t = su_a[a] + (unsigned)__shfl_xor((int)su_a[a], 32, 64);
      if (h == 0) out[65536 + 128 * mh + 32 * a + (lane & 31)] = (int)t;
    }
  }
  if (mh == 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const unsigned t = su_b[b] + (unsigned)__shfl_xor((int)su_b[b], 32, 64);
      if (h == 0) out[65536 + 256 + 32 * NB * nq + 32 * b + (lane & 31)] = (int)t;
    }
  }
}

__global__ void fill_kernel(u64* t, size_t n, u64 seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u64 z = seed + i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    t[i] = z ^ (z >> 31);
  }
}

typedef unsigned __int128 u128;
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 28;
  const int blocks = argc > 2 ? atoi(argv[2]) : 256;
  const int stages = argc > 3 ? atoi(argv[3]) : 5;
  const int reps = argc > 4 ? atoi(argv[4]) : 10;
  const int mode = argc > 5 ? atoi(argv[5]) : 0;
  const size_t len = (size_t)1 << n, n_steps = len / 32 / kStepRows;
  u64 *dA, *dB;
  int* dP;
  CK(hipMalloc(&dA, len * 8)); CK(hipMalloc(&dB, len * 8));
  CK(hipMalloc(&dP, (size_t)blocks * kPartialWords * 4));
  unsigned long long* dS;
  CK(hipMalloc(&dS, (size_t)blocks * 16));
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, dA, len, 0xA5A5000000000001ull);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, dB, len, 0xB6B6000000000002ull);
  if (mode & 8) { CK(hipMemset(dA, 0x80, len * 8)); CK(hipMemset(dB, 0x80, len * 8)); }
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nb = stages >= 20 ? 4 : 2, threads = nb == 4 ? 256 : 512;
  auto launch = [&]() {
    const size_t lds_bytes = (size_t)(stages % 10) * kStageBytes;
    switch (stages) {
      case 3: CK(hipFuncSetAttribute((const void*)gram5_kernel<3, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
              hipLaunchKernelGGL((gram5_kernel<3, 0, 2>), dim3(blocks), dim3(threads), lds_bytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, mode & 23, dS); break;
      case 4: CK(hipFuncSetAttribute((const void*)gram5_kernel<4, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
              hipLaunchKernelGGL((gram5_kernel<4, 0, 2>), dim3(blocks), dim3(threads), lds_bytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, mode & 23, dS); break;
      case 6: CK(hipFuncSetAttribute((const void*)gram5_kernel<6, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
              hipLaunchKernelGGL((gram5_kernel<6, 0, 2>), dim3(blocks), dim3(threads), lds_bytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, mode & 23, dS); break;
      case 8: CK(hipFuncSetAttribute((const void*)gram5_kernel<8, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
              hipLaunchKernelGGL((gram5_kernel<8, 0, 2>), dim3(blocks), dim3(threads), lds_bytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, mode & 23, dS); break;
      case 24: CK(hipFuncSetAttribute((const void*)gram5_kernel<4, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * kStageBytes)));
              hipLaunchKernelGGL((gram5_kernel<4, 0, 4>), dim3(blocks), dim3(threads), 4 * kStageBytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, mode & 7, dS); break;
      case 26: CK(hipFuncSetAttribute((const void*)gram5_kernel<6, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(6 * kStageBytes)));
              hipLaunchKernelGGL((gram5_kernel<6, 0, 4>), dim3(blocks), dim3(threads), 6 * kStageBytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, mode & 7, dS); break;
      case 14: CK(hipFuncSetAttribute((const void*)gram5_kernel<4, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * kStageBytes)));
              hipLaunchKernelGGL((gram5_kernel<4, 1, 2>), dim3(blocks), dim3(threads), 4 * kStageBytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, mode & 7, dS); break;
      default: CK(hipFuncSetAttribute((const void*)gram5_kernel<5, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
              hipLaunchKernelGGL((gram5_kernel<5, 0, 2>), dim3(blocks), dim3(threads), lds_bytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, mode & 23, dS); break;
    }
  };
  for (int w = 0; w < 3; ++w) launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double bytes = 2.0 * len * 8;
  printf("n=%d blocks=%d stages=%d mode=%d: best %.1f us, mean %.1f us -> %.2f TB/s (best), %.3f of 8 TB/s\n", n, blocks, stages, mode, best * 1e3, sum / reps * 1e3,
         bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 8e12);
  {
    std::vector<unsigned long long> hs((size_t)blocks * 2);
    CK(hipMemcpy(hs.data(), dS, hs.size() * 8, hipMemcpyDeviceToHost));
    double ghz = 0;
    for (int b = 0; b < blocks; ++b) ghz += (double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1;
    printf("  in-kernel clock of the last launch: %.2f GHz (mean over blocks), block 0: %llu cycles\n", ghz / blocks, hs[0]);
  }
  // check (small n): M[x][x'] = sum_rows a b as exact integers, from the limb products
  if (n <= 22 && (mode & 31) == 0) {
    std::vector<u64> hA(len), hB(len);
    CK(hipMemcpy(hA.data(), dA, len * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), dB, len * 8, hipMemcpyDeviceToHost));
    std::vector<int> hP((size_t)blocks * kPartialWords);
    CK(hipMemcpy(hP.data(), dP, hP.size() * 4, hipMemcpyDeviceToHost));
    std::vector<long long> G(65536, 0), SuA(256, 0), SuB(256, 0);
    for (int b = 0; b < blocks; ++b) {
      const int* P = hP.data() + (size_t)b * kPartialWords;
      const int NB = nb, nw = 8 / NB;
      for (int wave = 0; wave < 2 * nw; ++wave)
        for (int blk = 0; blk < 4 * NB; ++blk)
          for (int reg = 0; reg < 16; ++reg)
            for (int lane = 0; lane < 64; ++lane) {
              const int a = blk / NB, bb = blk % NB, mh = wave / nw, nq = wave % nw, hh = lane >> 5;
              const int m = 128 * mh + 32 * a + (reg & 3) + 8 * (reg >> 2) + 4 * hh, nn = 32 * NB * nq + 32 * bb + (lane & 31);
              G[m * 256 + nn] += P[((wave * (4 * NB) + blk) * 16 + reg) * 64 + lane];
            }
      for (int m = 0; m < 256; ++m) { SuA[m] += (unsigned)P[65536 + m]; SuB[m] += (unsigned)P[65536 + 256 + m]; }
    }
    const long long R = (long long)(len / 32);
    size_t bad = 0;
    for (int x = 0; x < 32; ++x)
      for (int y = 0; y < 32; ++y) {
        // exact: sum_rows a(x) b(y) as a 192-bit integer in three 64-bit words
        u64 w[3] = {0, 0, 0};
        for (size_t r = 0; r < (size_t)R; ++r) {
          const u128 pr = (u128)hA[32 * r + x] * hB[32 * r + y];
          const u128 s0 = (u128)w[0] + (u64)pr;
          w[0] = (u64)s0;
          const u128 s1 = (u128)w[1] + (u64)(pr >> 64) + (u64)(s0 >> 64);
          w[1] = (u64)s1;
          w[2] += (u64)(s1 >> 64);
        }
        // from the limbs: sum_{i,j} 2^(8(i+j)) T_ij,  T_ij = G + 128 SuA + 128 SuB - 16384 R
        u64 v[3] = {0, 0, 0};
        for (int i = 0; i < 8; ++i)
          for (int j = 0; j < 8; ++j) {
            const long long T = G[(8 * x + i) * 256 + 8 * y + j] + 128 * SuA[8 * x + i] + 128 * SuB[8 * y + j] - 16384 * R;
            if (T < 0) { ++bad; continue; }
            // add T << 8(i+j) into v
            const int sh = 8 * (i + j);
            u64 add[3] = {0, 0, 0};
            const int wd = sh / 64, bt = sh % 64;
            add[wd] = (u64)T << bt;
            if (bt && wd + 1 < 3) add[wd + 1] = (u64)T >> (64 - bt);
            u128 c = 0;
            for (int k = 0; k < 3; ++k) { c += (u128)v[k] + add[k]; v[k] = (u64)c; c >>= 64; }
          }
        if (v[0] != w[0] || v[1] != w[1] || v[2] != w[2]) ++bad;
      }
    printf("check: %zu of 1024 Gram entries differ from the exact integer sums\n", bad);
  }
  return 0;
}
