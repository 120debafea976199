// Prototype, second form (not part of the product): K1 = 4 or 5 rounds from one read, clean steady-state loop.
// See gram_bench.hip for the idea.  Stage = 8 KiB per table = 8192 / RB rows (RB = 8 * 2^K1 bytes per row); 8 waves = 2 (M) x 4 (N).
// build: hipcc --offload-arch=gfx950 -O3 -o gram4_bench gram4_bench.hip     usage: gram4_bench [n = 28] [K1 = 4] [blocks = 256] [reps = 100] [zero = 0]
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
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glob_ptr_t;

#ifndef GRAM_AUX
#define GRAM_AUX 0   // cache-policy bits of the LDS-DMA load (2: nontemporal)
#endif
#ifndef GRAM_NS
#define GRAM_NS 4
#endif
constexpr int NS = GRAM_NS;              // stages (even)
constexpr int kTabBytes = 8192;          // bytes per table and stage
constexpr int kStageBytes = 2 * kTabBytes;
constexpr int kThreads = 512;

template <int K1>
struct Geo {
  static constexpr int RB = 8 << K1;            // bytes per row
  static constexpr int ROWS = kTabBytes / RB;   // rows per stage: 32 (K1 = 5) or 64
  static constexpr int KSUB = ROWS / 32;        // MFMA k-steps per stage
  static constexpr int MB = RB / 64, NBK = RB / 128;   // 32-column blocks per wave along M and N
  static constexpr int NA = MB * KSUB * 2, NBR = NBK * KSUB * 2;   // transposed reads per stage: 8 and 4
  static constexpr int CPR = RB / 16;           // 16-byte chunks per row
  static constexpr int kWords = RB * RB + 2 * RB;   // partial words per block
};
// chunk position inside a row of the LDS image: eight consecutive rows x two adjacent chunks cover all 64 banks
template <int K1>
__device__ __forceinline__ int swz(int row, int chunk) {
  if constexpr (K1 == 5) return (((chunk >> 1) ^ (row & 7)) << 1) | (chunk & 1);
  else return (((chunk >> 1) ^ ((row >> 1) & 3)) << 1) | (chunk & 1);
}

#define TR8(R, AD, OFF) "ds_read_b64_tr_b8 %" #R ", %" #AD " offset:%" #OFF "\n\t"

template <int K1>
__global__ void __launch_bounds__(kThreads) gram_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, size_t n_steps,
                                                         int* __restrict__ partials, unsigned long long* __restrict__ stamps) {
  typedef Geo<K1> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mh = wave >> 2, nq = wave & 3;
  const int h = lane >> 5, g2 = (lane >> 4) & 1, ll = lane & 15, q = ll >> 1, p = ll & 1;
  // DMA: 16 instructions of 1 KiB per stage, two per wave: i < 8 -> table A
  size_t src_off[2];
  unsigned dst_off[2];
  const unsigned char* tab_of[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = 2 * wave + u, tab = i >> 3, piece = i & 7;          // piece: 1 KiB of the table's 8 KiB
    const int row = piece * (1024 / G::RB) + lane / G::CPR, pos = lane % G::CPR;
#ifdef GRAM_LINEAR
    src_off[u] = (size_t)row * G::RB + 16 * pos;   // timing experiment: unswizzled source (results are wrong)
#else
    src_off[u] = (size_t)row * G::RB + 16 * swz<K1>(row, pos);
#endif
    dst_off[u] = (unsigned)(tab * kTabBytes + piece * 1024);         // wave-uniform; the hardware adds lane * 16
    tab_of[u] = tab ? B : A;
  }
  auto issue = [&](size_t step, unsigned stage_off) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((glob_ptr_t)(tab_of[u] + step * (size_t)kTabBytes + src_off[u]), (lds_ptr_t)(lds + stage_off + dst_off[u]), 16, 0, GRAM_AUX);
  };
  // transposed reads: 32 columns from m0, rows 32 ks + 16 h + 8 t + q.  One address register per read; the stage is an immediate.
  auto tr_addr = [&](int tab, int m0, int ks, int t) -> unsigned {
    const int row = 32 * ks + 16 * h + 8 * t + q, chunk = (m0 >> 4) + g2;
    return (unsigned)(size_t)(lds_ptr_t)lds + (unsigned)(tab * kTabBytes + row * G::RB + 16 * swz<K1>(row, chunk) + 8 * p);
  };
  unsigned aa[G::NA], ba[G::NBR];   // index: ((block * KSUB) + ks) * 2 + t
#pragma unroll
  for (int a = 0; a < G::MB; ++a)
#pragma unroll
    for (int ks = 0; ks < G::KSUB; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) aa[(a * G::KSUB + ks) * 2 + t] = tr_addr(0, (G::RB / 2) * mh + 32 * a, ks, t);
#pragma unroll
  for (int b = 0; b < G::NBK; ++b)
#pragma unroll
    for (int ks = 0; ks < G::KSUB; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) ba[(b * G::KSUB + ks) * 2 + t] = tr_addr(1, (G::RB / 4) * nq + 32 * b, ks, t);
  static_assert(G::NA == 8 && G::NBR == 4, "eight + four reads per stage");
  unsigned aa2[G::NA], ba2[G::NBR];   // the stages beyond 64 KiB (the offset field of a DS instruction has 16 bits)
#pragma unroll
  for (int i = 0; i < G::NA; ++i) aa2[i] = aa[i] + 65536;
#pragma unroll
  for (int i = 0; i < G::NBR; ++i) ba2[i] = ba[i] + 65536;

  v16i acc[G::MB][G::NBK];
#pragma unroll
  for (int a = 0; a < G::MB; ++a)
#pragma unroll
    for (int b = 0; b < G::NBK; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
  unsigned su_a[G::MB] = {}, su_b[G::NBK] = {};

  const size_t first = blockIdx.x, stride = gridDim.x;
  const size_t my_steps = (n_steps > first) ? (n_steps - first + stride - 1) / stride : 0;   // the host sends multiples of NS
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();

  v2i r0[12], r1[12];
#define READ12X(R, AA, BA, OFF)                                                                                                           \
  asm volatile(TR8(0, 12, 24) TR8(1, 13, 24) TR8(2, 14, 24) TR8(3, 15, 24) TR8(4, 16, 24) TR8(5, 17, 24) TR8(6, 18, 24) TR8(7, 19, 24)   \
               TR8(8, 20, 24) TR8(9, 21, 24) TR8(10, 22, 24) "ds_read_b64_tr_b8 %11, %23 offset:%24"                               \
               : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]), "=&v"(R[5]), "=&v"(R[6]), "=&v"(R[7]), "=&v"(R[8]),   \
                 "=&v"(R[9]), "=&v"(R[10]), "=&v"(R[11])                                                                           \
               : "v"(AA[0]), "v"(AA[1]), "v"(AA[2]), "v"(AA[3]), "v"(AA[4]), "v"(AA[5]), "v"(AA[6]), "v"(AA[7]), "v"(BA[0]), "v"(BA[1]),   \
                 "v"(BA[2]), "v"(BA[3]), "n"(OFF)                                                                                   \
               : "memory")
#define READ12(R, OFF)                                               \
  do {                                                               \
    if ((OFF) < 65536) READ12X(R, aa, ba, (OFF) % 65536);            \
    else READ12X(R, aa2, ba2, (OFF) % 65536);                        \
  } while (0)
#define WAIT12(R)                                                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                               \
               : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]), "+v"(R[8]), "+v"(R[9]),   \
                 "+v"(R[10]), "+v"(R[11])                                                                                           \
               :                                                                                                                    \
               : "memory")
  auto compute = [&](v2i (&r)[12]) {
    v4i fa[G::MB][G::KSUB], fb[G::NBK][G::KSUB];
#pragma unroll
    for (int a = 0; a < G::MB; ++a)
#pragma unroll
      for (int ks = 0; ks < G::KSUB; ++ks) {
        const int i = (a * G::KSUB + ks) * 2;
        fa[a][ks] = v4i{r[i].x, r[i].y, r[i + 1].x, r[i + 1].y};
      }
#pragma unroll
    for (int b = 0; b < G::NBK; ++b)
#pragma unroll
      for (int ks = 0; ks < G::KSUB; ++ks) {
        const int i = 8 + (b * G::KSUB + ks) * 2;
        fb[b][ks] = v4i{r[i].x, r[i].y, r[i + 1].x, r[i + 1].y};
      }
    if (nq == 0) {
#pragma unroll
      for (int a = 0; a < G::MB; ++a)
#pragma unroll
        for (int ks = 0; ks < G::KSUB; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) su_a[a] = __builtin_amdgcn_sad_u8((unsigned)fa[a][ks][e], 0u, su_a[a]);
    }
    if (mh == 0) {
#pragma unroll
      for (int b = 0; b < G::NBK; ++b)
#pragma unroll
        for (int ks = 0; ks < G::KSUB; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) su_b[b] = __builtin_amdgcn_sad_u8((unsigned)fb[b][ks][e], 0u, su_b[b]);
    }
#pragma unroll
    for (int ks = 0; ks < G::KSUB; ++ks) {
#pragma unroll
      for (int b = 0; b < G::NBK; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) fb[b][ks][e] ^= 0x80808080;
#pragma unroll
      for (int a = 0; a < G::MB; ++a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) fa[a][ks][e] ^= 0x80808080;
#pragma unroll
        for (int b = 0; b < G::NBK; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a][ks], fb[b][ks], acc[a][b], 0, 0, 0);
      }
    }
  };
  // top of step s: step s + 1 becomes readable (this wave's part has landed; barrier: everyone's), stage of step s is refilled
#define TOP(S, STAGE)                                                                        \
  do {                                                                                       \
    if ((S) + NS - 1 < my_steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 2)) : "memory"); \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
    __builtin_amdgcn_s_barrier();                                                            \
    if ((S) + NS < my_steps) issue(first + ((S) + NS) * stride, (STAGE) * kStageBytes);      \
  } while (0)
  if (my_steps > 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) issue(first + (size_t)s * stride, s * kStageBytes);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    READ12(r0, 0);
    WAIT12(r0);
    for (size_t s = 0; s < my_steps; s += NS) {
      const bool more = s + NS < my_steps;
#define PAIR(I)                                                                                       \
      TOP(s + (I), (I));     READ12(r1, ((I) + 1) * kStageBytes); compute(r0); WAIT12(r1);            \
      TOP(s + (I) + 1, (I) + 1);                                                                      \
      if ((I) + 2 < NS) { READ12(r0, (((I) + 2) % NS) * kStageBytes); compute(r1); WAIT12(r0); }      \
      else { if (more) READ12(r0, 0); compute(r1); if (more) WAIT12(r0); }
      PAIR(0)
#if GRAM_NS >= 4
      PAIR(2)
#endif
#if GRAM_NS >= 6
      PAIR(4)
#endif
#if GRAM_NS >= 8
      PAIR(6)
#endif
#undef PAIR
    }
  }
  if (stamps && tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - w0;
  }
  // partials in accumulator order: word ((wave * MB * NBK + a * NBK + b) * 16 + reg) * 64 + lane, then the byte sums
  int* const out = partials + (size_t)blockIdx.x * G::kWords;
#pragma unroll
  for (int a = 0; a < G::MB; ++a)
#pragma unroll
    for (int b = 0; b < G::NBK; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) out[((wave * (G::MB * G::NBK) + a * G::NBK + b) * 16 + e) * 64 + lane] = acc[a][b][e];
  if (nq == 0) {
#pragma unroll
    for (int a = 0; a < G::MB; ++a) {
      const unsigned t = su_a[a] + (unsigned)__shfl_xor((int)su_a[a], 32, 64);
      if (h == 0) out[G::RB * G::RB + (G::RB / 2) * mh + 32 * a + (lane & 31)] = (int)t;
    }
  }
  if (mh == 0) {
#pragma unroll
    for (int b = 0; b < G::NBK; ++b) {
      const unsigned t = su_b[b] + (unsigned)__shfl_xor((int)su_b[b], 32, 64);
      if (h == 0) out[G::RB * G::RB + G::RB + (G::RB / 4) * nq + 32 * b + (lane & 31)] = (int)t;
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
template <int K1>
int run(int n, int blocks, int reps, int zero) {
  typedef Geo<K1> G;
  const size_t len = (size_t)1 << n, n_steps = len * 8 / kTabBytes;
  if (n_steps % ((size_t)blocks * NS) != 0) { printf("steps %zu not a multiple of blocks * %d\n", n_steps, NS); return 1; }
  u64 *dA, *dB;
  int* dP;
  unsigned long long* dS;
  CK(hipMalloc(&dA, len * 8)); CK(hipMalloc(&dB, len * 8));
  CK(hipMalloc(&dP, (size_t)blocks * G::kWords * 4));
  CK(hipMalloc(&dS, (size_t)blocks * 16));
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, dA, len, 0xA5A5000000000001ull);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, dB, len, 0xB6B6000000000002ull);
  if (zero) { CK(hipMemset(dA, 0x80, len * 8)); CK(hipMemset(dB, 0x80, len * 8)); }
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)gram_kernel<K1>, hipFuncAttributeMaxDynamicSharedMemorySize, NS * kStageBytes));
  auto launch = [&]() {
    hipLaunchKernelGGL(gram_kernel<K1>, dim3(blocks), dim3(kThreads), NS * kStageBytes, 0, (const unsigned char*)dA, (const unsigned char*)dB, n_steps, dP, dS);
  };
  for (int w = 0; w < 20; ++w) launch();
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
  std::vector<unsigned long long> hs((size_t)blocks * 2);
  CK(hipMemcpy(hs.data(), dS, hs.size() * 8, hipMemcpyDeviceToHost));
  double ghz = 0;
  for (int b = 0; b < blocks; ++b) ghz += (double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1;
  printf("n=%d K1=%d blocks=%d%s: best %.1f us, mean %.1f us -> %.2f TB/s (mean), %.3f of 8 TB/s; in-kernel clock %.2f GHz, block 0 %llu cycles\n", n, K1, blocks,
         zero ? " ZERO operands" : "", best * 1e3, sum / reps * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (sum / reps * 1e-3) / 8e12, ghz / blocks, hs[0]);
  if (n <= 22 && !zero) {
    constexpr int X = 1 << K1, RB = G::RB;
    std::vector<u64> hA(len), hB(len);
    CK(hipMemcpy(hA.data(), dA, len * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), dB, len * 8, hipMemcpyDeviceToHost));
    std::vector<int> hP((size_t)blocks * G::kWords);
    CK(hipMemcpy(hP.data(), dP, hP.size() * 4, hipMemcpyDeviceToHost));
    std::vector<long long> Gm((size_t)RB * RB, 0), SuA(RB, 0), SuB(RB, 0);
    for (int b = 0; b < blocks; ++b) {
      const int* P = hP.data() + (size_t)b * G::kWords;
      for (int wave = 0; wave < 8; ++wave)
        for (int blk = 0; blk < G::MB * G::NBK; ++blk)
          for (int reg = 0; reg < 16; ++reg)
            for (int lane = 0; lane < 64; ++lane) {
              const int a = blk / G::NBK, bb = blk % G::NBK, mh = wave >> 2, nq = wave & 3, hh = lane >> 5;
              const int m = (RB / 2) * mh + 32 * a + (reg & 3) + 8 * (reg >> 2) + 4 * hh, nn = (RB / 4) * nq + 32 * bb + (lane & 31);
              Gm[(size_t)m * RB + nn] += P[((wave * (G::MB * G::NBK) + blk) * 16 + reg) * 64 + lane];
            }
      for (int m = 0; m < RB; ++m) { SuA[m] += (unsigned)P[RB * RB + m]; SuB[m] += (unsigned)P[RB * RB + RB + m]; }
    }
    const long long R = (long long)(len / X);
    size_t bad = 0;
    for (int x = 0; x < X; ++x)
      for (int y = 0; y < X; ++y) {
        u64 w[3] = {0, 0, 0};
        for (size_t r = 0; r < (size_t)R; ++r) {
          const u128 pr = (u128)hA[X * r + x] * hB[X * r + y];
          const u128 s0 = (u128)w[0] + (u64)pr;
          w[0] = (u64)s0;
          const u128 s1 = (u128)w[1] + (u64)(pr >> 64) + (u64)(s0 >> 64);
          w[1] = (u64)s1;
          w[2] += (u64)(s1 >> 64);
        }
        u64 v[3] = {0, 0, 0};
        for (int i = 0; i < 8; ++i)
          for (int j = 0; j < 8; ++j) {
            const long long T = Gm[(size_t)(8 * x + i) * RB + 8 * y + j] + 128 * SuA[8 * x + i] + 128 * SuB[8 * y + j] - 16384 * R;
            if (T < 0) { ++bad; continue; }
            const int sh = 8 * (i + j), wd = sh / 64, bt = sh % 64;
            u64 add[3] = {0, 0, 0};
            add[wd] = (u64)T << bt;
            if (bt && wd + 1 < 3) add[wd + 1] = (u64)T >> (64 - bt);
            u128 c = 0;
            for (int k = 0; k < 3; ++k) { c += (u128)v[k] + add[k]; v[k] = (u64)c; c >>= 64; }
          }
        if (v[0] != w[0] || v[1] != w[1] || v[2] != w[2]) ++bad;
      }
    printf("check: %zu of %d Gram entries differ from the exact integer sums\n", bad, X * X);
  }
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dP)); CK(hipFree(dS));
  return 0;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 28;
  const int k1 = argc > 2 ? atoi(argv[2]) : 4;
  const int blocks = argc > 3 ? atoi(argv[3]) : 256;
  const int reps = argc > 4 ? atoi(argv[4]) : 100;
  const int zero = argc > 5 ? atoi(argv[5]) : 0;
  return k1 == 5 ? run<5>(n, blocks, reps, zero) : run<4>(n, blocks, reps, zero);
}
