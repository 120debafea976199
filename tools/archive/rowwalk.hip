// What the memory system gives a ROW-WALKING reader (column sums of a row-major matrix: gkr_phase1_kernel,
// coldot_kernel): two 2^13 x 2^13 tables of u64 (512 MiB each), every thread accumulates its columns down a chunk
// of rows.  Parameters: PW = 1 KiB spans a wave reads contiguously per row and table, RIF = rows in flight per
// thread, waves = grid size; how many row chunks follows.   hipcc -O3 --offload-arch=gfx950 -o rowwalk rowwalk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256;

template <int PW, int RIF, int TABLES>
__global__ void __launch_bounds__(kBlock)
walk(const ull2* __restrict__ A, const ull2* __restrict__ B, size_t rows, size_t mp, size_t rows_per_chunk, ull2* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t spans = mp / (64 * PW);                      // wave spans per row
  const size_t gw = (size_t)blockIdx.x * 4 + wave;          // global wave
  const size_t span = gw % spans, chunk = gw / spans;
  const size_t i0 = chunk * rows_per_chunk, i1 = std::min(rows, i0 + rows_per_chunk);
  if (i0 >= rows) return;
  u64 acc[PW][2];
#pragma unroll
  for (int j = 0; j < PW; ++j) acc[j][0] = acc[j][1] = 0;
  const size_t base = span * 64 * PW + lane;
  for (size_t i = i0; i + RIF <= i1; i += RIF) {
    ull2 a[RIF][PW], b[RIF][PW];
#pragma unroll
    for (int r = 0; r < RIF; ++r)
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        a[r][j] = __builtin_nontemporal_load(A + (i + r) * mp + base + j * 64);
        if (TABLES == 2) b[r][j] = __builtin_nontemporal_load(B + (i + r) * mp + base + j * 64);
      }
#pragma unroll
    for (int r = 0; r < RIF; ++r)
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        acc[j][0] += a[r][j].x * 3;
        acc[j][1] += a[r][j].y * 5;
        if (TABLES == 2) { acc[j][0] ^= b[r][j].x; acc[j][1] += b[r][j].y; }
      }
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) out[chunk * mp + base + j * 64] = ull2{acc[j][0], acc[j][1]};
}

template <int PW, int RIF, int TABLES>
void run(const ull2* A, const ull2* B, ull2* out, size_t rows, size_t M, int waves_target) {
  const size_t mp = M / 2, spans = mp / (64 * PW);
  size_t chunks = std::max<size_t>(1, waves_target / spans);
  size_t rpc = (rows + chunks - 1) / chunks;
  rpc = (rpc + RIF - 1) / RIF * RIF;
  chunks = (rows + rpc - 1) / rpc;
  const int grid = (int)((spans * chunks + 3) / 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int it = 0; it < 7; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((walk<PW, RIF, TABLES>), dim3(grid), dim3(kBlock), 0, 0, A, B, rows, mp, rpc, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = (double)TABLES * rows * M * 8;
  printf("tables %d  rows 2^%d x cols 2^%d  PW %d KiB  RIF %d  waves %6zu (chunks %4zu x %3zu rows)  %.1f us  %.2f TB/s\n", TABLES,
         (int)__builtin_ctzll(rows), (int)__builtin_ctzll(M), PW, RIF, spans * chunks, chunks, rpc, ms[3] * 1e3, bytes / (ms[3] * 1e-3) / 1e12);
}

int main() {
  const size_t words = (size_t)1 << 27;   // two tables of 2^26 u64 + room for 2^28 one-table runs
  ull2 *buf = nullptr, *out = nullptr;
  CK(hipMalloc(&buf, ((size_t)1 << 28) * 8));
  CK(hipMalloc(&out, (size_t)256 << 20));
  CK(hipMemset(buf, 1, ((size_t)1 << 28) * 8));
  const ull2* A = buf;
  const ull2* B = buf + words / 4;   // 2^26 u64 = 2^25 ull2 further on
  for (int waves : {1024, 4096}) {
    // gkr_phase1 shape: two 2^13 x 2^13 tables
    run<1, 2, 2>(A, B, out, 8192, 8192, waves);
    run<1, 4, 2>(A, B, out, 8192, 8192, waves);
    run<2, 2, 2>(A, B, out, 8192, 8192, waves);
    run<2, 4, 2>(A, B, out, 8192, 8192, waves);
    run<4, 2, 2>(A, B, out, 8192, 8192, waves);
    run<4, 4, 2>(A, B, out, 8192, 8192, waves);
    run<8, 2, 2>(A, B, out, 8192, 8192, waves);
    // coldot shape: one 2^14 x 2^14 table
    run<1, 4, 1>(A, B, out, 16384, 16384, waves);
    run<2, 4, 1>(A, B, out, 16384, 16384, waves);
    run<4, 4, 1>(A, B, out, 16384, 16384, waves);
    run<8, 2, 1>(A, B, out, 16384, 16384, waves);
    run<4, 8, 1>(A, B, out, 16384, 16384, waves);
  }
  return 0;
}
