// Stand-alone timing of wfold_pass_kernel<GoldilocksMont, 4, 5> (the fold behind the matrix-core first pass) and of
// wgrid_pass_kernel<., 5> - not part of the product.  Built against ANY copy of the kernel headers (-DKERNELS_HPP='"path"'), so
// two source states run side by side on one box and one set of tables: tools/run_wfbench.sh builds `base` from a saved copy
// and `new` from the tree, the GPU box alternates them.  Every run prints a checksum of the 243 cells and of the folded tables:
// a variant that is faster and wrong shows at once.
//   wfbench <tag> [log ...]      e.g. wfbench new 21 23 25 28
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef KERNELS_HPP
#define KERNELS_HPP "../thaler-study_amd/csrc/kernels.hpp"
#endif
#include KERNELS_HPP
using namespace sc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void xor_fold(const u64* t, size_t n, u64* out) {
  u64 acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= t[i] * (2 * i + 1);
  atomicXor((unsigned long long*)out, (unsigned long long)acc);
}

int main(int argc, char** argv) {
  const char* tag = argc > 1 ? argv[1] : "new";
  std::vector<int> logs;
  for (int i = 2; i < argc; ++i) logs.push_back(atoi(argv[i]));
  if (logs.empty()) logs = {25, 28};
  const int max_log = *std::max_element(logs.begin(), logs.end());
  const size_t N = (size_t)1 << max_log;
  u64 *A, *B, *A2, *B2, *P, *GR, *mb, *chk;
  unsigned* T;
  CK(hipMalloc(&A, N * 8)); CK(hipMalloc(&B, N * 8)); CK(hipMalloc(&A2, N / 2)); CK(hipMalloc(&B2, N / 2));
  CK(hipMalloc(&P, (size_t)1024 * kGridChunk * 8)); CK(hipMalloc(&GR, (size_t)32 * kGridChunk * 8));
  CK(hipMalloc(&T, 64 * sizeof(unsigned))); CK(hipMemset(T, 0, 64 * sizeof(unsigned)));
  CK(hipHostMalloc(&mb, kMailboxWords * 8 + 64, hipHostMallocMapped)); memset(mb, 0, kMailboxWords * 8 + 64);
  CK(hipMalloc(&chk, 16));
  GoldilocksMont f;
  hipLaunchKernelGGL((generate_kernel<GoldilocksMont>), dim3(2048), dim3(kBlock), 0, 0, f, (u64)0xA5A5000000000001ull, (u64)0, N, A);
  hipLaunchKernelGGL((generate_kernel<GoldilocksMont>), dim3(2048), dim3(kBlock), 0, 0, f, (u64)0xB6B6000000000002ull, (u64)0, N, B);
  CK(hipDeviceSynchronize());
  int per_cu = 0, cus = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&wfold_pass_kernel<GoldilocksMont, 4, 5, true>), kWfThreads, 0));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  FoldW fw;
  for (int c = 0; c < 16; ++c) fw.w[c] = 0x9E3779B97F4A7C15ull * (c + 3) % 0xFFFFFFFF00000001ull;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  u64 seq = 0;
  const int reps = getenv("WF_REPS") ? atoi(getenv("WF_REPS")) : 30;
  for (int lg : logs) {
    const size_t n_tiles = (size_t)1 << (lg - 12);
    const int grid = (int)std::min<size_t>(n_tiles, (size_t)per_cu * cus);
    std::vector<float> ts;
    for (int r = 0; r < reps + 3; ++r) {
      WgOut wo;
      wo.partials = P; wo.group_rows = GR; wo.tickets = T; wo.mailbox = mb; wo.seq = ++seq; wo.limbs_dev = nullptr;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((wfold_pass_kernel<GoldilocksMont, 4, 5, true>), dim3(grid), dim3(kWfThreads), 0, 0, f, A, B, A2, B2, fw, (u64)0, n_tiles, wo);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    u64 cells = 0;
    for (int c = 0; c < 243; ++c) cells ^= mb[kMailboxWide + c] * (2 * c + 1);
    CK(hipMemset(chk, 0, 16));
    hipLaunchKernelGGL(xor_fold, dim3(1024), dim3(256), 0, 0, A2, (size_t)1 << (lg - 4), chk);
    hipLaunchKernelGGL(xor_fold, dim3(1024), dim3(256), 0, 0, B2, (size_t)1 << (lg - 4), chk + 1);
    u64 h[2]; CK(hipMemcpy(h, chk, 16, hipMemcpyDeviceToHost));
    const double bytes = 16.0 * (double)((size_t)1 << lg) + 16.0 * (double)((size_t)1 << (lg - 4));
    printf("%-6s wfold(4,5)@%d grid=%d: median %.1f us  min %.1f  p90 %.1f -> %.2f TB/s | cells %016llx tables %016llx %016llx seq %s\n", tag, lg, grid,
           ts[ts.size() / 2] * 1e3, ts[0] * 1e3, ts[ts.size() * 9 / 10] * 1e3, bytes / (ts[ts.size() / 2] * 1e3) / 1e6,
           (unsigned long long)cells, (unsigned long long)h[0], (unsigned long long)h[1], mb[kMailboxSeq] == seq ? "ok" : "BAD");
  }
  // the five-round pass on cache-resident tables (wgrid_pass_kernel<., 5, false>, kf = 5): the launch behind it in a proof
  for (int lg : {21, 16}) {
    int pc = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, reinterpret_cast<const void*>(&wgrid_pass_kernel<GoldilocksMont, 5, false>), kBlock, 0));
    GridW gw;
    for (int c = 0; c < 32; ++c) gw.w[c] = 0x9E3779B97F4A7C15ull * (c + 5) % 0xFFFFFFFF00000001ull;
    const size_t n_out = (size_t)1 << (lg - 5);
    const size_t n_iter = (n_out + kWgEntries - 1) / kWgEntries;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((n_iter + 3) / 4, std::min(pc * cus, 1024)));
    std::vector<float> ts;
    for (int r = 0; r < reps + 3; ++r) {
      WgOut wo;
      wo.partials = P; wo.group_rows = GR; wo.tickets = T; wo.mailbox = mb; wo.seq = ++seq; wo.limbs_dev = nullptr;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((wgrid_pass_kernel<GoldilocksMont, 5, false>), dim3(grid), dim3(kBlock), 0, 0, f, A, B, A2, B2, gw, 5, n_out, wo);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    u64 cells = 0;
    for (int c = 0; c < 243; ++c) cells ^= mb[kMailboxWide + c] * (2 * c + 1);
    printf("%-6s wgrid(5,5)@%d grid=%d: median %.1f us  min %.1f | cells %016llx\n", tag, lg, grid, ts[ts.size() / 2] * 1e3, ts[0] * 1e3, (unsigned long long)cells);
  }
  return 0;
}
