// Standalone A/B harness for the pass kernels (not part of the product): times each pass
// shape on one GPU with HIP events, many repetitions, same process.  Build variants with
// -D flags (see tools/run_kbench.sh) and compare medians.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../thaler-study_amd/csrc/kernels.hpp"
using namespace sc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static unsigned g_ticket_base = 0;
template <int KF, int KS>
double run(const char* tag, u64* A, u64* B, u64* A2, u64* B2, u64* P, u64* S, int log_in, int grid_max, int nt_ld, int nt_st, int reps) {
  GoldilocksMont f;
  size_t n_units = (size_t)1 << (log_in - KF - KS);
  int grid = (int)std::min<size_t>((n_units + kBlock - 1) / kBlock, (size_t)grid_max);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0));
    FoldW fw; for (int c = 0; c < 8; ++c) fw.w[c] = 0x1234567ull * (c + 3);
    PassOut out; out.partials = P; out.n_rows = 4096; out.ticket = (unsigned*)(S + 40); out.ticket_base = g_ticket_base;
    out.sums_dev = S; out.mailbox = nullptr; out.seq = 0;
    if (grid > 1) g_ticket_base += grid;
    if (nt_ld && nt_st)
      hipLaunchKernelGGL((pass_kernel<GoldilocksMont, KF, KS, 3>), dim3(grid), dim3(kBlock), 0, 0, f, A, B, A2, B2, fw, n_units, out);
    else if (nt_ld)
      hipLaunchKernelGGL((pass_kernel<GoldilocksMont, KF, KS, 1>), dim3(grid), dim3(kBlock), 0, 0, f, A, B, A2, B2, fw, n_units, out);
    else
      hipLaunchKernelGGL((pass_kernel<GoldilocksMont, KF, KS, 0>), dim3(grid), dim3(kBlock), 0, 0, f, A, B, A2, B2, fw, n_units, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  double med = ts[ts.size() / 2] * 1e3;
  double bytes = 16.0 * (double)((size_t)1 << log_in) + (KF ? 16.0 * (double)((size_t)1 << (log_in - KF)) : 0.0);
  printf("%-10s kf=%d ks=%d log=%d grid=%d nt=%d/%d : median %.1f us  min %.1f us  -> %.2f TB/s\n", tag, KF, KS, log_in, grid, nt_ld, nt_st,
         med, ts[0] * 1e3, bytes / med / 1e6);
  return med;
}

int main(int argc, char** argv) {
  const char* tag = argc > 1 ? argv[1] : "base";
  int log_n = argc > 2 ? atoi(argv[2]) : 28;
  int reps = argc > 3 ? atoi(argv[3]) : 15;
  size_t N = (size_t)1 << log_n;
  u64 *A, *B, *A2, *B2, *P, *S;
  CK(hipMalloc(&A, N * 8)); CK(hipMalloc(&B, N * 8)); CK(hipMalloc(&A2, N * 4)); CK(hipMalloc(&B2, N * 4));
  CK(hipMalloc(&P, 4096 * 16 * 8)); CK(hipMalloc(&S, 64 * 8)); CK(hipMemset(S, 0, 64 * 8));
  GoldilocksMont f;
  hipLaunchKernelGGL((generate_kernel<GoldilocksMont>), dim3(2048), dim3(kBlock), 0, 0, f, (u64)1, (u64)0, N, A);
  hipLaunchKernelGGL((generate_kernel<GoldilocksMont>), dim3(2048), dim3(kBlock), 0, 0, f, (u64)2, (u64)0, N, B);
  CK(hipDeviceSynchronize());
  for (int gm : {1024, 2048, 4096}) {
    run<0, 1>(tag, A, B, A2, B2, P, S, log_n, gm, 1, 1, reps);
    run<0, 2>(tag, A, B, A2, B2, P, S, log_n, gm, 1, 1, reps);
    run<1, 1>(tag, A, B, A2, B2, P, S, log_n, gm, 1, 1, reps);
    run<2, 2>(tag, A, B, A2, B2, P, S, log_n, gm, 1, 1, reps);
  }
  run<0, 1>(tag, A, B, A2, B2, P, S, log_n, 2048, 0, 0, reps);
  run<0, 2>(tag, A, B, A2, B2, P, S, log_n, 2048, 0, 0, reps);
  run<2, 2>(tag, A, B, A2, B2, P, S, log_n, 2048, 0, 0, reps);
  run<2, 2>(tag, A, B, A2, B2, P, S, log_n - 2, 2048, 0, 0, reps);
  run<2, 2>(tag, A, B, A2, B2, P, S, log_n - 4, 2048, 0, 0, reps);
  run<2, 2>(tag, A, B, A2, B2, P, S, log_n - 6, 2048, 0, 0, reps);
  return 0;
}
