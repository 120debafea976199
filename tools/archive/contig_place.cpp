// Follow-up of tools/vmm_place.cpp: does PHYSICAL CONTIGUITY select the fold pass's speed?  Per process: one context with the
// pool's ordinary allocations and one with option "pool_contiguous" (every pool block of >= 1 MiB from
// hipExtMallocWithFlags(hipDeviceMallocContiguous): the caller's tables made by sc_table_generate AND the pass outputs),
// each proved over three times (fresh tables each time), plus contiguous caller tables with ordinary outputs and vice versa.
// build: hipcc -O2 -std=c++17 -o tools/build/contig_place tools/contig_place.cpp -Lthaler-study_amd -lsumcheck_hip -Wl,-rpath,$PWD/thaler-study_amd
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/sumcheck_hip.h"

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__);  \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)
#define SC(x)                                                                       \
  do {                                                                              \
    int rc_ = (x);                                                                  \
    if (rc_ != SC_OK) {                                                             \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, sc_last_error(ctx));               \
      exit(3);                                                                      \
    }                                                                               \
  } while (0)

static const int N = 28;
static const size_t kTableBytes = ((size_t)8) << N;

static void time_fold(sc_ctx* ctx, const sc_table* ta, const sc_table* tb, int reps, double* fold_us, double* first_us, double* proof_ms) {
  uint64_t c1;
  for (int i = 0; i < 12; ++i) SC(sc_prove(ctx, ta, tb, nullptr, nullptr, 0xC7C7000000000003ull, &c1, nullptr, nullptr));   // clocks up
  SC(sc_ctx_set_option(ctx, "time_kernels", 1));
  size_t n = 0;
  SC(sc_ctx_launch_log(ctx, nullptr, 0, &n, 1));
  std::vector<double> wall;
  for (int i = 0; i < reps; ++i) {
    auto t0 = std::chrono::steady_clock::now();
    SC(sc_prove(ctx, ta, tb, nullptr, nullptr, 0xC7C7000000000003ull, &c1, nullptr, nullptr));
    wall.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  std::vector<sc_launch_record> log(4096);
  SC(sc_ctx_launch_log(ctx, log.data(), log.size(), &n, 1));
  SC(sc_ctx_set_option(ctx, "time_kernels", 0));
  std::vector<double> f, g;
  for (size_t i = 0; i < std::min(n, log.size()); ++i) {
    if (log[i].kind == SC_KIND_PASS && log[i].kf == 3 && log[i].ks == 2 && log[i].log_in == N) f.push_back(log[i].ms * 1e3);
    if (log[i].kind == SC_KIND_PASS && log[i].kf == 0 && log[i].ks == 3 && log[i].log_in == N) g.push_back(log[i].ms * 1e3);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  *fold_us = med(f);
  *first_us = med(g);
  *proof_ms = med(wall);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 6;
  sc_field f;
  sc_field_from_modulus(0xFFFFFFFF00000001ull, &f);
  for (int round = 0; round < 2; ++round)
    for (int contiguous = 0; contiguous < 2; ++contiguous) {
      sc_ctx* ctx = nullptr;
      if (sc_ctx_create(&f, 0, &ctx) != SC_OK) return 1;
      SC(sc_ctx_set_option(ctx, "pool_contiguous", contiguous));
      for (int t = 0; t < 3; ++t) {
        sc_table *ga = nullptr, *gb = nullptr;
        SC(sc_table_generate(ctx, 0xA5A5000000000001ull, 0, (size_t)1 << N, &ga));
        SC(sc_table_generate(ctx, 0xB6B6000000000002ull, 0, (size_t)1 << N, &gb));
        double fu, gu, pm;
        time_fold(ctx, ga, gb, reps, &fu, &gu, &pm);
        printf("pool %-10s tables %d (pool's)         fold pass %7.1f us  first pass %7.1f us  proof %.4f ms\n", contiguous ? "contiguous" : "ordinary", t, fu, gu, pm);
        fflush(stdout);
        if (t == 2) {
          // the caller's tables the OTHER way, the pool's outputs as they are
          void *pa = nullptr, *pb = nullptr;
          const unsigned flag = contiguous ? 0u : (unsigned)hipDeviceMallocContiguous;
          hipError_t e = flag ? hipExtMallocWithFlags(&pa, kTableBytes, flag) : hipMalloc(&pa, kTableBytes);
          if (e == hipSuccess) e = flag ? hipExtMallocWithFlags(&pb, kTableBytes, flag) : hipMalloc(&pb, kTableBytes);
          if (e == hipSuccess) {
            CK(hipMemcpy(pa, sc_table_device_ptr(ga), kTableBytes, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(pb, sc_table_device_ptr(gb), kTableBytes, hipMemcpyDeviceToDevice));
            sc_table *ta = nullptr, *tb = nullptr;
            SC(sc_table_from_device(ctx, (const uint64_t*)pa, (size_t)1 << N, &ta));
            SC(sc_table_from_device(ctx, (const uint64_t*)pb, (size_t)1 << N, &tb));
            time_fold(ctx, ta, tb, reps, &fu, &gu, &pm);
            printf("pool %-10s tables %-22s fold pass %7.1f us  first pass %7.1f us  proof %.4f ms\n", contiguous ? "contiguous" : "ordinary",
                   contiguous ? "ordinary hipMalloc" : "contiguous", fu, gu, pm);
            fflush(stdout);
            sc_table_free(ctx, ta);
            sc_table_free(ctx, tb);
            SC(sc_ctx_synchronize(ctx));
          } else {
            printf("(allocation of the other kind failed: %s)\n", hipGetErrorString(e));
            (void)hipGetLastError();
          }
          if (pa) (void)hipFree(pa);
          if (pb) (void)hipFree(pb);
        }
        sc_table_free(ctx, ga);
        sc_table_free(ctx, gb);
      }
      sc_ctx_destroy(ctx);
    }
  return 0;
}
