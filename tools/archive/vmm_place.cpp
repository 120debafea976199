// does the PHYSICAL placement of the tables select the speed of the first folding pass?
// The caller's two 2 GiB tables (and, optionally, nothing else) are built from physical chunks obtained with the HIP
// virtual-memory API (hipMemCreate) and mapped into reserved address ranges (hipMemMap) in different ORDERS; the library
// proves over them through sc_table_from_device and the launch log gives the duration of pass_kernel<3,2> on 2^28 entries.
// If a mapping rule existed (e.g. "A and B from alternating chunks", "B's chunks rotated"), the pool and the table
// constructors could apply it; if every mapping of the same physical memory runs in the same mode, placement below the
// granule is the driver's and the experiment is a committed negative result.
//
// build: hipcc -O2 -std=c++17 -o tools/build/vmm_place tools/vmm_place.cpp -Lthaler-study_amd -lsumcheck_hip -Wl,-rpath,$PWD/thaler-study_amd
// run:   tools/build/vmm_place [log2 chunk bytes = 21 ...]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
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

static sc_ctx* ctx = nullptr;
static const int N = 28;
static const size_t kTableBytes = ((size_t)8) << N;

struct Chunks {
  size_t chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> h;
};

static Chunks make_chunks(size_t chunk, size_t count) {
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  Chunks c;
  c.chunk = chunk;
  c.h.resize(count);
  for (size_t i = 0; i < count; ++i) CK(hipMemCreate(&c.h[i], chunk, &prop, 0));
  return c;
}
static void map_range(void* base, const Chunks& c, const std::vector<size_t>& order) {
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (size_t i = 0; i < order.size(); ++i) CK(hipMemMap((char*)base + i * c.chunk, c.chunk, 0, c.h[order[i]], 0));
  CK(hipMemSetAccess(base, order.size() * c.chunk, &acc, 1));
}
static void unmap_range(void* base, size_t chunk, size_t count) {
  for (size_t i = 0; i < count; ++i) CK(hipMemUnmap((char*)base + i * chunk, chunk));
}

// duration of pass_kernel<3,2> on 2^28 entries (and of the first pass) over `reps` proofs: median
static void time_fold(const uint64_t* a, const uint64_t* b, int reps, double* fold_us, double* first_us, double* proof_ms) {
  sc_table *ta = nullptr, *tb = nullptr;
  SC(sc_table_from_device(ctx, a, (size_t)1 << N, &ta));
  SC(sc_table_from_device(ctx, b, (size_t)1 << N, &tb));
  uint64_t c1;
  for (int i = 0; i < 2; ++i) SC(sc_prove(ctx, ta, tb, nullptr, nullptr, 0xC7C7000000000003ull, &c1, nullptr, nullptr));
  SC(sc_ctx_set_option(ctx, "time_kernels", 1));
  size_t n = 0;
  SC(sc_ctx_launch_log(ctx, nullptr, 0, &n, 1));
  std::vector<double> wall;
  for (int i = 0; i < reps; ++i) {
    hipEvent_t e0;
    (void)e0;
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
  sc_table_free(ctx, ta);
  sc_table_free(ctx, tb);
}

int main(int argc, char** argv) {
  const int chunk_log = argc > 1 ? atoi(argv[1]) : 21;
  const int reps = argc > 2 ? atoi(argv[2]) : 6;
  sc_field f;
  sc_field_from_modulus(0xFFFFFFFF00000001ull, &f);
  if (sc_ctx_create(&f, 0, &ctx) != SC_OK) {
    fprintf(stderr, "ctx: %s\n", sc_last_error(nullptr));
    return 1;
  }
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  size_t chunk = (size_t)1 << chunk_log;
  if (chunk < gran) chunk = gran;
  const size_t per_table = kTableBytes / chunk, total = 2 * per_table;
  printf("granularity %zu B; chunk %zu B (%zu per 2 GiB table)\n", gran, chunk, per_table);

  // reference: the library's own tables (hipMalloc through the pool)
  sc_table *ga = nullptr, *gb = nullptr;
  SC(sc_table_generate(ctx, 0xA5A5000000000001ull, 0, (size_t)1 << N, &ga));
  SC(sc_table_generate(ctx, 0xB6B6000000000002ull, 0, (size_t)1 << N, &gb));
  double fu, gu, pm;
  time_fold(sc_table_device_ptr(ga), sc_table_device_ptr(gb), reps, &fu, &gu, &pm);
  printf("%-46s fold pass %7.1f us  first pass %7.1f us  proof %.4f ms\n", "hipMalloc (the pool's tables)", fu, gu, pm);
  fflush(stdout);

  Chunks c = make_chunks(chunk, total);
  void *va = nullptr, *vb = nullptr;
  CK(hipMemAddressReserve(&va, kTableBytes, 0, nullptr, 0));
  CK(hipMemAddressReserve(&vb, kTableBytes, 0, nullptr, 0));
  std::mt19937_64 rng(7);
  struct Mapping {
    std::string name;
    std::vector<size_t> oa, ob;
  };
  std::vector<Mapping> maps;
  std::vector<size_t> idx(total);
  std::iota(idx.begin(), idx.end(), 0);
  auto slice = [&](size_t from, size_t step) {
    std::vector<size_t> o;
    for (size_t i = 0; i < per_table; ++i) o.push_back(idx[from + i * step]);
    return o;
  };
  maps.push_back({"creation order: A = first half, B = second", slice(0, 1), slice(per_table, 1)});
  maps.push_back({"interleaved: A = even chunks, B = odd", slice(0, 2), slice(1, 2)});
  {
    Mapping m{"B's chunks reversed", slice(0, 1), slice(per_table, 1)};
    std::reverse(m.ob.begin(), m.ob.end());
    maps.push_back(m);
  }
  {
    Mapping m{"B's chunks rotated by one", slice(0, 1), slice(per_table, 1)};
    std::rotate(m.ob.begin(), m.ob.begin() + 1, m.ob.end());
    maps.push_back(m);
  }
  {
    Mapping m{"B's chunks rotated by half", slice(0, 1), slice(per_table, 1)};
    std::rotate(m.ob.begin(), m.ob.begin() + per_table / 2, m.ob.end());
    maps.push_back(m);
  }
  maps.push_back({"swapped: A = second half, B = first", slice(per_table, 1), slice(0, 1)});
  for (int r = 0; r < 3; ++r) {
    std::vector<size_t> p = idx;
    std::shuffle(p.begin(), p.end(), rng);
    Mapping m{"random permutation " + std::to_string(r), {}, {}};
    m.oa.assign(p.begin(), p.begin() + per_table);
    m.ob.assign(p.begin() + per_table, p.end());
    maps.push_back(m);
  }
  maps.push_back({"creation order again", slice(0, 1), slice(per_table, 1)});
  for (const Mapping& m : maps) {
    map_range(va, c, m.oa);
    map_range(vb, c, m.ob);
    CK(hipMemcpy(va, sc_table_device_ptr(ga), kTableBytes, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(vb, sc_table_device_ptr(gb), kTableBytes, hipMemcpyDeviceToDevice));
    CK(hipDeviceSynchronize());
    time_fold((const uint64_t*)va, (const uint64_t*)vb, reps, &fu, &gu, &pm);
    printf("%-46s fold pass %7.1f us  first pass %7.1f us  proof %.4f ms\n", m.name.c_str(), fu, gu, pm);
    fflush(stdout);
    SC(sc_ctx_synchronize(ctx));
    unmap_range(va, chunk, per_table);
    unmap_range(vb, chunk, per_table);
  }
  time_fold(sc_table_device_ptr(ga), sc_table_device_ptr(gb), reps, &fu, &gu, &pm);
  printf("%-46s fold pass %7.1f us  first pass %7.1f us  proof %.4f ms\n", "hipMalloc again", fu, gu, pm);
  for (auto h : c.h) CK(hipMemRelease(h));
  // more ways to obtain the two tables (round 4, after a box showed hipMalloc fast and every chunk mapping slow):
  // fresh hipMalloc pairs made now, ONE hipMemCreate per table, ONE for both, and fine-grained / uncached allocations
  auto fill_and_time = [&](const char* name, void* pa, void* pb) {
    CK(hipMemcpy(pa, sc_table_device_ptr(ga), kTableBytes, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(pb, sc_table_device_ptr(gb), kTableBytes, hipMemcpyDeviceToDevice));
    CK(hipDeviceSynchronize());
    time_fold((const uint64_t*)pa, (const uint64_t*)pb, reps, &fu, &gu, &pm);
    printf("%-46s fold pass %7.1f us  first pass %7.1f us  proof %.4f ms\n", name, fu, gu, pm);
    fflush(stdout);
  };
  for (int r = 0; r < 3; ++r) {
    void *pa = nullptr, *pb = nullptr;
    CK(hipMalloc(&pa, kTableBytes));
    CK(hipMalloc(&pb, kTableBytes));
    fill_and_time(("fresh hipMalloc pair " + std::to_string(r)).c_str(), pa, pb);
    SC(sc_ctx_synchronize(ctx));
    CK(hipFree(pa));
    CK(hipFree(pb));
  }
  {
    void* pab = nullptr;
    CK(hipMalloc(&pab, 2 * kTableBytes));
    fill_and_time("one hipMalloc of 4 GiB: A | B", pab, (char*)pab + kTableBytes);
    SC(sc_ctx_synchronize(ctx));
    CK(hipFree(pab));
  }
  {
    Chunks one = make_chunks(kTableBytes, 2);
    map_range(va, one, {0});
    map_range(vb, one, {1});
    fill_and_time("hipMemCreate: ONE 2 GiB handle per table", va, vb);
    SC(sc_ctx_synchronize(ctx));
    unmap_range(va, kTableBytes, 1);
    unmap_range(vb, kTableBytes, 1);
    for (auto h : one.h) CK(hipMemRelease(h));
  }
  {
    void *pa = nullptr, *pb = nullptr;
    if (hipExtMallocWithFlags(&pa, kTableBytes, hipDeviceMallocFinegrained) == hipSuccess &&
        hipExtMallocWithFlags(&pb, kTableBytes, hipDeviceMallocFinegrained) == hipSuccess) {
      fill_and_time("hipExtMallocWithFlags(fine-grained)", pa, pb);
      SC(sc_ctx_synchronize(ctx));
    }
    (void)hipGetLastError();
    if (pa) (void)hipFree(pa);
    if (pb) (void)hipFree(pb);
  }
  {
    void *pa = nullptr, *pb = nullptr;
    if (hipExtMallocWithFlags(&pa, kTableBytes, hipDeviceMallocUncached) == hipSuccess &&
        hipExtMallocWithFlags(&pb, kTableBytes, hipDeviceMallocUncached) == hipSuccess) {
      fill_and_time("hipExtMallocWithFlags(uncached)", pa, pb);
      SC(sc_ctx_synchronize(ctx));
    }
    (void)hipGetLastError();
    if (pa) (void)hipFree(pa);
    if (pb) (void)hipFree(pb);
  }
  time_fold(sc_table_device_ptr(ga), sc_table_device_ptr(gb), reps, &fu, &gu, &pm);
  printf("%-46s fold pass %7.1f us  first pass %7.1f us  proof %.4f ms\n", "the pool's tables, last", fu, gu, pm);
  sc_table_free(ctx, ga);
  sc_table_free(ctx, gb);
  sc_ctx_destroy(ctx);
  return 0;
}
