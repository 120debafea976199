// libsumcheck_hip.so - host engine + C ABI (include/sumcheck_hip.h) over the gfx950 kernels.
//
// Layout of this file:
//   1. context, error plumbing, device-buffer pool, workspace
//   2. collective transports (none / RCCL via dlopen / caller-supplied host callbacks)
//   3. kernel launch helpers (field dispatch Goldilocks vs generic Montgomery)
//   4. table API (upload/generate/clone/download/fix_variables/evaluate/relabel)
//   5. product-of-two-tables API (matrix_multiplication::G)
//   6. the prover state machine (sum_check_protocol::Prover) with the
//      two-variables-per-pass schedule and the sharded (one rank per GPU) mode
//
// There is deliberately no CPU path: without a usable HIP device every computing entry
// point fails with SC_ERR_HIP.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sumcheck_hip.h"
#include "kernels.hpp"

using sc::u64;

// =====================================================================================
// 1. context
// =====================================================================================

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;
std::string g_create_error;

bool load_rccl(std::string* why) {
  if (g_rccl.handle) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) {
    *why = std::string("dlopen(librccl) failed: ") + dlerror();
    return false;
  }
  RcclApi a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather) {
    *why = "librccl is missing a required symbol";
    return false;
  }
  g_rccl = a;
  return true;
}

enum class Transport { kNone, kRccl, kHost, kPeer };
constexpr int kWgMaxBlocks = 1024;   // blocks of a wgrid_pass_kernel launch at most: 32 groups of 32

}  // namespace

struct sc_ctx {
  sc::FieldParams fp;
  bool gold = false;
  int device = 0;
  hipStream_t stream = nullptr;
  mutable std::string err;
  // set by the first HIP failure other than an allocation failure (a faulted kernel, a lost mailbox
  // word): the stream, the ticket counter and the pool can no longer be trusted, so every later call
  // fails fast with SC_ERR_STATE instead of spinning on a hand-off that will never come
  mutable bool poisoned = false;

  // options
  int vars_per_pass = 2;
  // rounds served by the first pass (which folds nothing): 1..3, or 0 = by size - three for tables
  // of >= 2^18 entries (it saves an eighth of the traffic of a large proof and a pass of a small
  // one), two below
  int first_pass_vars = 0;
  // the passes on the smaller tables (kernels.hpp, wgrid_pass_kernel): up to five rounds each; 0: two rounds per pass
  // all the way down
  int grid_pass = 1;
  int grid_log = 20;        // largest FOLDED table (log2 entries) they take (measured: 21 costs n = 28 10 us, 19 costs n = 25 18 us)
  int grid_max_vars = 5;    // most rounds one of them serves (1..5)
  int grid_sharded = 1;     // sharded passes too (cells exchanged inside the kernel on the peer transport, summed by the
                            // collective on the others), down to shards that hold only their pending challenges
  int wgrid_blocks = 0;     // resident grid of wgrid_pass_kernel (0 = not asked yet)
  int grid_blocks = 0;      // cap on the blocks of such a launch (0 = as many as are resident; tests use it to reach both ticket levels)
  u64* d_wg_partials = nullptr;   // [kWgMaxBlocks][kGridChunk]
  u64* d_wg_groups = nullptr;     // [kWgMaxBlocks / 32][kGridChunk]
  unsigned* d_wg_tickets = nullptr;
  int tail_log = 16;  // shard log-size at which a sharded prover gathers: a 512 KiB all-gather per table is
                      // cheaper than the ~25 us of collective latency of each further sharded pass
  // grid cap of the streaming kernels: three 256-thread blocks per CU (set in sc_ctx_create).
  // Measured at n = 28: fix_variables k=1 685 us with 2048 blocks, 611 with 1024, 580 with 768,
  // 629 with 512 - more blocks than that only add concurrent DRAM streams and a longer final reduction
  int max_blocks = 768;
  int num_cus = 256;
  // blocks of each pass-kernel instantiation that fit on the chip at once ([generic|goldilocks][kf][ks],
  // 0 = not asked yet)
  int resident_blocks[2][4][4] = {};
  int time_kernels = 0;
  int nt_load_log = 22;   // tables of >= 2^this entries are loaded nontemporal (measured: 21-25 equal, 27 and off worse)
  int nt_store_log = 25;  // outputs of >= 2^this entries are stored nontemporal

  // workspace
  u64* d_partials = nullptr;  // [kMaxSums + spare][partial_rows]
  u64* d_sums = nullptr;      // 2*kMaxSums split limbs (+ spare)
  u64* h_sums = nullptr;      // pinned mirror
  size_t partial_rows = 0;
  unsigned* d_ticket = nullptr;  // arrival counter of finish_pass (only ever grows)
  unsigned ticket_base = 0;
  u64* h_mailbox = nullptr;   // pinned, device-mapped: sums + sequence word written by the kernel
  u64* d_mailbox = nullptr;   // device alias of h_mailbox
  u64 mailbox_seq = 0;
  int use_mailbox = 1;

  // device-buffer pool (free blocks by capacity in words; live blocks by pointer)
  std::multimap<size_t, u64*> pool_free;
  std::map<u64*, size_t> pool_live;

  // sharding
  Transport transport = Transport::kNone;
  int rank = 0, world = 1, log_world = 0;
  ncclComm_t comm = nullptr;
  sc_allreduce_fn host_allreduce = nullptr;
  sc_allgather_fn host_allgather = nullptr;
  void* host_user = nullptr;
  // peer transport (kernels.hpp, PeerX): this rank's region = inbox + two gather arenas, and every
  // rank's region as this process maps it
  u64* peer_region = nullptr;
  size_t peer_region_words = 0;
  bool peer_exported = false;
  u64* peer_base[sc::kMaxPeers] = {};
  bool peer_ipc_opened[sc::kMaxPeers] = {};
  int arena_log = 17;          // a gather arena holds world * 2^arena_log words per table (longer gathers go in chunks)
  unsigned xchg_tag = 0;       // exchange tag of the last sharded launch that reached the stream (the same on every rank)
  unsigned xchg_next = 0;      // tag handed to the launch being prepared (fill_peer); committed by commit_peer()
  unsigned gather_count = 0;   // gathers done: its parity selects the arena (NOT the tag's: passes advance the tag too)
  // bound of every in-kernel wait for a peer: the skew between the ranks' launches of the same pass.  The cold-start
  // lag of a freshly started job (seconds: code objects, first launches) is absorbed by the connect-time handshake
  // (peer_connect_ms), so this can be a real failure detector
  int peer_spin_ms = 2000;
  int peer_connect_ms = 120000;   // how long sc_ctx_comm_peer_connect waits for every peer's hello
  // fault injection (tests): delay every sharded launch of this rank by dbg_delay_ms on the host; dbg_skip_tag = 1
  // makes the next sharded launch skip a tag (a rank that is out of step with its peers)
  int dbg_delay_ms = 0;
  int dbg_skip_tag = 0;

  // kernel timing
  // pass-kernel timing (option "time_kernels"): a ring of event pairs, read back only when the
  // ring is full or the totals are queried, so that timing adds two event records per launch
  // and no host synchronisation to the rounds it measures
  static constexpr int kTimerRing = 64;
  static constexpr size_t kLaunchLogCap = 1 << 16;
  hipEvent_t kt_ev[kTimerRing][2] = {};
  sc_launch_record kt_meta[kTimerRing] = {};  // what each pending event pair brackets
  int kt_used = 0;
  double kt_ms = 0.0;
  long kt_n = 0;
  std::vector<sc_launch_record> launch_log;   // drained records (sc_ctx_launch_log)
};

struct sc_table {
  u64* d = nullptr;
  size_t len = 0;
};

namespace {

inline void poison(const sc_ctx* ctx) {
  if (ctx) ctx->poisoned = true;
}

int fail(const sc_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  else g_create_error = buf;
  return code;
}

#define SC_HIP(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      if (e_ != hipErrorOutOfMemory) poison(ctx);                                          \
      return fail(ctx, e_ == hipErrorOutOfMemory ? SC_ERR_OOM : SC_ERR_HIP, "%s: %s (%s:%d)", \
                  #call, hipGetErrorString(e_), __FILE__, __LINE__);                       \
    }                                                                                      \
  } while (0)

#define SC_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != SC_OK) return rc_; \
  } while (0)

inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
inline int log2_of(size_t x) {
  int l = 0;
  while (((size_t)1 << l) < x) ++l;
  return l;
}

// ---- host field arithmetic (O(1) work per round: Lagrange weights, limb recombination) --

struct HostField {
  sc::MontGeneric f;
  explicit HostField(const sc::FieldParams& p) : f(p) { c32R = f.mul((((u64)1) << 32) % p.p, p.r2_mod_p); }
  u64 add(u64 a, u64 b) const { return f.add(a, b); }
  u64 sub(u64 a, u64 b) const { return f.sub(a, b); }
  u64 mul(u64 a, u64 b) const { return f.mul(a, b); }
  u64 one() const { return f.r1; }
  u64 neg(u64 a) const { return a ? f.p - a : 0; }
  u64 pow(u64 a, u64 e) const {
    u64 r = one();
    while (e) {
      if (e & 1) r = mul(r, a);
      a = mul(a, a);
      e >>= 1;
    }
    return r;
  }
  u64 inv(u64 a) const { return pow(a, f.p - 2); }
  // (sum of low limbs) + 2^32 * (sum of high limbs)  mod p.  This runs 6-54 times per pass on the critical
  // path between a pass's sums and the next challenge: a 128-bit `%` (a libgcc call, ~100 ns) is replaced by
  // one Montgomery product with the constant 2^32 * R (hi * 2^32 = mont_mul(hi, 2^32 R)); limb sums are
  // below 2^36, so for moduli above that neither operand needs a reduction first.
  u64 recombine(u64 lo_sum, u64 hi_sum) const {
    const u64 lo = lo_sum < f.p ? lo_sum : lo_sum % f.p;
    const u64 hi = hi_sum < f.p ? hi_sum : hi_sum % f.p;
    return add(lo, mul(hi, c32R));
  }
  u64 c32R = 0;   // to_mont(2^32 mod p)
};

// ---- pool ---------------------------------------------------------------------------

int pool_alloc(sc_ctx* ctx, size_t words, u64** out) {
  if (words < 32) words = 32;
  auto it = ctx->pool_free.lower_bound(words);
  if (it != ctx->pool_free.end() && it->first <= 2 * words) {
    *out = it->second;
    ctx->pool_live[it->second] = it->first;
    ctx->pool_free.erase(it);
    return SC_OK;
  }
  u64* p = nullptr;
  hipError_t e = hipMalloc(&p, words * sizeof(u64));
  if (e != hipSuccess) {
    // release cached blocks and retry once
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
    ctx->pool_free.clear();
    e = hipMalloc(&p, words * sizeof(u64));
    if (e != hipSuccess)
      return fail(ctx, SC_ERR_OOM, "hipMalloc(%zu bytes): %s", words * sizeof(u64), hipGetErrorString(e));
  }
  ctx->pool_live[p] = words;
  *out = p;
  return SC_OK;
}

void pool_release(sc_ctx* ctx, u64* p) {
  if (!p) return;
  auto it = ctx->pool_live.find(p);
  if (it == ctx->pool_live.end()) return;
  ctx->pool_free.emplace(it->second, p);
  ctx->pool_live.erase(it);
}

int new_table(sc_ctx* ctx, size_t len, sc_table** out) {
  sc_table* t = new (std::nothrow) sc_table;
  if (!t) return fail(ctx, SC_ERR_OOM, "host allocation failed");
  int rc = pool_alloc(ctx, len, &t->d);
  if (rc != SC_OK) {
    delete t;
    return rc;
  }
  t->len = len;
  *out = t;
  return SC_OK;
}

int set_device(sc_ctx* ctx) {
  if (ctx->poisoned)
    return fail(ctx, SC_ERR_STATE, "context is unusable after an earlier HIP failure (%s); destroy it", ctx->err.c_str());
  SC_HIP(ctx, hipSetDevice(ctx->device));
  return SC_OK;
}

inline bool is_sharded(const sc_ctx* ctx) { return ctx->transport != Transport::kNone; }

int grid_for(const sc_ctx* ctx, size_t n_threads_needed) {
  size_t g = (n_threads_needed + sc::kBlock - 1) / sc::kBlock;
  if (g < 1) g = 1;
  if (g > (size_t)ctx->max_blocks) g = ctx->max_blocks;
  return (int)g;
}
// kernels that are bound by arithmetic or by scattered accesses, not by a stream: fill every
// wave slot (eight blocks per CU)
int grid_for_wide(const sc_ctx* ctx, size_t n_threads_needed) {
  size_t g = (n_threads_needed + sc::kBlock - 1) / sc::kBlock;
  const size_t cap = std::min<size_t>((size_t)8 * ctx->num_cus, ctx->partial_rows);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// =====================================================================================
// 3. launch helpers
// =====================================================================================

#define SC_DISPATCH_FIELD(ctx, F, f, ...)      \
  do {                                         \
    if ((ctx)->gold) {                         \
      typedef sc::GoldilocksMont F;            \
      F f((ctx)->fp);                          \
      __VA_ARGS__;                             \
    } else {                                   \
      typedef sc::MontGeneric F;               \
      F f((ctx)->fp);                          \
      __VA_ARGS__;                             \
    }                                          \
  } while (0)

// w[c] = prod_j (bit_j(c) ? r[j] : 1 - r[j]), c < 2^kf   (kernels.hpp: FoldW)
sc::FoldW make_fold_weights(const sc_ctx* ctx, const u64* r, int kf) {
  HostField hf(ctx->fp);
  sc::FoldW fw;
  for (int c = 0; c < 8; ++c) fw.w[c] = 0;
  fw.w[0] = hf.one();
  for (int j = 0; j < kf; ++j) {
    const int half = 1 << j;
    for (int c = half - 1; c >= 0; --c) {
      const u64 base = fw.w[c];
      fw.w[c + half] = hf.mul(base, r[j]);
      fw.w[c] = hf.mul(base, hf.sub(hf.one(), r[j]));
    }
  }
  return fw;
}

// the same for the passes on the smallest tables (kernels.hpp: GridW), kf <= 5
sc::GridW make_grid_weights(const sc_ctx* ctx, const u64* r, int kf) {
  HostField hf(ctx->fp);
  sc::GridW gw;
  for (int c = 0; c < (1 << sc::kGridMaxVars); ++c) gw.w[c] = 0;
  gw.w[0] = hf.one();
  for (int j = 0; j < kf; ++j) {
    const int half = 1 << j;
    for (int c = half - 1; c >= 0; --c) {
      const u64 base = gw.w[c];
      gw.w[c + half] = hf.mul(base, r[j]);
      gw.w[c] = hf.mul(base, hf.sub(hf.one(), r[j]));
    }
  }
  return gw;
}

// (H(0), H(1), H(inf)) -> H(2) = 2 H(1) - H(0) + 2 H(inf)   (H quadratic, inf = leading coefficient)
u64 eval2_from_inf(const HostField& hf, u64 e0, u64 e1, u64 einf) {
  u64 t = hf.add(e1, einf);
  return hf.sub(hf.add(t, t), e0);
}

// add the durations of the recorded launches to the totals (waits for the last of them)
void drain_kernel_timers(sc_ctx* ctx) {
  if (ctx->kt_used == 0) return;
  (void)hipEventSynchronize(ctx->kt_ev[ctx->kt_used - 1][1]);
  for (int i = 0; i < ctx->kt_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->kt_ev[i][0], ctx->kt_ev[i][1]) == hipSuccess) {
      ctx->kt_ms += ms;
      ctx->kt_n += 1;
      if (ctx->launch_log.size() < sc_ctx::kLaunchLogCap) {
        sc_launch_record r = ctx->kt_meta[i];
        r.ms = ms;
        ctx->launch_log.push_back(r);
      }
    }
  }
  ctx->kt_used = 0;
}

// Bracket the launch(es) that follow with an event pair on the context's stream (option
// "time_kernels"); bytes_read / bytes_written are what the launch has to move through HBM (every
// input once, every output once).  timer_end() after the launch.
int timer_begin(sc_ctx* ctx, int kind, int kf, int ks, int log_in, u64 bytes_read, u64 bytes_written) {
  if (!ctx->time_kernels) return SC_OK;
  if (ctx->kt_used == sc_ctx::kTimerRing) drain_kernel_timers(ctx);
  sc_launch_record& m = ctx->kt_meta[ctx->kt_used];
  m.kind = kind;
  m.kf = kf;
  m.ks = ks;
  m.log_in = log_in;
  m.bytes_read = bytes_read;
  m.bytes_written = bytes_written;
  m.ms = 0.0;
  SC_HIP(ctx, hipEventRecord(ctx->kt_ev[ctx->kt_used][0], ctx->stream));
  return SC_OK;
}
int timer_end(sc_ctx* ctx) {
  if (!ctx->time_kernels) return SC_OK;
  SC_HIP(ctx, hipEventRecord(ctx->kt_ev[ctx->kt_used][1], ctx->stream));
  ctx->kt_used += 1;
  return SC_OK;
}

// A pass is launched with at most as many blocks as are resident at once (occupancy x CUs) and
// grid-strides over the rest: with 4x more blocks than fit (the old fixed cap of 2048) the chip
// drains and refills between block generations and the last block has 4x more partial sums to
// reduce - 7 % of an n = 28 proof (measured with the max_blocks option: 2048 -> 2.31 ms,
// 1024 -> 2.20, 512 -> 2.13, 256 -> 2.27).
template <class F>
int pass_resident_blocks_t(sc_ctx* ctx, int kf, int ks) {
  const void* fn = nullptr;
#define SC_FN(KF, KS) fn = reinterpret_cast<const void*>(&sc::pass_kernel<F, KF, KS, 1>)
  switch (kf * 4 + ks) {
    case 0 * 4 + 1: SC_FN(0, 1); break;
    case 0 * 4 + 2: SC_FN(0, 2); break;
    case 0 * 4 + 3: SC_FN(0, 3); break;
    case 1 * 4 + 1: SC_FN(1, 1); break;
    case 1 * 4 + 2: SC_FN(1, 2); break;
    case 2 * 4 + 1: SC_FN(2, 1); break;
    case 2 * 4 + 2: SC_FN(2, 2); break;
    case 3 * 4 + 1: SC_FN(3, 1); break;
    case 3 * 4 + 2: SC_FN(3, 2); break;
    default: break;
  }
#undef SC_FN
  int per_cu = 0;
  if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, sc::pass_block_threads(kf, ks), 0) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    return ctx->max_blocks;
  }
  return per_cu * ctx->num_cus;
}

int pass_resident_blocks(sc_ctx* ctx, int kf, int ks) {
  const int fi = ctx->gold ? 1 : 0;
  int& slot = ctx->resident_blocks[fi][kf][ks];
  if (slot == 0) {
    int v = 0;
    SC_DISPATCH_FIELD(ctx, F, f, { (void)f; v = pass_resident_blocks_t<F>(ctx, kf, ks); });
    slot = v > 0 ? v : ctx->max_blocks;
  }
  return slot;
}

template <class F>
void launch_pass_t(sc_ctx* ctx, const F& f, int kf, int ks, const u64* A, const u64* B, u64* A2,
                   u64* B2, const sc::FoldW& fw, size_t n_units, int grid, int log_in, const sc::PassOut& out) {
  dim3 g(grid), b(sc::pass_block_threads(kf, ks));
  hipStream_t s = ctx->stream;
  // streaming hints are compile-time (kernels.hpp, ld16/st16): 0 = cached, 1 = stream the inputs,
  // 3 = stream inputs and outputs
  const int nt = (log_in >= ctx->nt_load_log ? 1 : 0) | ((kf > 0 && (log_in - kf) >= ctx->nt_store_log) ? 2 : 0);
#define SC_PASS(KF, KS)                                                                                            \
  do {                                                                                                             \
    if (nt == 3)                                                                                                   \
      hipLaunchKernelGGL((sc::pass_kernel<F, KF, KS, 3>), g, b, 0, s, f, A, B, A2, B2, fw, n_units, out);          \
    else if (nt & 1)                                                                                               \
      hipLaunchKernelGGL((sc::pass_kernel<F, KF, KS, 1>), g, b, 0, s, f, A, B, A2, B2, fw, n_units, out);          \
    else                                                                                                           \
      hipLaunchKernelGGL((sc::pass_kernel<F, KF, KS, 0>), g, b, 0, s, f, A, B, A2, B2, fw, n_units, out);          \
  } while (0)
  switch (kf * 4 + ks) {
    case 0 * 4 + 1: SC_PASS(0, 1); break;
    case 0 * 4 + 2: SC_PASS(0, 2); break;
    case 0 * 4 + 3: SC_PASS(0, 3); break;
    case 1 * 4 + 1: SC_PASS(1, 1); break;
    case 1 * 4 + 2: SC_PASS(1, 2); break;
    case 2 * 4 + 1: SC_PASS(2, 1); break;
    case 2 * 4 + 2: SC_PASS(2, 2); break;
    case 3 * 4 + 1: SC_PASS(3, 1); break;
    case 3 * 4 + 2: SC_PASS(3, 2); break;
    default: break;
  }
#undef SC_PASS
}

constexpr size_t kInboxRegionWords = 2 * (size_t)sc::kMaxPeers * sc::kInboxWords;   // two parities
// a rank's exported region: [inbox (two parities) | header | two gather arenas of two tables].  The header lets a peer
// check, when it maps the region, that both sides compute the same offsets: {magic, world, rank, arena_log}
constexpr size_t kPeerHeaderWords = 8;
constexpr u64 kPeerMagic = 0x7363706565723033ull;   // "scpeer03"

// 32-bit digest of what a sharded pass folds: identical on every rank unless the ranks were fed
// different challenges (FNV-1a over the words)
unsigned challenge_digest(const u64* r, int kf, int ks, int log_in) {
  u64 h = 0xcbf29ce484222325ull;
  auto mix = [&h](u64 v) {
    for (int i = 0; i < 8; ++i) {
      h ^= (v >> (8 * i)) & 0xFF;
      h *= 0x100000001b3ull;
    }
  };
  mix((u64)kf | ((u64)ks << 8) | ((u64)log_in << 16));
  for (int i = 0; i < kf; ++i) mix(r[i]);
  return (unsigned)(h ^ (h >> 32));
}

// Exchange fields of a sharded launch on the peer transport.  The tag advances with every such launch, in the
// same order on every rank - but only once the launch is known to be in the stream (commit_peer): a launch that
// failed locally must not leave this rank one tag ahead of its peers.
void fill_peer(sc_ctx* ctx, sc::PeerX& px, unsigned digest) {
  for (int q = 0; q < ctx->world; ++q) px.inbox[q] = ctx->peer_base[q];
  px.world = ctx->world;
  px.rank = ctx->rank;
  unsigned tag = ctx->xchg_tag + 1;
  if (ctx->dbg_skip_tag) {   // fault injection: this rank falls out of step
    tag += 1;
    ctx->dbg_skip_tag = 0;
  }
  if (tag == 0) tag = 1;
  ctx->xchg_next = tag;
  px.tag = tag;
  px.digest = digest;
  px.spin_ticks = (u64)ctx->peer_spin_ms * 100000ull;   // wall_clock64 runs at 100 MHz
  if (ctx->dbg_delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(ctx->dbg_delay_ms));   // fault injection: a late rank
}
inline void commit_peer(sc_ctx* ctx) {
  if (ctx->xchg_next) ctx->xchg_tag = ctx->xchg_next;
  ctx->xchg_next = 0;
}

sc::PassOut next_pass_out(sc_ctx* ctx, bool across_ranks = false, unsigned digest = 0, bool* from_mailbox = nullptr);
int commit_pass_out(sc_ctx* ctx, const sc::PassOut& out, int grid);

// Launch one pass over tables of 2^log_in entries.  The 2*NS split limbs end up in the
// host mailbox (*from_mailbox = true; wait with collect_sums) or in ctx->d_sums when they
// still have to be all-reduced on the device (RCCL transport).
int launch_pass(sc_ctx* ctx, int kf, int ks, const u64* A, const u64* B, u64* A2, u64* B2, const u64* r,
                int log_in, bool across_ranks, bool* from_mailbox) {
  if (kf < 0 || kf > 3 || ks < 1 || ks > 3 || (ks == 3 && kf != 0) || log_in < kf + ks)
    return fail(ctx, SC_ERR_ARG, "launch_pass: kf=%d ks=%d log_in=%d", kf, ks, log_in);
  const sc::FoldW fw = make_fold_weights(ctx, r, kf);
  size_t n_units = (size_t)1 << (log_in - kf - ks);
  const size_t bs = (size_t)sc::pass_block_threads(kf, ks);
  int grid = (int)std::min<size_t>(std::max<size_t>((n_units + bs - 1) / bs, 1), (size_t)ctx->max_blocks);
  grid = std::min(grid, pass_resident_blocks(ctx, kf, ks));
  const sc::PassOut out = next_pass_out(ctx, across_ranks, challenge_digest(r, kf, ks, log_in), from_mailbox);
  SC_TRY(timer_begin(ctx, SC_KIND_PASS, kf, ks, log_in, (u64)16 << log_in, kf > 0 ? (u64)16 << (log_in - kf) : 0));
  SC_DISPATCH_FIELD(ctx, F, f, launch_pass_t<F>(ctx, f, kf, ks, A, B, A2, B2, fw, n_units, grid, log_in, out));
  // the launch is in the stream: only now do the ticket base, the mailbox sequence and the exchange tag move (a
  // failed launch must leave them where the device-side counter and the peers still are)
  SC_TRY(commit_pass_out(ctx, out, grid));
  SC_TRY(timer_end(ctx));
  return SC_OK;
}

int wait_mailbox(sc_ctx* ctx, u64 seq);

// after the sequence word of a launch that exchanged with the peers: did the exchange succeed?
int check_exchange(sc_ctx* ctx, const char* what) {
  const u64 err = __atomic_load_n(ctx->h_mailbox + sc::kMailboxErr, __ATOMIC_ACQUIRE);
  if (err == 0) return SC_OK;
  if (err & (u64)sc::kXchgTimeout) {
    poison(ctx);
    return fail(ctx, SC_ERR_RCCL, "peer exchange: a rank's %s did not arrive within %d ms (rank %d waited for rank %d at tag %u; that rank's "
                "slot held tag ..%04x)", what, ctx->peer_spin_ms, ctx->rank, (int)((err >> 8) & 0xFF), ctx->xchg_tag, (unsigned)((err >> 16) & 0x3FFF));
  }
  return fail(ctx, SC_ERR_STATE, "the ranks of this sharded prover were given different challenges");
}

// One pass by wgrid_pass_kernel: folds kf <= 5 pending challenges of tables of 2^log_in entries and leaves the 3^ks
// cells of the next ks <= 5 rounds in the wide mailbox (collect_grid).  Its counters rest at zero.  Sharded passes:
// the peer transport exchanges the cells inside the kernel; RCCL sums their limbs on the stream behind the kernel and
// a small kernel hands the totals to the mailbox; a host transport gets this rank's residues and sums them itself.
int launch_grid_pass(sc_ctx* ctx, int kf, int ks, const u64* A, const u64* B, u64* A2, u64* B2, const u64* r, int log_in, bool across_ranks) {
  if (kf < 0 || kf > sc::kGridMaxVars || ks < 1 || ks > sc::kGridMaxVars || log_in < kf + ks)
    return fail(ctx, SC_ERR_ARG, "launch_grid_pass: kf=%d ks=%d log_in=%d", kf, ks, log_in);
  if (!ctx->use_mailbox) return fail(ctx, SC_ERR_STATE, "launch_grid_pass needs the host mailbox");
  const sc::GridW gw = make_grid_weights(ctx, r, kf);
  const size_t n_out = (size_t)1 << (log_in - kf);
  if (ctx->wgrid_blocks == 0) {
    int per_cu = 0;
    const void* fn = ctx->gold ? reinterpret_cast<const void*>(&sc::wgrid_pass_kernel<sc::GoldilocksMont, 5, false>)
                               : reinterpret_cast<const void*>(&sc::wgrid_pass_kernel<sc::MontGeneric, 5, false>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, sc::kBlock, 0) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 2;
    }
    ctx->wgrid_blocks = std::min(per_cu * ctx->num_cus, kWgMaxBlocks);
  }
  constexpr size_t kWaves = sc::kBlock / sc::kWave;
  const size_t n_iter = (n_out + sc::kWgEntries - 1) / sc::kWgEntries;   // one per wave
  const size_t cap = ctx->grid_blocks > 0 ? (size_t)std::min(ctx->grid_blocks, ctx->wgrid_blocks) : (size_t)ctx->wgrid_blocks;
  const int grid = (int)std::max<size_t>(1, std::min<size_t>((n_iter + kWaves - 1) / kWaves, cap));
  const bool peer = across_ranks && ctx->transport == Transport::kPeer;
  const bool rccl = across_ranks && ctx->transport == Transport::kRccl;
  int cells = 1;
  for (int i = 0; i < ks; ++i) cells *= 3;
  sc::WgOut wo;
  wo.partials = ctx->d_wg_partials;
  wo.group_rows = ctx->d_wg_groups;
  wo.tickets = ctx->d_wg_tickets;
  wo.mailbox = ctx->d_mailbox;
  wo.seq = ctx->mailbox_seq + 1;
  wo.limbs_dev = rccl ? ctx->d_sums : nullptr;
  if (peer) fill_peer(ctx, wo.px, challenge_digest(r, kf, ks, log_in));
  SC_TRY(timer_begin(ctx, SC_KIND_GRID_PASS, kf, ks, log_in, (u64)16 << log_in, kf > 0 ? (u64)16 << (log_in - kf) : 0));
  // the prefetching instantiation: fan-ins 0 and 2 on tables that give a wave more than one iteration (kernels.hpp)
  const bool pf = (kf == 0 || kf == 2) && n_iter > (size_t)grid * kWaves;
#define SC_WG(KS)                                                                                                                       \
  do {                                                                                                                                  \
    if (pf) hipLaunchKernelGGL((sc::wgrid_pass_kernel<F, KS, true>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream, f, A, B, A2, B2, gw, kf, n_out, wo); \
    else hipLaunchKernelGGL((sc::wgrid_pass_kernel<F, KS, false>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream, f, A, B, A2, B2, gw, kf, n_out, wo);   \
  } while (0)
  SC_DISPATCH_FIELD(ctx, F, f, {
    switch (ks) {
      case 1: SC_WG(1); break;
      case 2: SC_WG(2); break;
      case 3: SC_WG(3); break;
      case 4: SC_WG(4); break;
      default: SC_WG(5); break;
    }
  });
#undef SC_WG
  SC_HIP(ctx, hipGetLastError());
  if (peer) commit_peer(ctx);
  SC_TRY(timer_end(ctx));
  if (rccl) {
    ncclResult_t nr = g_rccl.AllReduce(ctx->d_sums, ctx->d_sums, 2 * (size_t)cells, ncclUint64, ncclSum, ctx->comm, ctx->stream);
    if (nr != ncclSuccess) return fail(ctx, SC_ERR_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?");
    hipLaunchKernelGGL(sc::mailbox_copy_wide_kernel, dim3(1), dim3(sc::kBlock), 0, ctx->stream, (const u64*)ctx->d_sums, 2 * cells,
                       ctx->d_mailbox, wo.seq);
    SC_HIP(ctx, hipGetLastError());
  }
  ctx->mailbox_seq += 1;
  return SC_OK;
}
// its cells: wait for the sequence word; the residues, or (sharded) the limb totals of all ranks recombined mod p
int collect_grid(sc_ctx* ctx, int ks, bool across_ranks, u64* out) {
  SC_TRY(wait_mailbox(ctx, ctx->mailbox_seq));
  int cells = 1;
  for (int i = 0; i < ks; ++i) cells *= 3;
  const u64* wide = ctx->h_mailbox + sc::kMailboxWide;
  if (!across_ranks) {
    for (int c = 0; c < cells; ++c) out[c] = wide[c];
    return SC_OK;
  }
  HostField hf(ctx->fp);
  if (ctx->transport == Transport::kHost) {
    // this rank's residues: split, sum over the ranks on the host, recombine
    std::vector<u64> limbs(2 * (size_t)cells);
    for (int c = 0; c < cells; ++c) {
      limbs[2 * c] = wide[c] & 0xFFFFFFFFull;
      limbs[2 * c + 1] = wide[c] >> 32;
    }
    if (ctx->host_allreduce(ctx->host_user, limbs.data(), limbs.size()) != 0)
      return fail(ctx, SC_ERR_RCCL, "host all-reduce callback failed");
    for (int c = 0; c < cells; ++c) out[c] = hf.recombine(limbs[2 * c], limbs[2 * c + 1]);
    return SC_OK;
  }
  if (ctx->transport == Transport::kPeer) SC_TRY(check_exchange(ctx, "sums"));
  for (int c = 0; c < cells; ++c) out[c] = hf.recombine(wide[2 * c], wide[2 * c + 1]);
  return SC_OK;
}

// The last pass of a sharded prover on the peer transport (kernels.hpp, rank_pass_kernel): folds the shard's 2^kf
// pending entries, exchanges the one entry left per table with the peers and leaves the world-entry tables in A2 / B2
// and the 3^log2(world) cells of the rank-bit rounds in out[].
int rank_pass(sc_ctx* ctx, int kf, const u64* A, const u64* B, u64* A2, u64* B2, const u64* r, u64* out) {
  if (ctx->transport != Transport::kPeer || !ctx->use_mailbox || kf < 0 || kf > sc::kGridMaxVars || ctx->log_world < 1 || ctx->log_world > 3)
    return fail(ctx, SC_ERR_STATE, "rank_pass: needs the peer transport, the mailbox and 2..8 ranks");
  const sc::GridW gw = make_grid_weights(ctx, r, kf);
  sc::WgOut wo;
  wo.partials = ctx->d_wg_partials;
  wo.group_rows = ctx->d_wg_groups;
  wo.tickets = ctx->d_wg_tickets;
  wo.mailbox = ctx->d_mailbox;
  wo.seq = ctx->mailbox_seq + 1;
  wo.limbs_dev = nullptr;
  fill_peer(ctx, wo.px, challenge_digest(r, kf, ctx->log_world, kf) ^ 0x72616e6bu);
  SC_TRY(timer_begin(ctx, SC_KIND_GRID_PASS, kf, ctx->log_world, kf, (u64)16 << kf, (u64)16 << ctx->log_world));
  SC_DISPATCH_FIELD(ctx, F, f,
                    hipLaunchKernelGGL((sc::rank_pass_kernel<F>), dim3(1), dim3(sc::kBlock), 0, ctx->stream, f, A, B, A2, B2, gw, kf, wo));
  SC_HIP(ctx, hipGetLastError());
  commit_peer(ctx);
  ctx->mailbox_seq += 1;
  SC_TRY(timer_end(ctx));
  SC_TRY(wait_mailbox(ctx, ctx->mailbox_seq));
  SC_TRY(check_exchange(ctx, "entries"));
  int cells = 1;
  for (int i = 0; i < ctx->log_world; ++i) cells *= 3;
  for (int c = 0; c < cells; ++c) out[c] = ctx->h_mailbox[sc::kMailboxWide + c];
  return SC_OK;
}

// PassOut of the next launch that ends in finish_pass.  across_ranks: its sums are summed over the ranks - inside the
// kernel on the peer transport (digest = what the ranks must agree on), by a collective on the stream for RCCL (the
// limbs stay in d_sums), by the host for a host transport.  *from_mailbox tells collect_sums where the limbs are.
// Nothing is committed here: call commit_pass_out() once the launch is known to be in the stream.
sc::PassOut next_pass_out(sc_ctx* ctx, bool across_ranks, unsigned digest, bool* from_mailbox) {
  const bool peer = across_ranks && ctx->transport == Transport::kPeer;
  const bool mailbox = (ctx->use_mailbox || peer) && !(across_ranks && ctx->transport == Transport::kRccl);
  sc::PassOut out;
  out.partials = ctx->d_partials;
  out.n_rows = (int)ctx->partial_rows;
  out.ticket = ctx->d_ticket;
  out.ticket_base = ctx->ticket_base;
  out.sums_dev = ctx->d_sums;
  out.mailbox = mailbox ? ctx->d_mailbox : nullptr;
  out.seq = mailbox ? ctx->mailbox_seq + 1 : 0;
  if (peer) fill_peer(ctx, out.px, digest);
  if (from_mailbox) *from_mailbox = mailbox;
  return out;
}
int commit_pass_out(sc_ctx* ctx, const sc::PassOut& out, int grid) {
  SC_HIP(ctx, hipGetLastError());
  if (out.px.world > 0) commit_peer(ctx);
  if (out.mailbox) ctx->mailbox_seq += 1;
  if (grid > 1) ctx->ticket_base += (unsigned)grid;
  return SC_OK;
}


// =====================================================================================
// 2. collectives
// =====================================================================================

// Spin on the mailbox sequence word the kernel writes last (system-scope release store).
int wait_mailbox(sc_ctx* ctx, u64 seq) {
  const u64* flag = ctx->h_mailbox + sc::kMailboxSeq;
  unsigned spins = 0;
  auto t0 = std::chrono::steady_clock::now();
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0x3FFF) == 0) {
      hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipSuccess && q != hipErrorNotReady) {
        poison(ctx);
        return fail(ctx, SC_ERR_HIP, "pass kernel failed: %s", hipGetErrorString(q));
      }
      double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (q == hipSuccess && el > 2.0) {
        poison(ctx);
        return fail(ctx, SC_ERR_HIP, "pass kernel finished but its mailbox word never arrived");
      }
      // (a kernel may itself wait up to peer_spin_ms for its peers: the host must outlast it)
      if (el > 120.0 + 1e-3 * ctx->peer_spin_ms) {
        poison(ctx);
        return fail(ctx, SC_ERR_HIP, "timed out waiting for the pass kernel");
      }
    }
    __builtin_ia32_pause();
  }
  return SC_OK;
}

// Finish a pass: bring the 2*ns split limbs to the host (mailbox, or d_sums after an
// optional RCCL all-reduce), sum them across ranks if a host transport is installed, and
// recombine into ns residues.
int collect_sums(sc_ctx* ctx, int ns, bool across_ranks, bool from_mailbox, u64* out) {
  const size_t count = 2 * (size_t)ns;
  u64* src = nullptr;
  if (from_mailbox) {
    SC_TRY(wait_mailbox(ctx, ctx->mailbox_seq));
    src = ctx->h_mailbox;
    // the kernel exchanged the limbs with the peers itself; the mailbox holds the totals, or why not
    if (across_ranks && ctx->transport == Transport::kPeer) SC_TRY(check_exchange(ctx, "sums"));
  } else if (across_ranks && ctx->transport == Transport::kPeer) {
    // limbs left in d_sums by a small kernel: one wave exchanges them with the peers and publishes
    const sc::PassOut po = next_pass_out(ctx, true, 0x5c5c5c5cu + (unsigned)ns);
    const u64* limbs = ctx->d_sums;
    switch (ns) {
      case 1: hipLaunchKernelGGL((sc::peer_exchange_kernel<1>), dim3(1), dim3(sc::kWave), 0, ctx->stream, limbs, po); break;
      case 3: hipLaunchKernelGGL((sc::peer_exchange_kernel<3>), dim3(1), dim3(sc::kWave), 0, ctx->stream, limbs, po); break;
      case 9: hipLaunchKernelGGL((sc::peer_exchange_kernel<9>), dim3(1), dim3(sc::kWave), 0, ctx->stream, limbs, po); break;
      default: hipLaunchKernelGGL((sc::peer_exchange_kernel<27>), dim3(1), dim3(sc::kWave), 0, ctx->stream, limbs, po); break;
    }
    SC_HIP(ctx, hipGetLastError());
    commit_peer(ctx);
    ctx->mailbox_seq += 1;
    return collect_sums(ctx, ns, true, true, out);
  } else {
    if (across_ranks && ctx->transport == Transport::kRccl) {
      ncclResult_t r = g_rccl.AllReduce(ctx->d_sums, ctx->d_sums, count, ncclUint64, ncclSum, ctx->comm,
                                        ctx->stream);
      if (r != ncclSuccess)
        return fail(ctx, SC_ERR_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    }
    if (ctx->use_mailbox && count <= (size_t)sc::kMailboxSeq) {
      // device -> pinned mailbox by a one-wave kernel; the host spins instead of a
      // memcpy + stream synchronise (about 8 us less per pass)
      const u64 seq = ctx->mailbox_seq + 1;
      hipLaunchKernelGGL(sc::mailbox_copy_kernel, dim3(1), dim3(sc::kWave), 0, ctx->stream, (const u64*)ctx->d_sums,
                         (int)count, ctx->d_mailbox, seq);
      SC_HIP(ctx, hipGetLastError());
      ctx->mailbox_seq = seq;
      SC_TRY(wait_mailbox(ctx, seq));
        src = ctx->h_mailbox;
    } else {
      SC_HIP(ctx, hipMemcpyAsync(ctx->h_sums, ctx->d_sums, count * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
      SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
      src = ctx->h_sums;
    }
  }
  if (across_ranks && ctx->transport == Transport::kHost) {
    if (src != ctx->h_sums) {
      memcpy(ctx->h_sums, src, count * sizeof(u64));
      src = ctx->h_sums;
    }
    if (ctx->host_allreduce(ctx->host_user, src, count) != 0)
      return fail(ctx, SC_ERR_RCCL, "host all-reduce callback failed");
  }
  HostField hf(ctx->fp);
  for (int s = 0; s < ns; ++s) out[s] = hf.recombine(src[2 * s], src[2 * s + 1]);
  return SC_OK;
}

// Peer transport: all-gather `len` words per rank of two device buffers (b may be null) into dst_a / dst_b
// (world * len words each, rank order, ordinary device memory).  The peers write into this rank's arena - at most
// 2^arena_log words per rank and table at a time, longer tables go in chunks - and every chunk is copied out on the
// stream before the next gather can touch that arena: two arenas are used alternately by the parity of a dedicated
// gather counter, and a peer can only be writing gather g + 1 once this rank's kernel of gather g has flagged, which
// sits behind the copy-out of gather g - 1 (the previous user of g + 1's arena) on this stream.
int peer_gather(sc_ctx* ctx, const u64* a, const u64* b, size_t len, u64* dst_a, u64* dst_b) {
  const size_t chunk_cap = (size_t)1 << ctx->arena_log;         // words per rank, table and chunk
  const size_t cap = (size_t)ctx->world << ctx->arena_log;      // words per table of an arena
  for (size_t off = 0; off < len; off += chunk_cap) {
    const size_t n = std::min(chunk_cap, len - off);
    const sc::PassOut out = next_pass_out(ctx, true, 0);
    const size_t arena_off = kInboxRegionWords + kPeerHeaderWords + (size_t)(ctx->gather_count & 1u) * 2 * cap;
    sc::PeerG pg;
    for (int q = 0; q < ctx->world; ++q) pg.arena[q] = ctx->peer_base[q] + arena_off;
    pg.table_stride = cap;
    const int grid = (int)std::min<size_t>(std::max<size_t>((n / 2 + sc::kBlock - 1) / sc::kBlock, 1), 256);
    hipLaunchKernelGGL(sc::peer_gather_kernel, dim3(grid), dim3(sc::kBlock), 0, ctx->stream, a + off, (b ? b : a) + off, n, pg, out);
    SC_HIP(ctx, hipGetLastError());
    commit_peer(ctx);
    ctx->gather_count += 1;
    ctx->mailbox_seq += 1;
    if (grid > 1) ctx->ticket_base += (unsigned)grid;
    SC_TRY(wait_mailbox(ctx, ctx->mailbox_seq));
    if (__atomic_load_n(ctx->h_mailbox + sc::kMailboxErr, __ATOMIC_ACQUIRE) != 0) {
      poison(ctx);
      return fail(ctx, SC_ERR_RCCL, "peer gather: a rank's tables did not arrive within %d ms", ctx->peer_spin_ms);
    }
    // arena rows [rank][n] -> dst rows [rank][len] at column `off`
    const u64* src = ctx->peer_base[ctx->rank] + arena_off;
    SC_HIP(ctx, hipMemcpy2DAsync(dst_a + off, len * sizeof(u64), src, n * sizeof(u64), n * sizeof(u64), (size_t)ctx->world,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    if (b && dst_b)
      SC_HIP(ctx, hipMemcpy2DAsync(dst_b + off, len * sizeof(u64), src + cap, n * sizeof(u64), n * sizeof(u64), (size_t)ctx->world,
                                   hipMemcpyDeviceToDevice, ctx->stream));
  }
  return SC_OK;
}

// All-gather `len` words per rank of a device buffer into a new pool buffer of len*world.
int gather_table(sc_ctx* ctx, const u64* local, size_t len, u64** out_full) {
  u64* full = nullptr;
  SC_TRY(pool_alloc(ctx, len * ctx->world, &full));
  if (ctx->transport == Transport::kPeer) {
    const int rc = peer_gather(ctx, local, nullptr, len, full, nullptr);
    if (rc != SC_OK) {
      pool_release(ctx, full);
      return rc;
    }
  } else if (ctx->transport == Transport::kRccl) {
    ncclResult_t r = g_rccl.AllGather(local, full, len, ncclUint64, ctx->comm, ctx->stream);
    if (r != ncclSuccess) {
      pool_release(ctx, full);
      return fail(ctx, SC_ERR_RCCL, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    }
  } else {
    if (!ctx->host_allgather) {
      pool_release(ctx, full);
      return fail(ctx, SC_ERR_STATE, "gather_table: no host collectives installed on this context");
    }
    std::vector<u64> send(len), recv(len * ctx->world);
    SC_HIP(ctx, hipMemcpyAsync(send.data(), local, len * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->host_allgather(ctx->host_user, send.data(), recv.data(), len) != 0) {
      pool_release(ctx, full);
      return fail(ctx, SC_ERR_RCCL, "host all-gather callback failed");
    }
    SC_HIP(ctx, hipMemcpyAsync(full, recv.data(), recv.size() * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  *out_full = full;
  return SC_OK;
}

// Sum `count` u64 words of a device buffer across ranks, in place (values are 32-bit limbs).
int allreduce_device(sc_ctx* ctx, u64* buf, size_t count) {
  if (ctx->transport == Transport::kPeer) {
    // gather every rank's vector, then sum the rows locally (plain u64 adds of limbs)
    u64* all = nullptr;
    SC_TRY(pool_alloc(ctx, count * ctx->world, &all));
    int rc = peer_gather(ctx, buf, nullptr, count, all, nullptr);
    if (rc == SC_OK) {
      hipLaunchKernelGGL(sc::sum_limb_rows_kernel, dim3(grid_for(ctx, count)), dim3(sc::kBlock), 0, ctx->stream, (const u64*)all, ctx->world,
                         count, buf);
      if (hipGetLastError() != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "sum_limb_rows_kernel launch failed");
    }
    pool_release(ctx, all);   // stream-ordered reuse
    return rc;
  }
  if (ctx->transport == Transport::kRccl) {
    ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclUint64, ncclSum, ctx->comm, ctx->stream);
    if (r != ncclSuccess)
      return fail(ctx, SC_ERR_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return SC_OK;
  }
  if (!ctx->host_allreduce) return fail(ctx, SC_ERR_STATE, "allreduce_device: no host collectives installed on this context");
  std::vector<u64> host(count);
  SC_HIP(ctx, hipMemcpyAsync(host.data(), buf, count * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->host_allreduce(ctx->host_user, host.data(), count) != 0)
    return fail(ctx, SC_ERR_RCCL, "host all-reduce callback failed");
  SC_HIP(ctx, hipMemcpyAsync(buf, host.data(), count * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

// =====================================================================================
// shared helpers for the table API
// =====================================================================================

sc::RVec make_rvec(const u64* r, size_t n) {
  sc::RVec rv;
  memset(&rv, 0, sizeof(rv));
  for (size_t i = 0; i < n && i < 64; ++i) rv.v[i] = r[i];
  return rv;
}

// eq table over `nbits` index bits: out[i] = prod_j (bit_j(i) ? r[j] : 1 - r[j]).
int build_eq_table(sc_ctx* ctx, const u64* r, int nbits, u64** out) {
  if (nbits > 40) return fail(ctx, SC_ERR_ARG, "eq table of 2^%d entries", nbits);
  u64* t = nullptr;
  SC_TRY(pool_alloc(ctx, (size_t)1 << nbits, &t));
  sc::RVec rv = make_rvec(r, (size_t)nbits);
  int grid = grid_for_wide(ctx, (size_t)1 << nbits);
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::eq_table_kernel<F>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream, f,
                                                  rv, 0, nbits, t));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pool_release(ctx, t);
    return fail(ctx, SC_ERR_HIP, "eq_table_kernel: %s", hipGetErrorString(e));
  }
  *out = t;
  return SC_OK;
}

// Launch shape of a row-walking kernel (kernels.hpp, coldot_kernel / gkr_phase1_kernel) over rows x M words: how many
// contiguous KiB a wave reads per row (PW: 4 where the row is long enough, measured +10-16 % over 1), how many blocks
// across the columns (gx) and how many chunks of rows (blockIdx.y) - ONE wave per SIMD in total (four waves per SIMD
// read 10-20 % slower with this access pattern, tools/rowwalk.hip), at least 16 rows per chunk so that the partial
// rows stay a small fraction of the table.
struct RowWalk {
  int pw;
  size_t gx, chunks, rows_per_chunk;
};
RowWalk row_walk_shape(const sc_ctx* ctx, size_t rows, size_t M) {
  const size_t mp = M / 2;
  RowWalk s;
  s.pw = (mp % 256 == 0 && mp >= 1024) ? 4 : (mp % 128 == 0 && mp >= 256) ? 2 : 1;
  const size_t n_spans = (mp + 64 * (size_t)s.pw - 1) / (64 * (size_t)s.pw);
  s.gx = std::min<size_t>((n_spans + 3) / 4, 1024);
  const size_t waves = 4 * (size_t)ctx->num_cus;   // one per SIMD
  size_t chunks = std::min<size_t>((waves + n_spans - 1) / n_spans, std::max<size_t>(rows / 16, 1));
  chunks = std::max<size_t>(1, std::min<size_t>(chunks, 1024));
  s.rows_per_chunk = (rows + chunks - 1) / chunks;
  s.rows_per_chunk = (s.rows_per_chunk + 3) / 4 * 4;   // whole batches of rows in flight
  s.chunks = (rows + s.rows_per_chunk - 1) / s.rows_per_chunk;
  return s;
}

// out[c] = sum_i w[i] * in[i*M + c], i < rows: one streaming pass (plus a small reduce when
// the rows are split over blockIdx.y for parallelism).  M must be even.
int coldot(sc_ctx* ctx, const u64* in, const u64* w, size_t rows, size_t M, u64* out) {
  const RowWalk rw = row_walk_shape(ctx, rows, M);
  const size_t gx = rw.gx, chunks = rw.chunks, rows_per_chunk = rw.rows_per_chunk;
  if (rows_per_chunk > sc::GoldilocksMont::kAccMaxTerms)   // one lazy accumulator sums rows_per_chunk products
    return fail(ctx, SC_ERR_UNSUPPORTED, "coldot: %zu rows per chunk exceed the lazy accumulator's capacity", rows_per_chunk);
  u64* partial = out;
  if (chunks > 1) SC_TRY(pool_alloc(ctx, chunks * M, &partial));
  const int nt = (rows * M) >= ((size_t)1 << ctx->nt_load_log) ? 1 : 0;
  SC_TRY(timer_begin(ctx, SC_KIND_COLDOT, log2_of(rows), 0, log2_of(rows * M), (u64)8 * rows * M + 8 * rows, (u64)8 * M));
#define SC_COLDOT(NT, PW)                                                                                                  \
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::coldot_kernel<F, NT, PW>), dim3((unsigned)gx, (unsigned)chunks),    \
                                                  dim3(sc::kBlock), 0, ctx->stream, f, in, w, rows, rows_per_chunk, M, partial))
  if (nt) {
    if (rw.pw == 4) SC_COLDOT(true, 4);
    else if (rw.pw == 2) SC_COLDOT(true, 2);
    else SC_COLDOT(true, 1);
  } else {
    if (rw.pw == 4) SC_COLDOT(false, 4);
    else if (rw.pw == 2) SC_COLDOT(false, 2);
    else SC_COLDOT(false, 1);
  }
#undef SC_COLDOT
  if (chunks > 1) {
    int grid = grid_for_wide(ctx, M);
    SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::sum_rows_kernel<F>), dim3(grid, 1), dim3(sc::kBlock), 0, ctx->stream, f,
                                                    (const u64*)partial, (const u64*)partial, chunks, M, out, out));
    pool_release(ctx, partial);
  }
  SC_HIP(ctx, hipGetLastError());
  SC_TRY(timer_end(ctx));
  return SC_OK;
}

// Fold k variables of a device table (LE or BE), producing a pool buffer.  `in` is never
// written.  LE: eight or more variables in one streaming segment-dot pass, fewer at up to three
// per pass (read N, write N/8).  BE: one "column dot" pass against the eq table of the k
// leading variables.
int fold_chain(sc_ctx* ctx, const u64* in, size_t len, const u64* r, size_t k, int order, u64** out,
               size_t* out_len) {
  if (k == 0) {
    u64* cp = nullptr;
    SC_TRY(pool_alloc(ctx, len, &cp));
    SC_HIP(ctx, hipMemcpyAsync(cp, in, len * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
    *out = cp;
    *out_len = len;
    return SC_OK;
  }
  if (order == SC_ORDER_BE && k >= 2 && (len >> k) >= 2) {
    // eq index i has variable 0 (r[0]) as its MSB: reverse r for the LE-bit table builder
    std::vector<u64> rr(r, r + k);
    std::reverse(rr.begin(), rr.end());
    u64* eq = nullptr;
    SC_TRY(build_eq_table(ctx, rr.data(), (int)k, &eq));
    u64* res = nullptr;
    int rc = pool_alloc(ctx, len >> k, &res);
    if (rc == SC_OK) rc = coldot(ctx, in, eq, (size_t)1 << k, len >> k, res);
    pool_release(ctx, eq);
    if (rc != SC_OK) {
      pool_release(ctx, res);
      return rc;
    }
    *out = res;
    *out_len = len >> k;
    return SC_OK;
  }
  const u64* cur = in;
  u64* owned = nullptr;  // intermediate we own (never `in`)
  size_t cur_len = len;
  size_t done = 0;
  // a failing step gives back the intermediate of the previous one (and its own output)
#define SC_CHAIN(expr)                \
  do {                                \
    int rc_ = (expr);                 \
    if (rc_ != SC_OK) {               \
      pool_release(ctx, nxt);         \
      pool_release(ctx, owned);       \
      return rc_;                     \
    }                                 \
  } while (0)
  while (done < k) {
    int step;
    u64* nxt = nullptr;
    if (order == SC_ORDER_LE && k - done >= 8) {
      // many variables left: one streaming pass over contiguous segments (kernels.hpp, fix_low_kernel)
      step = (int)std::min<size_t>(17, k - done);
      const size_t nlen = cur_len >> step;
      SC_CHAIN(pool_alloc(ctx, nlen, &nxt));
      const sc::RVec rv = make_rvec(r + done, (size_t)step);
      // one segment per wave: four-wave blocks while there is at most one segment per wave of one block per CU, beyond that
      // one block per CU with all sixteen waves, which draw their segments from a counter in LDS (kernels.hpp, evaluate_kernel)
      int grid = (int)std::min<size_t>((nlen + 3) / 4, (size_t)std::min(ctx->max_blocks, 1024)), threads = sc::kBlock;
      if (nlen > (size_t)4 * std::min(ctx->num_cus, ctx->max_blocks)) {
        threads = ctx->gold ? sc::stream_block<sc::GoldilocksMont>::fix_low : sc::stream_block<sc::MontGeneric>::fix_low;
        grid = std::min(ctx->num_cus, ctx->max_blocks);
      }
      const int nt = cur_len >= ((size_t)1 << ctx->nt_load_log) ? 1 : 0;
      SC_CHAIN(timer_begin(ctx, SC_KIND_FIX_LOW, step, 0, log2_of(cur_len), (u64)8 * cur_len, (u64)8 * nlen));
      if (nt)
        SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::fix_low_kernel<F, true>), dim3(grid), dim3(threads), 0,
                                                        ctx->stream, f, cur, nxt, step, rv, nlen));
      else
        SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::fix_low_kernel<F, false>), dim3(grid), dim3(threads), 0,
                                                        ctx->stream, f, cur, nxt, step, rv, nlen));
      SC_CHAIN(timer_end(ctx));
      cur_len = nlen;
    } else if (order == SC_ORDER_LE && k - done >= 4 && cur_len <= ((size_t)1 << 20) && (cur_len >> std::min<size_t>(5, k - done)) >= 1) {
      // a small table: four or five variables in one launch (kernels.hpp, fold_wide_kernel)
      step = (int)std::min<size_t>(5, k - done);
      const size_t nlen = cur_len >> step;
      SC_CHAIN(pool_alloc(ctx, nlen, &nxt));
      const sc::GridW gw = make_grid_weights(ctx, r + done, step);
      SC_CHAIN(timer_begin(ctx, SC_KIND_FOLD, step, 0, log2_of(cur_len), (u64)8 * cur_len, (u64)8 * nlen));
      const int grid = grid_for(ctx, nlen);
      SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::fold_wide_kernel<F>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream, f, cur, nxt,
                                                      gw, step, nlen));
      SC_CHAIN(timer_end(ctx));
      cur_len = nlen;
    } else if (order == SC_ORDER_LE) {
      step = (int)std::min<size_t>(3, k - done);
      while (step > 1 && (cur_len >> step) < 2) --step;
      size_t nlen = cur_len >> step;
      SC_CHAIN(pool_alloc(ctx, nlen, &nxt));
      const sc::FoldW fw = make_fold_weights(ctx, r + done, step);
      SC_CHAIN(timer_begin(ctx, SC_KIND_FOLD, step, 0, log2_of(cur_len), (u64)8 * cur_len, (u64)8 * nlen));
      if (nlen >= 2) {
        size_t n_units = nlen / 2;
        int grid = grid_for(ctx, n_units), threads = sc::kBlock;
        if (n_units > (size_t)ctx->max_blocks * sc::kBlock * sc::kFoldGrab) {   // several runs per wave: one block per CU with all sixteen waves (kernels.hpp)
          threads = sc::kFoldBlock;
          grid = std::min(ctx->num_cus, ctx->max_blocks);
        }
        const int nt = cur_len >= ((size_t)1 << ctx->nt_load_log) ? 1 : 0;
#define SC_FOLD(KF)                                                                                                  \
  do {                                                                                                               \
    if (sc::fold_kernel_lds_bytes(KF, threads) > 65536)   /* more dynamic LDS than a launch gets by default */        \
      SC_DISPATCH_FIELD(ctx, F, f, {                                                                                 \
        (void)f;                                                                                                     \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nt ? &sc::fold_kernel<F, KF, true> : &sc::fold_kernel<F, KF, false>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc::fold_kernel_lds_bytes(KF, threads)); \
      });                                                                                                            \
    if (nt)                                                                                                          \
      SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::fold_kernel<F, KF, true>), dim3(grid), dim3(threads), sc::fold_kernel_lds_bytes(KF, threads), \
                                                      ctx->stream, f, cur, nxt, fw, n_units));                       \
    else                                                                                                             \
      SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::fold_kernel<F, KF, false>), dim3(grid), dim3(threads), sc::fold_kernel_lds_bytes(KF, threads), \
                                                      ctx->stream, f, cur, nxt, fw, n_units));                       \
  } while (0)
        if (step == 3) SC_FOLD(3);
        else if (step == 2) SC_FOLD(2);
        else SC_FOLD(1);
#undef SC_FOLD
      } else {
        SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::fold_le_small_kernel<F>), dim3(1), dim3(64), 0,
                                                        ctx->stream, f, cur, nxt, r[done], nlen));
      }
      SC_CHAIN(timer_end(ctx));
      cur_len = nlen;
    } else {
      step = 1;
      size_t half = cur_len / 2;
      SC_CHAIN(pool_alloc(ctx, half, &nxt));
      int grid = grid_for(ctx, (half + 1) / 2);
      SC_CHAIN(timer_begin(ctx, SC_KIND_FOLD_BE, 1, 0, log2_of(cur_len), (u64)8 * cur_len, (u64)8 * half));
      SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::fold_be_kernel<F>), dim3(grid), dim3(sc::kBlock), 0,
                                                      ctx->stream, f, cur, nxt, r[done], half));
      SC_CHAIN(timer_end(ctx));
      cur_len = half;
    }
    {
      hipError_t le = hipGetLastError();
      if (le != hipSuccess) {
        poison(ctx);
        pool_release(ctx, nxt);
        pool_release(ctx, owned);
        return fail(ctx, SC_ERR_HIP, "fold launch: %s", hipGetErrorString(le));
      }
    }
    if (owned) pool_release(ctx, owned);  // stream-ordered reuse: single stream per context
    owned = nxt;
    cur = nxt;
    done += step;
  }
#undef SC_CHAIN
  *out = owned;
  *out_len = cur_len;
  return SC_OK;
}

int check_table(const sc_ctx* ctx, const sc_table* t, const char* what) {
  if (!t || !t->d || !is_pow2(t->len)) return fail(ctx, SC_ERR_ARG, "%s: table is null or not 2^k long", what);
  return SC_OK;
}

}  // namespace

// =====================================================================================
// C ABI: field helpers
// =====================================================================================

extern "C" int sc_field_from_modulus(uint64_t p, sc_field* out) {
  if (!out) return SC_ERR_ARG;
  sc::FieldParams fp;
  if (!sc::field_params_from_modulus(p, &fp)) return SC_ERR_ARG;
  out->p = fp.p;
  out->p_inv_neg = fp.p_inv_neg;
  out->r_mod_p = fp.r_mod_p;
  out->r2_mod_p = fp.r2_mod_p;
  return SC_OK;
}

static sc::FieldParams to_params(const sc_field* f) {
  sc::FieldParams fp;
  fp.p = f->p;
  fp.p_inv_neg = f->p_inv_neg;
  fp.r_mod_p = f->r_mod_p;
  fp.r2_mod_p = f->r2_mod_p;
  return fp;
}

extern "C" uint64_t sc_field_to_mont(const sc_field* f, uint64_t canonical) {
  HostField hf(to_params(f));
  return hf.mul(canonical % f->p, f->r2_mod_p);
}
extern "C" uint64_t sc_field_from_mont(const sc_field* f, uint64_t mont) {
  HostField hf(to_params(f));
  return hf.mul(mont, 1);
}

// matrix-multiplication/src/lib.rs:17-60 with x = 0, 1, 2: Lagrange basis polynomials
// scaled by y_i / denominator_i and summed coefficient-wise.
extern "C" int sc_interpolate_quadratic(const sc_field* f, const uint64_t e[3], uint64_t c[3]) {
  if (!f || !e || !c || f->p < 3) return SC_ERR_ARG;
  HostField hf(to_params(f));
  u64 x[3] = {0, hf.one(), hf.add(hf.one(), hf.one())};
  c[0] = c[1] = c[2] = 0;
  for (int i = 0; i < 3; ++i) {
    int j = (i + 1) % 3, k = (i + 2) % 3;
    u64 dinv = hf.inv(hf.mul(hf.sub(x[i], x[j]), hf.sub(x[i], x[k])));
    u64 w = hf.mul(e[i], dinv);
    c[0] = hf.add(c[0], hf.mul(hf.mul(x[j], x[k]), w));
    c[1] = hf.add(c[1], hf.mul(hf.sub(hf.neg(x[j]), x[k]), w));
    c[2] = hf.add(c[2], w);
  }
  return SC_OK;
}

// =====================================================================================
// C ABI: context
// =====================================================================================

extern "C" int sc_ctx_create(const sc_field* f, int device, sc_ctx** out) {
  if (!f || !out) return fail(nullptr, SC_ERR_ARG, "sc_ctx_create: null argument");
  sc_field chk;
  if (sc_field_from_modulus(f->p, &chk) != SC_OK || chk.p_inv_neg != f->p_inv_neg ||
      chk.r_mod_p != f->r_mod_p || chk.r2_mod_p != f->r2_mod_p)
    return fail(nullptr, SC_ERR_ARG, "sc_ctx_create: inconsistent field constants for p=%llu",
                (unsigned long long)f->p);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, SC_ERR_HIP, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(nullptr, SC_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
  sc_ctx* ctx = new (std::nothrow) sc_ctx;
  if (!ctx) return fail(nullptr, SC_ERR_OOM, "host allocation failed");
  ctx->fp = to_params(f);
  ctx->gold = (f->p == sc::GoldilocksMont::P);
  ctx->device = device;
#define SC_CREATE_HIP(call)                                                                \
  do {                                                                                     \
    hipError_t e2_ = (call);                                                               \
    if (e2_ != hipSuccess) {                                                               \
      int rc_ = fail(nullptr, SC_ERR_HIP, "%s: %s", #call, hipGetErrorString(e2_));        \
      sc_ctx_destroy(ctx);                                                                 \
      return rc_;                                                                          \
    }                                                                                      \
  } while (0)
  SC_CREATE_HIP(hipSetDevice(device));
  SC_CREATE_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->num_cus = cus;
    ctx->max_blocks = 3 * ctx->num_cus;
  }
  ctx->partial_rows = 4096;
  SC_CREATE_HIP(hipMalloc(&ctx->d_partials, ctx->partial_rows * 32 * sizeof(u64)));
  SC_CREATE_HIP(hipMalloc(&ctx->d_sums, 512 * sizeof(u64)));   // up to 2 x 243 split limbs of a five-round pass
  SC_CREATE_HIP(hipHostMalloc(&ctx->h_sums, 512 * sizeof(u64), hipHostMallocDefault));
  SC_CREATE_HIP(hipMalloc(&ctx->d_ticket, 64));
  SC_CREATE_HIP(hipMemset(ctx->d_ticket, 0, 64));
  SC_CREATE_HIP(hipHostMalloc(&ctx->h_mailbox, sc::kMailboxWords * sizeof(u64), hipHostMallocMapped | hipHostMallocCoherent));
  memset(ctx->h_mailbox, 0, sc::kMailboxWords * sizeof(u64));
  SC_CREATE_HIP(hipMalloc(&ctx->d_wg_partials, (size_t)kWgMaxBlocks * sc::kGridChunk * sizeof(u64)));
  SC_CREATE_HIP(hipMalloc(&ctx->d_wg_groups, (size_t)(kWgMaxBlocks / sc::kWgGroupBlocks) * sc::kGridChunk * sizeof(u64)));
  SC_CREATE_HIP(hipMalloc(&ctx->d_wg_tickets, 64 * sizeof(unsigned)));
  SC_CREATE_HIP(hipMemset(ctx->d_wg_tickets, 0, 64 * sizeof(unsigned)));
  SC_CREATE_HIP(hipHostGetDevicePointer((void**)&ctx->d_mailbox, ctx->h_mailbox, 0));
  SC_CREATE_HIP(hipDeviceSynchronize());
  for (int i = 0; i < sc_ctx::kTimerRing; ++i) {
    SC_CREATE_HIP(hipEventCreate(&ctx->kt_ev[i][0]));
    SC_CREATE_HIP(hipEventCreate(&ctx->kt_ev[i][1]));
  }
#undef SC_CREATE_HIP
  *out = ctx;
  return SC_OK;
}

extern "C" int sc_ctx_destroy(sc_ctx* ctx) {
  if (!ctx) return SC_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
  for (int q = 0; q < sc::kMaxPeers; ++q)
    if (ctx->peer_ipc_opened[q] && ctx->peer_base[q]) (void)hipIpcCloseMemHandle(ctx->peer_base[q]);
  if (ctx->peer_region) (void)hipFree(ctx->peer_region);
  for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
  for (auto& kv : ctx->pool_live) (void)hipFree(kv.first);
  if (ctx->d_partials) (void)hipFree(ctx->d_partials);
  if (ctx->d_wg_partials) (void)hipFree(ctx->d_wg_partials);
  if (ctx->d_wg_groups) (void)hipFree(ctx->d_wg_groups);
  if (ctx->d_wg_tickets) (void)hipFree(ctx->d_wg_tickets);
  if (ctx->d_sums) (void)hipFree(ctx->d_sums);
  if (ctx->h_sums) (void)hipHostFree(ctx->h_sums);
  if (ctx->h_mailbox) (void)hipHostFree(ctx->h_mailbox);
  if (ctx->d_ticket) (void)hipFree(ctx->d_ticket);
  for (int i = 0; i < sc_ctx::kTimerRing; ++i)
    for (int k = 0; k < 2; ++k)
      if (ctx->kt_ev[i][k]) (void)hipEventDestroy(ctx->kt_ev[i][k]);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return SC_OK;
}

extern "C" const char* sc_last_error(const sc_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" int sc_ctx_set_option(sc_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return SC_ERR_ARG;
  std::string k(key);
  if (k == "vars_per_pass") {
    if (value != 1 && value != 2) return fail(ctx, SC_ERR_ARG, "vars_per_pass must be 1 or 2");
    ctx->vars_per_pass = (int)value;
  } else if (k == "first_pass_vars") {
    if (value < 0 || value > 3) return fail(ctx, SC_ERR_ARG, "first_pass_vars must be 0 (auto), 1, 2 or 3");
    ctx->first_pass_vars = (int)value;
  } else if (k == "grid_pass") {
    ctx->grid_pass = value ? 1 : 0;
  } else if (k == "grid_log") {
    if (value < 0 || value > 26) return fail(ctx, SC_ERR_ARG, "grid_log out of range (0..26)");
    ctx->grid_log = (int)value;
  } else if (k == "grid_max_vars") {
    if (value < 1 || value > sc::kGridMaxVars) return fail(ctx, SC_ERR_ARG, "grid_max_vars must be 1..5");
    ctx->grid_max_vars = (int)value;
  } else if (k == "grid_sharded") {
    ctx->grid_sharded = value ? 1 : 0;
  } else if (k == "grid_blocks") {
    if (value < 0 || value > kWgMaxBlocks) return fail(ctx, SC_ERR_ARG, "grid_blocks must be 0..%d", kWgMaxBlocks);
    ctx->grid_blocks = (int)value;
  } else if (k == "tail_log") {
    if (value < 0 || value > 40) return fail(ctx, SC_ERR_ARG, "tail_log out of range");
    ctx->tail_log = (int)value;
  } else if (k == "max_blocks") {
    if (value < 1 || value > (int64_t)ctx->partial_rows) return fail(ctx, SC_ERR_ARG, "max_blocks out of range");
    ctx->max_blocks = (int)value;
  } else if (k == "time_kernels") {
    ctx->time_kernels = value ? 1 : 0;  // recorded pairs stay in the ring until it fills or the totals are read
  } else if (k == "use_mailbox") {
    ctx->use_mailbox = value ? 1 : 0;
  } else if (k == "arena_log") {
    if (value < 4 || value > 26) return fail(ctx, SC_ERR_ARG, "arena_log must be in [4, 26]");
    if (ctx->peer_region) return fail(ctx, SC_ERR_STATE, "arena_log must be set before sc_ctx_comm_peer_export");
    ctx->arena_log = (int)value;
  } else if (k == "peer_spin_ms") {
    if (value < 1 || value > 600000) return fail(ctx, SC_ERR_ARG, "peer_spin_ms out of range");
    ctx->peer_spin_ms = (int)value;
  } else if (k == "peer_connect_ms") {
    if (value < 1 || value > 3600000) return fail(ctx, SC_ERR_ARG, "peer_connect_ms out of range");
    ctx->peer_connect_ms = (int)value;
  } else if (k == "dbg_delay_ms") {
    if (value < 0 || value > 10000) return fail(ctx, SC_ERR_ARG, "dbg_delay_ms out of range");
    ctx->dbg_delay_ms = (int)value;
  } else if (k == "dbg_skip_tag") {
    ctx->dbg_skip_tag = value ? 1 : 0;
  } else if (k == "nt_load_log") {
    ctx->nt_load_log = (int)value;
  } else if (k == "nt_store_log") {
    ctx->nt_store_log = (int)value;
  } else {
    return fail(ctx, SC_ERR_ARG, "unknown option '%s'", key);
  }
  return SC_OK;
}

extern "C" int sc_ctx_get_option(const sc_ctx* ctx, const char* key, int64_t* value) {
  if (!ctx || !key || !value) return SC_ERR_ARG;
  std::string k(key);
  if (k == "vars_per_pass") *value = ctx->vars_per_pass;
  else if (k == "first_pass_vars") *value = ctx->first_pass_vars;
  else if (k == "grid_pass") *value = ctx->grid_pass;
  else if (k == "grid_log") *value = ctx->grid_log;
  else if (k == "grid_max_vars") *value = ctx->grid_max_vars;
  else if (k == "grid_sharded") *value = ctx->grid_sharded;
  else if (k == "grid_blocks") *value = ctx->grid_blocks;
  else if (k == "tail_log") *value = ctx->tail_log;
  else if (k == "max_blocks") *value = ctx->max_blocks;
  else if (k == "time_kernels") *value = ctx->time_kernels;
  else if (k == "use_mailbox") *value = ctx->use_mailbox;
  else if (k == "arena_log") *value = ctx->arena_log;
  else if (k == "peer_spin_ms") *value = ctx->peer_spin_ms;
  else if (k == "nt_load_log") *value = ctx->nt_load_log;
  else if (k == "nt_store_log") *value = ctx->nt_store_log;
  else if (k == "peer_connect_ms") *value = ctx->peer_connect_ms;
  else if (k == "dbg_delay_ms") *value = ctx->dbg_delay_ms;
  else if (k == "transport") *value = (int64_t)ctx->transport;   // 0 none, 1 RCCL, 2 host callbacks, 3 peer
  else if (k == "comm_nranks") {
    // how many ranks the data plane really spans: RCCL's own count (ncclCommCount) when it is the transport
    int n = ctx->world;
    if (ctx->transport == Transport::kRccl) {
      n = 0;
      if (!g_rccl.CommCount || !ctx->comm || g_rccl.CommCount(ctx->comm, &n) != ncclSuccess)
        return fail(ctx, SC_ERR_RCCL, "ncclCommCount failed");
    }
    *value = n;
  }
  else return fail(ctx, SC_ERR_ARG, "unknown option '%s'", key);
  return SC_OK;
}

extern "C" int sc_ctx_synchronize(sc_ctx* ctx) {
  if (!ctx) return SC_ERR_ARG;
  SC_TRY(set_device(ctx));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

extern "C" void* sc_ctx_stream(const sc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int sc_ctx_kernel_time(sc_ctx* ctx, double out[2], int reset) {
  if (!ctx || !out) return SC_ERR_ARG;
  SC_TRY(set_device(ctx));
  drain_kernel_timers(ctx);
  out[0] = (double)ctx->kt_n;
  out[1] = ctx->kt_ms;
  if (reset) {
    ctx->kt_n = 0;
    ctx->kt_ms = 0.0;
  }
  return SC_OK;
}

extern "C" int sc_ctx_launch_log(sc_ctx* ctx, sc_launch_record* out, size_t cap, size_t* n_out, int reset) {
  if (!ctx || !n_out || (cap && !out)) return SC_ERR_ARG;
  SC_TRY(set_device(ctx));
  drain_kernel_timers(ctx);
  const size_t n = std::min(cap, ctx->launch_log.size());
  for (size_t i = 0; i < n; ++i) out[i] = ctx->launch_log[i];
  *n_out = ctx->launch_log.size();
  if (reset) ctx->launch_log.clear();
  return SC_OK;
}

// =====================================================================================
// C ABI: sharding
// =====================================================================================

extern "C" int sc_comm_unique_id(uint8_t id[128]) {
  std::string why;
  if (!id) return SC_ERR_ARG;
  if (!load_rccl(&why)) return fail(nullptr, SC_ERR_RCCL, "%s", why.c_str());
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return fail(nullptr, SC_ERR_RCCL, "ncclGetUniqueId failed (%d)", (int)r);
  memcpy(id, &u, 128);
  return SC_OK;
}

static int set_world(sc_ctx* ctx, int rank, int world) {
  if (world < 1 || !is_pow2((size_t)world) || rank < 0 || rank >= world)
    return fail(ctx, SC_ERR_ARG, "rank %d / world %d: world must be a power of two", rank, world);
  if (ctx->transport != Transport::kNone) return fail(ctx, SC_ERR_STATE, "communicator already initialised");
  ctx->rank = rank;
  ctx->world = world;
  ctx->log_world = log2_of((size_t)world);
  return SC_OK;
}

extern "C" int sc_ctx_comm_init_rccl(sc_ctx* ctx, const uint8_t id[128], int rank, int world) {
  if (!ctx || !id) return SC_ERR_ARG;
  std::string why;
  if (!load_rccl(&why)) return fail(ctx, SC_ERR_RCCL, "%s", why.c_str());
  SC_TRY(set_device(ctx));
  SC_TRY(set_world(ctx, rank, world));
  ncclUniqueId u;
  memcpy(&u, id, 128);
  ncclResult_t r = g_rccl.CommInitRank(&ctx->comm, world, u, rank);
  if (r != ncclSuccess) {
    ctx->world = 1;
    ctx->rank = 0;
    ctx->log_world = 0;
    return fail(ctx, SC_ERR_RCCL, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  }
  ctx->transport = Transport::kRccl;
  return SC_OK;
}

extern "C" int sc_ctx_comm_init_host(sc_ctx* ctx, int rank, int world, sc_allreduce_fn allreduce,
                                     sc_allgather_fn allgather, void* user) {
  if (!ctx || !allreduce || !allgather) return SC_ERR_ARG;
  SC_TRY(set_world(ctx, rank, world));
  ctx->host_allreduce = allreduce;
  ctx->host_allgather = allgather;
  ctx->host_user = user;
  ctx->transport = Transport::kHost;
  return SC_OK;
}

extern "C" int sc_ctx_comm_peer_export(sc_ctx* ctx, int rank, int world, uint8_t handle[64]) {
  if (!ctx || !handle) return SC_ERR_ARG;
  SC_TRY(set_device(ctx));
  if (world > sc::kMaxPeers) return fail(ctx, SC_ERR_ARG, "the peer transport serves up to %d ranks (one node)", sc::kMaxPeers);
  if (ctx->peer_region) return fail(ctx, SC_ERR_STATE, "sc_ctx_comm_peer_export: already exported");
  SC_TRY(set_world(ctx, rank, world));
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  // inbox + header + two arenas of two tables
  const size_t words = kInboxRegionWords + kPeerHeaderWords + 2 * 2 * ((size_t)world << ctx->arena_log);
  // fine-grained device memory: written by the peers over xGMI while kernels of this rank poll it
  // (no coarse-grained fallback: a peer's stores into ordinary device memory are not guaranteed to be seen by
  // this device's caches across launches; a caller without fine-grained memory uses another transport)
  hipError_t e = hipExtMallocWithFlags((void**)&ctx->peer_region, words * sizeof(u64), hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ctx->peer_region = nullptr;
    ctx->world = 1; ctx->rank = 0; ctx->log_world = 0;
    return fail(ctx, SC_ERR_OOM, "peer region of %zu bytes: %s", words * sizeof(u64), hipGetErrorString(e));
  }
  ctx->peer_region_words = words;
  e = hipMemset(ctx->peer_region, 0, (kInboxRegionWords + kPeerHeaderWords) * sizeof(u64));
  const u64 header[4] = {kPeerMagic, (u64)world, (u64)rank, (u64)ctx->arena_log};
  if (e == hipSuccess) e = hipMemcpy(ctx->peer_region + kInboxRegionWords, header, sizeof(header), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, ctx->peer_region);
  if (e != hipSuccess) {
    (void)hipFree(ctx->peer_region);
    ctx->peer_region = nullptr;
    ctx->world = 1; ctx->rank = 0; ctx->log_world = 0;
    return fail(ctx, SC_ERR_HIP, "exporting the peer region: %s", hipGetErrorString(e));
  }
  memcpy(handle, &h, 64);
  ctx->peer_exported = true;
  return SC_OK;
}

// Every peer region is mapped.  Before the transport is declared usable:
//  1. each mapped region's header must name the same world and arena size and the rank it is mapped as (offsets into a
//     peer's region are computed locally);
//  2. hello: one granule into every peer's inbox; the HOST then polls its own inbox until every peer's hello is there
//     (up to peer_connect_ms).  A peer that said hello has mapped this region, loaded its code object and run a kernel,
//     so from here on the ranks are in step and the in-kernel waits can be short (peer_spin_ms);
//  3. a self-test through the real paths - an in-kernel exchange of known limbs and a gather of known words - so that a
//     node whose fine-grained memory does not behave as the kernels assume (peer stores visible to a polling kernel,
//     gathered data visible to the next launch) fails HERE, with SC_ERR_RCCL, and not inside a proof.
static int peer_finish_connect(sc_ctx* ctx) {
  for (int q = 0; q < ctx->world; ++q) {
    u64 header[4] = {0, 0, 0, 0};
    SC_HIP(ctx, hipMemcpyAsync(header, ctx->peer_base[q] + kInboxRegionWords, sizeof(header), hipMemcpyDeviceToHost, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (header[0] != kPeerMagic || header[1] != (u64)ctx->world || header[2] != (u64)q || header[3] != (u64)ctx->arena_log)
      return fail(ctx, SC_ERR_ARG, "peer %d's region was exported as rank %llu of %llu with arena_log %llu (expected rank %d of %d, arena_log %d)",
                  q, (unsigned long long)header[2], (unsigned long long)header[1], (unsigned long long)header[3], q, ctx->world, ctx->arena_log);
  }
  ctx->transport = Transport::kPeer;
  auto unusable = [ctx](int rc) {
    ctx->transport = Transport::kNone;
    return rc;
  };
  sc::PeerX px;
  for (int q = 0; q < ctx->world; ++q) px.inbox[q] = ctx->peer_base[q];
  px.world = ctx->world;
  px.rank = ctx->rank;
  hipLaunchKernelGGL(sc::peer_hello_kernel, dim3(1), dim3(sc::kWave), 0, ctx->stream, px);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
    return unusable(fail(ctx, SC_ERR_HIP, "peer hello launch failed"));
  {
    std::vector<u64> inbox((size_t)sc::kMaxPeers * sc::kInboxWords);
    const auto t0 = std::chrono::steady_clock::now();
    while (true) {
      if (hipMemcpyAsync(inbox.data(), ctx->peer_region, (size_t)ctx->world * sc::kInboxWords * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        return unusable(fail(ctx, SC_ERR_HIP, "reading the inbox failed"));
      int missing = -1;
      for (int q = 0; q < ctx->world; ++q)
        if (inbox[(size_t)q * sc::kInboxWords + sc::kInboxHello] != (((u64)sc::kHelloTag << 32) | (u64)(q + 1))) missing = q;
      if (missing < 0) break;
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el * 1e3 > ctx->peer_connect_ms)
        return unusable(fail(ctx, SC_ERR_RCCL, "peer connect: rank %d did not say hello within %d ms", missing, ctx->peer_connect_ms));
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  }
  // self-test 1: in-kernel exchange.  limbs (rank + 1, 7) -> totals (world (world + 1) / 2, 7 world)
  {
    const u64 mine[2] = {(u64)ctx->rank + 1, 7};
    SC_HIP(ctx, hipMemcpyAsync(ctx->d_sums, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    u64 got = 0;
    const int keep_delay = ctx->dbg_delay_ms;
    ctx->dbg_delay_ms = 0;
    const int rc = collect_sums(ctx, 1, true, false, &got);
    ctx->dbg_delay_ms = keep_delay;
    if (rc != SC_OK) return unusable(rc == SC_ERR_STATE ? fail(ctx, SC_ERR_RCCL, "peer self-test: digest mismatch") : rc);
    HostField hf(ctx->fp);
    const u64 want = hf.recombine((u64)ctx->world * (ctx->world + 1) / 2, 7 * (u64)ctx->world);
    if (got != want) return unusable(fail(ctx, SC_ERR_RCCL, "peer self-test: the in-kernel exchange summed to the wrong value"));
  }
  // self-test 2: gather (two chunk-sized rounds, so both arenas are exercised)
  {
    const size_t len = 64;
    u64 *src = nullptr, *dst = nullptr;
    SC_TRY(pool_alloc(ctx, len, &src));
    int rc = pool_alloc(ctx, len * ctx->world, &dst);
    std::vector<u64> host(len * ctx->world);
    for (int round = 0; round < 2 && rc == SC_OK; ++round) {
      for (size_t i = 0; i < len; ++i) host[i] = ((u64)(ctx->rank + 1) << 32) | ((u64)round << 16) | i;
      if (hipMemcpyAsync(src, host.data(), len * sizeof(u64), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "self-test upload failed");
      if (rc == SC_OK) rc = peer_gather(ctx, src, nullptr, len, dst, nullptr);
      if (rc == SC_OK && (hipMemcpyAsync(host.data(), dst, len * ctx->world * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                          hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = fail(ctx, SC_ERR_HIP, "self-test download failed");
      for (int q = 0; q < ctx->world && rc == SC_OK; ++q)
        for (size_t i = 0; i < len; ++i)
          if (host[(size_t)q * len + i] != (((u64)(q + 1) << 32) | ((u64)round << 16) | i)) {
            rc = fail(ctx, SC_ERR_RCCL, "peer self-test: gathered word %zu of rank %d is wrong", i, q);
            break;
          }
    }
    pool_release(ctx, src);
    pool_release(ctx, dst);
    if (rc != SC_OK) return unusable(rc);
  }
  return SC_OK;
}

extern "C" int sc_ctx_comm_peer_connect(sc_ctx* ctx, const uint8_t* handles) {
  if (!ctx || !handles) return SC_ERR_ARG;
  if (!ctx->peer_exported || ctx->transport != Transport::kNone)
    return fail(ctx, SC_ERR_STATE, "sc_ctx_comm_peer_connect: call sc_ctx_comm_peer_export first (once)");
  SC_TRY(set_device(ctx));
  for (int q = 0; q < ctx->world; ++q) {
    if (q == ctx->rank) {
      ctx->peer_base[q] = ctx->peer_region;
      continue;
    }
    if (ctx->peer_ipc_opened[q]) continue;   // a retried connect
    hipIpcMemHandle_t h;
    memcpy(&h, handles + 64 * (size_t)q, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(ctx, SC_ERR_HIP, "mapping rank %d's peer region: %s", q, hipGetErrorString(e));
    ctx->peer_base[q] = (u64*)p;
    ctx->peer_ipc_opened[q] = true;
  }
  return peer_finish_connect(ctx);
}

extern "C" int sc_ctx_comm_peer_connect_local(sc_ctx* ctx, sc_ctx* const* peers) {
  if (!ctx || !peers) return SC_ERR_ARG;
  if (!ctx->peer_exported || ctx->transport != Transport::kNone)
    return fail(ctx, SC_ERR_STATE, "sc_ctx_comm_peer_connect_local: call sc_ctx_comm_peer_export first (once)");
  SC_TRY(set_device(ctx));
  for (int q = 0; q < ctx->world; ++q) {
    const sc_ctx* pq = (q == ctx->rank) ? ctx : peers[q];
    if (!pq || !pq->peer_region || pq->world != ctx->world || pq->rank != q || pq->arena_log != ctx->arena_log)
      return fail(ctx, SC_ERR_ARG, "peer %d is not an exported context of the same world", q);
    if (pq->device != ctx->device) {
      // contexts of one process on different GPUs: this device must be allowed to reach the peer's memory
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, ctx->device, pq->device) != hipSuccess || !can)
        return fail(ctx, SC_ERR_UNSUPPORTED, "device %d cannot access device %d's memory", ctx->device, pq->device);
      const hipError_t e = hipDeviceEnablePeerAccess(pq->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return fail(ctx, SC_ERR_HIP, "hipDeviceEnablePeerAccess(%d): %s", pq->device, hipGetErrorString(e));
      (void)hipGetLastError();
    }
    ctx->peer_base[q] = pq->peer_region;
  }
  return peer_finish_connect(ctx);
}

extern "C" int sc_ctx_comm_rank(const sc_ctx* ctx, int* rank, int* world) {
  if (!ctx) return SC_ERR_ARG;
  if (rank) *rank = ctx->rank;
  if (world) *world = ctx->world;
  return SC_OK;
}

// =====================================================================================
// C ABI: tables
// =====================================================================================

extern "C" int sc_table_upload(sc_ctx* ctx, const uint64_t* host, size_t len, sc_table** out) {
  if (!ctx || !host || !out) return SC_ERR_ARG;
  if (!is_pow2(len)) return fail(ctx, SC_ERR_ARG, "sc_table_upload: len %zu is not a power of two", len);
  SC_TRY(set_device(ctx));
  sc_table* t = nullptr;
  SC_TRY(new_table(ctx, len, &t));
  hipError_t e = hipMemcpyAsync(t->d, host, len * sizeof(u64), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    sc_table_free(ctx, t);
    return fail(ctx, SC_ERR_HIP, "upload: %s", hipGetErrorString(e));
  }
  *out = t;
  return SC_OK;
}

extern "C" int sc_table_generate(sc_ctx* ctx, uint64_t seed, uint64_t start, size_t len, sc_table** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  if (!is_pow2(len)) return fail(ctx, SC_ERR_ARG, "sc_table_generate: len %zu is not a power of two", len);
  SC_TRY(set_device(ctx));
  sc_table* t = nullptr;
  SC_TRY(new_table(ctx, len, &t));
  int grid = grid_for_wide(ctx, len);
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::generate_kernel<F>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream,
                                                  f, (u64)seed, (u64)start, len, t->d));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    sc_table_free(ctx, t);
    return fail(ctx, SC_ERR_HIP, "generate: %s", hipGetErrorString(e));
  }
  *out = t;
  return SC_OK;
}

extern "C" int sc_table_clone(sc_ctx* ctx, const sc_table* t, sc_table** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, t, "sc_table_clone"));
  SC_TRY(set_device(ctx));
  sc_table* c = nullptr;
  SC_TRY(new_table(ctx, t->len, &c));
  hipError_t e = hipMemcpyAsync(c->d, t->d, t->len * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    sc_table_free(ctx, c);
    return fail(ctx, SC_ERR_HIP, "clone: %s", hipGetErrorString(e));
  }
  *out = c;
  return SC_OK;
}

extern "C" int sc_table_download(sc_ctx* ctx, const sc_table* t, uint64_t* host, size_t len) {
  if (!ctx || !host) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, t, "sc_table_download"));
  if (len != t->len) return fail(ctx, SC_ERR_ARG, "sc_table_download: len %zu != table len %zu", len, t->len);
  SC_TRY(set_device(ctx));
  SC_HIP(ctx, hipMemcpyAsync(host, t->d, len * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

extern "C" size_t sc_table_len(const sc_table* t) { return t ? t->len : 0; }
extern "C" const uint64_t* sc_table_device_ptr(const sc_table* t) { return t ? t->d : nullptr; }

extern "C" int sc_table_free(sc_ctx* ctx, sc_table* t) {
  if (!t) return SC_OK;
  if (!ctx) return SC_ERR_ARG;
  pool_release(ctx, t->d);
  delete t;
  return SC_OK;
}

extern "C" int sc_table_fix_variables(sc_ctx* ctx, const sc_table* in, const uint64_t* r, size_t k, int order,
                                      sc_table** out) {
  if (!ctx || !out || (k && !r)) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, in, "sc_table_fix_variables"));
  if (order != SC_ORDER_LE && order != SC_ORDER_BE) return fail(ctx, SC_ERR_ARG, "bad order %d", order);
  int nv = log2_of(in->len);
  if (k > (size_t)nv) {
    if (ctx->world > 1)
      return fail(ctx, SC_ERR_UNSUPPORTED, "fix_variables of %zu variables crosses shards (local has %d)", k, nv);
    return fail(ctx, SC_ERR_ARG, "fix_variables: k=%zu > num_vars=%d", k, nv);
  }
  if (ctx->world > 1 && order == SC_ORDER_BE && k > 0)
    return fail(ctx, SC_ERR_UNSUPPORTED, "BE fix_variables pairs entries of different shards");
  SC_TRY(set_device(ctx));
  sc_table* t = new (std::nothrow) sc_table;
  if (!t) return fail(ctx, SC_ERR_OOM, "host allocation failed");
  int rc = fold_chain(ctx, in->d, in->len, r, k, order, &t->d, &t->len);
  if (rc != SC_OK) {
    delete t;
    return rc;
  }
  // no synchronisation: every consumer of the new table (and every release of the old one) is work on the
  // context's stream, behind these launches; a fault of theirs surfaces at the next call that waits
  *out = t;
  return SC_OK;
}

// Local evaluate of a device table at a full LE point, times `w_extra`; leaves the split limbs
// of the result in the mailbox (*from_mailbox) or in ctx->d_sums.
static int evaluate_local(sc_ctx* ctx, const u64* d, size_t len, const u64* pt_le, u64 w_extra, bool across,
                          bool* from_mailbox) {
  const int nv = log2_of(len);
  *from_mailbox = false;
  if (nv < 8) {
    // tiny table: fold chain, then scale into d_sums
    u64* v1 = nullptr;
    size_t out_len = 0;
    SC_TRY(fold_chain(ctx, d, len, pt_le, (size_t)nv, SC_ORDER_LE, &v1, &out_len));
    SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::scale_split_kernel<F>), dim3(1), dim3(64), 0, ctx->stream, f,
                                                    (const u64*)v1, w_extra, ctx->d_sums));
    pool_release(ctx, v1);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
  }
  const int ta = std::min(nv - 7, 10);
  // a wave streams 2^chunk_log consecutive 1 KiB tiles: long contiguous runs per wave read faster
  // (n = 28: 409 us with 16 tiles, 382 with 128), as long as there are chunks for every wave.  (Round 3 swept the grid
  // and the chunk size at 2^24 entries - 256 to 1024 blocks, 8 to 128 tiles per chunk: 36.4-38.8 us whatever the shape,
  // profiles/r03_mle24_sweep.txt; at that size the launch is its ~8 us floor plus 22 us of stream.)
  int chunk_log = std::min({ta, 7, std::max(3, nv - 17)});
  sc::RVec rv = make_rvec(pt_le, (size_t)nv);
  int grid = (int)std::min<size_t>(((((size_t)1 << (nv - 7)) >> chunk_log) + 3) / 4, (size_t)std::min(ctx->max_blocks, 1024));
  if (grid < 1) grid = 1;
  // more chunks than one per wave of a four-wave block per CU: one block per CU with all the waves it holds, which draw
  // chunks of 32 tiles from a counter in LDS (kernels.hpp; n = 28: 351 -> 333 us on one box, chunks of 128 tiles 340)
  int threads = sc::kBlock;
  if ((((size_t)1 << (nv - 7)) >> chunk_log) > (size_t)4 * std::min(ctx->num_cus, ctx->max_blocks)) {
    threads = ctx->gold ? sc::stream_block<sc::GoldilocksMont>::evaluate : sc::stream_block<sc::MontGeneric>::evaluate;
    grid = std::min(ctx->num_cus, ctx->max_blocks);
    chunk_log = std::min(chunk_log, 5);
  }
  const sc::PassOut out = next_pass_out(ctx, across, challenge_digest(pt_le, std::min(nv, 3), 0, nv), from_mailbox);
  const int nt = nv >= ctx->nt_load_log ? 1 : 0;
  SC_TRY(timer_begin(ctx, SC_KIND_EVALUATE, nv, 0, nv, (u64)8 << nv, 0));
  if (nt)
    SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::evaluate_kernel<F, true>), dim3(grid), dim3(threads), 0, ctx->stream,
                                                    f, d, nv, rv, ta, chunk_log, w_extra, out));
  else
    SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::evaluate_kernel<F, false>), dim3(grid), dim3(threads), 0, ctx->stream,
                                                    f, d, nv, rv, ta, chunk_log, w_extra, out));
  SC_TRY(commit_pass_out(ctx, out, grid));
  SC_TRY(timer_end(ctx));
  return SC_OK;
}

extern "C" int sc_table_evaluate(sc_ctx* ctx, const sc_table* t, const uint64_t* r, size_t n, int order,
                                 uint64_t* out) {
  if (!ctx || !out || (n && !r)) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, t, "sc_table_evaluate"));
  if (order != SC_ORDER_LE && order != SC_ORDER_BE) return fail(ctx, SC_ERR_ARG, "bad order %d", order);
  int nl = log2_of(t->len);
  if (n != (size_t)(nl + ctx->log_world))
    return fail(ctx, SC_ERR_ARG, "evaluate: point has %zu entries, table has %d variables", n, nl + ctx->log_world);
  SC_TRY(set_device(ctx));
  const int g = ctx->log_world;
  // local part: LE -> the low nl variables are r[0..nl); BE -> the shard index is the leading
  // variables r[0..g), the local ones are r[g..n).  A BE evaluate is the LE evaluate at the
  // reversed point (same multilinear polynomial, index bits named in the opposite order).
  std::vector<u64> pt;
  if (order == SC_ORDER_LE) pt.assign(r, r + nl);
  else {
    pt.assign(r + g, r + n);
    std::reverse(pt.begin(), pt.end());
  }
  HostField hf(ctx->fp);
  u64 w = hf.one();
  for (int i = 0; i < g; ++i) {
    // rank bit i (LE) is variable nl+i; in BE order rank bit (g-1-i) is variable i
    u64 ri = (order == SC_ORDER_LE) ? r[nl + i] : r[g - 1 - i];
    bool bit = (ctx->rank >> i) & 1;
    w = hf.mul(w, bit ? ri : hf.sub(hf.one(), ri));
  }
  bool mb = false;
  SC_TRY(evaluate_local(ctx, t->d, t->len, pt.data(), w, is_sharded(ctx), &mb));
  u64 res = 0;
  SC_TRY(collect_sums(ctx, 1, is_sharded(ctx), mb, &res));
  *out = res;
  return SC_OK;
}

extern "C" int sc_table_relabel(sc_ctx* ctx, const sc_table* in, size_t a, size_t b, size_t k, sc_table** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, in, "sc_table_relabel"));
  if (ctx->world > 1) return fail(ctx, SC_ERR_UNSUPPORTED, "relabel on a sharded table");
  int nv = log2_of(in->len);
  if (a > b) std::swap(a, b);
  if (a + k > b || b + k > (size_t)nv) return fail(ctx, SC_ERR_ARG, "relabel(%zu,%zu,%zu) on %d variables", a, b, k, nv);
  SC_TRY(set_device(ctx));
  sc_table* t = nullptr;
  SC_TRY(new_table(ctx, in->len, &t));
  int grid = grid_for_wide(ctx, in->len);
  hipLaunchKernelGGL(sc::relabel_kernel, dim3(grid), dim3(sc::kBlock), 0, ctx->stream, (const u64*)in->d, t->d, in->len,
                     (unsigned)a, (unsigned)b, (unsigned)k);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    sc_table_free(ctx, t);
    return fail(ctx, SC_ERR_HIP, "relabel: %s", hipGetErrorString(e));
  }
  *out = t;
  return SC_OK;
}

// =====================================================================================
// C ABI: product of two tables
// =====================================================================================

static int check_pair(const sc_ctx* ctx, const sc_table* a, const sc_table* b, const char* what) {
  SC_TRY(check_table(ctx, a, what));
  SC_TRY(check_table(ctx, b, what));
  if (a->len != b->len) return fail(ctx, SC_ERR_ARG, "%s: table lengths differ (%zu vs %zu)", what, a->len, b->len);
  return SC_OK;
}

extern "C" int sc_matmul_g_new(sc_ctx* ctx, const sc_table* A, const sc_table* B, size_t n, const uint64_t* point,
                               sc_table** a_out, sc_table** b_out) {
  if (!ctx || !point || !a_out || !b_out) return SC_ERR_ARG;
  SC_TRY(check_pair(ctx, A, B, "sc_matmul_g_new"));
  if (is_sharded(ctx)) {
    // Row-block shards (top log2(world) bits of the row index = rank).  f_b = B~(z, r2) has
    // z = row, so the local rows ARE this rank's shard of f_b: no exchange.  f_a = A~(r1, z)
    // has z = column and sums over rows: every rank holds a partial vector over all columns;
    // they are summed as 32-bit limbs (one all-reduce of 2*2^n words) and each rank keeps
    // its own column range.  ("G::new".)
    SC_TRY(set_device(ctx));
    const int g = ctx->log_world;
    if (n < (size_t)g) return fail(ctx, SC_ERR_ARG, "sc_matmul_g_new: 2^%zu rows cannot be split over %d ranks", n, ctx->world);
    const size_t side = (size_t)1 << n, rows_local = side >> g;
    if (A->len != rows_local * side) return fail(ctx, SC_ERR_ARG, "sc_matmul_g_new: shard must hold 2^(2n)/world entries");
    if (side < 2) return fail(ctx, SC_ERR_UNSUPPORTED, "sc_matmul_g_new: sharded 1x1 matrices");
    u64 *eq = nullptr, *partial = nullptr, *limbs = nullptr;
    sc_table *ta = nullptr, *tb = nullptr;
    int rc = build_eq_table(ctx, point, (int)n, &eq);
    if (rc == SC_OK) rc = pool_alloc(ctx, side, &partial);
    if (rc == SC_OK) rc = pool_alloc(ctx, 2 * side, &limbs);
    if (rc == SC_OK) rc = coldot(ctx, A->d, eq + (size_t)ctx->rank * rows_local, rows_local, side, partial);
    if (rc == SC_OK) {
      hipLaunchKernelGGL(sc::split_limbs_kernel, dim3(grid_for(ctx, side)), dim3(sc::kBlock), 0, ctx->stream,
                         (const u64*)partial, side, limbs);
      rc = allreduce_device(ctx, limbs, 2 * side);
    }
    if (rc == SC_OK) rc = new_table(ctx, rows_local, &ta);
    if (rc == SC_OK) {
      const u64* mine = limbs + 2 * (size_t)ctx->rank * rows_local;
      SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::recombine_limbs_kernel<F>), dim3(grid_for(ctx, rows_local)),
                                                      dim3(sc::kBlock), 0, ctx->stream, f, mine, rows_local, ta->d));
      if (hipGetLastError() != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "recombine_limbs_kernel launch failed");
    }
    if (rc == SC_OK) {
      tb = new (std::nothrow) sc_table;
      if (!tb) rc = fail(ctx, SC_ERR_OOM, "host allocation failed");
      else rc = fold_chain(ctx, B->d, B->len, point + n, n, SC_ORDER_LE, &tb->d, &tb->len);
    }
    if (rc == SC_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "g_new: sync failed");
    pool_release(ctx, eq);
    pool_release(ctx, partial);
    pool_release(ctx, limbs);
    if (rc != SC_OK) {
      sc_table_free(ctx, ta);
      if (tb) { pool_release(ctx, tb->d); delete tb; }
      return rc;
    }
    *a_out = ta;
    *b_out = tb;
    return SC_OK;
  }
  if (A->len != ((size_t)1 << (2 * n))) return fail(ctx, SC_ERR_ARG, "sc_matmul_g_new: tables must have 2^(2n) entries");
  // matrix-multiplication/src/lib.rs:81-86.  relabel(0,n,n) + fix_variables(point[..n]) folds
  // the ROW index of A with LE weights: f_a[col] = sum_row eq(point[..n])[row] * A[row][col]
  // - one "column dot" pass over A, no transposed copy.  f_b folds the column index of B.
  SC_TRY(set_device(ctx));
  const size_t side = (size_t)1 << n;
  if (n == 0) {
    SC_TRY(sc_table_clone(ctx, A, a_out));
    int rc0 = sc_table_clone(ctx, B, b_out);
    if (rc0 != SC_OK) { sc_table_free(ctx, *a_out); *a_out = nullptr; }
    return rc0;
  }
  sc_table* ta = nullptr;
  if (side >= 2) {
    u64* eq = nullptr;
    SC_TRY(build_eq_table(ctx, point, (int)n, &eq));
    int rc1 = new_table(ctx, side, &ta);
    if (rc1 == SC_OK) rc1 = coldot(ctx, A->d, eq, side, side, ta->d);
    pool_release(ctx, eq);
    if (rc1 != SC_OK) {
      sc_table_free(ctx, ta);
      return rc1;
    }
  }
  int rc = sc_table_fix_variables(ctx, B, point + n, n, SC_ORDER_LE, b_out);
  if (rc != SC_OK) {
    sc_table_free(ctx, ta);
    return rc;
  }
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *a_out = ta;
  return SC_OK;
}

extern "C" int sc_prod2_to_evaluations(sc_ctx* ctx, const sc_table* a, const sc_table* b, sc_table** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  SC_TRY(check_pair(ctx, a, b, "sc_prod2_to_evaluations"));
  SC_TRY(set_device(ctx));
  sc_table* t = nullptr;
  SC_TRY(new_table(ctx, a->len, &t));
  int grid = grid_for(ctx, a->len);
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::mul_kernel<F>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream, f,
                                                  (const u64*)a->d, (const u64*)b->d, t->d, a->len));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    sc_table_free(ctx, t);
    return fail(ctx, SC_ERR_HIP, "to_evaluations: %s", hipGetErrorString(e));
  }
  *out = t;
  return SC_OK;
}

// Round sums of (a, b) as they are (no fold); handles the degenerate 1-entry tables.
static int round_sums_now(sc_ctx* ctx, const u64* a, const u64* b, int log_len, bool across, u64 e[3]) {
  if (log_len < 1) return fail(ctx, SC_ERR_ARG, "round sums need at least one variable");
  bool mb = false;
  SC_TRY(launch_pass(ctx, 0, 1, a, b, nullptr, nullptr, nullptr, log_len, across, &mb));
  SC_TRY(collect_sums(ctx, 3, across, mb, e));
  HostField hf(ctx->fp);
  e[2] = eval2_from_inf(hf, e[0], e[1], e[2]);  // the kernel sums H(0), H(1), H(inf)
  return SC_OK;
}

extern "C" int sc_prod2_round_sums(sc_ctx* ctx, const sc_table* a, const sc_table* b, uint64_t out_e[3]) {
  if (!ctx || !out_e) return SC_ERR_ARG;
  SC_TRY(check_pair(ctx, a, b, "sc_prod2_round_sums"));
  SC_TRY(set_device(ctx));
  if (a->len < 2) return fail(ctx, SC_ERR_ARG, "sc_prod2_round_sums: tables have no variable left");
  return round_sums_now(ctx, a->d, b->d, log2_of(a->len), is_sharded(ctx), out_e);
}

extern "C" int sc_prod2_sum(sc_ctx* ctx, const sc_table* a, const sc_table* b, uint64_t* out_c1) {
  if (!ctx || !out_c1) return SC_ERR_ARG;
  SC_TRY(check_pair(ctx, a, b, "sc_prod2_sum"));
  SC_TRY(set_device(ctx));
  HostField hf(ctx->fp);
  if (a->len == 1) {
    // zero variables on this rank: the (partial) sum is the single product
    u64* prod = nullptr;
    SC_TRY(pool_alloc(ctx, 1, &prod));
    SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::mul_kernel<F>), dim3(1), dim3(sc::kBlock), 0, ctx->stream, f,
                                                    (const u64*)a->d, (const u64*)b->d, prod, (size_t)1));
    SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::scale_split_kernel<F>), dim3(1), dim3(64), 0, ctx->stream, f,
                                                    (const u64*)prod, hf.one(), ctx->d_sums));
    pool_release(ctx, prod);
    SC_HIP(ctx, hipGetLastError());
    return collect_sums(ctx, 1, is_sharded(ctx), false, out_c1);
  }
  u64 e[3];
  SC_TRY(round_sums_now(ctx, a->d, b->d, log2_of(a->len), is_sharded(ctx), e));
  *out_c1 = hf.add(e[0], e[1]);  // c_1 = H(0) + H(1)
  return SC_OK;
}

extern "C" int sc_prod2_fold_and_sums(sc_ctx* ctx, const sc_table* a, const sc_table* b, const uint64_t r[1],
                                      sc_table** a_out, sc_table** b_out, uint64_t out_e[3]) {
  if (!ctx || !r || !a_out || !b_out || !out_e) return SC_ERR_ARG;
  SC_TRY(check_pair(ctx, a, b, "sc_prod2_fold_and_sums"));
  SC_TRY(set_device(ctx));
  int nv = log2_of(a->len);
  if (nv < 2) return fail(ctx, SC_ERR_ARG, "sc_prod2_fold_and_sums: need >= 2 variables (fold one, sum over one)");
  sc_table *ta = nullptr, *tb = nullptr;
  SC_TRY(new_table(ctx, a->len / 2, &ta));
  int rc = new_table(ctx, a->len / 2, &tb);
  bool mb = false;
  if (rc == SC_OK) rc = launch_pass(ctx, 1, 1, a->d, b->d, ta->d, tb->d, r, nv, is_sharded(ctx), &mb);
  if (rc == SC_OK) rc = collect_sums(ctx, 3, is_sharded(ctx), mb, out_e);
  if (rc == SC_OK) {
    HostField hf(ctx->fp);
    out_e[2] = eval2_from_inf(hf, out_e[0], out_e[1], out_e[2]);
  }
  if (rc != SC_OK) {
    sc_table_free(ctx, ta);
    sc_table_free(ctx, tb);
    return rc;
  }
  *a_out = ta;
  *b_out = tb;
  return SC_OK;
}

extern "C" int sc_prod2_evaluate(sc_ctx* ctx, const sc_table* a, const sc_table* b, const uint64_t* point, size_t n,
                                 uint64_t* out) {
  if (!ctx || !out) return SC_ERR_ARG;
  SC_TRY(check_pair(ctx, a, b, "sc_prod2_evaluate"));
  u64 va = 0, vb = 0;
  SC_TRY(sc_table_evaluate(ctx, a, point, n, SC_ORDER_LE, &va));
  SC_TRY(sc_table_evaluate(ctx, b, point, n, SC_ORDER_LE, &vb));
  HostField hf(ctx->fp);
  *out = hf.mul(va, vb);
  return SC_OK;
}

// =====================================================================================
// C ABI: the prover
// =====================================================================================
//
// Schedule.  The reference folds one variable and re-sums every round
// (sum-check-protocol/src/lib.rs:105-112).  Here one device pass serves up to two rounds:
// it folds the (<= 3) challenges received since the previous pass and accumulates the 3x3
// grid S[u][v] of the folded tables (3x3x3 for the first pass, which has nothing to fold).  Round j is H(u) = S[u][0] + S[u][1]; round j+1,
// once r_j is known, is H'(v) = sum_u L_u(r_j) S[u][v] with the Lagrange basis on {0,1,2}
// - exact field identities, so every round polynomial equals the reference's bit for bit.
struct sc_prover {
  sc_ctx* ctx = nullptr;
  const u64* a0 = nullptr;  // caller's tables (borrowed, never written)
  const u64* b0 = nullptr;
  const u64* cur_a = nullptr;
  const u64* cur_b = nullptr;
  u64* own_a = nullptr;  // pool buffers backing cur_* when they are not the caller's
  u64* own_b = nullptr;
  int cur_log = 0;       // log2 length of cur_* on this rank
  bool sharded = false;  // still one shard per rank (sums need the all-reduce)
  size_t num_vars = 0;   // global
  size_t next_round = 0;
  std::vector<u64> pending;  // challenges not yet folded into cur_*
  // cache of the last pass
  int cache_ks = 0;
  size_t cache_round = 0;
  // the grid S with its leading `g_known` axes collapsed at the challenges received since the pass (prover_answer's
  // working copy: each round collapses one more axis instead of starting from S again); g_known < 0: not built
  mutable u64 G[sc::kGridMaxCells];
  mutable int g_known = -1;
  u64 S[sc::kGridMaxCells];
  u64 c1 = 0;
};

namespace {

// ---- the schedule: PURE host logic (no device, no context), shared by the engine and by sc_plan_proof ---------------

// measured (two vs three rounds from the first pass, ms): n = 16 0.109 / 0.116, 18 0.128 / 0.125, 20 0.167 / 0.152,
// 22 0.204 / 0.198, 24 0.307 / 0.294, 26 0.75 / 0.68, 28 2.36 / 2.13
constexpr int kFirstPass3Log = 18;

// the options the schedule depends on + the communicator's shape
struct PlanOpts {
  int vars_per_pass, first_pass_vars, grid_pass, grid_log, grid_max_vars, grid_sharded, tail_log, use_mailbox;
  int transport;   // sc_plan_options.h numbering = Transport: 0 none, 1 RCCL, 2 host callbacks, 3 peer
  int log_world;
};
PlanOpts plan_opts_of(const sc_ctx* ctx) {
  return PlanOpts{ctx->vars_per_pass, ctx->first_pass_vars, ctx->grid_pass, ctx->grid_log, ctx->grid_max_vars, ctx->grid_sharded,
                  ctx->tail_log, ctx->use_mailbox, (int)ctx->transport, ctx->log_world};
}

// rounds a pass_kernel launch at round j serves (the schedule of DESIGN.md section 4)
int pass_rounds(const PlanOpts& o, size_t num_vars, size_t j, int kf, int cur_log) {
  const size_t remaining = num_vars - j;  // variables left including round j's
  int ks = (o.vars_per_pass == 2 && remaining >= 2) ? 2 : 1;
  if (j == 0 && kf == 0) {
    const int first = o.first_pass_vars ? o.first_pass_vars : (cur_log >= kFirstPass3Log ? 3 : 2);
    if (o.vars_per_pass == 2 && remaining >= 3 && first == 3) ks = 3;
    if (first < ks) ks = first;
  }
  return ks;
}

// Rounds a grid pass (kernels.hpp, wgrid_pass_kernel) serves when `vars` variables are left, i.e. the folded table
// has 2^vars entries: as few passes as grid_max_vars allows, the rounds shared evenly among them.
int grid_passes_needed(const PlanOpts& o, int vars) { return (vars + o.grid_max_vars - 1) / o.grid_max_vars; }
int grid_rounds(const PlanOpts& o, int vars) {
  const int need = grid_passes_needed(o, vars);
  return std::max(1, std::min((vars + need - 1) / need, vars));
}
// does the pass at round j go to wgrid_pass_kernel?  The folded table (on a sharded prover: the folded shard) must be
// small enough and keep at least one variable.  (The one-round-per-pass mode and an explicit first_pass_vars are
// requests for those schedules.)
bool takes_grid_pass(const PlanOpts& o, bool sharded, int cur_log, int kf, size_t j) {
  if (!o.grid_pass || !o.use_mailbox || o.vars_per_pass != 2) return false;
  if (sharded && !o.grid_sharded) return false;
  if (j == 0 && kf == 0 && o.first_pass_vars != 0) return false;
  return cur_log - kf >= 1 && cur_log - kf <= o.grid_log;
}

// What the prover does at round j when its cache does not cover the round: the state is (kf pending challenges, local
// tables of 2^cur_log entries, still sharded or not).
//   Sharded: pairs (2b, 2b+1) stay shard-local while the local table still has the kf+ks variables this pass
//   touches.  A shard that can go on with five-round passes (exchange inside the kernel / one collective per pass)
//   is gathered only when it is down to its pending challenges: 2^kf <= 32 entries.  Otherwise (grid_sharded 0)
//   it is gathered at tail_log: below that the latency of a collective per two-round pass costs more than finishing
//   redundantly on every rank.  On the peer transport, when nothing but the pending challenges is left, the rounds of
//   the rank bits are ONE small launch: fold, exchange the single entries (the gather), cells.
struct PassPlan {
  enum Kind { kPass = SC_PLAN_PASS, kGridPass = SC_PLAN_GRID_PASS, kRankPass = SC_PLAN_RANK_PASS } kind;
  bool gather_first;   // all-gather both tables (cur_log += log_world, unsharded from here on) before the launch
  int ks;              // rounds the launch serves
  const char* error;   // non-null: the state cannot be continued (a caller bug or an option combination without a kernel)
};
PassPlan plan_pass(const PlanOpts& o, size_t num_vars, size_t j, int kf, int cur_log, bool sharded) {
  PassPlan p{PassPlan::kPass, false, 0, nullptr};
  if (kf > sc::kGridMaxVars) {
    p.error = "more unfolded challenges than a pass can fold";
    return p;
  }
  int ks = pass_rounds(o, num_vars, j, kf, cur_log);
  const bool shard_grid = sharded && takes_grid_pass(o, true, cur_log, kf, j);
  if (sharded && !shard_grid && o.transport == (int)Transport::kPeer && cur_log == kf && o.log_world >= 1 && o.log_world <= 3 &&
      num_vars - j == (size_t)o.log_world && takes_grid_pass(o, true, cur_log + 1, kf, j)) {
    p.kind = PassPlan::kRankPass;
    p.ks = o.log_world;
    return p;
  }
  if (sharded && !shard_grid && (cur_log < kf + ks || cur_log <= o.tail_log)) {
    p.gather_first = true;
    cur_log += o.log_world;
    sharded = false;
  }
  // the smallest tables: up to five rounds per pass (after a gather the table is whole: decided on that)
  const bool by_grid = takes_grid_pass(o, sharded, cur_log, kf, j);
  if (by_grid) ks = grid_rounds(o, cur_log - kf);   // sharded: planned on the shard's own variables
  if (kf > 3 && !by_grid) p.error = "unfolded challenges and no grid pass to fold them";
  else if (cur_log < kf + ks) p.error = "the table has fewer variables than the pass needs";
  p.kind = by_grid ? PassPlan::kGridPass : PassPlan::kPass;
  p.ks = ks;
  return p;
}

// all-gather both tables of a sharded prover into pool buffers (any transport)
int gather_pair(sc_ctx* ctx, const u64* a, const u64* b, size_t len, u64** fa, u64** fb) {
  if (ctx->transport == Transport::kPeer) {
    u64 *ga = nullptr, *gb = nullptr;
    SC_TRY(pool_alloc(ctx, len * ctx->world, &ga));
    int rc = pool_alloc(ctx, len * ctx->world, &gb);
    if (rc == SC_OK) rc = peer_gather(ctx, a, b, len, ga, gb);
    if (rc != SC_OK) {
      pool_release(ctx, ga);
      pool_release(ctx, gb);
      return rc;
    }
    *fa = ga;
    *fb = gb;
    return SC_OK;
  }
  SC_TRY(gather_table(ctx, a, len, fa));
  const int rc = gather_table(ctx, b, len, fb);
  if (rc != SC_OK) {
    pool_release(ctx, *fa);
    *fa = nullptr;
  }
  return rc;
}

int prover_pass(sc_prover* pr, size_t j) {
  sc_ctx* ctx = pr->ctx;
  const int kf = (int)pr->pending.size();
  const PassPlan plan = plan_pass(plan_opts_of(ctx), pr->num_vars, j, kf, pr->cur_log, pr->sharded);
  if (plan.error) return fail(ctx, SC_ERR_STATE, "prover (round %zu, %d pending, 2^%d entries): %s", j, kf, pr->cur_log, plan.error);
  const int ks = plan.ks;
  if (plan.kind == PassPlan::kRankPass) {
    u64 *na = nullptr, *nb = nullptr;
    SC_TRY(pool_alloc(ctx, (size_t)ctx->world, &na));
    int rc = pool_alloc(ctx, (size_t)ctx->world, &nb);
    if (rc == SC_OK) rc = rank_pass(ctx, kf, pr->cur_a, pr->cur_b, na, nb, pr->pending.data(), pr->S);
    if (rc != SC_OK) {
      pool_release(ctx, na);
      pool_release(ctx, nb);
      return rc;
    }
    pool_release(ctx, pr->own_a);
    pool_release(ctx, pr->own_b);
    pr->own_a = na;
    pr->own_b = nb;
    pr->cur_a = na;
    pr->cur_b = nb;
    pr->cur_log = ctx->log_world;
    pr->pending.clear();
    pr->sharded = false;
    pr->cache_ks = ctx->log_world;
    pr->cache_round = j;
    pr->g_known = -1;
    return SC_OK;
  }
  if (plan.gather_first) {
    u64 *fa = nullptr, *fb = nullptr;
    SC_TRY(gather_pair(ctx, pr->cur_a, pr->cur_b, (size_t)1 << pr->cur_log, &fa, &fb));
    pool_release(ctx, pr->own_a);
    pool_release(ctx, pr->own_b);
    pr->own_a = fa;
    pr->own_b = fb;
    pr->cur_a = fa;
    pr->cur_b = fb;
    pr->cur_log += ctx->log_world;
    pr->sharded = false;
  }
  const bool by_grid = plan.kind == PassPlan::kGridPass;

  u64 *na = nullptr, *nb = nullptr;
  if (kf > 0) {
    size_t out_len = (size_t)1 << (pr->cur_log - kf);
    SC_TRY(pool_alloc(ctx, out_len, &na));
    int rc = pool_alloc(ctx, out_len, &nb);
    if (rc != SC_OK) {
      pool_release(ctx, na);
      return rc;
    }
  }
  bool mb = false;
  int rc;
  if (by_grid) {
    rc = launch_grid_pass(ctx, kf, ks, pr->cur_a, pr->cur_b, na, nb, pr->pending.data(), pr->cur_log, pr->sharded);
    if (rc == SC_OK) rc = collect_grid(ctx, ks, pr->sharded, pr->S);
  } else {
    rc = launch_pass(ctx, kf, ks, pr->cur_a, pr->cur_b, na, nb, pr->pending.data(), pr->cur_log, pr->sharded, &mb);
    if (rc == SC_OK) rc = collect_sums(ctx, ks == 1 ? 3 : ks == 2 ? 9 : 27, pr->sharded, mb, pr->S);
  }
  if (rc != SC_OK) {
    pool_release(ctx, na);
    pool_release(ctx, nb);
    return rc;
  }
  if (kf > 0) {
    pool_release(ctx, pr->own_a);
    pool_release(ctx, pr->own_b);
    pr->own_a = na;
    pr->own_b = nb;
    pr->cur_a = na;
    pr->cur_b = nb;
    pr->cur_log -= kf;
    pr->pending.clear();
  }
  pr->cache_ks = ks;
  pr->cache_round = j;
  pr->g_known = -1;
  return SC_OK;
}

// answer round j from the cache (which must cover it).  The cache is the grid of the last pass
// in the {0,1,inf} basis, first variable on the slowest axis: S[(3u + v)*3 + w] for a
// three-round pass.  For each fixed value of the other axes a line along the leading axis is a
// quadratic q(X) = s0 + X (s1 - s0 - sinf) + X^2 sinf; the challenges received since the pass
// collapse the leading axes one by one, the round's variable is the next axis, and the axes
// after it are summed over {0,1}.
void prover_answer(const sc_prover* pr, size_t j, u64 e[3]) {
  HostField hf(pr->ctx->fp);
  int total = 1;
  for (int i = 0; i < pr->cache_ks; ++i) total *= 3;
  const int known = (int)(j - pr->cache_round);  // == pr->pending.size()
  u64* g = pr->G;
  if (pr->g_known < 0 || pr->g_known > known) {
    for (int i = 0; i < total; ++i) g[i] = pr->S[i];
    pr->g_known = 0;
  }
  int cells = total;
  for (int i = 0; i < pr->g_known; ++i) cells /= 3;
  for (int i = pr->g_known; i < known; ++i) {
    const u64 r = pr->pending[i];
    const u64 r2 = hf.mul(r, r);
    cells /= 3;
    for (int c = 0; c < cells; ++c) {
      const u64 s0 = g[c], s1 = g[cells + c], si = g[2 * cells + c];
      const u64 lin = hf.sub(hf.sub(s1, s0), si);
      g[c] = hf.add(hf.add(s0, hf.mul(r, lin)), hf.mul(r2, si));
    }
  }
  pr->g_known = known;
  const int rest = cells / 3;  // cells per value of the round's variable
  u64 h[3];
  for (int x = 0; x < 3; ++x) {
    u64 t = 0;
    for (int c = 0; c < rest; ++c) {
      // keep the cells whose remaining axes are all in {0,1}
      bool boolean = true;
      for (int d = c; d > 0; d /= 3) boolean = boolean && (d % 3 != 2);
      if (boolean) t = hf.add(t, g[x * rest + c]);
    }
    h[x] = t;
  }
  e[0] = h[0];
  e[1] = h[1];
  e[2] = eval2_from_inf(hf, h[0], h[1], h[2]);
}

bool cache_covers(const sc_prover* pr, size_t j) {
  if (pr->cache_ks == 0 || j < pr->cache_round) return false;
  const size_t known = j - pr->cache_round;
  return known < (size_t)pr->cache_ks && pr->pending.size() == known;
}

}  // namespace

// `replicated`: on a sharded context, treat a and b as whole tables held identically by every rank (the
// small tables of the GKR phases): no exchange, every rank proves the same thing
static int prover_create_impl(sc_ctx* ctx, const sc_table* a, const sc_table* b, bool replicated, sc_prover** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  SC_TRY(check_pair(ctx, a, b, "sc_prover_create"));
  SC_TRY(set_device(ctx));
  sc_prover* pr = new (std::nothrow) sc_prover;
  if (!pr) return fail(ctx, SC_ERR_OOM, "host allocation failed");
  pr->ctx = ctx;
  pr->a0 = pr->cur_a = a->d;
  pr->b0 = pr->cur_b = b->d;
  pr->cur_log = log2_of(a->len);
  pr->sharded = is_sharded(ctx) && !replicated;
  pr->num_vars = (size_t)pr->cur_log + (pr->sharded ? ctx->log_world : 0);
  HostField hf(ctx->fp);
  if (pr->num_vars == 0) {
    // no variable: c_1 is the single product, no rounds follow
    int rc = sc_prod2_sum(ctx, a, b, &pr->c1);
    if (rc != SC_OK) {
      delete pr;
      return rc;
    }
    *out = pr;
    return SC_OK;
  }
  // Prover::new's claim rides on round 0's pass: c_1 = H(0) + H(1)
  int rc = prover_pass(pr, 0);
  if (rc != SC_OK) {
    sc_prover_destroy(pr);
    return rc;
  }
  u64 e[3];
  prover_answer(pr, 0, e);
  pr->c1 = hf.add(e[0], e[1]);
  *out = pr;
  return SC_OK;
}

extern "C" void sc_plan_options_default(sc_plan_options* o) {
  if (!o) return;
  const sc_ctx d{};   // the defaults are the context's member initialisers (no device is touched)
  o->vars_per_pass = d.vars_per_pass;
  o->first_pass_vars = d.first_pass_vars;
  o->grid_pass = d.grid_pass;
  o->grid_log = d.grid_log;
  o->grid_max_vars = d.grid_max_vars;
  o->grid_sharded = d.grid_sharded;
  o->tail_log = d.tail_log;
  o->use_mailbox = d.use_mailbox;
}

// The launches of a whole proof, by the planner the engine itself runs (plan_pass): a dry run of sc_prove's state machine.
extern "C" int sc_plan_proof(const sc_plan_options* opt, size_t num_vars, int world, int transport, sc_plan_step* out, size_t cap,
                             size_t* n_out) {
  if (!opt || !n_out || (cap && !out)) return SC_ERR_ARG;
  if (world < 1 || !is_pow2((size_t)world) || transport < 0 || transport > 3 || (world > 1 && transport == 0)) return SC_ERR_ARG;
  if (opt->vars_per_pass < 1 || opt->vars_per_pass > 2 || opt->first_pass_vars < 0 || opt->first_pass_vars > 3 || opt->grid_log < 0 ||
      opt->grid_log > 26 || opt->grid_max_vars < 1 || opt->grid_max_vars > sc::kGridMaxVars || opt->tail_log < 0)
    return SC_ERR_ARG;
  const int g = log2_of((size_t)world);
  if (num_vars < (size_t)g || num_vars > 62) return SC_ERR_ARG;
  const PlanOpts o{opt->vars_per_pass, opt->first_pass_vars, opt->grid_pass, opt->grid_log, opt->grid_max_vars, opt->grid_sharded,
                   opt->tail_log, opt->use_mailbox, transport, g};
  // the prover's state: local table size, pending challenges, still sharded?
  int cur_log = (int)num_vars - g, kf = 0;
  bool sharded = transport != 0;
  size_t n = 0, j = 0;
  auto emit = [&](int action, int kf_, int ks_, int log_in, bool sh) {
    if (n < cap) out[n] = sc_plan_step{action, kf_, ks_, log_in, sh ? 1 : 0};
    ++n;
  };
  while (j < num_vars) {
    const PassPlan p = plan_pass(o, num_vars, j, kf, cur_log, sharded);
    if (p.error) return SC_ERR_STATE;
    if (p.kind == PassPlan::kRankPass) {
      emit(SC_PLAN_RANK_PASS, kf, p.ks, cur_log, true);
      cur_log = g;
      sharded = false;
    } else {
      if (p.gather_first) {
        emit(SC_PLAN_GATHER, 0, 0, cur_log, true);
        cur_log += g;
        sharded = false;
      }
      emit(p.kind, kf, p.ks, cur_log, sharded);
      cur_log -= kf;
    }
    kf = p.ks;   // the rounds the launch serves are answered from its cache; their challenges are pending at the next one
    j += (size_t)p.ks;
  }
  *n_out = n;
  return SC_OK;
}

extern "C" int sc_prover_create(sc_ctx* ctx, const sc_table* a, const sc_table* b, sc_prover** out) {
  return prover_create_impl(ctx, a, b, false, out);
}

extern "C" int sc_prover_c1(const sc_prover* pr, uint64_t* out) {
  if (!pr || !out) return SC_ERR_ARG;
  *out = pr->c1;
  return SC_OK;
}

extern "C" int sc_prover_num_vars(const sc_prover* pr, size_t* out) {
  if (!pr || !out) return SC_ERR_ARG;
  *out = pr->num_vars;
  return SC_OK;
}

extern "C" int sc_prover_round(sc_prover* pr, uint64_t r_prev, size_t j, uint64_t out_e[3]) {
  if (!pr || !out_e) return SC_ERR_ARG;
  sc_ctx* ctx = pr->ctx;
  if (j != pr->next_round)
    return fail(ctx, SC_ERR_STATE, "sc_prover_round: expected round %zu, got %zu", pr->next_round, j);
  if (j >= pr->num_vars) return fail(ctx, SC_ERR_STATE, "sc_prover_round: all %zu rounds done", pr->num_vars);
  if (r_prev >= ctx->fp.p && j != 0) return fail(ctx, SC_ERR_ARG, "sc_prover_round: challenge is not reduced");
  SC_TRY(set_device(ctx));
  if (j != 0) pr->pending.push_back(r_prev);  // sum-check-protocol/src/lib.rs:106-109
  if (!cache_covers(pr, j)) {
    const int rc = prover_pass(pr, j);
    if (rc != SC_OK) {
      if (j != 0 && !pr->pending.empty()) pr->pending.pop_back();  // a retried round must not fold r_prev twice
      return rc;
    }
  }
  prover_answer(pr, j, out_e);
  pr->next_round = j + 1;
  return SC_OK;
}

extern "C" int sc_prover_destroy(sc_prover* pr) {
  if (!pr) return SC_OK;
  pool_release(pr->ctx, pr->own_a);
  pool_release(pr->ctx, pr->own_b);
  delete pr;
  return SC_OK;
}

extern "C" int sc_prove(sc_ctx* ctx, const sc_table* a, const sc_table* b, sc_draw_fn draw, void* user,
                        uint64_t seed_r, uint64_t* c1, uint64_t* evals, uint64_t* challenges) {
  if (!ctx) return SC_ERR_ARG;
  sc_prover* pr = nullptr;
  SC_TRY(sc_prover_create(ctx, a, b, &pr));
  if (c1) *c1 = pr->c1;
  HostField hf(ctx->fp);
  u64 r_j = hf.one();  // callers pass F::one() for round 0 (matrix-multiplication/src/lib.rs:356)
  int rc = SC_OK;
  for (size_t j = 0; j < pr->num_vars; ++j) {
    u64 e[3];
    rc = sc_prover_round(pr, r_j, j, e);
    if (rc != SC_OK) break;
    if (evals) memcpy(evals + 3 * j, e, sizeof(e));
    r_j = draw ? draw(user, j, e)
               : hf.mul(sc::splitmix64(seed_r + j + 1) % ctx->fp.p, ctx->fp.r2_mod_p);
    if (r_j >= ctx->fp.p) {
      rc = fail(ctx, SC_ERR_ARG, "sc_prove: draw() returned an unreduced challenge");
      break;
    }
    if (challenges) challenges[j] = r_j;
  }
  sc_prover_destroy(pr);
  return rc;
}

// =====================================================================================
// C ABI: gkr_protocol::round_polynomial::W
// =====================================================================================

namespace {

struct WView {
  const u64 *add, *mul, *w_b, *w_c;
  int kb, kc;       // variables of b and c (global)
  size_t rows = 0;  // values of c this rank holds (2^kc unsharded)
};

int check_w(const sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b, const sc_table* w_c,
            WView* v) {
  SC_TRY(check_table(ctx, add, "gkr W"));
  SC_TRY(check_table(ctx, mul, "gkr W"));
  SC_TRY(check_table(ctx, w_b, "gkr W"));
  SC_TRY(check_table(ctx, w_c, "gkr W"));
  // sharded contexts: add and mul are this rank's rows of c (top log2(world) bits of the index = rank), w_b and w_c are
  // whole on every rank - the layout of sc_gkr_prover_create
  v->kb = log2_of(w_b->len);
  v->kc = log2_of(w_c->len);
  if (add->len != mul->len || add->len * (size_t)ctx->world != ((size_t)1 << (v->kb + v->kc)))
    return fail(ctx, SC_ERR_ARG, "gkr W: add/mul must have num_vars(w_b) + num_vars(w_c) variables (over all ranks)");
  if (ctx->world > 1 && v->kc < ctx->log_world)
    return fail(ctx, SC_ERR_UNSUPPORTED, "sharded gkr W: fewer rows of c than ranks");
  v->rows = add->len >> v->kb;
  v->add = add->d;
  v->mul = mul->d;
  v->w_b = w_b->d;
  v->w_c = w_c->d;
  return SC_OK;
}

// (H(0), H(1), H(2)) of the current round; the summed variable is w_b's while it has any.  Sharded: every rank sums
// over its own rows of c (the pairs of the summed variable are shard-local: index bit 0), the limbs are added across
// the ranks like a pass's.
int w_round_sums(sc_ctx* ctx, const WView& w, u64 e[3]) {
  if (w.kb + w.kc < 1) return fail(ctx, SC_ERR_ARG, "gkr W: no variable left");
  const bool across = is_sharded(ctx);
  const size_t c0 = across ? (size_t)ctx->rank * w.rows : 0;   // first value of c on this rank
  const u64 *V, *Fx;
  int logV;
  if (w.kb >= 1) {
    V = w.w_b;
    logV = w.kb;
    Fx = w.w_c + c0;
  } else {
    if (w.rows < 2) return fail(ctx, SC_ERR_UNSUPPORTED, "sharded gkr W: the summed variable crosses shards");
    V = w.w_c + c0;
    logV = log2_of(w.rows);
    Fx = w.w_b;
  }
  const size_t n_pieces = (w.rows << w.kb) / 2;
  int grid = grid_for(ctx, n_pieces);
  bool mb = false;
  const u64 dg[1] = {(u64)w.kb << 32 | (u64)w.kc};
  sc::PassOut out = next_pass_out(ctx, across, challenge_digest(dg, 1, 0, 77), &mb);
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::gkr_sums_kernel<F>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream, f,
                                                  w.add, w.mul, V, logV, Fx, n_pieces, out));
  SC_TRY(commit_pass_out(ctx, out, grid));
  SC_TRY(collect_sums(ctx, 3, across, mb, e));
  HostField hf(ctx->fp);
  e[2] = eval2_from_inf(hf, e[0], e[1], e[2]);
  return SC_OK;
}

int evaluate_replicated(sc_ctx* ctx, const u64* d, size_t len, const u64* pt, u64* out);
int fetch_word(sc_ctx* ctx, const u64* d, u64* out);

}  // namespace

extern "C" int sc_gkr_wiring(sc_ctx* ctx, const int32_t* gate_type, const uint32_t* in0, const uint32_t* in1, size_t k_i,
                             size_t k_next, const uint64_t* r_i, sc_table** add_out, sc_table** mul_out) {
  if (!ctx || !gate_type || !in0 || !in1 || (k_i && !r_i) || !add_out || !mul_out) return SC_ERR_ARG;
  if (k_i > 30 || k_next > 15) return fail(ctx, SC_ERR_ARG, "sc_gkr_wiring: layer too large");
  if (ctx->world > 1 && k_next < (size_t)ctx->log_world) return fail(ctx, SC_ERR_UNSUPPORTED, "sc_gkr_wiring: fewer rows of c than ranks");
  SC_TRY(set_device(ctx));
  // sharded: every rank gets the whole gate list and keeps the gates whose c (= in1) falls into its rows - the shard
  // (top log2(world) index bits = rank) of the tables, with no exchange
  const size_t n_gates = (size_t)1 << k_i, n_next = (size_t)1 << k_next, rows = n_next / (size_t)ctx->world, len = rows * n_next;
  const unsigned row_lo = (unsigned)((size_t)ctx->rank * rows);
  for (size_t a = 0; a < n_gates; ++a) {
    if ((gate_type[a] != 0 && gate_type[a] != 1) || in0[a] >= n_next || in1[a] >= n_next)
      return fail(ctx, SC_ERR_ARG, "sc_gkr_wiring: gate %zu is malformed", a);
  }
  u64* eq = nullptr;
  u64* gates = nullptr;  // [type | in0 | in1] as 32-bit words
  sc_table *ta = nullptr, *tm = nullptr;
  int rc = build_eq_table(ctx, r_i, (int)k_i, &eq);
  if (rc == SC_OK) rc = pool_alloc(ctx, (3 * n_gates * 4 + 7) / 8 + 1, &gates);
  if (rc == SC_OK) rc = new_table(ctx, len, &ta);
  if (rc == SC_OK) rc = new_table(ctx, len, &tm);
  if (rc == SC_OK) {
    int* d_type = (int*)gates;
    unsigned* d_in0 = (unsigned*)gates + n_gates;
    unsigned* d_in1 = (unsigned*)gates + 2 * n_gates;
    hipError_t e = hipMemcpyAsync(d_type, gate_type, n_gates * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in0, in0, n_gates * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in1, in1, n_gates * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ta->d, 0, len * sizeof(u64), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(tm->d, 0, len * sizeof(u64), ctx->stream);
    if (e == hipSuccess) {
      SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::gkr_wiring_scatter_kernel<F>), dim3(grid_for_wide(ctx, n_gates)),
                                                      dim3(sc::kBlock), 0, ctx->stream, f, (const u64*)eq, (const int*)d_type,
                                                      (const unsigned*)d_in0, (const unsigned*)d_in1, n_gates, (int)k_next,
                                                      row_lo, (unsigned)rows, ta->d, tm->d));
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "sc_gkr_wiring: %s", hipGetErrorString(e));
  }
  pool_release(ctx, eq);
  pool_release(ctx, gates);
  if (rc != SC_OK) {
    sc_table_free(ctx, ta);
    sc_table_free(ctx, tm);
    return rc;
  }
  *add_out = ta;
  *mul_out = tm;
  return SC_OK;
}

extern "C" int sc_gkr_w_to_evaluations(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                                       const sc_table* w_c, sc_table** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  WView w;
  SC_TRY(check_w(ctx, add, mul, w_b, w_c, &w));
  // the reference's output order is b-major while the shards are rows of c: a sharded result would need an all-to-all
  // nobody consumes (Prover::new only sums it: sc_gkr_prover_c1 / the round sums give that)
  if (is_sharded(ctx) && ctx->world > 1) return fail(ctx, SC_ERR_UNSUPPORTED, "gkr W to_evaluations on a sharded context");
  SC_TRY(set_device(ctx));
  sc_table* t = nullptr;
  SC_TRY(new_table(ctx, add->len, &t));
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::gkr_to_evaluations_kernel<F>), dim3(grid_for_wide(ctx, add->len)),
                                                  dim3(sc::kBlock), 0, ctx->stream, f, w.add, w.mul, w.w_b, w.kb, w.w_c, w.kc,
                                                  t->d));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    sc_table_free(ctx, t);
    return fail(ctx, SC_ERR_HIP, "gkr to_evaluations: %s", hipGetErrorString(e));
  }
  *out = t;
  return SC_OK;
}

extern "C" int sc_gkr_w_round_sums(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                                   const sc_table* w_c, uint64_t out_e[3]) {
  if (!ctx || !out_e) return SC_ERR_ARG;
  WView w;
  SC_TRY(check_w(ctx, add, mul, w_b, w_c, &w));
  SC_TRY(set_device(ctx));
  return w_round_sums(ctx, w, out_e);
}

extern "C" int sc_gkr_w_fix_variables(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                                      const sc_table* w_c, const uint64_t* r, size_t k, sc_table** add_out,
                                      sc_table** mul_out, sc_table** w_b_out, sc_table** w_c_out) {
  if (!ctx || (k && !r) || !add_out || !mul_out || !w_b_out || !w_c_out) return SC_ERR_ARG;
  WView w;
  SC_TRY(check_w(ctx, add, mul, w_b, w_c, &w));
  if (k > (size_t)(w.kb + w.kc)) return fail(ctx, SC_ERR_ARG, "gkr fix_variables: k=%zu > num_vars=%d", k, w.kb + w.kc);
  const size_t k_b = std::min<size_t>(k, (size_t)w.kb), k_c = k - k_b;   // round_polynomial.rs:60-63
  sc_table *oa = nullptr, *om = nullptr, *ob = nullptr, *oc = nullptr;
  int rc = sc_table_fix_variables(ctx, add, r, k, SC_ORDER_LE, &oa);
  if (rc == SC_OK) rc = sc_table_fix_variables(ctx, mul, r, k, SC_ORDER_LE, &om);
  if (rc == SC_OK) rc = sc_table_fix_variables(ctx, w_b, r, k_b, SC_ORDER_LE, &ob);
  if (rc == SC_OK) rc = sc_table_fix_variables(ctx, w_c, r + k_b, k_c, SC_ORDER_LE, &oc);
  if (rc != SC_OK) {
    sc_table_free(ctx, oa);
    sc_table_free(ctx, om);
    sc_table_free(ctx, ob);
    sc_table_free(ctx, oc);
    return rc;
  }
  *add_out = oa;
  *mul_out = om;
  *w_b_out = ob;
  *w_c_out = oc;
  return SC_OK;
}

extern "C" int sc_gkr_w_evaluate(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                                 const sc_table* w_c, const uint64_t* point, size_t n, uint64_t* out) {
  if (!ctx || !out || (n && !point)) return SC_ERR_ARG;
  WView w;
  SC_TRY(check_w(ctx, add, mul, w_b, w_c, &w));
  if (n != (size_t)(w.kb + w.kc)) return fail(ctx, SC_ERR_ARG, "gkr evaluate: point has %zu entries, W has %d variables", n, w.kb + w.kc);
  u64 ae = 0, me = 0, wb = 0, wc = 0;
  SC_TRY(sc_table_evaluate(ctx, add, point, n, SC_ORDER_LE, &ae));
  SC_TRY(sc_table_evaluate(ctx, mul, point, n, SC_ORDER_LE, &me));
  SC_TRY(evaluate_replicated(ctx, w_b->d, w_b->len, point, &wb));          // whole on every rank
  SC_TRY(evaluate_replicated(ctx, w_c->d, w_c->len, point + w.kb, &wc));
  HostField hf(ctx->fp);
  *out = hf.add(hf.mul(ae, hf.add(wb, wc)), hf.mul(me, hf.mul(wb, wc)));   // round_polynomial.rs:56
  return SC_OK;
}

// The W prover in its two-phase form (kernels.hpp, "Two-phase form of the W sumcheck"): the rounds over
// the b variables are ONE product-of-two-tables proof on [P | L] x [W_b | 1], the rounds over the c
// variables one on [Q | w* add_r] x [W_c | 1]; both run on the pass engine of sc_prover.  add and mul are
// streamed twice per layer (dense form) or the gate list is scattered twice (sparse form).
struct sc_gkr_prover {
  sc_ctx* ctx = nullptr;
  // dense form: borrowed tables, index (c << kb) | b (this rank's rows of c on a sharded context)
  const u64 *add = nullptr, *mul = nullptr;
  size_t add_len = 0;
  // both forms: W_b and W_c, whole on every rank
  const u64 *w_b = nullptr, *w_c = nullptr;
  int kb = 0, kc = 0;   // variables of b / c when the prover was created
  // sparse form: the gate list on the device and eq(r_i, a)
  bool sparse = false;
  size_t n_gates = 0;
  int* sp_type = nullptr;
  unsigned *sp_in0 = nullptr, *sp_in1 = nullptr;
  u64* sp_val = nullptr;
  u64* sp_words = nullptr;   // pool block backing sp_type / sp_in0 / sp_in1
  size_t num_vars = 0, next_round = 0;
  u64 c1 = 0;
  std::vector<u64> r;        // challenges received
  // current phase
  sc_prover* sub = nullptr;
  sc_table ta, tb;
  u64 *TA = nullptr, *TB = nullptr;
};

namespace {

// value of a whole (unsharded) device table at an LE point, on this rank alone
int evaluate_replicated(sc_ctx* ctx, const u64* d, size_t len, const u64* pt, u64* out) {
  if (len == 1) return fetch_word(ctx, d, out);
  HostField hf(ctx->fp);
  bool mb = false;
  SC_TRY(evaluate_local(ctx, d, len, pt, hf.one(), false, &mb));
  return collect_sums(ctx, 1, false, mb, out);
}

// P and L of the b phase (dense form): one streaming pass over add and mul
int gkr_dense_phase1(sc_gkr_prover* pr, u64* P, u64* L) {
  sc_ctx* ctx = pr->ctx;
  const size_t M = (size_t)1 << pr->kb;
  const size_t rows = pr->add_len / M;   // this rank's values of c
  const u64* w = pr->w_c + (is_sharded(ctx) ? (size_t)ctx->rank * rows : 0);
  const RowWalk rw = row_walk_shape(ctx, rows, M);
  const size_t gx = rw.gx, chunks = rw.chunks, rows_per_chunk = rw.rows_per_chunk;
  if (rows_per_chunk > sc::GoldilocksMont::kAccMaxTerms)
    return fail(ctx, SC_ERR_UNSUPPORTED, "gkr: %zu rows per chunk exceed the lazy accumulator's capacity", rows_per_chunk);
  u64 *pP = P, *pL = L;
  if (chunks > 1) {
    SC_TRY(pool_alloc(ctx, chunks * M, &pP));
    int rc = pool_alloc(ctx, chunks * M, &pL);
    if (rc != SC_OK) {
      pool_release(ctx, pP);
      return rc;
    }
  }
  const int nt = pr->add_len >= ((size_t)1 << ctx->nt_load_log) ? 1 : 0;
  int rc = timer_begin(ctx, SC_KIND_GKR, pr->kc, 0, log2_of(pr->add_len), (u64)16 * pr->add_len + 8 * rows, (u64)16 * M);
  if (rc == SC_OK) {
#define SC_PHASE1(NT, PW)                                                                                                   \
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::gkr_phase1_kernel<F, NT, PW>), dim3((unsigned)gx, (unsigned)chunks),  \
                                                  dim3(sc::kBlock), 0, ctx->stream, f, pr->add, pr->mul, w, rows, rows_per_chunk, M, pP, pL))
    if (nt) {
      if (rw.pw == 4) SC_PHASE1(true, 4);
      else if (rw.pw == 2) SC_PHASE1(true, 2);
      else SC_PHASE1(true, 1);
    } else {
      if (rw.pw == 4) SC_PHASE1(false, 4);
      else if (rw.pw == 2) SC_PHASE1(false, 2);
      else SC_PHASE1(false, 1);
    }
#undef SC_PHASE1
    if (chunks > 1) {
      const int grid = grid_for_wide(ctx, M);
      SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::sum_rows_kernel<F>), dim3(grid, 2), dim3(sc::kBlock), 0, ctx->stream, f,
                                                      (const u64*)pP, (const u64*)pL, chunks, M, P, L));
    }
    if (hipGetLastError() != hipSuccess) {
      poison(ctx);
      rc = fail(ctx, SC_ERR_HIP, "gkr phase-1 launch failed");
    }
  }
  if (rc == SC_OK) rc = timer_end(ctx);
  if (chunks > 1) {
    pool_release(ctx, pP);
    pool_release(ctx, pL);
  }
  return rc;
}

// sum a vector of residues over the ranks of a sharded context (split limbs, exact), in place
int allreduce_residues(sc_ctx* ctx, u64* v, size_t n) {
  if (!is_sharded(ctx) || ctx->world == 1) return SC_OK;
  u64* limbs = nullptr;
  SC_TRY(pool_alloc(ctx, 2 * n, &limbs));
  hipLaunchKernelGGL(sc::split_limbs_kernel, dim3(grid_for(ctx, n)), dim3(sc::kBlock), 0, ctx->stream, (const u64*)v, n, limbs);
  int rc = allreduce_device(ctx, limbs, 2 * n);
  if (rc == SC_OK) {
    SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::recombine_limbs_kernel<F>), dim3(grid_for(ctx, n)), dim3(sc::kBlock), 0,
                                                    ctx->stream, f, (const u64*)limbs, n, v));
    if (hipGetLastError() != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "recombine_limbs_kernel launch failed");
  }
  pool_release(ctx, limbs);
  return rc;
}

// Build the phase's tables and start its product prover.  which = 0: b variables; 1: c variables (b fixed at
// pr->r[0 .. kb)).
int gkr_start_phase(sc_gkr_prover* pr, int which) {
  sc_ctx* ctx = pr->ctx;
  HostField hf(ctx->fp);
  if (pr->sub) sc_prover_destroy(pr->sub);
  pr->sub = nullptr;
  pool_release(ctx, pr->TA);
  pool_release(ctx, pr->TB);
  pr->TA = pr->TB = nullptr;
  const int kv = which == 0 ? pr->kb : pr->kc;   // variables of this phase
  const size_t n = (size_t)1 << kv;
  u64 *X = nullptr, *Y = nullptr;   // P, L  or  add_r, mul_r
  u64 wstar = 0;
  int rc = SC_OK;
  const bool sharded_dense = !pr->sparse && is_sharded(ctx) && ctx->world > 1;
  if (which == 0) {
    rc = pool_alloc(ctx, n, &X);
    if (rc == SC_OK) rc = pool_alloc(ctx, n, &Y);
    if (rc == SC_OK && pr->sparse) {
      hipError_t e = hipMemsetAsync(X, 0, n * sizeof(u64), ctx->stream);
      if (e == hipSuccess) e = hipMemsetAsync(Y, 0, n * sizeof(u64), ctx->stream);
      if (e == hipSuccess) {
        SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::gkr_sparse_phase1_kernel<F>), dim3(grid_for_wide(ctx, pr->n_gates)),
                                                        dim3(sc::kBlock), 0, ctx->stream, f, (const u64*)pr->sp_val,
                                                        (const int*)pr->sp_type, (const unsigned*)pr->sp_in0,
                                                        (const unsigned*)pr->sp_in1, pr->n_gates, pr->w_c, X, Y));
        e = hipGetLastError();
      }
      if (e != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "gkr sparse phase 1: %s", hipGetErrorString(e));
    } else if (rc == SC_OK) {
      rc = gkr_dense_phase1(pr, X, Y);
      // sharded: every rank summed its own rows of c; P and L are the sums over the ranks
      if (rc == SC_OK && sharded_dense) rc = allreduce_residues(ctx, X, n);
      if (rc == SC_OK && sharded_dense) rc = allreduce_residues(ctx, Y, n);
    }
  } else {
    // w* = W_b(r_b)
    rc = evaluate_replicated(ctx, pr->w_b, (size_t)1 << pr->kb, pr->r.data(), &wstar);
    if (rc == SC_OK && pr->sparse) {
      u64* eqb = nullptr;
      rc = build_eq_table(ctx, pr->r.data(), pr->kb, &eqb);
      if (rc == SC_OK) rc = pool_alloc(ctx, n, &X);
      if (rc == SC_OK) rc = pool_alloc(ctx, n, &Y);
      if (rc == SC_OK) {
        hipError_t e = hipMemsetAsync(X, 0, n * sizeof(u64), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(Y, 0, n * sizeof(u64), ctx->stream);
        if (e == hipSuccess) {
          SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::gkr_sparse_phase2_kernel<F>), dim3(grid_for_wide(ctx, pr->n_gates)),
                                                          dim3(sc::kBlock), 0, ctx->stream, f, (const u64*)pr->sp_val,
                                                          (const int*)pr->sp_type, (const unsigned*)pr->sp_in0,
                                                          (const unsigned*)pr->sp_in1, pr->n_gates, (const u64*)eqb, X, Y));
          e = hipGetLastError();
        }
        if (e != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "gkr sparse phase 2: %s", hipGetErrorString(e));
      }
      pool_release(ctx, eqb);
    } else if (rc == SC_OK) {
      // add(r_b, .) and mul(r_b, .): fix the kb low variables of both tables (one streaming pass each)
      size_t la = 0, lm = 0;
      rc = fold_chain(ctx, pr->add, pr->add_len, pr->r.data(), (size_t)pr->kb, SC_ORDER_LE, &X, &la);
      if (rc == SC_OK) rc = fold_chain(ctx, pr->mul, pr->add_len, pr->r.data(), (size_t)pr->kb, SC_ORDER_LE, &Y, &lm);
      if (rc == SC_OK && sharded_dense) {
        // every rank holds its rows of c: gather the whole 2^kc-entry tables
        u64 *gx = nullptr, *gy = nullptr;
        rc = gather_pair(ctx, X, Y, la, &gx, &gy);
        pool_release(ctx, X);
        pool_release(ctx, Y);
        X = gx;
        Y = gy;
      }
    }
  }
  if (rc == SC_OK) rc = pool_alloc(ctx, 2 * n, &pr->TA);
  if (rc == SC_OK) rc = pool_alloc(ctx, 2 * n, &pr->TB);
  if (rc == SC_OK) {
    const u64 sY = which == 0 ? 0 : wstar, sZ = which == 0 ? hf.one() : wstar;
    const u64* Z = which == 0 ? Y : X;
    const u64* V = which == 0 ? pr->w_b : pr->w_c;
    SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::gkr_combine_kernel<F>), dim3(grid_for_wide(ctx, n)), dim3(sc::kBlock), 0,
                                                    ctx->stream, f, (const u64*)X, (const u64*)(which == 0 ? X : Y), sY, Z, sZ, V, n,
                                                    pr->TA, pr->TB));
    if (hipGetLastError() != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "gkr_combine_kernel launch failed");
  }
  pool_release(ctx, X);   // stream-ordered: the combine kernel is ahead of any reuse
  pool_release(ctx, Y);
  if (rc != SC_OK) return rc;
  pr->ta.d = pr->TA;
  pr->ta.len = 2 * n;
  pr->tb.d = pr->TB;
  pr->tb.len = 2 * n;
  return prover_create_impl(ctx, &pr->ta, &pr->tb, true, &pr->sub);
}

int gkr_prover_begin(sc_gkr_prover* pr) {
  pr->num_vars = (size_t)(pr->kb + pr->kc);
  SC_TRY(gkr_start_phase(pr, pr->kb >= 1 ? 0 : 1));
  pr->c1 = pr->sub->c1;   // sum of [X | lin] . [V | 1] = sum_b W(b) P(b) + L(b) = sum of f
  return SC_OK;
}

}  // namespace

extern "C" int sc_gkr_prover_create_sparse(sc_ctx* ctx, const int32_t* gate_type, const uint32_t* in0, const uint32_t* in1,
                                           size_t k_i, size_t k_next, const uint64_t* r_i, const sc_table* w_next,
                                           sc_gkr_prover** out) {
  if (!ctx || !gate_type || !in0 || !in1 || (k_i && !r_i) || !out) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, w_next, "sc_gkr_prover_create_sparse"));
  if (k_i > 30 || k_next < 1 || k_next > 26 || w_next->len != ((size_t)1 << k_next))
    return fail(ctx, SC_ERR_ARG, "sc_gkr_prover_create_sparse: bad layer sizes");
  SC_TRY(set_device(ctx));
  const size_t n_gates = (size_t)1 << k_i, n_next = (size_t)1 << k_next;
  for (size_t a = 0; a < n_gates; ++a)
    if ((gate_type[a] != 0 && gate_type[a] != 1) || in0[a] >= n_next || in1[a] >= n_next)
      return fail(ctx, SC_ERR_ARG, "sc_gkr_prover_create_sparse: gate %zu is malformed", a);
  sc_gkr_prover* pr = new (std::nothrow) sc_gkr_prover;
  if (!pr) return fail(ctx, SC_ERR_OOM, "host allocation failed");
  pr->ctx = ctx;
  pr->sparse = true;
  pr->n_gates = n_gates;
  pr->w_b = pr->w_c = w_next->d;   // on a sharded context: the whole table, on every rank
  pr->kb = pr->kc = (int)k_next;
  int rc = pool_alloc(ctx, (3 * n_gates * 4 + 7) / 8 + 1, &pr->sp_words);   // type | in0 | in1 as 32-bit words
  if (rc == SC_OK) rc = build_eq_table(ctx, r_i, (int)k_i, &pr->sp_val);  // eq(r_i, a): the gate's weight
  if (rc == SC_OK) {
    pr->sp_type = (int*)pr->sp_words;
    pr->sp_in0 = (unsigned*)pr->sp_words + n_gates;
    pr->sp_in1 = (unsigned*)pr->sp_words + 2 * n_gates;
    hipError_t e = hipMemcpyAsync(pr->sp_type, gate_type, n_gates * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pr->sp_in0, in0, n_gates * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pr->sp_in1, in1, n_gates * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // host arrays may go away after return
    if (e != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "sc_gkr_prover_create_sparse: %s", hipGetErrorString(e));
  }
  if (rc == SC_OK) rc = gkr_prover_begin(pr);
  if (rc != SC_OK) {
    sc_gkr_prover_destroy(pr);
    return rc;
  }
  *out = pr;
  return SC_OK;
}

extern "C" int sc_gkr_prover_create(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b,
                                    const sc_table* w_c, sc_gkr_prover** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, add, "gkr W"));
  SC_TRY(check_table(ctx, mul, "gkr W"));
  SC_TRY(check_table(ctx, w_b, "gkr W"));
  SC_TRY(check_table(ctx, w_c, "gkr W"));
  SC_TRY(set_device(ctx));
  const int kb = log2_of(w_b->len), kc = log2_of(w_c->len);
  // sharded: add and mul are this rank's rows of c (top log2(world) bits of the index = rank); W_b and W_c whole
  if (add->len != mul->len || add->len * (size_t)ctx->world != ((size_t)1 << (kb + kc)))
    return fail(ctx, SC_ERR_ARG, "gkr W: add/mul must have num_vars(w_b) + num_vars(w_c) variables");
  if (kb + kc < 1) return fail(ctx, SC_ERR_ARG, "sc_gkr_prover_create: W has no variables");
  if (ctx->world > 1 && (kc < ctx->log_world || kb < 1))
    return fail(ctx, SC_ERR_UNSUPPORTED, "sharded gkr W needs at least log2(world) variables of c and one of b");
  sc_gkr_prover* pr = new (std::nothrow) sc_gkr_prover;
  if (!pr) return fail(ctx, SC_ERR_OOM, "host allocation failed");
  pr->ctx = ctx;
  pr->add = add->d;
  pr->mul = mul->d;
  pr->add_len = add->len;
  pr->w_b = w_b->d;
  pr->w_c = w_c->d;
  pr->kb = kb;
  pr->kc = kc;
  int rc = gkr_prover_begin(pr);
  if (rc != SC_OK) {
    sc_gkr_prover_destroy(pr);
    return rc;
  }
  *out = pr;
  return SC_OK;
}

extern "C" int sc_gkr_prover_c1(const sc_gkr_prover* pr, uint64_t* out) {
  if (!pr || !out) return SC_ERR_ARG;
  *out = pr->c1;
  return SC_OK;
}

extern "C" int sc_gkr_prover_round(sc_gkr_prover* pr, uint64_t r_prev, size_t j, uint64_t out_e[3]) {
  if (!pr || !out_e) return SC_ERR_ARG;
  sc_ctx* ctx = pr->ctx;
  if (j != pr->next_round) return fail(ctx, SC_ERR_STATE, "sc_gkr_prover_round: expected round %zu, got %zu", pr->next_round, j);
  if (j >= pr->num_vars) return fail(ctx, SC_ERR_STATE, "sc_gkr_prover_round: all %zu rounds done", pr->num_vars);
  if (j != 0 && r_prev >= ctx->fp.p) return fail(ctx, SC_ERR_ARG, "sc_gkr_prover_round: challenge is not reduced");
  SC_TRY(set_device(ctx));
  HostField hf(ctx->fp);
  if (j != 0) pr->r.push_back(r_prev);
  size_t local_j = j;
  bool phase_start = (j == 0);
  if (pr->kb >= 1 && j >= (size_t)pr->kb) {
    local_j = j - (size_t)pr->kb;
    if (local_j == 0) {
      // b is fixed at r[0 .. kb): build the c phase
      const int rc = gkr_start_phase(pr, 1);
      if (rc != SC_OK) {
        pr->r.pop_back();
        return rc;
      }
      phase_start = true;
    }
  }
  const int rc = sc_prover_round(pr->sub, phase_start ? hf.one() : r_prev, local_j, out_e);
  if (rc != SC_OK) {
    if (j != 0) pr->r.pop_back();
    return rc;
  }
  pr->next_round = j + 1;
  return SC_OK;
}

// Whole W sumcheck in one call (sc_prove's contract): create, 2k rounds, the challenge of each round drawn on the host
// after that round's sums have been read back.
template <class P, class RoundFn>
static int run_rounds(sc_ctx* ctx, P* pr, size_t num_vars, RoundFn round, sc_draw_fn draw, void* user, uint64_t seed_r,
                      uint64_t* evals, uint64_t* challenges) {
  HostField hf(ctx->fp);
  u64 r_j = hf.one();   // callers pass F::one() for round 0
  for (size_t j = 0; j < num_vars; ++j) {
    u64 e[3];
    SC_TRY(round(pr, r_j, j, e));
    if (evals) memcpy(evals + 3 * j, e, sizeof(e));
    r_j = draw ? draw(user, j, e) : hf.mul(sc::splitmix64(seed_r + j + 1) % ctx->fp.p, ctx->fp.r2_mod_p);
    if (r_j >= ctx->fp.p) return fail(ctx, SC_ERR_ARG, "draw() returned an unreduced challenge");
    if (challenges) challenges[j] = r_j;
  }
  return SC_OK;
}

extern "C" int sc_gkr_prove(sc_ctx* ctx, const sc_table* add, const sc_table* mul, const sc_table* w_b, const sc_table* w_c,
                            sc_draw_fn draw, void* user, uint64_t seed_r, uint64_t* c1, uint64_t* evals, uint64_t* challenges) {
  if (!ctx) return SC_ERR_ARG;
  sc_gkr_prover* pr = nullptr;
  SC_TRY(sc_gkr_prover_create(ctx, add, mul, w_b, w_c, &pr));
  if (c1) *c1 = pr->c1;
  const int rc = run_rounds(ctx, pr, pr->num_vars, sc_gkr_prover_round, draw, user, seed_r, evals, challenges);
  sc_gkr_prover_destroy(pr);
  return rc;
}

extern "C" int sc_gkr_prover_destroy(sc_gkr_prover* pr) {
  if (!pr) return SC_OK;
  if (pr->sub) sc_prover_destroy(pr->sub);
  pool_release(pr->ctx, pr->TA);
  pool_release(pr->ctx, pr->TB);
  pool_release(pr->ctx, pr->sp_words);
  pool_release(pr->ctx, pr->sp_val);
  delete pr;
  return SC_OK;
}

// =====================================================================================
// C ABI: triangle_counting::G
// =====================================================================================

namespace {

struct TriView {
  const u64 *f1, *f2, *f3;
  int n1, n2, n3, xv, yv, zv;
};

int check_tri(const sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len, TriView* v) {
  SC_TRY(check_table(ctx, f1, "triangle G"));
  SC_TRY(check_table(ctx, f2, "triangle G"));
  SC_TRY(check_table(ctx, f3, "triangle G"));
  if (is_sharded(ctx)) return fail(ctx, SC_ERR_UNSUPPORTED, "triangle G on a sharded context");
  const int k = (int)var_len;
  v->f1 = f1->d; v->f2 = f2->d; v->f3 = f3->d;
  v->n1 = log2_of(f1->len); v->n2 = log2_of(f2->len); v->n3 = log2_of(f3->len);
  v->xv = v->n1 > k ? v->n1 - k : 0;          // triangle-counting/src/lib.rs:53-55
  v->yv = v->n2 > k ? v->n2 - k : 0;          // :57-59
  v->zv = v->n3 < k ? v->n3 : k;              // :61-67
  // the three copies must describe one consistent state (x fixed before y before z)
  const bool ok = (v->xv > 0) ? (v->n1 == v->xv + k && v->n2 == 2 * k && v->n3 == v->xv + k)
                : (v->yv > 0) ? (v->n1 == v->yv && v->n2 == v->yv + k && v->n3 == k)
                              : (v->n1 == 0 && v->n2 == v->zv && v->n3 == v->zv);
  if (!ok) return fail(ctx, SC_ERR_ARG, "triangle G: inconsistent table sizes (%d,%d,%d) for var_len %d", v->n1, v->n2, v->n3, k);
  return SC_OK;
}


}  // namespace

extern "C" int sc_tri_to_evaluations(sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3,
                                     size_t var_len, sc_table** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  TriView v;
  SC_TRY(check_tri(ctx, f1, f2, f3, var_len, &v));
  SC_TRY(set_device(ctx));
  if (v.xv + v.yv + v.zv > 34) return fail(ctx, SC_ERR_ARG, "triangle to_evaluations: 2^%d entries", v.xv + v.yv + v.zv);
  const size_t total = (size_t)1 << (v.xv + v.yv + v.zv);
  sc_table* t = nullptr;
  SC_TRY(new_table(ctx, total, &t));
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::tri_to_evaluations_kernel<F>), dim3(grid_for_wide(ctx, total)),
                                                  dim3(sc::kBlock), 0, ctx->stream, f, v.f1, v.f2, v.f3, v.xv, v.yv, v.zv, t->d));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    sc_table_free(ctx, t);
    return fail(ctx, SC_ERR_HIP, "triangle to_evaluations: %s", hipGetErrorString(e));
  }
  *out = t;
  return SC_OK;
}

extern "C" int sc_tri_round_sums(sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len,
                                 uint64_t out_e[3]) {
  if (!ctx || !out_e) return SC_ERR_ARG;
  TriView v;
  SC_TRY(check_tri(ctx, f1, f2, f3, var_len, &v));
  SC_TRY(set_device(ctx));
  if (v.xv + v.yv + v.zv < 1) return fail(ctx, SC_ERR_ARG, "triangle G: no variable left");
  const size_t total = (size_t)1 << (v.xv + v.yv + v.zv - 1);
  const int grid = grid_for_wide(ctx, total);
  sc::PassOut out = next_pass_out(ctx);
  SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::tri_sums_kernel<F>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream, f,
                                                  v.f1, v.f2, v.f3, v.xv, v.yv, v.zv, out));
  SC_TRY(commit_pass_out(ctx, out, grid));
  SC_TRY(collect_sums(ctx, 3, false, ctx->use_mailbox != 0, out_e));
  HostField hf(ctx->fp);
  out_e[2] = eval2_from_inf(hf, out_e[0], out_e[1], out_e[2]);
  return SC_OK;
}

extern "C" int sc_tri_fix_variables(sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3,
                                    size_t var_len, const uint64_t* r, size_t k, sc_table** f1_out, sc_table** f2_out,
                                    sc_table** f3_out) {
  if (!ctx || (k && !r) || !f1_out || !f2_out || !f3_out) return SC_ERR_ARG;
  TriView v;
  SC_TRY(check_tri(ctx, f1, f2, f3, var_len, &v));
  const size_t xv = v.xv, yv = v.yv;
  if (k > (size_t)(v.xv + v.yv + v.zv)) return fail(ctx, SC_ERR_ARG, "triangle fix_variables: k=%zu > num_vars", k);
  // triangle-counting/src/lib.rs:90-105
  const size_t n_xy = std::min(xv + yv, k);
  const size_t n_yz = k > xv ? k - xv : 0;
  std::vector<u64> xz(r, r + std::min(xv, k));
  if (k > xv + yv) xz.insert(xz.end(), r + xv + yv, r + k);
  sc_table *o1 = nullptr, *o2 = nullptr, *o3 = nullptr;
  int rc = sc_table_fix_variables(ctx, f1, r, n_xy, SC_ORDER_LE, &o1);
  if (rc == SC_OK) rc = sc_table_fix_variables(ctx, f2, r + std::min(xv, k), n_yz, SC_ORDER_LE, &o2);
  if (rc == SC_OK) rc = sc_table_fix_variables(ctx, f3, xz.data(), xz.size(), SC_ORDER_LE, &o3);
  if (rc != SC_OK) {
    sc_table_free(ctx, o1);
    sc_table_free(ctx, o2);
    sc_table_free(ctx, o3);
    return rc;
  }
  *f1_out = o1;
  *f2_out = o2;
  *f3_out = o3;
  return SC_OK;
}

extern "C" int sc_tri_evaluate(sc_ctx* ctx, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len,
                               const uint64_t* point, size_t n, uint64_t* out) {
  if (!ctx || !out || (n && !point)) return SC_ERR_ARG;
  TriView v;
  SC_TRY(check_tri(ctx, f1, f2, f3, var_len, &v));
  if (n != (size_t)(v.xv + v.yv + v.zv)) return fail(ctx, SC_ERR_ARG, "triangle evaluate: point has %zu entries, G has %d variables", n, v.xv + v.yv + v.zv);
  // :72-84
  std::vector<u64> xz(point, point + v.xv);
  xz.insert(xz.end(), point + v.xv + v.yv, point + n);
  u64 e1 = 0, e2 = 0, e3 = 0;
  SC_TRY(sc_table_evaluate(ctx, f1, point, (size_t)(v.xv + v.yv), SC_ORDER_LE, &e1));
  SC_TRY(sc_table_evaluate(ctx, f2, point + v.xv, (size_t)(v.yv + v.zv), SC_ORDER_LE, &e2));
  SC_TRY(sc_table_evaluate(ctx, f3, xz.data(), xz.size(), SC_ORDER_LE, &e3));
  HostField hf(ctx->fp);
  *out = hf.mul(hf.mul(e1, e3), e2);   // :86
  return SC_OK;
}

// Three product-of-two-tables sumchecks in a row (see include/sumcheck_hip.h).
struct sc_tri_prover {
  sc_ctx* ctx = nullptr;
  const u64* adj = nullptr;  // borrowed 2^(2k) table
  int k = 0;
  size_t next_round = 0;
  std::vector<u64> r;        // every challenge received
  sc_prover* sub = nullptr;  // current phase's engine
  sc_table ta, tb;           // views handed to sc_prover_create
  u64 *P = nullptr, *f3r = nullptr, *f1y = nullptr, *Q = nullptr, *f2r = nullptr;  // pool buffers
  u64* adj_full = nullptr;   // sharded contexts: the gathered adjacency table
  u64 scale = 0;             // f1(r_x, r_y) in the z phase
  u64 c1 = 0;
};

namespace {

// Fold the challenges the sub-prover has not applied yet plus `r_last`; hand back the table(s) asked for (a null
// a_out / b_out: that table is not needed, its folds are not launched).
int prover_finish(sc_prover* pr, u64 r_last, u64** a_out, u64** b_out, size_t* len_out) {
  sc_ctx* ctx = pr->ctx;
  std::vector<u64> rs(pr->pending);   // the sub-prover's own state is left untouched
  rs.push_back(r_last);
  const size_t len = (size_t)1 << pr->cur_log;
  u64 *na = nullptr, *nb = nullptr;
  size_t la = 0, lb = 0;
  if (a_out) SC_TRY(fold_chain(ctx, pr->cur_a, len, rs.data(), rs.size(), SC_ORDER_LE, &na, &la));
  if (b_out) {
    const int rc = fold_chain(ctx, pr->cur_b, len, rs.data(), rs.size(), SC_ORDER_LE, &nb, &lb);
    if (rc != SC_OK) {
      pool_release(ctx, na);
      return rc;
    }
  }
  if (a_out) *a_out = na;
  if (b_out) *b_out = nb;
  *len_out = a_out ? la : lb;
  return SC_OK;
}

// one word of device memory to the host: through the pinned mailbox (a one-wave kernel + a spin on its sequence word,
// ~5 us) rather than hipMemcpyAsync + hipStreamSynchronize (~40 us of host time around an 8-byte copy)
int fetch_word(sc_ctx* ctx, const u64* d, u64* out) {
  if (ctx->use_mailbox) {
    const u64 seq = ctx->mailbox_seq + 1;
    hipLaunchKernelGGL(sc::mailbox_copy_kernel, dim3(1), dim3(sc::kWave), 0, ctx->stream, d, 1, ctx->d_mailbox, seq);
    SC_HIP(ctx, hipGetLastError());
    ctx->mailbox_seq = seq;
    SC_TRY(wait_mailbox(ctx, seq));
    *out = ctx->h_mailbox[0];
    return SC_OK;
  }
  SC_HIP(ctx, hipMemcpyAsync(out, d, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int tri_start_phase(sc_tri_prover* tp, const u64* a, const u64* b, size_t len) {
  tp->ta.d = const_cast<u64*>(a);
  tp->ta.len = len;
  tp->tb.d = const_cast<u64*>(b);
  tp->tb.len = len;
  if (tp->sub) sc_prover_destroy(tp->sub);
  tp->sub = nullptr;
  return prover_create_impl(tp->ctx, &tp->ta, &tp->tb, /*replicated=*/true, &tp->sub);   // whole tables on every rank
}

}  // namespace

// On a sharded context `adj` is this rank's rows of the adjacency table (top log2(world) bits of the row index
// = rank).  The n^3 work - the matrix square - is split by rows of P across the ranks; the adjacency table and
// P are gathered (n^2 words each) and the three product sumchecks on 2^(2k)- and 2^k-entry tables run
// replicated on every rank with no further exchange.
extern "C" int sc_tri_prover_create(sc_ctx* ctx, const sc_table* adj, size_t var_len, sc_tri_prover** out) {
  if (!ctx || !out) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, adj, "sc_tri_prover_create"));
  const bool sharded = is_sharded(ctx) && ctx->world > 1;
  const size_t full_len = (size_t)1 << (2 * var_len);
  if (var_len < 1 || var_len > 15 || adj->len * (size_t)(sharded ? ctx->world : 1) != full_len)
    return fail(ctx, SC_ERR_ARG, "sc_tri_prover_create: adjacency table must have 2^(2*var_len) entries (over all ranks)");
  if (sharded && var_len < (size_t)ctx->log_world)
    return fail(ctx, SC_ERR_ARG, "sc_tri_prover_create: fewer rows than ranks");
  SC_TRY(set_device(ctx));
  sc_tri_prover* tp = new (std::nothrow) sc_tri_prover;
  if (!tp) return fail(ctx, SC_ERR_OOM, "host allocation failed");
  tp->ctx = ctx;
  tp->adj = adj->d;
  tp->k = (int)var_len;
  int rc = SC_OK;
  if (sharded) {
    rc = gather_table(ctx, adj->d, adj->len, &tp->adj_full);
    tp->adj = tp->adj_full;
  }
  const size_t n = (size_t)1 << var_len;
  const size_t z_rows = sharded ? n / ctx->world : n, z_begin = sharded ? (size_t)ctx->rank * z_rows : 0;
  if (rc == SC_OK) rc = pool_alloc(ctx, full_len, &tp->P);
  if (rc == SC_OK) {
    rc = timer_begin(ctx, SC_KIND_MATSQ, tp->k, 0, (int)(2 * var_len), (u64)8 * full_len, (u64)8 * z_rows * n);
    if (rc == SC_OK) {
      if (var_len >= 6 && z_rows >= 64) {
        // adjacency tables are 0/1 (G::new_adj_matrix): their square is a count, computed exactly by the int8 matrix
        // cores.  Bytes + flag, the MFMA kernel, and behind it the generic kernel, which runs only if the flag says
        // the table held something else - no host round trip decides.
        u64* bytes = nullptr;   // T8 | T8t | flag
        rc = pool_alloc(ctx, 2 * (full_len / 8) + 8, &bytes);
        if (rc == SC_OK) {
          unsigned char* t8 = reinterpret_cast<unsigned char*>(bytes);
          unsigned char* t8t = t8 + full_len;
          unsigned* flag = reinterpret_cast<unsigned*>(bytes + 2 * (full_len / 8));
          const size_t tiles = (z_rows / 64) * (n / 64);   // 64 x 64 output tiles of this rank's rows
          const int grid = (int)std::min<size_t>(tiles, (size_t)4 * ctx->num_cus);
          const size_t tiles32 = (z_rows / 32) * (n / 32);
          if (hipMemsetAsync(flag, 0, sizeof(unsigned), ctx->stream) != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "matsq: memset failed");
          SC_DISPATCH_FIELD(ctx, F, f, {
            hipLaunchKernelGGL((sc::matsq_bytes_kernel<F>), dim3((unsigned)std::min<size_t>((n / 64) * (n / 64), (size_t)8 * ctx->num_cus)),
                               dim3(sc::kBlock), 0, ctx->stream, f, tp->adj, tp->k, t8, t8t, flag);
            hipLaunchKernelGGL((sc::matsq_mfma_kernel<F>), dim3((unsigned)std::min<size_t>((tiles32 + 3) / 4, (size_t)8 * ctx->num_cus)),
                               dim3(sc::kBlock), 0, ctx->stream, f, (const unsigned char*)t8, (const unsigned char*)t8t, tp->k, tp->P, z_begin,
                               z_rows, (const unsigned*)flag);
            hipLaunchKernelGGL((sc::matsq_tiled_kernel<F>), dim3(grid), dim3(sc::kBlock), 0, ctx->stream, f, tp->adj, tp->k, tp->P, z_begin,
                               z_rows, (const unsigned*)flag);
          });
          pool_release(ctx, bytes);   // stream-ordered reuse
        }
      } else {
        SC_DISPATCH_FIELD(ctx, F, f, hipLaunchKernelGGL((sc::matsq_kernel<F>), dim3(grid_for_wide(ctx, z_rows * n)), dim3(sc::kBlock),
                                                        0, ctx->stream, f, tp->adj, tp->k, tp->P, z_begin, z_rows));
      }
      if (hipGetLastError() != hipSuccess) rc = fail(ctx, SC_ERR_HIP, "matsq kernel launch failed");
    }
    if (rc == SC_OK) rc = timer_end(ctx);
  }
  if (rc == SC_OK && sharded) {
    // every rank computed its rows of P: gather the whole square
    u64* full = nullptr;
    rc = gather_table(ctx, tp->P + z_begin * n, z_rows * n, &full);
    if (rc == SC_OK) {
      pool_release(ctx, tp->P);
      tp->P = full;
    }
  }
  // x phase: sum_{x,z} P(x,z) f3(x,z), both indexed (z << k) | x
  if (rc == SC_OK) rc = tri_start_phase(tp, tp->P, tp->adj, full_len);
  if (rc != SC_OK) {
    sc_tri_prover_destroy(tp);
    return rc;
  }
  tp->c1 = tp->sub->c1;
  *out = tp;
  return SC_OK;
}

extern "C" int sc_tri_prover_c1(const sc_tri_prover* pr, uint64_t* out) {
  if (!pr || !out) return SC_ERR_ARG;
  *out = pr->c1;
  return SC_OK;
}

extern "C" int sc_tri_prover_round(sc_tri_prover* tp, uint64_t r_prev, size_t j, uint64_t out_e[3]) {
  if (!tp || !out_e) return SC_ERR_ARG;
  sc_ctx* ctx = tp->ctx;
  const size_t k = (size_t)tp->k;
  if (j != tp->next_round) return fail(ctx, SC_ERR_STATE, "sc_tri_prover_round: expected round %zu, got %zu", tp->next_round, j);
  if (j >= 3 * k) return fail(ctx, SC_ERR_STATE, "sc_tri_prover_round: all %zu rounds done", 3 * k);
  if (j != 0 && r_prev >= ctx->fp.p) return fail(ctx, SC_ERR_ARG, "sc_tri_prover_round: challenge is not reduced");
  SC_TRY(set_device(ctx));
  if (j != 0) tp->r.push_back(r_prev);
  // a failed round leaves the challenge list as it was, so the round can be retried
#define SC_TRY_POP(expr)                        \
  do {                                          \
    int rc_ = (expr);                           \
    if (rc_ != SC_OK) {                         \
      if (j != 0) tp->r.pop_back();             \
      return rc_;                               \
    }                                           \
  } while (0)
  HostField hf(ctx->fp);
  const size_t n = (size_t)1 << k;
  if (j == k) {
    // x fully fixed at r_x = r[0..k): P(r_x, .) is not needed any more, f3(r_x, .) is
    // every buffer lands in a member of tp at once, so an error return leaks nothing (destroy frees them)
    u64* pb = nullptr;
    size_t len = 0;
    SC_TRY_POP(prover_finish(tp->sub, r_prev, nullptr, &pb, &len));   // P(r_x, .) is not needed
    pool_release(ctx, tp->f3r);
    tp->f3r = pb;  // f3(r_x, z), 2^k entries
    size_t l1 = 0;
    pool_release(ctx, tp->f1y);
    tp->f1y = nullptr;
    SC_TRY_POP(fold_chain(ctx, tp->adj, n * n, tp->r.data(), k, SC_ORDER_LE, &tp->f1y, &l1));  // f1(r_x, y)
    if (!tp->Q) SC_TRY_POP(pool_alloc(ctx, n, &tp->Q));
    SC_TRY_POP(coldot(ctx, tp->adj, tp->f3r, n, n, tp->Q));  // Q[y] = sum_z f2[(z<<k)|y] f3r[z]
    SC_TRY_POP(tri_start_phase(tp, tp->f1y, tp->Q, n));
  } else if (j == 2 * k) {
    u64* pa = nullptr;
    size_t len = 0;
    SC_TRY_POP(prover_finish(tp->sub, r_prev, &pa, nullptr, &len));   // Q(r_y) is not needed
    const int fr = fetch_word(ctx, pa, &tp->scale);                    // f1(r_x, r_y)
    pool_release(ctx, pa);
    SC_TRY_POP(fr);
    size_t l2 = 0;
    pool_release(ctx, tp->f2r);
    tp->f2r = nullptr;
    SC_TRY_POP(fold_chain(ctx, tp->adj, n * n, tp->r.data() + k, k, SC_ORDER_LE, &tp->f2r, &l2));  // f2(r_y, z)
    SC_TRY_POP(tri_start_phase(tp, tp->f2r, tp->f3r, n));
  }
  const size_t local_j = j % k;
  const bool phase_start = (local_j == 0);
  u64 e[3];
  SC_TRY_POP(sc_prover_round(tp->sub, phase_start ? hf.one() : r_prev, local_j, e));
#undef SC_TRY_POP
  if (j >= 2 * k) {
    for (int i = 0; i < 3; ++i) e[i] = hf.mul(e[i], tp->scale);
  }
  memcpy(out_e, e, sizeof(e));
  tp->next_round = j + 1;
  return SC_OK;
}

extern "C" int sc_tri_prove(sc_ctx* ctx, const sc_table* adj, size_t var_len, sc_draw_fn draw, void* user, uint64_t seed_r,
                            uint64_t* c1, uint64_t* evals, uint64_t* challenges) {
  if (!ctx) return SC_ERR_ARG;
  sc_tri_prover* tp = nullptr;
  SC_TRY(sc_tri_prover_create(ctx, adj, var_len, &tp));
  if (c1) *c1 = tp->c1;
  const int rc = run_rounds(ctx, tp, 3 * (size_t)tp->k, sc_tri_prover_round, draw, user, seed_r, evals, challenges);
  sc_tri_prover_destroy(tp);
  return rc;
}

extern "C" int sc_tri_prover_destroy(sc_tri_prover* tp) {
  if (!tp) return SC_OK;
  if (tp->sub) sc_prover_destroy(tp->sub);
  pool_release(tp->ctx, tp->P);
  pool_release(tp->ctx, tp->f3r);
  pool_release(tp->ctx, tp->f1y);
  pool_release(tp->ctx, tp->Q);
  pool_release(tp->ctx, tp->f2r);
  pool_release(tp->ctx, tp->adj_full);
  delete tp;
  return SC_OK;
}

// =====================================================================================
// C ABI: restrict_poly (gkr-protocol/src/lib.rs:291-321)
// =====================================================================================

extern "C" int sc_table_restrict_to_line(sc_ctx* ctx, const sc_table* t, const uint64_t* b, const uint64_t* c, size_t k,
                                         uint64_t* out_coeffs) {
  if (!ctx || !out_coeffs || (k && (!b || !c))) return SC_ERR_ARG;
  SC_TRY(check_table(ctx, t, "sc_table_restrict_to_line"));
  if (is_sharded(ctx)) return fail(ctx, SC_ERR_UNSUPPORTED, "restrict_to_line on a sharded table");
  if ((size_t)log2_of(t->len) != k) return fail(ctx, SC_ERR_ARG, "restrict_to_line: table has %d variables, k = %zu", log2_of(t->len), k);
  if (ctx->fp.p <= k) return fail(ctx, SC_ERR_UNSUPPORTED, "restrict_to_line needs k+1 distinct points: p = %llu <= k", (unsigned long long)ctx->fp.p);
  HostField hf(ctx->fp);
  // q(j) = W~(b + j (c - b)), j = 0..k
  std::vector<u64> xs(k + 1), ys(k + 1), pt(k), d(k);
  for (size_t i = 0; i < k; ++i) d[i] = hf.sub(c[i], b[i]);
  u64 x = 0;
  for (size_t j = 0; j <= k; ++j) {
    xs[j] = x;
    for (size_t i = 0; i < k; ++i) pt[i] = hf.add(b[i], hf.mul(x, d[i]));
    SC_TRY(sc_table_evaluate(ctx, t, pt.data(), k, SC_ORDER_LE, &ys[j]));
    x = hf.add(x, hf.one());
  }
  // Lagrange: q(X) = sum_j y_j prod_{m != j} (X - x_m) / (x_j - x_m); master polynomial once,
  // synthetic division per node.
  std::vector<u64> master(k + 2, 0);
  master[0] = hf.one();
  for (size_t m = 0; m <= k; ++m) {           // multiply by (X - x_m)
    for (size_t i = m + 1; i > 0; --i) master[i] = hf.sub(master[i - 1], hf.mul(xs[m], master[i]));
    master[0] = hf.neg(hf.mul(xs[m], master[0]));
  }
  std::vector<u64> coeffs(k + 1, 0), quot(k + 1);
  for (size_t j = 0; j <= k; ++j) {
    u64 carry = master[k + 1];                // divide master by (X - x_j)
    for (size_t i = k + 1; i > 0; --i) {
      quot[i - 1] = carry;
      carry = hf.add(master[i - 1], hf.mul(carry, xs[j]));
    }
    u64 den = hf.one();
    for (size_t m = 0; m <= k; ++m)
      if (m != j) den = hf.mul(den, hf.sub(xs[j], xs[m]));
    const u64 w = hf.mul(ys[j], hf.inv(den));
    for (size_t i = 0; i <= k; ++i) coeffs[i] = hf.add(coeffs[i], hf.mul(w, quot[i]));
  }
  memcpy(out_coeffs, coeffs.data(), (k + 1) * sizeof(u64));
  return SC_OK;
}
