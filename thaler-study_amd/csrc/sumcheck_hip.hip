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
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
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
  // optional (every RCCL has them; a stand-in may not): what bounds a collective whose peer is gone
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
};

RcclApi g_rccl;
std::string g_create_error;

bool load_rccl(std::string* why) {
  if (g_rccl.handle) return true;
  // SC_RCCL_LIBRARY: the RCCL build to use (a path; e.g. a site's own build) - that one or nothing, never a silent second choice
  const char* chosen = getenv("SC_RCCL_LIBRARY");
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  if (chosen && *chosen) {
    h = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      *why = std::string("dlopen(SC_RCCL_LIBRARY=") + chosen + ") failed: " + dlerror();
      return false;
    }
  } else {
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
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
  a.CommAbort = (decltype(a.CommAbort))dlsym(h, "ncclCommAbort");
  a.CommGetAsyncError = (decltype(a.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather) {
    *why = "librccl is missing a required symbol";
    return false;
  }
  g_rccl = a;
  return true;
}

// kLocal: the shards are the per-device contexts of ONE multi-device handle of this process (sc_ctx_create_multi): a sharded
// launch leaves this shard's own sums in its own mailbox and the handle's host thread adds them - no collective at all
enum class Transport { kNone, kRccl, kHost, kPeer, kLocal };
constexpr int kMaxSubs = 8;          // devices behind one multi-device handle (one node)
constexpr int kTailLogMax = 12;
constexpr int kTailSoftLog = 11;      // the planner aims at <= 2^this where that costs no launch (the host's share doubles with every step above it)
constexpr int kTailEntries = 1 << kTailLogMax;   // a pass whose outputs have <= 2^host_tail_log <= this many entries per table writes them to
                                     // pinned host memory (sc_ctx::h_tail): the host finishes the proof from them (option "host_tail_log")
constexpr int kTailSmallLog = 5;     // ... and what a pass_kernel launch may hand over: one wave's stores (finish_pass drains wave 0 only)
constexpr int kTailSlots = 32;       // provers of one context / handle that can hold a tail slot at a time (the others stay on the device)
constexpr size_t kTailSlotWords = 2 * (size_t)kTailEntries;   // [table][kTailEntries]
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
  u64* d_gram_rows = nullptr;     // gram_pass_kernel: the blocks' rows of 81 cells + its ticket (zero at rest); allocated on first use
  int fold_dma = 1;       // pass_kernel<4,2>: the LDS-DMA form (kernels/pass.hpp; Goldilocks)
  int pipe32 = 1, pipe32_log = 20, pipe32_blocks = 0;   // pass_kernel<3,2>: the pipelined whole-tile form on tables of >= 2^pipe32_log entries
  int gram_log = 21;              // first pass of an unsharded proof on tables of >= 2^gram_log entries: kernels/gram.hpp (0: never)
  int host_tail_log = kTailLogMax;   // folded tables of <= 2^this entries go to pinned host memory and the host finishes the proof (0: off)
  int wfold_log = 40;                // the fold behind the matrix-core first pass serves FIVE rounds (wfold_pass_kernel) on tables of <= 2^this entries (0: never)
  int wfold_min_log = 21;            // ... and of >= 2^this entries (below, a five-round grid pass folds the four challenges)
  int wfold5_min_log = 24;           // a grid pass with FIVE challenges to fold over tables of >= 2^this entries runs in the same kernel's (5, ks) form
  int wfold_always = 0;              // 0: where the proof then needs fewer launches (the planner counts both ways); 1: wherever it can run
  int wfold_blocks = 0;              // its resident grid (0 = not asked yet)
  int tail_log = 16;  // shard log-size at which a sharded prover gathers: a 512 KiB all-gather per table is
                      // cheaper than the ~25 us of collective latency of each further sharded pass
  // grid cap of the streaming kernels: three 256-thread blocks per CU (set in sc_ctx_create).
  // Measured at n = 28: fix_variables k=1 685 us with 2048 blocks, 611 with 1024, 580 with 768,
  // 629 with 512 - more blocks than that only add concurrent DRAM streams and a longer final reduction
  int max_blocks = 768;
  int num_cus = 256;
  // blocks of each pass-kernel instantiation that fit on the chip at once ([generic|goldilocks][kf][ks],
  // 0 = not asked yet)
  int resident_blocks[2][5][4] = {};
  int time_kernels = 0;
  // where a proof's wall time goes on the HOST side (always on: four clock reads per pass): ns spent spinning on the mailbox (the
  // kernels + their launch latency) and ns spent inside the pass launches (buffers, weights, hipLaunchKernelGGL); the rest of a
  // proof's wall time is the host's arithmetic between them.  Options "stat_wait_ns" / "stat_launch_ns" (get), "stat_reset" (set)
  uint64_t stat_wait_ns = 0, stat_launch_ns = 0;
  int nt_load_log = 22;   // tables of >= 2^this entries are loaded nontemporal (measured: 21-25 equal, 27 and off worse)
  int nt_store_log = 25;  // outputs of >= 2^this entries are stored nontemporal

  // workspace
  u64* d_partials = nullptr;  // [kMaxSums + spare][partial_rows]
  u64* d_sums = nullptr;      // 2*kMaxSums split limbs (+ spare)
  u64* h_sums = nullptr;      // pinned mirror
  size_t partial_rows = 0;
  unsigned* d_ticket = nullptr;  // arrival counter of finish_pass (only ever grows)
  bool fold_lds_allowed[4][2] = {};   // fold_kernel<KF, NT>: its dynamic LDS above 64 KiB has been requested
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
  // RCCL plane: how long the host waits for work queued behind a collective (a pass's sums, a gathered table) before it gives
  // the communicator up - ncclCommAbort, the context poisoned, SC_ERR_RCCL.  A collective whose peer is gone never finishes
  // on its own: its kernel spins on the device.  Generous by default (a healthy collective takes microseconds; a peer may be
  // seconds late into its first launch)
  int rccl_timeout_ms = 30000;
  mutable bool comm_failed = false;   // the communicator was given up: every later failure of this context reports SC_ERR_RCCL
  int peer_connect_ms = 120000;   // how long sc_ctx_comm_peer_connect waits for every peer's hello
  // fault injection (tests): delay every sharded launch of this rank by dbg_delay_ms on the host; dbg_skip_tag = 1
  // makes the next sharded launch skip a tag (a rank that is out of step with its peers)
  int dbg_delay_ms = 0;
  int dbg_skip_tag = 0;
  int dbg_fold_grab = 0;
  int pool_contiguous = 0;   // pool blocks of >= 1 MiB are asked for as PHYSICALLY contiguous VRAM (hipDeviceMallocContiguous)   // measurements: tiles per draw of fold_kernel's four-wave launches (0 = 1, the default; 4 = round 3's behaviour)

  // multi-device handle (sc_ctx_create_multi, engine/multi.inc).  The handle itself owns no device state: `subs` are ordinary
  // contexts, one per entry of devices[], shard d = rank d of world subs.size() on Transport::kLocal; `mrt` holds one
  // launcher thread per further device.  In a sub, `parent` points back.
  // h_tail / d_tail (every context): pinned, device-mapped tail buffers, kTailSlots slots of [table][kTailEntries] words; a
  // prover holds one slot from creation to destruction (on a handle: the same index on every shard, handed out from the
  // handle's tail_free)
  std::vector<sc_ctx*> subs;
  sc_ctx* parent = nullptr;
  struct MultiRuntime* mrt = nullptr;
  u64* h_tail = nullptr;
  u64* d_tail = nullptr;
  std::vector<int> tail_free;
  // sc_table_evaluate_many: the points of one launch (16 x 64 words), pinned staging and its device copy (allocated on first use)
  u64* h_points = nullptr;
  u64* d_points = nullptr;

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
  // a table of a multi-device handle: parts[d] = the d-th contiguous shard, a table of subs[d]; d stays null, len = whole length
  std::vector<sc_table*> parts;
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
  if (ctx && ctx->comm_failed && code == SC_ERR_HIP) code = SC_ERR_RCCL;   // (whatever broke behind a communicator that was given up)
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

// entry points a multi-device handle does not serve (the callers' own polynomial types, SURVEY 8f: they run on an ordinary
// context of one device)
#define SC_NO_MULTI(ctx, what)                                                                                        \
  do {                                                                                                                \
    if ((ctx) && is_multi(ctx))                                                                                       \
      return fail(ctx, SC_ERR_UNSUPPORTED, "%s: not available on a multi-device handle (use a context of one device)", what); \
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
  hipError_t e = hipErrorOutOfMemory;
  if (ctx->pool_contiguous && words * sizeof(u64) >= ((size_t)1 << 20)) {
    e = hipExtMallocWithFlags((void**)&p, words * sizeof(u64), hipDeviceMallocContiguous);
    if (e != hipSuccess) (void)hipGetLastError();   // no contiguous range of that size: an ordinary allocation below
  }
  if (e != hipSuccess) e = hipMalloc(&p, words * sizeof(u64));
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

// connecting: the call is the peer connect itself (the only thing an exported, not yet connected context may do)
int set_device(sc_ctx* ctx, bool connecting = false) {
  if (ctx->poisoned)
    return fail(ctx, SC_ERR_STATE, "context is unusable after an earlier HIP failure (%s); destroy it", ctx->err.c_str());
  if (ctx->world > 1 && ctx->transport == Transport::kNone && !connecting)
    return fail(ctx, SC_ERR_STATE, "rank %d of %d has exported its peer region but is not connected (sc_ctx_comm_peer_connect "
                "failed or was not called): connect, or destroy the context", ctx->rank, ctx->world);
  SC_HIP(ctx, hipSetDevice(ctx->device));
  return SC_OK;
}

inline bool is_sharded(const sc_ctx* ctx) { return ctx->transport != Transport::kNone; }
inline bool is_multi(const sc_ctx* ctx) { return !ctx->subs.empty(); }

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

}  // namespace

// the same calls on a multi-device handle (engine/abi_multi.inc); the single-device entry points dispatch to them
static int multi_destroy(sc_ctx* m);
static int multi_set_option(sc_ctx* m, const char* key, int64_t value);
static int multi_synchronize(sc_ctx* m);
static int multi_table_upload(sc_ctx* m, const uint64_t* host, size_t len, sc_table** out);
static int multi_table_generate(sc_ctx* m, uint64_t seed, uint64_t start, size_t len, sc_table** out);
static int multi_table_clone(sc_ctx* m, const sc_table* t, sc_table** out);
static int multi_table_download(sc_ctx* m, const sc_table* t, uint64_t* host, size_t len);
static int multi_table_free(sc_ctx* m, sc_table* t);
static int multi_table_fix_variables(sc_ctx* m, const sc_table* in, const uint64_t* r, size_t k, int order, sc_table** out);
static int multi_table_evaluate(sc_ctx* m, const sc_table* t, const uint64_t* r, size_t n, int order, uint64_t* out);
static int multi_matmul_g_new(sc_ctx* m, const sc_table* A, const sc_table* B, size_t n, const uint64_t* point, sc_table** a_out,
                              sc_table** b_out);
static int multi_table_relabel(sc_ctx* m, const sc_table* in, size_t a, size_t b, size_t k, sc_table** out);
static int multi_gkr_w_to_evaluations(sc_ctx* m, const sc_table* add, const sc_table* mul, const sc_table* w_b, const sc_table* w_c, sc_table** out);
static int multi_gkr_w_round_sums(sc_ctx* m, const sc_table* add, const sc_table* mul, const sc_table* w_b, const sc_table* w_c, uint64_t out_e[3]);
static int multi_tri_to_evaluations(sc_ctx* m, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len, sc_table** out);
static int multi_tri_round_sums(sc_ctx* m, const sc_table* f1, const sc_table* f2, const sc_table* f3, size_t var_len, uint64_t out_e[3]);
static int multi_prod2_to_evaluations(sc_ctx* m, const sc_table* a, const sc_table* b, sc_table** out);
static int multi_prod2_round_sums(sc_ctx* m, const sc_table* a, const sc_table* b, uint64_t out_e[3]);
static int multi_prod2_sum(sc_ctx* m, const sc_table* a, const sc_table* b, uint64_t* out_c1);
static int multi_prod2_fold_and_sums(sc_ctx* m, const sc_table* a, const sc_table* b, const uint64_t r[1], sc_table** a_out,
                                     sc_table** b_out, uint64_t out_e[3]);

// The rest of the engine, in the order it builds on itself (one translation unit; the parts are not stand-alone headers)
#include "engine/launch.inc"
#include "engine/collectives.inc"
#include "engine/multi.inc"
#include "engine/table_helpers.inc"
#include "engine/abi_context.inc"
#include "engine/abi_sharding.inc"
#include "engine/abi_tables.inc"
#include "engine/abi_prover.inc"
#include "engine/abi_gkr.inc"
#include "engine/abi_triangle.inc"
#include "engine/abi_restrict.inc"
#include "engine/abi_multi.inc"
