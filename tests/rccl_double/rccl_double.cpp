// TEST INFRASTRUCTURE ONLY - never on the product path.
//
// A stand-in for librccl between PROCESSES THAT SHARE ONE GPU.  RCCL refuses two ranks on one device ("Duplicate GPU detected"),
// so on the one-GPU boxes of this pool the library's RCCL plane - Transport::kRccl: ncclAllReduce of the split limbs behind
// every sharded pass, ncclAllGather at the tail, ncclCommCount (engine/collectives.inc, launch.inc) - could only ever run with
// one rank, i.e. its N > 1 control flow was dead code until a real multi-GPU node ran it (missing 2).  This
// library exports the seven entry points the product dlopen()s (thaler-study_amd/csrc/sumcheck_hip.hip: load_rccl) with RCCL's
// signatures and semantics as far as the product uses them - ncclUint64 / ncclSum, in-place or out-of-place, "the result is in
// recvbuf in stream order" - and implements them through a POSIX shared-memory segment: copy to the rank's slot, barrier, add
// (or concatenate) the slots on the host, barrier, copy back.  The product selects it the way it would select any other RCCL
// build: environment variable SC_RCCL_LIBRARY = path of the shared object.  Nothing under thaler-study_amd/ knows it exists.
//
// What it is for: 2 / 4 / 8 processes on one device through sc_ctx_comm_init_rccl - sharded proofs, the tail gather,
// comm_nranks, a rank that dies (the survivors' collectives fail with ncclSystemError after SC_RCCL_DOUBLE_TIMEOUT_MS,
// default 20 s, and the library turns that into SC_ERR_RCCL).  What it is NOT: a measurement of RCCL or of xGMI.
//
// SC_RCCL_DOUBLE_ASYNC_HANG=1 makes a missing rank look the way it looks under the real library: the collective call returns
// ncclSuccess and what it queued on the stream never completes (here: a host function that blocks the stream until
// ncclCommAbort), so the product's own bound - option "rccl_timeout_ms": abort, poison, SC_ERR_RCCL - is what ends the wait.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kSlotWords = (size_t)1 << 20;   // 8 MiB per rank and chunk
constexpr uint32_t kMagic = 0x5cdb1e05u;
struct Shm {
  std::atomic<uint32_t> magic, attached, detached, dead;
  std::atomic<uint32_t> count, sense;   // barrier: arrivals of this generation, generation number
  std::atomic<uint64_t> calls[kMaxRanks];   // collectives entered per rank (a mismatch = ranks out of step)
  alignas(4096) uint64_t slot[kMaxRanks][kSlotWords];
};
int timeout_ms() {
  const char* e = getenv("SC_RCCL_DOUBLE_TIMEOUT_MS");
  return e ? atoi(e) : 20000;
}
}  // namespace

struct ncclComm {
  Shm* shm = nullptr;
  int rank = 0, nranks = 1;
  char name[64] = {0};
  std::atomic<int>* aborted = new std::atomic<int>(0);   // (never freed: a blocked host function may still look at it)
};

namespace {
bool barrier(ncclComm* c) {
  Shm* s = c->shm;
  const uint32_t gen = s->sense.load(std::memory_order_acquire);
  if (s->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
    s->count.store(0, std::memory_order_relaxed);
    s->sense.store(gen + 1, std::memory_order_release);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (s->sense.load(std::memory_order_acquire) == gen) {
    if (s->dead.load(std::memory_order_relaxed)) return false;
    if ((++spins & 1023) == 0) {
      if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) {
        s->dead.store(1, std::memory_order_relaxed);   // a rank is missing: every later collective fails everywhere
        return false;
      }
      std::this_thread::yield();
    }
  }
  return true;
}
bool usable(ncclComm_t c, ncclDataType_t t) { return c && c->shm && t == ncclUint64 && !c->shm->dead.load(); }

void block_until_abort(void* p) {
  std::atomic<int>* aborted = (std::atomic<int>*)p;
  const auto t0 = std::chrono::steady_clock::now();
  while (!aborted->load(std::memory_order_acquire) &&
         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 120.0)   // (a test that forgets to abort still ends)
    std::this_thread::sleep_for(std::chrono::microseconds(200));
}
// a rank is missing: fail at once (the default), or behave like the real library - success now, a stream that never gets on
ncclResult_t rank_missing(ncclComm* c, hipStream_t stream) {
  const char* e = getenv("SC_RCCL_DOUBLE_ASYNC_HANG");
  if (!e || !*e || *e == '0') return ncclSystemError;
  return hipLaunchHostFunc(stream, block_until_abort, c->aborted) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  static std::atomic<unsigned> serial{0};
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, sizeof(id->internal), "/sc_rccl_double_%d_%u_%llx", (int)getpid(), serial.fetch_add(1), (unsigned long long)now);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  if (strncmp(id.internal, "/sc_rccl_double_", 16) != 0) return ncclInvalidArgument;   // not an id of this library
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, sizeof(Shm)) != 0) {
    close(fd);
    return ncclSystemError;
  }
  void* p = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  ncclComm* c = new ncclComm;
  c->shm = (Shm*)p;   // (a fresh segment is zero-filled: the atomics start at 0)
  c->rank = rank;
  c->nranks = nranks;
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  c->shm->magic.store(kMagic);
  c->shm->attached.fetch_add(1);
  // like ncclCommInitRank, return when every rank has joined
  const auto t0 = std::chrono::steady_clock::now();
  while (c->shm->attached.load() < (uint32_t)nranks) {
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) {
      c->shm->dead.store(1);
      munmap(p, sizeof(Shm));
      shm_unlink(c->name);
      delete c;
      return ncclSystemError;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->shm) {
    const bool last = c->shm->detached.fetch_add(1) + 1 == (uint32_t)c->nranks;
    munmap(c->shm, sizeof(Shm));
    if (last) shm_unlink(c->name);
  }
  delete c;
  return ncclSuccess;
}

// ncclCommAbort: whatever this communicator left on a stream gives way; the communicator is gone afterwards
ncclResult_t ncclCommAbort(ncclComm_t c) {
  if (!c) return ncclSuccess;
  c->aborted->store(1, std::memory_order_release);
  if (c->shm) c->shm->dead.store(1);
  return ncclCommDestroy(c);
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t* st) {
  if (!c || !st) return ncclInvalidArgument;
  *st = ncclSuccess;   // (like the real library between ranks of one node: a peer that stops is not noticed, its kernel just waits)
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count) {
  if (!c || !count) return ncclInvalidArgument;
  *count = c->nranks;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t stream) {
  if (!usable(c, datatype) || op != ncclSum || !sendbuff || !recvbuff) return c && c->shm && c->shm->dead.load() ? rank_missing(c, stream) : ncclInvalidArgument;
  c->shm->calls[c->rank].fetch_add(1);
  std::vector<uint64_t> total;
  for (size_t off = 0; off < count; off += kSlotWords) {
    const size_t n = count - off < kSlotWords ? count - off : kSlotWords;
    if (hipMemcpyAsync(c->shm->slot[c->rank], (const uint64_t*)sendbuff + off, n * 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return ncclUnhandledCudaError;
    if (!barrier(c)) return rank_missing(c, stream);
    total.assign(n, 0);
    for (int q = 0; q < c->nranks; ++q)
      for (size_t i = 0; i < n; ++i) total[i] += c->shm->slot[q][i];
    if (!barrier(c)) return ncclSystemError;   // everyone has read: the slots may be rewritten
    if (hipMemcpyAsync((uint64_t*)recvbuff + off, total.data(), n * 8, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t c, hipStream_t stream) {
  if (!usable(c, datatype) || !sendbuff || !recvbuff) return c && c->shm && c->shm->dead.load() ? rank_missing(c, stream) : ncclInvalidArgument;
  c->shm->calls[c->rank].fetch_add(1);
  for (size_t off = 0; off < sendcount; off += kSlotWords) {
    const size_t n = sendcount - off < kSlotWords ? sendcount - off : kSlotWords;
    if (hipMemcpyAsync(c->shm->slot[c->rank], (const uint64_t*)sendbuff + off, n * 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return ncclUnhandledCudaError;
    if (!barrier(c)) return rank_missing(c, stream);
    for (int q = 0; q < c->nranks; ++q)
      if (hipMemcpyAsync((uint64_t*)recvbuff + (size_t)q * sendcount + off, c->shm->slot[q], n * 8, hipMemcpyHostToDevice, stream) != hipSuccess)
        return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
  }
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "rccl double: a HIP copy failed";
    case ncclSystemError: return "rccl double: a rank did not arrive (dead or out of step)";
    case ncclInvalidArgument: return "rccl double: invalid argument (only ncclUint64 / ncclSum are implemented)";
    default: return "rccl double: error";
  }
}

}  // extern "C"
