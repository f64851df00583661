// dalm_comm_*: the few collectives of the sharded in-batch negatives on RCCL, owned by the extension.
//
// What crosses GPUs per step (the reference's DDP all-reduce is dalm/training/rag_e2e/
// train_rage2e.py:416-418,471 - it never gathers embeddings): all-gather of the [B_l, D] f32 embeddings, a
// [B_l,4] stats all-gather, the 1-float token count and the LoRA gradient buckets (SUM).  All of them run on a
// stream the caller names (dalm_comm_*_on: what the Python host code uses - stream-ordered, capturable, no library-owned
// stream), or on a side HIP stream owned by the communicator with explicit ordering against the caller's streams
// (dalm_comm_wait_stream / dalm_comm_stream_wait: hipEvents; created on first use), so a gather can overlap the other tower.
//
// RCCL is bound at run time (dlopen/dlsym): the copy already loaded in the process (torch ships one) is reused,
// /opt/rocm/lib/librccl.so is the fallback; libdalm_hip.so itself has no link-time dependency on RCCL.
// One process per GPU; the 128-byte ncclUniqueId travels through the caller (file / store / env).
#include "common.hpp"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclInt8 = 0, kNcclFloat32 = 7 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {   // a copy already mapped into the process first (no second RCCL instance)
      x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (x.handle) break;
    }
    if (!x.handle)
      for (const char* n : names) {
        x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (x.handle) break;
      }
    if (!x.handle) {
      const char* why = dlerror();   // a second dlerror() call returns NULL: read it once
      x.error = std::string("librccl.so not found: ") + (why ? why : "");
      return x;
    }
    auto sym = [&](const char* s) { return dlsym(x.handle, s); };
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(sym("ncclAllReduce"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.AllGather || !x.AllReduce)
      x.error = "librccl.so lacks one of ncclGetUniqueId/CommInitRank/CommDestroy/AllGather/AllReduce";
    return x;
  }();
  return r;
}

int nccl_fail(int rc, const char* fn) {
  Rccl& r = rccl();
  dalm::set_error(std::string(fn) + ": RCCL: " + (r.GetErrorString ? r.GetErrorString(rc) : "error") + " (" + std::to_string(rc) + ")");
  return 1000 + rc;   // positive: runtime failure (1000 + ncclResult_t)
}
int hip_fail(hipError_t e, const char* fn) {
  dalm::set_error(std::string(fn) + ": " + hipGetErrorString(e));
  return static_cast<int>(e);
}

}  // namespace

struct dalm_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  int rank = 0, world = 1, device = 0;
  // record + wait on the shared event pair is one critical section: the main thread's gathers and the autograd thread's
  // gradient-bucket hooks may both order themselves against the side stream
  std::mutex mu;
};

using namespace dalm;

extern "C" int dalm_comm_unique_id(void* id128) {
  DALM_REQUIRE(id128, DALM_E_NULL, "null pointer argument");
  Rccl& r = rccl();
  if (!r.error.empty()) { set_error(std::string(__func__) + ": " + r.error); return 999; }
  ncclUniqueId id;
  if (int rc = r.GetUniqueId(&id); rc != kNcclSuccess) return nccl_fail(rc, __func__);
  memcpy(id128, id.internal, 128);
  return 0;
}

extern "C" int dalm_comm_init(dalm_comm_t** out, const void* id128, int rank, int world, int device) {
  DALM_REQUIRE(out && id128, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(world >= 1 && rank >= 0 && rank < world && device >= 0, DALM_E_SHAPE, "need 0 <= rank < world, device >= 0");
  Rccl& r = rccl();
  if (!r.error.empty()) { set_error(std::string(__func__) + ": " + r.error); return 999; }
  if (hipError_t e = hipSetDevice(device); e != hipSuccess) return hip_fail(e, __func__);
  dalm_comm* c = new dalm_comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  if (int rc = r.CommInitRank(&c->comm, world, id, rank); rc != kNcclSuccess) { delete c; return nccl_fail(rc, __func__); }
  // the side stream and its events are created on first use of the side-stream entry points (ensure_side_stream): a
  // process that only uses the *_on forms never owns an extra HIP stream
  *out = c;
  return 0;
}

namespace {
hipError_t ensure_side_stream(dalm_comm* c) {      // call with c->mu held
  if (c->stream) return hipSuccess;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
  return e;
}
}  // namespace

extern "C" int dalm_comm_destroy(dalm_comm_t* c) {
  if (!c) return 0;
  hipError_t e = c->stream ? hipStreamSynchronize(c->stream) : hipDeviceSynchronize();
  rccl().CommDestroy(c->comm);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return e == hipSuccess ? 0 : hip_fail(e, __func__);
}

extern "C" int dalm_comm_rank(const dalm_comm_t* c) { return c ? c->rank : -1; }
extern "C" int dalm_comm_world(const dalm_comm_t* c) { return c ? c->world : -1; }

// the communicator's stream waits for everything queued so far on `producer`
extern "C" int dalm_comm_wait_stream(dalm_comm_t* c, dalm_stream_t producer) {
  DALM_REQUIRE(c, DALM_E_NULL, "null communicator");
  std::lock_guard<std::mutex> lock(c->mu);
  if (hipError_t e = ensure_side_stream(c); e != hipSuccess) return hip_fail(e, __func__);
  if (hipError_t e = hipEventRecord(c->ev_in, as_stream(producer)); e != hipSuccess) return hip_fail(e, __func__);
  if (hipError_t e = hipStreamWaitEvent(c->stream, c->ev_in, 0); e != hipSuccess) return hip_fail(e, __func__);
  return 0;
}
// `consumer` waits for everything queued so far on the communicator's stream
extern "C" int dalm_comm_stream_wait(dalm_comm_t* c, dalm_stream_t consumer) {
  DALM_REQUIRE(c, DALM_E_NULL, "null communicator");
  std::lock_guard<std::mutex> lock(c->mu);
  if (hipError_t e = ensure_side_stream(c); e != hipSuccess) return hip_fail(e, __func__);
  if (hipError_t e = hipEventRecord(c->ev_out, c->stream); e != hipSuccess) return hip_fail(e, __func__);
  if (hipError_t e = hipStreamWaitEvent(as_stream(consumer), c->ev_out, 0); e != hipSuccess) return hip_fail(e, __func__);
  return 0;
}

// recv[world * bytes_per_rank] = concatenation over ranks of send[bytes_per_rank]; enqueued on the comm stream
extern "C" int dalm_comm_allgather(dalm_comm_t* c, const void* send, void* recv, size_t bytes_per_rank) {
  DALM_REQUIRE(c && send && recv, DALM_E_NULL, "null pointer argument");
  if (bytes_per_rank == 0) return 0;
  std::lock_guard<std::mutex> lock(c->mu);
  if (hipError_t e = ensure_side_stream(c); e != hipSuccess) return hip_fail(e, __func__);
  if (int rc = rccl().AllGather(send, recv, bytes_per_rank, kNcclInt8, c->comm, c->stream); rc != kNcclSuccess)
    return nccl_fail(rc, __func__);
  return 0;
}

// buf[n] <- sum over ranks (in place); enqueued on the comm stream
extern "C" int dalm_comm_allreduce_sum_f32(dalm_comm_t* c, float* buf, size_t n) {
  DALM_REQUIRE(c && buf, DALM_E_NULL, "null pointer argument");
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lock(c->mu);
  if (hipError_t e = ensure_side_stream(c); e != hipSuccess) return hip_fail(e, __func__);
  if (int rc = rccl().AllReduce(buf, buf, n, kNcclFloat32, kNcclSum, c->comm, c->stream); rc != kNcclSuccess)
    return nccl_fail(rc, __func__);
  return 0;
}

// The same collectives enqueued on a stream of the CALLER's choice (no side stream, no events): stream-ordered like any
// kernel launch, capturable into the caller's hipGraph, and the caller decides which of its streams overlaps what
// (round 2 measured the library-owned extra stream at +19 % step time on one GPU: one more stream shifts the
// stream -> hardware-queue mapping of the whole process).  RCCL serialises operations of one communicator itself.
extern "C" int dalm_comm_allgather_on(dalm_comm_t* c, const void* send, void* recv, size_t bytes_per_rank,
                                      dalm_stream_t stream) {
  DALM_REQUIRE(c && send && recv, DALM_E_NULL, "null pointer argument");
  if (bytes_per_rank == 0) return 0;
  std::lock_guard<std::mutex> lock(c->mu);
  if (int rc = rccl().AllGather(send, recv, bytes_per_rank, kNcclInt8, c->comm, as_stream(stream)); rc != kNcclSuccess)
    return nccl_fail(rc, __func__);
  return 0;
}

extern "C" int dalm_comm_allreduce_sum_f32_on(dalm_comm_t* c, float* buf, size_t n, dalm_stream_t stream) {
  DALM_REQUIRE(c && buf, DALM_E_NULL, "null pointer argument");
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lock(c->mu);
  if (int rc = rccl().AllReduce(buf, buf, n, kNcclFloat32, kNcclSum, c->comm, as_stream(stream)); rc != kNcclSuccess)
    return nccl_fail(rc, __func__);
  return 0;
}
