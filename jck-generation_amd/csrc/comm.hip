// RCCL gradient all-reduce behind the C ABI: {init, enqueue, wait} (include/jckgan.h).
//
// The reference's multi-GPU form is torch's DistributedDataParallel around the two networks: the averaged gradients are what
// optimizer_d.step() / optimizer_g.step() consume (train/dcgan_trainer.py:180,189, train/cgan_trainer.py:204,212).  Here a
// network's gradients are ONE flat fp32 arena, so the exchange is one ncclAllReduce(SUM) per arena (or per slice of one), enqueued
// on the communicator's OWN HIP stream behind an event of the stream that produced the gradients, and awaited by an event - no host
// thread, no host synchronisation: the engine's streams keep running underneath.
//
// librccl is resolved at the first jck_comm_* call, not at load time: a single-GPU process never maps it.  A copy that is already
// in the process (PyTorch's) is preferred, so that one RCCL serves both.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/jckgan.h"

void jck_set_error(const std::string& s);

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)
    if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);      // a copy the process already holds
  const char* env = getenv("JCK_RCCL_LIB");
  if (!g_rccl.h && env) g_rccl.h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
  for (const char* n : names)
    if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!g_rccl.h) g_rccl.h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!g_rccl.h) { g_rccl.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
  auto sym = [&](const char* s) -> void* {
    void* p = dlsym(g_rccl.h, s);
    if (!p && g_rccl.why.empty()) g_rccl.why = std::string("librccl lacks ") + s;
    return p;
  };
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(sym("ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
}
bool rccl_ready() {
  std::call_once(g_rccl_once, load_rccl);
  return g_rccl.why.empty();
}
}  // namespace

#define CFAIL(code, msg)                                    \
  do {                                                      \
    jck_set_error(std::string(__func__) + ": " + (msg));    \
    return (code);                                          \
  } while (0)
#define CHIP(expr)                                                                                            \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) CFAIL(JCK_E_HIP, std::string("HIP error ") + hipGetErrorString(e_) + " at " #expr); \
  } while (0)
#define CNCCL(expr)                                                                                                   \
  do {                                                                                                                \
    ncclResult_t r_ = (expr);                                                                                         \
    if (r_ != ncclSuccess) CFAIL(JCK_E_HIP, std::string("RCCL error ") + g_rccl.GetErrorString(r_) + " at " #expr);   \
  } while (0)

#define JCK_COMM_TICKETS 8
struct jck_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;                   // the collectives run here, never on a caller's stream
  hipEvent_t ready[JCK_COMM_TICKETS] = {}, done[JCK_COMM_TICKETS] = {};
  int world = 1, rank = 0, next = 0, device = 0;
};

extern "C" int jck_comm_unique_id(unsigned char* id128) {
  if (!id128) CFAIL(JCK_E_ARG, "id is NULL");
  if (!rccl_ready()) CFAIL(JCK_E_HIP, g_rccl.why);
  ncclUniqueId id;
  CNCCL(g_rccl.GetUniqueId(&id));
  static_assert(sizeof(id.internal) == JCK_COMM_ID_BYTES, "JCK_COMM_ID_BYTES must be RCCL's NCCL_UNIQUE_ID_BYTES");
  memcpy(id128, id.internal, JCK_COMM_ID_BYTES);
  return JCK_OK;
}

extern "C" int jck_comm_create(jck_comm** out, const unsigned char* id128, int world, int rank) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) CFAIL(JCK_E_ARG, "need out, id, 0 <= rank < world");
  if (!rccl_ready()) CFAIL(JCK_E_HIP, g_rccl.why);
  jck_comm* c = new jck_comm();
  c->world = world; c->rank = rank;
  auto fail = [&](int code) { jck_comm_destroy(c); return code; };
  if (hipGetDevice(&c->device) != hipSuccess) { jck_set_error("jck_comm_create: hipGetDevice failed"); return fail(JCK_E_HIP); }
  ncclUniqueId id;
  memcpy(id.internal, id128, JCK_COMM_ID_BYTES);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);      // collective: returns when every rank has called it
  if (r != ncclSuccess) { jck_set_error(std::string("jck_comm_create: ncclCommInitRank: ") + g_rccl.GetErrorString(r)); c->comm = nullptr; return fail(JCK_E_HIP); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { jck_set_error("jck_comm_create: stream"); return fail(JCK_E_HIP); }
  for (int i = 0; i < JCK_COMM_TICKETS; ++i)
    if (hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) != hipSuccess) { jck_set_error("jck_comm_create: events"); return fail(JCK_E_HIP); }
  *out = c;
  return JCK_OK;
}

extern "C" int jck_comm_world(const jck_comm* c) { return c ? c->world : 0; }

// SUM all-reduce of buf[0, count) in place, behind everything `producer_stream` holds at the time of the call.
extern "C" int jck_comm_allreduce_enqueue(jck_comm* c, float* buf, size_t count, void* producer_stream, int* ticket) {
  if (!c || !c->comm || !buf || !ticket) CFAIL(JCK_E_ARG, "need comm, buf, ticket");
  const int t = c->next;
  c->next = (c->next + 1) % JCK_COMM_TICKETS;
  CHIP(hipEventRecord(c->ready[t], (hipStream_t)producer_stream));
  CHIP(hipStreamWaitEvent(c->stream, c->ready[t], 0));
  if (count) CNCCL(g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, c->comm, c->stream));
  CHIP(hipEventRecord(c->done[t], c->stream));
  *ticket = t;
  return JCK_OK;
}

// `consumer_stream` waits (on the device) for the all-reduce of `ticket`; the host does not block.
extern "C" int jck_comm_wait(jck_comm* c, int ticket, void* consumer_stream) {
  if (!c || ticket < 0 || ticket >= JCK_COMM_TICKETS) CFAIL(JCK_E_ARG, "bad ticket");
  CHIP(hipStreamWaitEvent((hipStream_t)consumer_stream, c->done[ticket], 0));
  return JCK_OK;
}

extern "C" int jck_comm_destroy(jck_comm* c) {
  if (!c) return JCK_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  for (int i = 0; i < JCK_COMM_TICKETS; ++i) {
    if (c->ready[i]) (void)hipEventDestroy(c->ready[i]);
    if (c->done[i]) (void)hipEventDestroy(c->done[i]);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return JCK_OK;
}
