// Data-parallel exchange points of the step behind the C ABI: an RCCL communicator (xGMI inside a node) with its OWN
// side HIP stream and hipEvent fences, so a caller that is not PyTorch gets the same "all-reduce under the backward pass" overlap the
// engine gets from torch.distributed.  Replaces DeepSpeed ZeRO-2's bucketed gradient reduction (scripts/zero2.json:16-22) and
// diffdist's target all_gather (ola_utils.py:96-106).
//
//   vp_comm_allreduce_async(buf): [compute stream] record `ready` -> [side stream] wait(ready); ncclAllReduce(sum, in place); record `done`
//   vp_comm_wait:                 [compute stream] wait(done of every bucket issued since the last wait)   (the host never blocks)
//   vp_comm_allgather:            on the caller's stream (its result is needed by the next kernel anyway)
//
// RCCL is resolved at run time (dlsym on the already-loaded librccl — PyTorch ships its own copy and a process must not hold two —
// else dlopen("librccl.so" / "librccl.so.1")), so libvisper_hip.so carries no link-time dependency on it.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
  bool ok;
};

Rccl g_rccl = {};

bool load_rccl() {
  if (g_rccl.ok) return true;
  void* h = RTLD_DEFAULT;
  if (!dlsym(h, "ncclCommInitRank")) {
    h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      vp_set_error("vp_comm: RCCL not found (%s)", dlerror());
      return false;
    }
  }
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.AllGather;
  if (!g_rccl.ok) vp_set_error("vp_comm: RCCL symbols missing");
  return g_rccl.ok;
}

constexpr int MAX_PENDING = 64;

struct VpComm {
  ncclComm_t comm;
  hipStream_t side;
  hipEvent_t ready[MAX_PENDING], done[MAX_PENDING];
  int n_pending, rank, world;
};

#define VP_HIP(x)                                                          \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      vp_set_error("vp_comm: %s -> %s", #x, hipGetErrorString(e_));        \
      return VP_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)
#define VP_NCCL(x)                                                                                          \
  do {                                                                                                      \
    ncclResult_t r_ = (x);                                                                                  \
    if (r_ != ncclSuccess) {                                                                                \
      vp_set_error("vp_comm: %s -> %s", #x, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); \
      return VP_ERR_HIP;                                                                                    \
    }                                                                                                       \
  } while (0)

bool dtype_of(int dtype, ncclDataType_t* out) {
  if (dtype == 0) *out = ncclFloat32;
  else if (dtype == 1) *out = ncclBfloat16;
  else if (dtype == 2) *out = ncclInt32;
  else return false;
  return true;
}

}  // namespace

extern "C" {

int vp_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

// rank 0 creates the id (128 bytes) and hands it to the other ranks out of band (environment, file, MPI, torch.distributed store)
int vp_comm_unique_id(void* id_out) {
  VP_REQUIRE(id_out, VP_ERR_BAD_ARG, "vp_comm_unique_id: null");
  if (!load_rccl()) return VP_ERR_HIP;
  ncclUniqueId id;
  VP_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return VP_OK;
}

// collective over all ranks: one process per GPU, the calling thread's current HIP device is the rank's GPU
int vp_comm_init(int rank, int world, const void* id, void** comm_out) {
  VP_REQUIRE(comm_out && id && world >= 1 && rank >= 0 && rank < world, VP_ERR_BAD_ARG, "vp_comm_init: bad args");
  if (!load_rccl()) return VP_ERR_HIP;
  VpComm* c = new VpComm();
  c->rank = rank; c->world = world; c->n_pending = 0;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  {
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
      vp_set_error("vp_comm_init: ncclCommInitRank -> %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
      delete c;
      return VP_ERR_HIP;
    }
  }
  bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
  int made = 0;
  for (; ok && made < MAX_PENDING; ++made)
    ok = hipEventCreateWithFlags(&c->ready[made], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->done[made], hipEventDisableTiming) == hipSuccess;
  if (!ok) {                                                // roll back whatever was created: nothing leaks on a failed init
    vp_set_error("vp_comm_init: could not create the side stream / fence events");
    (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return VP_ERR_HIP;
  }
  *comm_out = c;
  return VP_OK;
}

// In-place sum all-reduce of `count` elements (dtype 0 = fp32, 1 = bf16, 2 = int32) on the communicator's side stream, ordered after
// everything already queued on `compute_stream`; returns at once.  The buffer must not be touched until vp_comm_wait.
int vp_comm_allreduce_async(void* comm, void* buf, long count, int dtype, hipStream_t compute_stream) {
  VpComm* c = (VpComm*)comm;
  ncclDataType_t dt;
  VP_REQUIRE(c && buf && count > 0 && dtype_of(dtype, &dt), VP_ERR_BAD_ARG, "vp_comm_allreduce_async: bad args");
  if (c->n_pending == MAX_PENDING) {                                        // recycle: fold the outstanding buckets into the stream order
    for (int i = 0; i < c->n_pending; ++i) VP_HIP(hipStreamWaitEvent(c->side, c->done[i], 0));
    VP_HIP(hipEventRecord(c->done[0], c->side));
    c->n_pending = 1;
  }
  const int k = c->n_pending;
  VP_HIP(hipEventRecord(c->ready[k], compute_stream));
  VP_HIP(hipStreamWaitEvent(c->side, c->ready[k], 0));
  VP_NCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dt, ncclSum, c->comm, c->side));
  VP_HIP(hipEventRecord(c->done[k], c->side));
  c->n_pending = k + 1;
  return VP_OK;
}

// `compute_stream` waits (on the device) for every all-reduce issued since the last wait
int vp_comm_wait(void* comm, hipStream_t compute_stream) {
  VpComm* c = (VpComm*)comm;
  VP_REQUIRE(c, VP_ERR_BAD_ARG, "vp_comm_wait: null communicator");
  for (int i = 0; i < c->n_pending; ++i) VP_HIP(hipStreamWaitEvent(compute_stream, c->done[i], 0));
  c->n_pending = 0;
  return VP_OK;
}

// recv[world * count] = rank-ordered concatenation of every rank's send[count] (dist_collect's order: ola_utils.py:104-106), on `stream`
int vp_comm_allgather(void* comm, const void* send, void* recv, long count, int dtype, hipStream_t stream) {
  VpComm* c = (VpComm*)comm;
  ncclDataType_t dt;
  VP_REQUIRE(c && send && recv && count > 0 && dtype_of(dtype, &dt), VP_ERR_BAD_ARG, "vp_comm_allgather: bad args");
  VP_NCCL(g_rccl.AllGather(send, recv, (size_t)count, dt, c->comm, stream));
  return VP_OK;
}

int vp_comm_info(void* comm, int* rank, int* world) {
  VpComm* c = (VpComm*)comm;
  VP_REQUIRE(c, VP_ERR_BAD_ARG, "vp_comm_info: null communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return VP_OK;
}

int vp_comm_destroy(void* comm) {
  VpComm* c = (VpComm*)comm;
  if (!c) return VP_OK;
  (void)hipStreamSynchronize(c->side);
  for (int i = 0; i < MAX_PENDING; ++i) {
    (void)hipEventDestroy(c->ready[i]);
    (void)hipEventDestroy(c->done[i]);
  }
  if (g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
  (void)hipStreamDestroy(c->side);
  delete c;
  return VP_OK;
}

}  // extern "C"
