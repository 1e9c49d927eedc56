// cvh_comm_*: the data-parallel exchange of the hot path on a communicator of its OWN (RCCL over xGMI), not on torch.distributed's
// ProcessGroupNCCL.  Replaces what utils/ddp_utils.py:47-89 (rendezvous + communicator creation by a dummy all-reduce) and
// main_train.py:91-96 (DistributedDataParallel's reducer) reach through torch.distributed:
//
//   rank 0:    cvh_comm_unique_id(id)          -> 128 opaque bytes, handed to the other ranks through whatever key-value rendezvous the
//                                                 launcher already has (cvnets_amd/comm.py: the TCP store of the env:// rendezvous)
//   every rank: cvh_comm_init(&comm, world, rank, id)    on the device that is current (one process per GPU)
//               cvh_comm_allreduce / broadcast / allgather / reducescatter (comm, ..., stream)   stream-ordered, hipGraph-capturable
//               cvh_comm_destroy(comm)
//
// librccl is opened at run time (dlopen; the copy PyTorch already mapped is reused when there is one), so single-GPU use of
// libcvnets_hip.so neither links nor loads it; types and prototypes come from RCCL's public header.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library itself is opened with dlopen below

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "cvnets_hip.h"

namespace {

typedef ncclUniqueId UniqueId;
typedef ncclComm_t comm_t;
typedef decltype(&ncclGetUniqueId) get_unique_id_fn;
typedef decltype(&ncclCommInitRank) comm_init_rank_fn;
typedef decltype(&ncclCommDestroy) comm_destroy_fn;
typedef decltype(&ncclAllReduce) all_reduce_fn;
typedef decltype(&ncclBroadcast) broadcast_fn;
typedef decltype(&ncclAllGather) all_gather_fn;
typedef decltype(&ncclReduceScatter) reduce_scatter_fn;
typedef decltype(&ncclGetErrorString) get_error_string_fn;

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  all_reduce_fn all_reduce = nullptr;
  broadcast_fn broadcast = nullptr;
  all_gather_fn all_gather = nullptr;
  reduce_scatter_fn reduce_scatter = nullptr;
  get_error_string_fn error_string = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)  // a copy that is already mapped (PyTorch ships one) first: two RCCL instances in one process would not share topology state
      if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL)) != nullptr) break;
    if (r.handle == nullptr)
      for (const char* n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (r.handle == nullptr) {
      fprintf(stderr, "[cvh_comm] librccl not found: %s\n", dlerror());
      return;
    }
    r.get_unique_id = (get_unique_id_fn)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (comm_init_rank_fn)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (comm_destroy_fn)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (all_reduce_fn)dlsym(r.handle, "ncclAllReduce");
    r.broadcast = (broadcast_fn)dlsym(r.handle, "ncclBroadcast");
    r.all_gather = (all_gather_fn)dlsym(r.handle, "ncclAllGather");
    r.reduce_scatter = (reduce_scatter_fn)dlsym(r.handle, "ncclReduceScatter");
    r.error_string = (get_error_string_fn)dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce && r.broadcast && r.all_gather && r.reduce_scatter;
    if (!r.ok) fprintf(stderr, "[cvh_comm] librccl lacks a collective entry point\n");
  });
  return r;
}

inline int nccl_type(int dtype) {
  return dtype == CVH_DT_F32 ? (int)ncclFloat32 : (dtype == CVH_DT_BF16 ? (int)ncclBfloat16 : (dtype == CVH_COMM_BYTES ? (int)ncclUint8 : -1));
}

struct Comm {
  comm_t comm;
  int world, rank, device;
};

std::atomic<long long> g_calls[4];  // all-reduce, broadcast, all-gather, reduce-scatter launches (cvh_comm_counters)

int report(ncclResult_t rc, const char* what) {
  if (rc != ncclSuccess) {
    Rccl& r = rccl();
    fprintf(stderr, "[cvh_comm] %s failed: %s\n", what, r.error_string ? r.error_string(rc) : "?");
    return 1000 + (int)rc;  // RCCL result codes are reported above the hipError_t range
  }
  return 0;
}

}  // namespace

extern "C" int cvh_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int cvh_comm_unique_id(void* id128) {
  Rccl& r = rccl();
  if (!r.ok) return -3;
  if (id128 == nullptr) return -2;
  UniqueId id;
  int rc = report(r.get_unique_id(&id), "ncclGetUniqueId");
  if (rc == 0) memcpy(id128, id.internal, sizeof(id.internal));
  return rc;
}

extern "C" int cvh_comm_init(void** comm, int world, int rank, const void* id128) {
  Rccl& r = rccl();
  if (!r.ok) return -3;
  if (comm == nullptr || id128 == nullptr || world < 1 || rank < 0 || rank >= world) return -2;
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  comm_t c = nullptr;
  int rc = report(r.comm_init_rank(&c, world, id, rank), "ncclCommInitRank");
  if (rc != 0) return rc;
  *comm = new Comm{c, world, rank, dev};
  return 0;
}

extern "C" int cvh_comm_destroy(void* comm) {
  if (comm == nullptr) return 0;
  Comm* c = static_cast<Comm*>(comm);
  int rc = report(rccl().comm_destroy(c->comm), "ncclCommDestroy");
  delete c;
  return rc;
}

extern "C" int cvh_comm_world(void* comm) { return comm ? static_cast<Comm*>(comm)->world : -2; }
extern "C" int cvh_comm_rank(void* comm) { return comm ? static_cast<Comm*>(comm)->rank : -2; }

extern "C" int cvh_comm_allreduce(void* comm, void* buf, long long count, int dtype, int average, void* stream) {
  if (comm == nullptr || buf == nullptr || count < 0 || nccl_type(dtype) < 0 || dtype == CVH_COMM_BYTES) return -2;
  if (count == 0) return 0;
  Comm* c = static_cast<Comm*>(comm);
  g_calls[0].fetch_add(1, std::memory_order_relaxed);
  return report(rccl().all_reduce(buf, buf, (size_t)count, (ncclDataType_t)nccl_type(dtype), average ? ncclAvg : ncclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int cvh_comm_broadcast(void* comm, void* buf, long long count, int dtype, int root, void* stream) {
  if (comm == nullptr || buf == nullptr || count < 0 || nccl_type(dtype) < 0) return -2;
  Comm* c = static_cast<Comm*>(comm);
  if (root < 0 || root >= c->world) return -2;
  if (count == 0) return 0;
  g_calls[1].fetch_add(1, std::memory_order_relaxed);
  return report(rccl().broadcast(buf, buf, (size_t)count, (ncclDataType_t)nccl_type(dtype), root, c->comm, (hipStream_t)stream), "ncclBroadcast");
}

extern "C" int cvh_comm_allgather(void* comm, const void* send, void* recv, long long count_per_rank, int dtype, void* stream) {
  if (comm == nullptr || send == nullptr || recv == nullptr || count_per_rank < 0 || nccl_type(dtype) < 0) return -2;
  if (count_per_rank == 0) return 0;
  Comm* c = static_cast<Comm*>(comm);
  g_calls[2].fetch_add(1, std::memory_order_relaxed);
  return report(rccl().all_gather(send, recv, (size_t)count_per_rank, (ncclDataType_t)nccl_type(dtype), c->comm, (hipStream_t)stream), "ncclAllGather");
}

extern "C" int cvh_comm_reducescatter(void* comm, const void* send, void* recv, long long count_per_rank, int dtype, void* stream) {
  if (comm == nullptr || send == nullptr || recv == nullptr || count_per_rank < 0 || nccl_type(dtype) < 0 || dtype == CVH_COMM_BYTES) return -2;
  if (count_per_rank == 0) return 0;
  Comm* c = static_cast<Comm*>(comm);
  g_calls[3].fetch_add(1, std::memory_order_relaxed);
  return report(rccl().reduce_scatter(send, recv, (size_t)count_per_rank, (ncclDataType_t)nccl_type(dtype), ncclSum, c->comm, (hipStream_t)stream), "ncclReduceScatter");
}

extern "C" int cvh_comm_counters(int reset, long long* out) {
  for (int i = 0; i < 4; ++i) {
    if (out) out[i] = g_calls[i].load(std::memory_order_relaxed);
    if (reset) g_calls[i].store(0, std::memory_order_relaxed);
  }
  return 0;
}
