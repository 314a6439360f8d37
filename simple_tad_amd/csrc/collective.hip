// RCCL behind the C ABI: tad_rccl_{unique_id, init, allreduce, broadcast, destroy}.
//
// The reference exchanges gradients through torch.nn.parallel.DistributedDataParallel (run_class_finetuning.py:446-448; process group
// from utils.init_distributed_mode, utils.py:283-333, backend 'nccl' at :325) and broadcasts the initial parameters inside the DDP
// constructor.  The Python host of this library keeps that route (parallel.DataParallel -> torch.distributed, whose "nccl" backend IS
// RCCL on ROCm); these entry points give a C / C++ host the same two collectives over xGMI without torch: one communicator per process
// (one process per GPU), in-place sum / average all-reduce of a flat gradient bucket on a caller-supplied stream, broadcast from a root.
//
// librccl.so.1 is resolved at the first call with dlopen by SONAME, so a process that already carries RCCL (torch's bundled copy has the
// same soname) shares that instance and the library has no link-time dependency: it loads on hosts without RCCL and the other entry
// points stay usable.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include "common.h"

TAD_NAMESPACE_BEGIN
namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

RcclApi& api() {
  static RcclApi a = [] {
    RcclApi r;
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) return r;
#define TAD_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym))
    TAD_SYM(GetUniqueId, "ncclGetUniqueId");
    TAD_SYM(CommInitRank, "ncclCommInitRank");
    TAD_SYM(CommDestroy, "ncclCommDestroy");
    TAD_SYM(CommCount, "ncclCommCount");
    TAD_SYM(AllReduce, "ncclAllReduce");
    TAD_SYM(Broadcast, "ncclBroadcast");
    TAD_SYM(GetErrorString, "ncclGetErrorString");
#undef TAD_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.AllReduce && r.Broadcast && r.GetErrorString;
    return r;
  }();
  return a;
}

int need_api(const char* what) {
  if (!api().ok) {
    set_error("%s: librccl.so.1 could not be loaded (%s)", what, api().handle ? "symbols missing" : dlerror());
    return TAD_ELAUNCH;
  }
  return TAD_OK;
}

int rccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return TAD_OK;
  set_error("%s: RCCL error %d (%s)", what, (int)r, api().GetErrorString(r));
  return TAD_ELAUNCH;
}

bool dtype_of(int dtype, ncclDataType_t* out) {
  if (dtype == TAD_F32) { *out = ncclFloat32; return true; }
  if (dtype == TAD_BF16) { *out = ncclBfloat16; return true; }
  return false;
}

}  // namespace
TAD_NAMESPACE_END

using namespace tad;

extern "C" {

int tad_rccl_unique_id(void* id128) {
  TAD_REQUIRE(id128, "rccl_unique_id: null pointer");
  static_assert(sizeof(ncclUniqueId) == TAD_RCCL_UNIQUE_ID_BYTES, "unique id size");
  if (int rc = need_api("rccl_unique_id")) return rc;
  ncclUniqueId id;
  if (int rc = rccl_check(api().GetUniqueId(&id), "rccl_unique_id")) return rc;
  memcpy(id128, &id, sizeof(id));
  return TAD_OK;
}

int tad_rccl_init(const void* id128, int nranks, int rank, tad_comm_t* comm) {
  TAD_REQUIRE(id128 && comm, "rccl_init: null pointer");
  TAD_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rccl_init: rank %d outside [0, %d)", rank, nranks);
  if (int rc = need_api("rccl_init")) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  if (int rc = rccl_check(api().CommInitRank(&c, nranks, id, rank), "rccl_init")) return rc;  // binds to the calling thread's current device
  *comm = (tad_comm_t)c;
  return TAD_OK;
}

int tad_rccl_world_size(tad_comm_t comm, int* nranks) {
  TAD_REQUIRE(comm && nranks, "rccl_world_size: null pointer");
  if (int rc = need_api("rccl_world_size")) return rc;
  return rccl_check(api().CommCount((ncclComm_t)comm, nranks), "rccl_world_size");
}

int tad_rccl_allreduce(tad_comm_t comm, void* buf, size_t count, int dtype, int average, tad_stream_t stream) {
  TAD_REQUIRE(comm && buf, "rccl_allreduce: null pointer");
  ncclDataType_t dt;
  TAD_REQUIRE(dtype_of(dtype, &dt), "rccl_allreduce: bad dtype %d", dtype);
  if (int rc = need_api("rccl_allreduce")) return rc;
  if (count == 0) return TAD_OK;
  return rccl_check(api().AllReduce(buf, buf, count, dt, average ? ncclAvg : ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "rccl_allreduce");
}

int tad_rccl_broadcast(tad_comm_t comm, void* buf, size_t count, int dtype, int root, tad_stream_t stream) {
  TAD_REQUIRE(comm && buf, "rccl_broadcast: null pointer");
  ncclDataType_t dt;
  TAD_REQUIRE(dtype_of(dtype, &dt), "rccl_broadcast: bad dtype %d", dtype);
  TAD_REQUIRE(root >= 0, "rccl_broadcast: bad root %d", root);
  if (int rc = need_api("rccl_broadcast")) return rc;
  if (count == 0) return TAD_OK;
  return rccl_check(api().Broadcast(buf, buf, count, dt, root, (ncclComm_t)comm, (hipStream_t)stream), "rccl_broadcast");
}

int tad_rccl_destroy(tad_comm_t comm) {
  if (!comm) return TAD_OK;
  if (int rc = need_api("rccl_destroy")) return rc;
  return rccl_check(api().CommDestroy((ncclComm_t)comm), "rccl_destroy");
}

}  // extern "C"
