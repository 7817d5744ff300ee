// The contrastive exchange of the hot path over RCCL, behind the C ABI (`vl_allgather_embed`,
// `vl_reducescatter_grad`): what a NON-Python host bound to include/vitlens_hip.h needs to run the multi-GPU path - one process
// per GPU, one communicator per process.  (The Python host of this repository makes the same calls through torch.distributed,
// backend "nccl" = RCCL: vitlens_hip/step.py TorchComm; `AbiComm` there drives THESE entries instead.)
//
// Replaces: gather_features (open_clip/loss.py:20-78: dist.all_gather of the [b, 768] features, the reduce-scatter of the
// feature gradients under --gather-with-grad) and DDP's gradient all-reduce (training/*_main.py: DistributedDataParallel).
//
// RCCL is resolved at RUN time (dlopen of librccl.so.1, the soname PyTorch-ROCm's own copy carries as well: inside a torch process
// the already loaded library is used): libvitlens_hip.so has no link-time dependency on it and loads on a box without RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "vitlens_hip.h"

extern "C" int vl_set_error(const char* msg);

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
    }
    if (!r.h) return;
    auto sym = [&](const char* n) { return dlsym(r.h, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.ReduceScatter = (decltype(r.ReduceScatter))sym("ncclReduceScatter");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.ReduceScatter && r.AllReduce && r.GetErrorString;
  });
  return r;
}

int fail(const char* what, ncclResult_t e) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error");
  return vl_set_error(buf);
}

}  // namespace

struct vl_comm {
  ncclComm_t c;
  int rank, world;
};

#define VL_NEED_RCCL(fn) do { if (!rccl().ok) return vl_set_error(fn ": librccl.so.1 could not be loaded (RCCL is resolved at run time)"); } while (0)

extern "C" int vl_comm_unique_id(void* id128) {
  VL_NEED_RCCL("vl_comm_unique_id");
  if (!id128) return vl_set_error("vl_comm_unique_id: null buffer");
  ncclUniqueId id;
  const ncclResult_t e = rccl().GetUniqueId(&id);
  if (e != ncclSuccess) return fail("vl_comm_unique_id", e);
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int vl_comm_create(vl_comm_t* comm, const void* id128, int rank, int world) {
  VL_NEED_RCCL("vl_comm_create");
  if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return vl_set_error("vl_comm_create: need comm, a 128-byte id and 0 <= rank < world");
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  const ncclResult_t e = rccl().CommInitRank(&c, world, id, rank);          // (collective: every rank of the node calls it)
  if (e != ncclSuccess) return fail("vl_comm_create", e);
  *comm = new vl_comm{c, rank, world};
  return 0;
}

extern "C" int vl_comm_destroy(vl_comm_t comm) {
  if (!comm) return 0;
  const ncclResult_t e = rccl().ok ? rccl().CommDestroy(comm->c) : ncclSuccess;
  delete comm;
  return e == ncclSuccess ? 0 : fail("vl_comm_destroy", e);
}

extern "C" int vl_allgather_embed(vl_comm_t comm, const float* local, float* gathered, long count, hipStream_t stream) {
  VL_NEED_RCCL("vl_allgather_embed");
  if (!comm || !local || !gathered || count <= 0) return vl_set_error("vl_allgather_embed: null operand or empty payload");
  const ncclResult_t e = rccl().AllGather(local, gathered, (size_t)count, ncclFloat, comm->c, stream);
  return e == ncclSuccess ? 0 : fail("vl_allgather_embed", e);
}

extern "C" int vl_reducescatter_grad(vl_comm_t comm, const float* full, float* mine, long count_per_rank, hipStream_t stream) {
  VL_NEED_RCCL("vl_reducescatter_grad");
  if (!comm || !full || !mine || count_per_rank <= 0) return vl_set_error("vl_reducescatter_grad: null operand or empty payload");
  const ncclResult_t e = rccl().ReduceScatter(full, mine, (size_t)count_per_rank, ncclFloat, ncclSum, comm->c, stream);
  return e == ncclSuccess ? 0 : fail("vl_reducescatter_grad", e);
}

extern "C" int vl_allreduce_grad(vl_comm_t comm, float* buf, long count, hipStream_t stream) {
  VL_NEED_RCCL("vl_allreduce_grad");
  if (!comm || !buf || count <= 0) return vl_set_error("vl_allreduce_grad: null operand or empty payload");
  const ncclResult_t e = rccl().AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, comm->c, stream);
  return e == ncclSuccess ? 0 : fail("vl_allreduce_grad", e);
}
