// RCCL handle of the C ABI ("no global state beyond a per-device handle for RCCL"): tl_comm_* / tl_allreduce /
// tl_reduce_scatter / tl_all_gather.  What the data-parallel trainer exchanges per step (models/synthesis_trainer.py:226-227 of the
// reference sit between loss.backward() and optimizer.step(); here: one flat fp32 gradient buffer, the gathered W_hh factor rows, the
// slices of the gate-row-sharded label LSTM) are plain fp32 device buffers, so the handle takes float pointers and element counts.
//
// librccl is NOT a link-time dependency of libtonal_hip.so: a single-GPU user must be able to load the library on a box without
// it, and a PyTorch process already holds a copy (torch/lib/librccl.so) that the handle should share instead of loading the system
// one beside it.  The entry points are resolved on first use: from the librccl the process has loaded, else librccl.so.1 / .so.
#include "tonal_common.h"
#include <dlfcn.h>
#include <link.h>
#include <string.h>

namespace tl {
namespace {

// the slice of rccl.h this file needs (ABI of RCCL 2.x: opaque communicator, 128-byte id, C enums)
typedef struct ncclComm* comm_t;
typedef struct { char internal[128]; } unique_id;
enum { NCCL_SUCCESS = 0, NCCL_FLOAT32 = 7, NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3 };

struct api {
  int (*GetUniqueId)(unique_id*);
  int (*CommInitRank)(comm_t*, int, unique_id, int);
  int (*CommDestroy)(comm_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
  int (*ReduceScatter)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t);
  const char* (*GetErrorString)(int);
  bool ok;
};

int find_loaded(struct dl_phdr_info* info, size_t, void* data) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
    strncpy((char*)data, info->dlpi_name, 1023);
    return 1;
  }
  return 0;
}

// state: 0 loaded, 1 no librccl in the process or on the loader path, 2 a librccl without one of the entry points
struct loaded { api a; int state; const char* missing; };

loaded load_rccl() {
  loaded l = {};
  char path[1024] = {0};
  dl_iterate_phdr(find_loaded, path);
  void* h = path[0] ? dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD) : nullptr;
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    l.state = 1;
    return l;
  }
  api& a = l.a;
#define TL_SYM(field, name)                                              \
  a.field = (decltype(a.field))dlsym(h, name);                           \
  if (!a.field && !l.missing) l.missing = name;
  TL_SYM(GetUniqueId, "ncclGetUniqueId")
  TL_SYM(CommInitRank, "ncclCommInitRank")
  TL_SYM(CommDestroy, "ncclCommDestroy")
  TL_SYM(AllReduce, "ncclAllReduce")
  TL_SYM(ReduceScatter, "ncclReduceScatter")
  TL_SYM(AllGather, "ncclAllGather")
  TL_SYM(GetErrorString, "ncclGetErrorString")
#undef TL_SYM
  a.ok = l.missing == nullptr;
  l.state = a.ok ? 0 : 2;
  return l;
}

// resolved once, by whichever thread comes first (C++11 function-local static: concurrent first calls wait for it)
const loaded& rccl_state() {
  static const loaded l = load_rccl();
  return l;
}

int fail(const api* a, const char* what, int rc) {
  set_error("%s: %s", what, a->GetErrorString(rc));
  return TL_ELAUNCH;
}

}  // namespace
}  // namespace tl

#define TL_RCCL(a)                                                                                     \
  const tl::loaded& a##_l = tl::rccl_state();                                                          \
  if (a##_l.state == 1) {                                                                              \
    tl::set_error("RCCL is not available in this process (librccl.so.1 could not be loaded)");         \
    return TL_ENODEV;                                                                                  \
  }                                                                                                    \
  if (a##_l.state == 2) {                                                                              \
    tl::set_error("the librccl of this process has no %s (RCCL 2.x entry points needed)", a##_l.missing); \
    return TL_ENODEV;                                                                                  \
  }                                                                                                    \
  const tl::api* a = &a##_l.a;

extern "C" int tl_comm_unique_id(void* id128) {
  using namespace tl;
  TL_REQUIRE(id128 != nullptr, "comm_unique_id: null buffer (128 bytes)");
  TL_RCCL(a);
  unique_id id;
  const int rc = a->GetUniqueId(&id);
  if (rc != NCCL_SUCCESS) return fail(a, "ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return TL_OK;
}

extern "C" int tl_comm_init(void** comm, int rank, int nranks, const void* id128) {
  using namespace tl;
  TL_REQUIRE(comm != nullptr && id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: comm, id and 0 <= rank < nranks needed");
  TL_RCCL(a);
  unique_id id;
  memcpy(id.internal, id128, sizeof(id.internal));
  comm_t c = nullptr;
  const int rc = a->CommInitRank(&c, nranks, id, rank);       // on the calling thread's current HIP device
  if (rc != NCCL_SUCCESS) return fail(a, "ncclCommInitRank", rc);
  *comm = (void*)c;
  return TL_OK;
}

extern "C" int tl_comm_destroy(void* comm) {
  using namespace tl;
  TL_REQUIRE(comm != nullptr, "comm_destroy: null communicator");
  TL_RCCL(a);
  const int rc = a->CommDestroy((comm_t)comm);
  return rc == NCCL_SUCCESS ? TL_OK : fail(a, "ncclCommDestroy", rc);
}

extern "C" int tl_allreduce(void* comm, const float* send, float* recv, int64_t count, int op, void* stream) {
  using namespace tl;
  TL_REQUIRE(comm && send && recv && count > 0, "allreduce: communicator, buffers and count > 0 needed");
  TL_REQUIRE(op >= 0 && op <= 2, "allreduce: op 0 sum, 1 max, 2 min");
  TL_RCCL(a);
  const int nop = op == 0 ? NCCL_SUM : op == 1 ? NCCL_MAX : NCCL_MIN;
  const int rc = a->AllReduce(send, recv, (size_t)count, NCCL_FLOAT32, nop, (comm_t)comm, (hipStream_t)stream);
  return rc == NCCL_SUCCESS ? TL_OK : fail(a, "ncclAllReduce", rc);
}

extern "C" int tl_reduce_scatter(void* comm, const float* send, float* recv, int64_t recv_count, void* stream) {
  using namespace tl;
  TL_REQUIRE(comm && send && recv && recv_count > 0, "reduce_scatter: communicator, buffers and recv_count > 0 needed");
  TL_RCCL(a);
  const int rc = a->ReduceScatter(send, recv, (size_t)recv_count, NCCL_FLOAT32, NCCL_SUM, (comm_t)comm, (hipStream_t)stream);
  return rc == NCCL_SUCCESS ? TL_OK : fail(a, "ncclReduceScatter", rc);
}

extern "C" int tl_all_gather(void* comm, const float* send, float* recv, int64_t send_count, void* stream) {
  using namespace tl;
  TL_REQUIRE(comm && send && recv && send_count > 0, "all_gather: communicator, buffers and send_count > 0 needed");
  TL_RCCL(a);
  const int rc = a->AllGather(send, recv, (size_t)send_count, NCCL_FLOAT32, (comm_t)comm, (hipStream_t)stream);
  return rc == NCCL_SUCCESS ? TL_OK : fail(a, "ncclAllGather", rc);
}
