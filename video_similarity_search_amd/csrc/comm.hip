// slic_comm / slic_allreduce_*: the one collective of the sharded k-means iteration (the all-reduce of
// [K*D sums | K counts | n_changed] over xGMI) behind the C ABI — a thin wrapper over RCCL's ncclAllReduce on the caller's stream.
// RCCL is bound at first use with dlopen (the copy the process already has — PyTorch-ROCm's — or /opt/rocm's), so the library
// itself has no link-time dependency on it and loads on hosts without RCCL; every entry point then fails loudly.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int bind_rccl() {
  if (g_rccl.h) return SLIC_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    slic_set_error("slic_comm: librccl.so not found (%s)", dlerror());
    return SLIC_EHIP;
  }
  Rccl r;
  r.h = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) {
    slic_set_error("slic_comm: librccl.so lacks an expected symbol");
    return SLIC_EHIP;
  }
  g_rccl = r;
  return SLIC_OK;
}
}  // namespace

struct slic_comm {
  ncclComm_t comm;
  int world, rank;
};

#define SLIC_NCCL_CHECK(expr)                                                                        \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess) {                                                                        \
      slic_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));      \
      return SLIC_EHIP;                                                                             \
    }                                                                                               \
  } while (0)

extern "C" int slic_comm_unique_id(void* id_out) {
  SLIC_REQUIRE(id_out, "slic_comm_unique_id: null pointer");
  int rc = bind_rccl();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == SLIC_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  SLIC_NCCL_CHECK(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return SLIC_OK;
}

extern "C" int slic_comm_create(const void* id, int world, int rank, slic_comm** out) {
  SLIC_REQUIRE(id && out && world >= 1 && rank >= 0 && rank < world, "slic_comm_create: bad args");
  int rc = bind_rccl();
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c;
  SLIC_NCCL_CHECK(g_rccl.CommInitRank(&c, world, uid, rank));       // on the calling thread's current HIP device
  slic_comm* s = new slic_comm{c, world, rank};
  *out = s;
  return SLIC_OK;
}

static int allreduce(slic_comm* c, void* buf, int64_t n, ncclDataType_t dt, void* stream) {
  SLIC_REQUIRE(c && buf && n > 0, "slic_allreduce: bad args");
  SLIC_NCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)n, dt, ncclSum, c->comm, (hipStream_t)stream));
  return SLIC_OK;
}

extern "C" int slic_allreduce_f32(slic_comm* c, float* buf, int64_t n, void* stream) { return allreduce(c, buf, n, ncclFloat32, stream); }
extern "C" int slic_allreduce_f64(slic_comm* c, double* buf, int64_t n, void* stream) { return allreduce(c, buf, n, ncclFloat64, stream); }

extern "C" int slic_comm_destroy(slic_comm* c) {
  if (!c) return SLIC_OK;
  if (g_rccl.h) SLIC_NCCL_CHECK(g_rccl.CommDestroy(c->comm));
  delete c;
  return SLIC_OK;
}
