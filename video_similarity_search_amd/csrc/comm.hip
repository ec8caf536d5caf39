// slic_comm / slic_allreduce_*: the one collective of the sharded k-means iteration (the all-reduce of
// [K*D sums | K counts | n_changed] over xGMI) behind the C ABI — a thin wrapper over RCCL's ncclAllReduce on the caller's stream.
// Stands in for the NCCL process group the reference sets up in /root/reference/misc/distributed_helper.py:30-64 (and replaces the
// rank-0 k-means + barrier of online_train.py:625-662).
// RCCL is bound at first use with dlopen — the copy the process ALREADY has (PyTorch-ROCm's: RTLD_NOLOAD first, so that no second
// RCCL is mapped beside it), else /opt/rocm's — so the library itself has no link-time dependency on it and loads on hosts without
// RCCL; every entry point then fails loudly.
// Bounded waits: a collective that waits for a lost peer must not hang the job.  The communicator is created NON-BLOCKING
// (ncclCommInitRankConfig, blocking = 0) and polled (ncclCommGetAsyncError) against a deadline; slic_comm_wait does the same for
// work already enqueued on a stream (event + async-error poll).  On a timeout or an asynchronous error the communicator is ABORTED
// (ncclCommAbort: enqueued kernels are released) and the call returns SLIC_ETIMEOUT / SLIC_EHIP: every rank can then exit non-zero.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <time.h>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRankConfig)(ncclComm_t*, int, ncclUniqueId, int, ncclConfig_t*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommFinalize)(ncclComm_t) = nullptr;          // optional (RCCL >= 2.14): flushes a non-blocking communicator before the destroy
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int bind_rccl() {
  if (g_rccl.h) return SLIC_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {        // the copy already mapped into the process, if any
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (h) break;
  }
  if (!h)
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
  r.CommInitRankConfig = (decltype(r.CommInitRankConfig))dlsym(h, "ncclCommInitRankConfig");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.CommFinalize = (decltype(r.CommFinalize))dlsym(h, "ncclCommFinalize");
  r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
  r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRankConfig || !r.AllReduce || !r.CommDestroy || !r.CommAbort || !r.CommGetAsyncError || !r.GetErrorString) {
    slic_set_error("slic_comm: librccl.so lacks an expected symbol");
    return SLIC_EHIP;
  }
  g_rccl = r;
  return SLIC_OK;
}

double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
void nap() {
  timespec ts{0, 200000};      // 0.2 ms
  nanosleep(&ts, nullptr);
}
}  // namespace

struct slic_comm {
  ncclComm_t comm;
  int world, rank;
  int timeout_ms;              // deadline of every wait of this communicator's own (0 = none)
  bool dead;                   // aborted: every later call fails at once
};

#define SLIC_NCCL_CHECK(expr)                                                                        \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess && _r != ncclInProgress) {                                                \
      slic_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));      \
      return SLIC_EHIP;                                                                             \
    }                                                                                               \
  } while (0)

// poll the communicator until its pending operation has finished; on a deadline or an asynchronous error abort it
static int settle(slic_comm* c, int timeout_ms, const char* who) {
  const double t0 = now_ms();
  for (;;) {
    ncclResult_t st = ncclSuccess;
    ncclResult_t rc = g_rccl.CommGetAsyncError(c->comm, &st);
    if (rc != ncclSuccess) st = rc;
    if (st == ncclSuccess) return SLIC_OK;
    if (st != ncclInProgress) {
      slic_set_error("%s: RCCL reports %s on rank %d of %d; communicator aborted", who, g_rccl.GetErrorString(st), c->rank, c->world);
      g_rccl.CommAbort(c->comm);
      c->dead = true;
      return SLIC_EHIP;
    }
    if (timeout_ms > 0 && now_ms() - t0 > timeout_ms) {
      slic_set_error("%s: no progress within %d ms on rank %d of %d (a peer is missing or stuck); communicator aborted", who, timeout_ms,
                     c->rank, c->world);
      g_rccl.CommAbort(c->comm);
      c->dead = true;
      return SLIC_ETIMEOUT;
    }
    nap();
  }
}

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

extern "C" int slic_comm_create_timeout(const void* id, int world, int rank, int timeout_ms, slic_comm** out) {
  SLIC_REQUIRE(id && out && world >= 1 && rank >= 0 && rank < world && timeout_ms >= 0, "slic_comm_create: bad args");
  *out = nullptr;
  int rc = bind_rccl();
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
  cfg.blocking = 0;                                                 // every call returns at once; progress is polled
  slic_comm* s = new slic_comm{nullptr, world, rank, timeout_ms, false};
  ncclResult_t r = g_rccl.CommInitRankConfig(&s->comm, world, uid, rank, &cfg);      // on the calling thread's current HIP device
  if (r != ncclSuccess && r != ncclInProgress) {
    slic_set_error("slic_comm_create: ncclCommInitRankConfig -> %s", g_rccl.GetErrorString(r));
    delete s;
    return SLIC_EHIP;
  }
  rc = settle(s, timeout_ms, "slic_comm_create");
  if (rc) {
    delete s;                                                       // already aborted
    return rc;
  }
  *out = s;
  return SLIC_OK;
}

extern "C" int slic_comm_create(const void* id, int world, int rank, slic_comm** out) {
  return slic_comm_create_timeout(id, world, rank, 0, out);         // 0 = wait as long as it takes (the original contract)
}

static int allreduce(slic_comm* c, void* buf, int64_t n, ncclDataType_t dt, void* stream) {
  SLIC_REQUIRE(c && buf && n > 0, "slic_allreduce: bad args");
  SLIC_REQUIRE(!c->dead, "slic_allreduce: the communicator was aborted");
  SLIC_NCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)n, dt, ncclSum, c->comm, (hipStream_t)stream));
  // non-blocking communicator: the enqueue itself may still be in progress (the first collective also connects the peers): settle it
  // here, under the communicator's deadline
  return settle(c, c->timeout_ms, "slic_allreduce");
}

extern "C" int slic_allreduce_f32(slic_comm* c, float* buf, int64_t n, void* stream) { return allreduce(c, buf, n, ncclFloat32, stream); }
extern "C" int slic_allreduce_f64(slic_comm* c, double* buf, int64_t n, void* stream) { return allreduce(c, buf, n, ncclFloat64, stream); }

// Wait, with a deadline, until `event` (a hipEvent_t the caller recorded behind the work it wants to see finished — e.g. behind ONE
// iteration's collective and status read-back, so that work enqueued later keeps running ahead) has completed.
// SLIC_OK: done.  SLIC_ETIMEOUT: the deadline passed (a peer never joined a collective): the communicator is aborted, which releases
// the stuck kernel, and every later call on it fails.  SLIC_EHIP: RCCL reported an asynchronous error (communicator aborted) or HIP did.
static int wait_event(slic_comm* c, hipEvent_t ev, int timeout_ms, const char* who) {
  const double t0 = now_ms();
  int rc = SLIC_OK;
  for (;;) {
    hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) {
      slic_set_error("%s: hipEventQuery -> %s", who, hipGetErrorString(e));
      rc = SLIC_EHIP;
      break;
    }
    ncclResult_t st = ncclSuccess;
    g_rccl.CommGetAsyncError(c->comm, &st);
    if (st != ncclSuccess && st != ncclInProgress) {
      slic_set_error("%s: RCCL reports %s on rank %d of %d; communicator aborted", who, g_rccl.GetErrorString(st), c->rank, c->world);
      g_rccl.CommAbort(c->comm);
      c->dead = true;
      rc = SLIC_EHIP;
      break;
    }
    if (timeout_ms > 0 && now_ms() - t0 > timeout_ms) {
      slic_set_error("%s: the awaited work did not finish within %d ms on rank %d of %d (a peer is missing or stuck); communicator aborted",
                     who, timeout_ms, c->rank, c->world);
      g_rccl.CommAbort(c->comm);
      c->dead = true;
      rc = SLIC_ETIMEOUT;
      break;
    }
    nap();
  }
  (void)hipGetLastError();                                          // hipErrorNotReady from the queries is not an error of ours
  return rc;
}

extern "C" int slic_comm_wait_event(slic_comm* c, void* event, int timeout_ms) {
  SLIC_REQUIRE(c && event && timeout_ms >= 0, "slic_comm_wait_event: bad args");
  SLIC_REQUIRE(!c->dead, "slic_comm_wait_event: the communicator was aborted");
  return wait_event(c, (hipEvent_t)event, timeout_ms, "slic_comm_wait_event");
}

// The same for everything enqueued on `stream` so far (an event recorded here, now).
extern "C" int slic_comm_wait(slic_comm* c, void* stream, int timeout_ms) {
  SLIC_REQUIRE(c && timeout_ms >= 0, "slic_comm_wait: bad args");
  SLIC_REQUIRE(!c->dead, "slic_comm_wait: the communicator was aborted");
  hipEvent_t ev;
  SLIC_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, (hipStream_t)stream);
  if (e != hipSuccess) {
    (void)hipEventDestroy(ev);
    slic_set_error("slic_comm_wait: hipEventRecord -> %s", hipGetErrorString(e));
    return SLIC_EHIP;
  }
  const int rc = wait_event(c, ev, timeout_ms, "slic_comm_wait");
  (void)hipEventDestroy(ev);
  return rc;
}

// give up on a communicator at once (ncclCommAbort): kernels it has enqueued are released, the handle is freed.  This is what a
// process-exit finalizer calls: it never blocks on a peer.
extern "C" int slic_comm_abort(slic_comm* c) {
  if (!c) return SLIC_OK;
  if (g_rccl.h && !c->dead) g_rccl.CommAbort(c->comm);
  delete c;
  return SLIC_OK;
}

// Orderly teardown of a NON-BLOCKING communicator: finalize (flush what it has enqueued), poll that to completion under the
// communicator's deadline (10 s when it was created without one), then destroy.  A peer that is gone, or an asynchronous error,
// ends in ncclCommAbort instead of a wait without a deadline; the handle is freed either way.
extern "C" int slic_comm_destroy(slic_comm* c) {
  if (!c) return SLIC_OK;
  int rc = SLIC_OK;
  if (g_rccl.h && !c->dead) {
    const int deadline = c->timeout_ms > 0 ? c->timeout_ms : 10000;
    if (g_rccl.CommFinalize) {
      ncclResult_t r = g_rccl.CommFinalize(c->comm);
      if (r != ncclSuccess && r != ncclInProgress) {
        slic_set_error("slic_comm_destroy: ncclCommFinalize -> %s; communicator aborted", g_rccl.GetErrorString(r));
        g_rccl.CommAbort(c->comm);
        c->dead = true;
        rc = SLIC_EHIP;
      }
    }
    if (!c->dead) rc = settle(c, deadline, "slic_comm_destroy");   // aborts on a deadline / asynchronous error
    if (!c->dead) {
      ncclResult_t r = g_rccl.CommDestroy(c->comm);
      if (r != ncclSuccess && r != ncclInProgress) {
        slic_set_error("slic_comm_destroy: ncclCommDestroy -> %s", g_rccl.GetErrorString(r));
        rc = SLIC_EHIP;
      }
    }
  }
  delete c;
  return rc;
}
