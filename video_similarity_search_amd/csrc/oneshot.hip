// slic_oneshot / slic_allreduce_oneshot_f64: the sharded k-means iteration's ONE exchange as a one-shot all-to-all over peer-mapped
// memory instead of a ring collective (the all-reduce of [K*D sums | K counts | n_changed], 2 MB in fp64 at
// K = 500, D = 512, is latency-bound at 8 GPUs: a ring all-reduce is 2 (W - 1) dependent steps over xGMI's point-to-point links, where
// every GPU can simply WRITE its partial to its 7 peers at once — one link each — and add the 8 partials it holds itself).
// Replaces, with slic_kmeans_lloyd_local / _global around it, the rank-0 k-means + barrier of /root/reference/online_train.py:625-662;
// the process model is the reference's (one process per GPU, misc/distributed_helper.py:30-37).
//
//   setup   : every rank allocates an INBOX [parity 2][source rank W][max_n doubles] + FLAGS [parity 2][source W][chunk] in device
//             memory that other devices may write while a kernel of this one reads it (uncached / fine-grained allocation), exports it
//             with hipIpcGetMemHandle; the W handles travel by any channel (torch.distributed, a file); every rank maps its peers' inboxes
//             (hipIpcOpenMemHandle: xGMI peer access between GPUs; between two processes on ONE GPU — how a one-GPU box tests this — the same
//             memory through a second mapping).
//   exchange: ONE kernel of at most 128 workgroups, exchange number seq = 1, 2, ...: workgroup b copies chunk b (4096 doubles; then b + grid, ...) of the rank's payload into
//             inbox[seq & 1][rank] of every peer, makes the stores visible at system scope (release fence), stores seq into the peer's flag
//             (seq & 1, rank, b); then waits — one lane per peer, a bounded spin — until its OWN flags (seq & 1, peer, b) hold seq, acquires,
//             and writes the sum of the W chunks, added in RANK order (the rank's own from the payload itself), back to the payload.
//             fp64 sums of a few fp32-born values are exact, so every rank ends with bit-identical numbers (the oracle's n_shards = -W).
//             Double buffering by parity is enough: a peer's push of seq + 2 into the same slot follows ITS wait for this rank's push of
//             seq + 1, which stream order puts behind this rank's reads of seq.
//   bounded : a peer that never pushes (a lost rank) ends the spin after timeout_ms (wall_clock64); the kernel then writes seq to a host-
//             mapped status word and completes with whatever the inbox holds — nothing hangs; slic_oneshot_check reports SLIC_ETIMEOUT
//             from then on and the caller raises.
#include "common.h"
#include <string.h>

#define ONESHOT_MAX_WORLD 16
#define ONESHOT_CHUNK 4096            // doubles per workgroup step (32 KB)
#define ONESHOT_MAX_GRID 128          // workgroups per exchange kernel: larger payloads walk their chunks (see the kernel)

struct slic_oneshot {
  int world, rank;
  int64_t max_n;                      // doubles per payload
  int nchunk_max;
  char* local;                        // this rank's region: flags, then the inbox
  size_t bytes, inbox_off;
  char* peer[ONESHOT_MAX_WORLD];      // every rank's region as mapped here (peer[rank] == local)
  unsigned seq;                       // exchanges issued so far
  int* status_host;                   // host-mapped word: 0, or the first exchange number whose wait ran out
  int* status_dev;
  int timeout_ms;
  int mem_kind;                       // 0 uncached, 1 fine-grained, 2 plain hipMalloc
  bool connected;
};

struct OneshotPeers {
  char* p[ONESHOT_MAX_WORLD];
};

__global__ __launch_bounds__(256) void oneshot_allreduce_f64_kernel(OneshotPeers peers, int world, int rank, int64_t max_n, int nchunk_max,
                                                                    size_t inbox_off, double* __restrict__ buf, int64_t n, unsigned seq,
                                                                    long long timeout_ticks, int* __restrict__ status) {
  const int tid = threadIdx.x;
  const int par = (int)(seq & 1u);
  const int nchunk = (int)((n + ONESHOT_CHUNK - 1) / ONESHOT_CHUNK);
  // The grid is BOUNDED (<= ONESHOT_MAX_GRID workgroups, far below what one GPU keeps resident even when several ranks share it) and a
  // workgroup walks chunks b, b + grid, ...: the wait for chunk b needs workgroup (b mod grid) of every peer to have run its earlier
  // chunks, never a workgroup that is not resident yet — the spin cannot depend on co-residency of a large grid.
  for (int b = blockIdx.x; b < nchunk; b += gridDim.x) {
  const int64_t e0 = (int64_t)b * ONESHOT_CHUNK;
  const int64_t e1 = e0 + ONESHOT_CHUNK < n ? e0 + ONESHOT_CHUNK : n;
  // ---- 1. this rank's chunk into every peer's inbox[par][rank] (16 bytes per lane)
  for (int p = 0; p < world; ++p) {
    if (p == rank) continue;
    double* dst = (double*)(peers.p[p] + inbox_off) + ((int64_t)par * world + rank) * max_n;
    for (int64_t e = e0 + 2 * tid; e < e1; e += 2 * 256) {
      if (e + 1 < e1) {
        const double2 v = *(const double2*)(buf + e);
        *(double2*)(dst + e) = v;
      } else dst[e] = buf[e];                                   // the last element of an odd-length payload
    }
  }
  __threadfence_system();                                      // the stores above are visible to the peers' devices before the flags are
  __syncthreads();
  if (tid < world && tid != rank) {
    unsigned* f = (unsigned*)peers.p[tid] + ((int64_t)par * world + rank) * nchunk_max + b;
    __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- 2. wait for chunk b of every peer (one lane per peer); bounded
  int timed_out = 0;
  if (tid < world && tid != rank) {
    const unsigned* f = (const unsigned*)peers.p[rank] + ((int64_t)par * world + tid) * nchunk_max + b;
    const long long t0 = wall_clock64();
    for (;;) {
      const unsigned v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((int)(v - seq) >= 0) break;
      if (wall_clock64() - t0 > timeout_ticks) { timed_out = 1; break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (__syncthreads_or(timed_out)) {
    // (a plain system-scope store to the host-mapped word: the first exchange that ran out is what matters, and the host refuses further
    //  exchanges once the word is set — a later overwrite by a concurrent workgroup of the same exchange writes the same number)
    if (tid == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0)
      __hip_atomic_store(status, (int)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                // system scope: no stale line of the inbox is read below
  // ---- 3. the W chunks added in rank order
  const double* inbox = (const double*)(peers.p[rank] + inbox_off) + (int64_t)par * world * max_n;
  for (int64_t e = e0 + 2 * tid; e < e1; e += 2 * 256) {
    const bool pair = e + 1 < e1;
    double2 acc = {0.0, 0.0};
    double2 own;
    own.x = buf[e];
    own.y = pair ? buf[e + 1] : 0.0;
    for (int s = 0; s < world; ++s) {
      double2 v;
      if (s == rank) v = own;
      else {
        const double* q = inbox + (int64_t)s * max_n + e;
        v.x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v.y = pair ? __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
      }
      acc.x += v.x;
      acc.y += v.y;
    }
    buf[e] = acc.x;
    if (pair) buf[e + 1] = acc.y;
  }
  __syncthreads();                                             // the next chunk's timed_out / flag lanes start from a quiet workgroup
  }
}

static size_t oneshot_flag_bytes(int world, int nchunk_max) { return slic_align_up((size_t)2 * world * nchunk_max * sizeof(unsigned), 4096); }

extern "C" int slic_oneshot_create(int world, int rank, int64_t max_n, int timeout_ms, slic_oneshot** out, void* handle_out) {
  SLIC_REQUIRE(out && handle_out && world >= 1 && world <= ONESHOT_MAX_WORLD && rank >= 0 && rank < world && max_n > 0 && timeout_ms >= 0,
               "slic_oneshot_create: bad args (1 <= world <= %d)", ONESHOT_MAX_WORLD);
  static_assert(sizeof(hipIpcMemHandle_t) == SLIC_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  *out = nullptr;
  slic_oneshot* c = new slic_oneshot();
  c->world = world; c->rank = rank;
  c->max_n = (max_n + 1) / 2 * 2;
  c->nchunk_max = (int)slic_cdiv(c->max_n, ONESHOT_CHUNK);
  c->inbox_off = oneshot_flag_bytes(world, c->nchunk_max);
  c->bytes = c->inbox_off + (size_t)2 * world * c->max_n * sizeof(double);
  c->seq = 0; c->timeout_ms = timeout_ms; c->connected = false;
  for (int p = 0; p < ONESHOT_MAX_WORLD; ++p) c->peer[p] = nullptr;
  // memory another device writes while kernels of this one read it must be uncached or fine-grained: plain (coarse-grained) device memory
  // is not kept coherent with remote writers — a system-scope load may be served by the XCD's L2 — so a rank could spin on a stale flag or
  // add a stale payload.  With peers there is NO plain-memory fallback: the create fails and the caller keeps the RCCL
  // all-reduce.  A world of one has no remote writer; plain memory is fine there.
  void* mem = nullptr;
  c->mem_kind = 0;
  if (hipExtMallocWithFlags(&mem, c->bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    c->mem_kind = 1;
    if (hipExtMallocWithFlags(&mem, c->bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      const hipError_t e2 = hipGetLastError();
      c->mem_kind = 2;
      if (world > 1) {
        slic_set_error("slic_oneshot_create: neither uncached nor fine-grained device memory is available for the %zu-byte inbox (%s); "
                       "coarse-grained memory is not coherent with peer writes — use the RCCL all-reduce (exchange='allreduce')",
                       c->bytes, hipGetErrorString(e2));
        delete c;
        return SLIC_EHIP;
      }
      if (hipMalloc(&mem, c->bytes) != hipSuccess) {
        slic_set_error("slic_oneshot_create: cannot allocate %zu bytes of device memory: %s", c->bytes, hipGetErrorString(hipGetLastError()));
        delete c;
        return SLIC_EHIP;
      }
    }
  }
  c->local = (char*)mem;
  c->peer[rank] = c->local;
  hipError_t e = hipMemset(mem, 0, c->inbox_off);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->status_host, 64, hipHostMallocMapped);
  if (e == hipSuccess) {
    c->status_host[0] = 0;
    e = hipHostGetDevicePointer((void**)&c->status_dev, c->status_host, 0);
  }
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, mem);
  if (e != hipSuccess) {
    slic_set_error("slic_oneshot_create: %s (memory kind %d)", hipGetErrorString(e), c->mem_kind);
    (void)hipFree(mem);
    if (c->status_host) (void)hipHostFree(c->status_host);
    delete c;
    return SLIC_EHIP;
  }
  memcpy(handle_out, &h, sizeof(h));
  *out = c;
  return SLIC_OK;
}

extern "C" int slic_oneshot_connect(slic_oneshot* c, const void* all_handles) {
  SLIC_REQUIRE(c && all_handles && !c->connected, "slic_oneshot_connect: bad args / already connected");
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)all_handles + (size_t)p * SLIC_IPC_HANDLE_BYTES, sizeof(h));
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      slic_set_error("slic_oneshot_connect: hipIpcOpenMemHandle of rank %d's inbox on rank %d -> %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", p, c->rank,
                     hipGetErrorString(e));
      for (int q = 0; q < p; ++q)
        if (q != c->rank && c->peer[q]) { (void)hipIpcCloseMemHandle(c->peer[q]); c->peer[q] = nullptr; }
      return SLIC_EHIP;
    }
    c->peer[p] = (char*)ptr;
  }
  c->connected = true;
  return SLIC_OK;
}

extern "C" int slic_allreduce_oneshot_f64(slic_oneshot* c, double* buf, int64_t n, void* stream) {
  SLIC_REQUIRE(c && buf && n > 0 && n <= c->max_n && ((uintptr_t)buf % 16) == 0,
               "slic_allreduce_oneshot_f64: bad args (n <= the communicator's max_n, buf 16-byte aligned)");
  SLIC_REQUIRE(c->connected || c->world == 1, "slic_allreduce_oneshot_f64: slic_oneshot_connect has not run");
  SLIC_REQUIRE(c->status_host[0] == 0, "slic_allreduce_oneshot_f64: exchange %d of this communicator timed out (a peer is missing); create a new one",
               c->status_host[0]);
  OneshotPeers pp;
  for (int p = 0; p < ONESHOT_MAX_WORLD; ++p) pp.p[p] = c->peer[p];
  const unsigned seq = ++c->seq;
  const long long ticks = c->timeout_ms > 0 ? (long long)c->timeout_ms * 100000ll : (1ll << 62);      // wall_clock64: 100 MHz
  const int64_t nchunks = slic_cdiv(n, ONESHOT_CHUNK);
  const unsigned nb = (unsigned)(nchunks < ONESHOT_MAX_GRID ? nchunks : ONESHOT_MAX_GRID);
  oneshot_allreduce_f64_kernel<<<dim3(nb), dim3(256), 0, (hipStream_t)stream>>>(pp, c->world, c->rank, c->max_n, c->nchunk_max, c->inbox_off, buf, n,
                                                                               seq, ticks, c->status_dev);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// After the stream (or an event behind the exchange) has completed: SLIC_OK, or SLIC_ETIMEOUT when a wait of this communicator ran out —
// the payloads since then are garbage and the caller must raise.  Never blocks.
extern "C" int slic_oneshot_check(slic_oneshot* c) {
  SLIC_REQUIRE(c, "slic_oneshot_check: null");
  const int s = *(volatile int*)c->status_host;
  if (s != 0) {
    slic_set_error("slic_oneshot: exchange %d on rank %d of %d did not receive every peer's payload within %d ms (a peer is missing or stuck)", s,
                   c->rank, c->world, c->timeout_ms);
    return SLIC_ETIMEOUT;
  }
  return SLIC_OK;
}

extern "C" int slic_oneshot_info(const slic_oneshot* c, int* out /* [4]: world, rank, memory kind (0 uncached, 1 fine-grained, 2 plain), exchanges issued */) {
  SLIC_REQUIRE(c && out, "slic_oneshot_info: null");
  out[0] = c->world; out[1] = c->rank; out[2] = c->mem_kind; out[3] = (int)c->seq;
  return SLIC_OK;
}

extern "C" int slic_oneshot_destroy(slic_oneshot* c) {
  if (!c) return SLIC_OK;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  if (c->local) (void)hipFree(c->local);
  if (c->status_host) (void)hipHostFree(c->status_host);
  (void)hipGetLastError();
  delete c;
  return SLIC_OK;
}
