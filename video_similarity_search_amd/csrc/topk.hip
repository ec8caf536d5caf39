// Cosine top-k retrieval: fused fp32-MFMA similarity GEMM + per-query streaming top-k.
//
// Replaces the host path of /root/reference/iic_retrieve_clips.py:295-296 (sklearn cosine_distances over
// 10k x 100k, then a full np.argsort per row) and evaluate.py:208-231 (cosine_distances + argpartition):
// the 4 GB distance matrix is never written.  Rows are L2-normalised first (sklearn normalize: zero rows
// stay zero), similarity s = q_hat . g_hat on v_mfma_f32_32x32x2_f32, distance = clip(1 - s, 0, 2).
//
// Workgroup = 128 queries x one slice of the gallery, walked in 128-row tiles (same LDS staging as
// kmeans.hip).  MFMA roles: A = gallery tile (rows), B = query tile (cols)  =>  lane (r, h) holds query r and
// 16 gallery rows per accumulator; the h = 0 lane of each pair owns the query's running top-k (an unsorted
// k-slot binary heap in LDS, root = worst kept entry, mirrored in registers) and also consumes its partner's 16 values.
// A candidate replaces the root and sifts down: O(log k) per insertion, O(k log(Ng/k)) insertions per query and slice in
// expectation.  Order: larger s first, ties -> lower gallery index.
// slic_topk_merge folds the per-slice lists into the final sorted [Nq, k].
#include "common.h"
#include <math.h>
#include <limits.h>
#include <stdlib.h>

#define TK_BQ 128
#define TK_BG 128
#define TK_BK 32
#define TK_PC_MAX 32   // pending candidates per lane between heap drains (fewer when k leaves less LDS)
#define TK_KMAX 88

__device__ __forceinline__ int tk_off(int row, int chunk) {
  return row * TK_BK + ((chunk ^ ((row >> 1) & 7)) << 2);
}

// (score, gallery index) as ONE 64-bit key whose unsigned order is the retrieval order — larger score first, ties -> lower
// index: high word = the order-preserving image of the float, low word = ~index.  A heap step then costs one 8-byte LDS access
// and one 64-bit compare per entry instead of two accesses and a three-way test.
__device__ __forceinline__ unsigned long long tk_key(float s, int i) {
  const unsigned u = __float_as_uint(s);
  const unsigned o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)o << 32) | (unsigned)(~i);
}
__device__ __forceinline__ float tk_key_score(unsigned long long key) {
  const unsigned o = (unsigned)(key >> 32);
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ int tk_key_index(unsigned long long key) { return (int)(~(unsigned)key); }

__device__ __forceinline__ bool tk_better(float s, int i, float t, int ti) {
  return s > t || (s == t && i < ti);
}

__global__ __launch_bounds__(256) void topk_partial_kernel(
    const float* __restrict__ Q, int Nq, const float* __restrict__ G, int Ng, int D, int k, int self_mask,
    int g_per_slice, int TK_PC, float* __restrict__ pval, int32_t* __restrict__ pidx /* [slices][Nq][k] */) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* stage = lds;                                   // [2][2][128*32]
  float* lval = lds + 2 * 2 * TK_BQ * TK_BK;             // [4 waves][k][32]
  int* lidx = (int*)(lval + 4 * k * 32);
  float* pval_l = (float*)(lidx + 4 * k * 32);          // [4 waves][TK_PC][32] pending candidates (value)
  int* pidx_l = (int*)(pval_l + 4 * TK_PC * 32);        // [4 waves][TK_PC][32]              (gallery index)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * TK_BQ;
  const int gbeg = blockIdx.y * g_per_slice;
  const int gend = min(gbeg + g_per_slice, Ng);
  float* myv = lval + wave * k * 32;
  int* myi = lidx + wave * k * 32;
  float* pv = pval_l + wave * TK_PC * 32;
  int* pi = pidx_l + wave * TK_PC * 32;
  int pc = 0;                // this lane's pending count
  if (h == 0)
    for (int s = 0; s < k; ++s) { myv[s * 32 + r] = -INFINITY; myi[s * 32 + r] = INT_MAX; }
  float thr = -INFINITY;     // worst kept entry
  int thr_i = INT_MAX;
  const int q = q0 + 32 * wave + r;

  const int srow = tid >> 2, scp = tid & 3;
  const float* qr[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int qq = q0 + srow + 64 * p;
    if (qq > Nq - 1) qq = Nq - 1;
    qr[p] = Q + (int64_t)qq * D + scp * 8;
  }
  const int nk = (D + TK_BK - 1) / TK_BK;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  for (int g0 = gbeg; g0 < gend; g0 += TK_BG) {
    const float* gr[2];
    bool gv[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int gg = g0 + srow + 64 * p;
      gv[p] = gg < gend;
      if (gg > Ng - 1) gg = Ng - 1;
      gr[p] = G + (int64_t)gg * D + scp * 8;
    }
    f32x4 xq[2][2], xg[2][2];
    auto gload = [&](int kt) {
      const int k0 = kt * TK_BK;
      const bool kin = (k0 + scp * 8) < D;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (kin) { xq[p][0] = *(const f32x4*)(qr[p] + k0); xq[p][1] = *(const f32x4*)(qr[p] + k0 + 4); }
        else { xq[p][0] = z4; xq[p][1] = z4; }
        if (kin && gv[p]) { xg[p][0] = *(const f32x4*)(gr[p] + k0); xg[p][1] = *(const f32x4*)(gr[p] + k0 + 4); }
        else { xg[p][0] = z4; xg[p][1] = z4; }
      }
    };
    auto lwrite = [&](int buf) {
      float* Gs = stage + (buf * 2 + 0) * TK_BQ * TK_BK;
      float* Qs = stage + (buf * 2 + 1) * TK_BQ * TK_BK;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = srow + 64 * p;
        *(f32x4*)&Qs[tk_off(row, 2 * scp)] = xq[p][0];
        *(f32x4*)&Qs[tk_off(row, 2 * scp + 1)] = xq[p][1];
        *(f32x4*)&Gs[tk_off(row, 2 * scp)] = xg[p][0];
        *(f32x4*)&Gs[tk_off(row, 2 * scp + 1)] = xg[p][1];
      }
    };
    f32x16 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
    __syncthreads();           // previous tile's readers are done with the staging buffers
    gload(0);
    lwrite(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) gload(kt + 1);
      const float* Gs = stage + (buf * 2 + 0) * TK_BQ * TK_BK;
      const float* Qs = stage + (buf * 2 + 1) * TK_BQ * TK_BK;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 b = *(const f32x4*)&Qs[tk_off(32 * wave + r, 2 * qd + h)];
        f32x4 a[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) a[ct] = *(const f32x4*)&Gs[tk_off(32 * ct + r, 2 * qd + h)];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct][t], b[t], acc[ct], 0, 0, 0);
      }
      if (kt + 1 < nk) lwrite(buf ^ 1);
      __syncthreads();
    }
    // ---- top-k update: the h == 0 lane of a pair consumes both halves' values.  A value that beats the lane's current
    // worst kept entry is only APPENDED to a small pending buffer (O(1), no dependent LDS chain); the buffers are drained
    // into the heaps when one fills up and at the end of every tile, so the wave pays one sift round per pending entry
    // of its busiest lane instead of one per element step in which any of its 32 queries has a candidate.
    auto flush = [&]() {
      for (int j = 0; j < TK_PC; ++j) {
        const bool act = j < pc;
        if (!__any(act)) break;
        if (act) {
          const float s = pv[j * 32 + r];
          const int gi = pi[j * 32 + r];
          if (tk_better(s, gi, thr, thr_i)) {              // the root may have improved since the append
            // the list is a binary heap whose root (slot 0) is the WORST kept entry: the candidate replaces the root
            // and sifts down (<= log2 k levels)
            int pos = 0;
            for (;;) {
              const int c1 = 2 * pos + 1, c2 = c1 + 1;
              if (c1 >= k) break;
              float cv = myv[c1 * 32 + r];
              int ci = myi[c1 * 32 + r], cs = c1;
              if (c2 < k) {
                const float v2 = myv[c2 * 32 + r];
                const int i2 = myi[c2 * 32 + r];
                if (tk_better(cv, ci, v2, i2)) { cv = v2; ci = i2; cs = c2; }       // the worse child
              }
              if (!tk_better(s, gi, cv, ci)) break;                                    // candidate is not better: stays here
              myv[pos * 32 + r] = cv; myi[pos * 32 + r] = ci;                          // worse child moves up
              pos = cs;
            }
            myv[pos * 32 + r] = s; myi[pos * 32 + r] = gi;
            thr = myv[r]; thr_i = myi[r];
          }
        }
      }
      pc = 0;
    };
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float own = acc[ct][v];
        const float oth = __shfl_xor(own, 32);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const float s = half ? oth : own;
          const int gi = g0 + ct * 32 + (v & 3) + 8 * (v >> 2) + 4 * half;
          const bool cand = (h == 0) && (q < Nq) && (gi < gend) && !(self_mask && gi == q) &&
                            tk_better(s, gi, thr, thr_i);
          if (cand) {
            pv[pc * 32 + r] = s;
            pi[pc * 32 + r] = gi;
            ++pc;
          }
        }
        if (__any(pc > TK_PC - 2)) flush();               // at most two appends per step: pc never exceeds TK_PC
      }
    flush();
  }
  if (h == 0 && q < Nq) {
    float* ov = pval + ((int64_t)blockIdx.y * Nq + q) * k;
    int32_t* oi = pidx + ((int64_t)blockIdx.y * Nq + q) * k;
    for (int s = 0; s < k; ++s) { ov[s] = myv[s * 32 + r]; oi[s] = myi[s * 32 + r]; }
  }
}

// The same kernel with both operand tiles moved HBM -> LDS by the DMA path (buffer_load ... lds): a 2-stage ring that runs
// CONTINUOUSLY over (gallery tile, k-tile) pairs — the first k-tile of the next gallery tile is already in flight while
// the wave drains the current tile's scores into its heaps — counted vmcnt waits, one barrier per k-tile, no branch in the
// k loop (k-tiles past D are all-zero DMAs).  Natural k order: lane half h of MFMA step t takes k = 8 qd + 4 h + t from
// both operands, which is just another order of the same dot product.
//
// Draining a tile's 128 x 128 scores into the heaps (a third of the kernel's time when every element was shuffled to the list
// owner and tested there): every lane — both halves of a pair — now tests its OWN 64 scores against the query's bound, a group
// of GS at a time: one max over the group and ONE compare decide whether the group holds a candidate at all (after the first
// tiles it rarely does); only then are the group's elements tested one by one and appended to the lane's own pending buffer.
// The owner half drains its own and its partner's pending entries into the heap.
#ifdef TK_COUNT
__device__ unsigned long long tk_cnt[8];
extern "C" int slic_debug_topk_counters(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(tk_cnt), sizeof(tk_cnt));
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(tk_cnt), z, sizeof(z)); }
  return 0;
}
// counters are kept in registers and added to the global ones once, at the end of the kernel (an atomic per event would sit
// in front of the next vmcnt(0) and be measured itself)
#define TKC(i, n) do { tkc_[i] += (unsigned long long)(n); } while (0)
#define TKC_DECL unsigned long long tkc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TKC_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) if (tkc_[i_]) atomicAdd(&tk_cnt[i_], tkc_[i_]); } while (0)
static __device__ inline unsigned long long tk_now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#else
#define TKC(i, n) do {} while (0)
#define TKC_DECL
#define TKC_FLUSH do {} while (0)
#endif
template <int GS>
__global__ __launch_bounds__(256) void topk_partial_dma(
    const float* __restrict__ Q, int Nq, const float* __restrict__ G, int Ng, int D, int k, int self_mask,
    int g_per_slice, int TK_PC, float* __restrict__ pval, int32_t* __restrict__ pidx /* [slices][Nq][k] */,
    int* __restrict__ gthr /* [Nq] order-preserving int image of a lower bound of query q's final k-th best score */) {
  TKC_DECL
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int STAGE_FLOATS = (TK_BQ + TK_BG) * TK_BK;
  // heaps: [4 waves][kh][32] keys, kh = k rounded up to 4 m + 1 so that every node has four children (the padding holds the
  // best possible key and is never picked as a node's worst child); pending candidates: [4 waves][TK_PC][64] keys, one column per LANE
  const int kh = 1 + ((k + 2) / 4) * 4;
  unsigned long long* heaps = (unsigned long long*)(lds + 2 * STAGE_FLOATS);
  unsigned long long* pend = heaps + 4 * kh * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * TK_BQ;
  const int gbeg = blockIdx.y * g_per_slice;
  const int gend = min(gbeg + g_per_slice, Ng);
  unsigned long long* hp = heaps + wave * kh * 32;           // this wave's heaps, entry e of query r at hp[e * 32 + r]
  unsigned long long* pq = pend + wave * TK_PC * 64;
  // drains are taken by all four waves TOGETHER: a wave that drains alone keeps the other three waiting at the next k-tile
  // barrier, so four independent drains cost the workgroup four stalls.  wflag[tile & 1] is raised at the end of a tile by any
  // wave whose pending buffers run short and read by everybody after the first barrier of the next tile.
  int* wflag = (int*)(pend + 4 * TK_PC * 64);
  if (tid < 2) wflag[tid] = 0;
  int pc = 0;                                                // this lane's pending count
  const unsigned long long KEY_EMPTY = tk_key(-INFINITY, INT_MAX), KEY_PAD = ~0ull;
  if (h == 0)
    for (int s = 0; s < kh; ++s) hp[s * 32 + r] = s < k ? KEY_EMPTY : KEY_PAD;
  const int q = q0 + 32 * wave + r;
  const bool owner = h == 0 && q < Nq;                       // this lane keeps query q's list
  unsigned long long root = KEY_EMPTY;                       // worst kept entry (heap root)
  float thr = owner ? -INFINITY : INFINITY;                  // its score; +inf on lanes that own no list: never a candidate

  const int srow = tid >> 3;
  const int cq = (tid & 7) ^ ((srow >> 1) & 7);              // SOURCE chunk of this lane (LDS slot = tid & 7)
  const int qrows = min(TK_BQ, Nq - q0);
  const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(Q + (int64_t)q0 * D), 0, (int)((int64_t)qrows * D * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(G + (int64_t)gbeg * D), 0, (int)((int64_t)(gend - gbeg) * D * 4), 0x00020000);     // rows past the slice: zeros
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned qoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) qoff[i] = ((unsigned)(srow + 32 * i) * (unsigned)D + cq * 4) * 4u;
  const int klim = D - cq * 4;
  const int nk = (D + TK_BK - 1) / TK_BK;
  const int nkp = (nk + 1) & ~1;                              // k-tiles per gallery tile, rounded up to the ring length
  const int ntile = (gend - gbeg + TK_BG - 1) / TK_BG;
  // issue the DMAs of ring step (tile, kt) — kt may run into the zero padding, tile past the end is all out of range
  auto issue = [&](int tile, int kt, int stage) {
    float* Gs = lds + stage * STAGE_FLOATS;
    float* Qs = Gs + TK_BG * TK_BK;
    const bool kin = kt * TK_BK < klim && tile < ntile;
    const unsigned kb = (unsigned)kt * (TK_BK * 4u);
    const unsigned gb = (unsigned)tile * (unsigned)(TK_BG * D * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)(Gs + (8 * wave + 32 * i) * TK_BK),
                                               16, (int)(kin ? gb + qoff[i] + kb : OOB), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (__attribute__((address_space(3))) void*)(Qs + (8 * wave + 32 * i) * TK_BK),
                                               16, (int)(kin ? qoff[i] + kb : OOB), 0, 0, 0);
  };
  f32x16 acc[4];
  int gimg = (int)0x807FFFFF;                                  // order-preserving int image of -inf
  auto compute = [&](int stage) {
    const float* Gs = lds + stage * STAGE_FLOATS;
    const float* Qs = Gs + TK_BG * TK_BK;
    f32x4 b[2], a[2][4];
    b[0] = *(const f32x4*)&Qs[tk_off(32 * wave + r, h)];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) a[0][ct] = *(const f32x4*)&Gs[tk_off(32 * ct + r, h)];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int cur = qd & 1, nxt = cur ^ 1;
      if (qd < 3) {
        b[nxt] = *(const f32x4*)&Qs[tk_off(32 * wave + r, 2 * (qd + 1) + h)];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) a[nxt][ct] = *(const f32x4*)&Gs[tk_off(32 * ct + r, 2 * (qd + 1) + h)];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][ct][t], b[cur][t], acc[ct], 0, 0, 0);
      if (qd < 3) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto flush = [&]() {
    const int pcp = __shfl_xor(pc, 32);                        // the partner half's count
    const int tot = h == 0 ? pc + pcp : 0;                     // the owner drains its own entries, then its partner's
    TKC(0, 1);
    for (int j = 0; j < 2 * TK_PC; ++j) {
      const bool act = j < tot;
      if (!__any(act)) break;
      TKC(1, 1);
      if (act) {
        const int jj = j < pc ? j : j - pc;
        const unsigned long long cand = pq[jj * 64 + (j < pc ? lane : lane + 32)];
        if (cand > root) {
#if defined(TK_COUNT) && TK_COUNT > 1
          atomicAdd(&tk_cnt[5], 1ull);
#endif
          // 4-ary heap, root (slot 0) = worst kept entry: the four children of a node are read together (one LDS round
          // trip per level, log4 k levels), the worst of them moves up while it is worse than the candidate
          int pos = 0;
          unsigned long long newroot = cand;
          for (;;) {
            const int c0 = 4 * pos + 1;
            if (c0 >= kh) break;
            const unsigned long long k0 = hp[c0 * 32 + r], k1 = hp[(c0 + 1) * 32 + r], k2 = hp[(c0 + 2) * 32 + r],
                                     k3 = hp[(c0 + 3) * 32 + r];
            unsigned long long w = k0; int ws = c0;
            if (k1 < w) { w = k1; ws = c0 + 1; }
            if (k2 < w) { w = k2; ws = c0 + 2; }
            if (k3 < w) { w = k3; ws = c0 + 3; }
            if (!(cand > w)) break;
            hp[pos * 32 + r] = w;
            if (pos == 0) newroot = w;
            pos = ws;
          }
          hp[pos * 32 + r] = cand;
          root = newroot;
        }
      }
    }
    pc = 0;
    thr = owner ? tk_key_score(root) : INFINITY;
    // slices of the same query prune for each other: the k-th best score any slice holds is a lower bound of the final k-th
    // best, so a full heap publishes its root (atomic max on an order-preserving int image).  Published HERE, where the root
    // changes — a drain is followed by a k-tile of matrix work, which hides the atomic's round trip; at the end of a tile it
    // would sit right in front of the next k-tile's vmcnt(0).
    if (owner && thr > -INFINITY) {
      const int b = __float_as_int(thr);
      atomicMax(gthr + q, b >= 0 ? b : b ^ 0x7FFFFFFF);
    }
  };
  issue(0, 0, 0);
  for (int tile = 0; tile < ntile; ++tile) {
#ifdef TK_COUNT
    const unsigned long long ts0_ = tk_now();
#endif
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
    for (int s0 = 0; s0 < nkp; s0 += 2) {
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const int kt = s0 + sidx;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool wrap = kt + 1 >= nkp;                       // the next ring step opens the next gallery tile
        issue(wrap ? tile + 1 : tile, wrap ? 0 : kt + 1, sidx ^ 1);
        // the best bound the other slices have published: asked for two k-tiles before the tile ends, so the round trip
        // to memory is over by then (the vmcnt(0) of the last k-tile covers it)
        if (s0 == nkp - 2 && sidx == 0 && owner) gimg = __builtin_nontemporal_load(gthr + q);
        if (s0 == 0) {
          if (sidx == 0) {
            if (wflag[(tile + 1) & 1]) flush();                // raised at the end of the previous tile (0 before the first)
          } else if (tid == 0) wflag[(tile + 1) & 1] = 0;      // everybody has read it: clear it for the tile after this one
        }
        compute(sidx);
      }
    }
#ifdef TK_COUNT
    const unsigned long long ts1_ = tk_now();
    TKC(6, ts1_ - ts0_);                                       // k loop (incl. a collective flush)
#endif
    const int g0 = gbeg + tile * TK_BG;
    // rare exclusions are applied to the scores up front (wave-uniform tests), so the per-element test below is one compare:
    // gallery rows past the slice (last tile only) and the query itself (self_mask, only where the ranges meet)
    const bool ragged = g0 + TK_BG > gend;
    const bool selfhit = self_mask && g0 < q0 + 32 * wave + 32 && g0 + TK_BG > q0 + 32 * wave;
    if (ragged || selfhit) {
      const int qo = q0 + 32 * wave + r;                       // both halves of a lane pair serve query r
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int gi = g0 + ct * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
          if (gi >= gend || (self_mask && gi == qo)) acc[ct][v] = -INFINITY;
        }
    }
    // Slices of the same query prune for each other: the k-th best score any slice holds is a lower bound of the final
    // k-th best, so it is published (atomic max on an order-preserving int image) and every slice drops scores strictly
    // below the best bound published so far.  Exact whatever the timing: a dropped score cannot be in the final top-k.
    const float gb = __int_as_float(gimg >= 0 ? gimg : gimg ^ 0x7FFFFFFF);
    // the bound both halves of a pair test against: the owner's heap root (or the published bound); lanes of a query slot
    // past Nq test against +inf
    const float gbb = __shfl(gb, r);
    float filt = fmaxf(__shfl(thr, r), gbb);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int v0 = 0; v0 < 16; v0 += GS) {
        if (__any(pc > TK_PC - GS)) {                           // room for a whole group in every lane's pending buffer
          flush();
          filt = fmaxf(__shfl(thr, r), gbb);
        }
        float gm = acc[ct][v0];
#pragma unroll
        for (int v = v0 + 1; v < v0 + GS; ++v) gm = fmaxf(gm, acc[ct][v]);
        if (gm >= filt && gm > -INFINITY) {
          TKC(2, 1);
#pragma unroll
          for (int v = v0; v < v0 + GS; ++v) {
            const float sc = acc[ct][v];
            if (sc >= filt && sc > -INFINITY) {                 // ties with the root are sorted out by the owner (index order)
              TKC(3, 1);
#if defined(TK_COUNT) && TK_COUNT > 1
              atomicAdd(&tk_cnt[4], 1ull);
#endif
              pq[pc * 64 + lane] = tk_key(sc, g0 + ct * 32 + (v & 3) + 8 * (v >> 2) + 4 * h);
              ++pc;
            }
          }
        }
      }
    // no drain at the end of a tile: the pending entries ride along until a buffer runs short of room for two more groups —
    // a drain costs max-over-lanes rounds, and the fuller the buffers the closer that maximum is to the mean
    if (__any(pc > TK_PC - 2 * GS) && lane == 0) wflag[tile & 1] = 1;
#ifdef TK_COUNT
    TKC(7, tk_now() - ts1_);               // everything between two k loops
#endif
  }
  flush();
  TKC_FLUSH;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (h == 0 && q < Nq) {
    float* ov = pval + ((int64_t)blockIdx.y * Nq + q) * k;
    int32_t* oi = pidx + ((int64_t)blockIdx.y * Nq + q) * k;
    for (int s = 0; s < k; ++s) {
      const unsigned long long e = hp[s * 32 + r];
      ov[s] = tk_key_score(e);
      oi[s] = tk_key_index(e);
    }
  }
}

// D <= 512: the QUERY operand lives in registers.  At one workgroup per CU (the lists fill the LDS) a SIMD holds a single wave,
// so whatever a k-tile does before its first MFMA — wait for the DMA, barrier, issue the next DMA, fetch the first fragments
// from LDS — leaves the matrix pipe idle (the 2-stage kernel above spends 20 % of its k loop that way).  Here
//  * a wave's 32 queries x D are loaded once into NK * 16 registers per lane (NK = ceil(D / 32) k-tiles, unrolled, so every
//    MFMA's B operand is a fixed register): no query DMA, no query LDS reads, half the DMA instructions;
//  * the freed LDS makes the gallery ring 4 stages deep at the same footprint: the DMA of step s + 3 is issued at step s and
//    waited for (counted vmcnt) at step s + 2 — two k-tiles in flight instead of one;
//  * the barrier of step s publishes stage s + 1, so the last quarter of step s already fetches the first fragments of step
//    s + 1: after the next barrier the MFMAs start at once.
// Same k order inside every accumulator as topk_partial_dma, same scan / drain / bound logic: bit-identical lists.
template <int NK, int GS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void topk_partial_qreg(
    const float* __restrict__ Q, int Nq, const float* __restrict__ G, int Ng, int D, int k, int self_mask,
    int g_per_slice, int TK_PC, float* __restrict__ pval, int32_t* __restrict__ pidx /* [slices][Nq][k] */,
    int* __restrict__ gthr, int gstep /* gallery row r of this launch = row r * gstep of G (a strided SAMPLE; 1 = all rows) */,
    const int* __restrict__ qmap /* query slot q reads row qmap[q] of Q (NULL: q) */,
    const int* __restrict__ nq_dev /* number of query slots in use, read on the device (NULL: Nq) */) {
  TKC_DECL
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(NK % 4 == 0, "a gallery tile is a whole number of ring turns");
  const int Nq_cap = Nq;                                       // row pitch of the partial lists
  if (nq_dev) {
    Nq = *nq_dev;
    if ((int)blockIdx.x * TK_BQ >= Nq) return;                 // (workgroup-uniform, before any barrier)
  }
  constexpr int STAGE_FLOATS = TK_BG * TK_BK;                  // gallery rows only
  const int kh = 1 + ((k + 2) / 4) * 4;
  unsigned long long* heaps = (unsigned long long*)(lds + 4 * STAGE_FLOATS);
  unsigned long long* pend = heaps + 4 * kh * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * TK_BQ;
  const int gbeg = blockIdx.y * g_per_slice;
  const int gend = min(gbeg + g_per_slice, Ng);
  unsigned long long* hp = heaps + wave * kh * 32;
  unsigned long long* pq = pend + wave * TK_PC * 64;
  int* wflag = (int*)(pend + 4 * TK_PC * 64);                  // collective drains, see topk_partial_dma
  if (tid < 2) wflag[tid] = 0;
  int pc = 0;
  const unsigned long long KEY_EMPTY = tk_key(-INFINITY, INT_MAX), KEY_PAD = ~0ull;
  if (h == 0)
    for (int s = 0; s < kh; ++s) hp[s * 32 + r] = s < k ? KEY_EMPTY : KEY_PAD;
  const int q = q0 + 32 * wave + r;
  const bool owner = h == 0 && q < Nq;
  unsigned long long root = KEY_EMPTY;
  float thr = owner ? -INFINITY : INFINITY;

  // lane (r, h) of MFMA step (kt, qd, t) multiplies by Q[q][32 kt + 8 qd + 4 h + t]
  f32x4 qr[NK][4];
  int qself;                                                   // the gallery index that IS this query (self_mask)
  {
    const int qs = q < Nq ? q : Nq - 1;
    qself = qmap ? qmap[qs] : q;
    const float* qrow = Q + (int64_t)(qmap ? qmap[qs] : qs) * D;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = 32 * kt + 8 * qd + 4 * h;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        qr[kt][qd] = c < D ? *(const f32x4*)(qrow + c) : z;    // D % 8 == 0: a chunk is inside the row or past it
      }
  }

  const int srow = tid >> 3;
  const int cq = (tid & 7) ^ ((srow >> 1) & 7);              // SOURCE chunk of this lane (LDS slot = tid & 7)
  const unsigned rowb = (unsigned)D * (unsigned)gstep * 4u;    // bytes between two rows of this launch's gallery
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(G + (int64_t)gbeg * gstep * D), 0, (int)(((int64_t)(gend - gbeg - 1) * gstep + 1) * D * 4), 0x00020000);     // rows past the slice: zeros
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned goff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) goff[i] = (unsigned)(srow + 32 * i) * rowb + (unsigned)(cq * 16);
  const int klim = D - cq * 4;
  const int ntile = (gend - gbeg + TK_BG - 1) / TK_BG;
  // the DMAs of ring step (tile, kt): k-tiles past D are all-zero, a tile past the end is all out of range
  auto issue = [&](int tile, int kt, int stage) {
    float* Gs = lds + stage * STAGE_FLOATS;
    const bool kin = kt * TK_BK < klim && tile < ntile;
    const unsigned kb = (unsigned)kt * (TK_BK * 4u);
    const unsigned gb = (unsigned)tile * ((unsigned)TK_BG * rowb);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)(Gs + (8 * wave + 32 * i) * TK_BK),
                                               16, (int)(kin ? gb + goff[i] + kb : OOB), 0, 0, 0);
  };
  auto flush = [&]() {
    const int pcp = __shfl_xor(pc, 32);                        // the partner half's count
    const int tot = h == 0 ? pc + pcp : 0;                     // the owner drains its own entries, then its partner's
    TKC(0, 1);
    for (int j = 0; j < 2 * TK_PC; ++j) {
      const bool act = j < tot;
      if (!__any(act)) break;
      TKC(1, 1);
      if (act) {
        const int jj = j < pc ? j : j - pc;
        const unsigned long long cand = pq[jj * 64 + (j < pc ? lane : lane + 32)];
        if (cand > root) {
          int pos = 0;                                         // 4-ary heap, see topk_partial_dma
          unsigned long long newroot = cand;
          for (;;) {
            const int c0 = 4 * pos + 1;
            if (c0 >= kh) break;
            const unsigned long long k0 = hp[c0 * 32 + r], k1 = hp[(c0 + 1) * 32 + r], k2 = hp[(c0 + 2) * 32 + r],
                                     k3 = hp[(c0 + 3) * 32 + r];
            unsigned long long w = k0; int ws = c0;
            if (k1 < w) { w = k1; ws = c0 + 1; }
            if (k2 < w) { w = k2; ws = c0 + 2; }
            if (k3 < w) { w = k3; ws = c0 + 3; }
            if (!(cand > w)) break;
            hp[pos * 32 + r] = w;
            if (pos == 0) newroot = w;
            pos = ws;
          }
          hp[pos * 32 + r] = cand;
          root = newroot;
        }
      }
    }
    pc = 0;
    thr = owner ? tk_key_score(root) : INFINITY;
    if (owner && gthr && thr > -INFINITY) {                    // publish the new root (see topk_partial_dma)
      const int b = __float_as_int(thr);
      atomicMax(gthr + q, b >= 0 ? b : b ^ 0x7FFFFFFF);
    }
  };
  f32x16 acc[4];
  f32x4 a[2][4];                                               // gallery fragments, double-buffered ACROSS k-tiles
  int gimg = (int)0x807FFFFF;                                  // order-preserving int image of -inf
  issue(0, 0, 0);
  issue(NK > 1 ? 0 : 1, NK > 1 ? 1 : 0, 1);
  issue(NK > 2 ? 0 : 1, NK > 2 ? 2 : 0, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // step 0 has landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) a[0][ct] = *(const f32x4*)&lds[tk_off(32 * ct + r, h)];
  for (int tile = 0; tile < ntile; ++tile) {
#ifdef TK_COUNT
    const unsigned long long ts0_ = tk_now();
#endif
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      // step s = (tile, kt) computes from stage kt & 3.  Outstanding here: the DMAs of steps s + 1 and s + 2 (4 each).
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // step s + 1 has landed (this wave's part of it)
      __builtin_amdgcn_s_barrier();                            // ... everybody's; and stage (kt + 3) & 3 has been read by all
      {
        const int kn = kt + 3;                                  // step s + 3
        issue(kn >= NK ? tile + 1 : tile, kn >= NK ? kn - NK : kn, kn & 3);
      }
      if (kt == 0) {
        if (wflag[(tile + 1) & 1]) flush();                    // raised at the end of the previous tile (0 before the first)
      } else if (kt == 1) {
        if (tid == 0) wflag[(tile + 1) & 1] = 0;               // everybody has read it: clear it for the tile after this one
      } else if (kt == NK - 2) {
        if (owner && gthr) gimg = __builtin_nontemporal_load(gthr + q); // the other slices' best bound, needed when the tile ends (NULL: the slices keep to themselves)
      }
      const float* Gs = lds + (kt & 3) * STAGE_FLOATS;
      const float* Gn = lds + ((kt + 1) & 3) * STAGE_FLOATS;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int cur = qd & 1, nxt = cur ^ 1;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          a[nxt][ct] = qd < 3 ? *(const f32x4*)&Gs[tk_off(32 * ct + r, 2 * (qd + 1) + h)]
                              : *(const f32x4*)&Gn[tk_off(32 * ct + r, h)];       // first fragments of the next step
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][ct][t], qr[kt][qd][t], acc[ct], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
#ifdef TK_COUNT
    const unsigned long long ts1_ = tk_now();
    TKC(6, ts1_ - ts0_);
#endif
    const int g0 = gbeg + tile * TK_BG;
    const bool ragged = g0 + TK_BG > gend;
    // (with a query map the slots of a block are arbitrary rows: the range test that skips the per-element comparison does not apply)
    const bool selfhit = self_mask && (qmap != nullptr || (g0 < q0 + 32 * wave + 32 && g0 + TK_BG > q0 + 32 * wave));
    if (ragged || selfhit) {
      const int qo = qself;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int gi = g0 + ct * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
          if (gi >= gend || (self_mask && gi == qo)) acc[ct][v] = -INFINITY;
        }
    }
    const float gb = __int_as_float(gimg >= 0 ? gimg : gimg ^ 0x7FFFFFFF);
    const float gbb = __shfl(gb, r);
    float filt = fmaxf(__shfl(thr, r), gbb);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int v0 = 0; v0 < 16; v0 += GS) {
        if (__any(pc > TK_PC - GS)) {                           // room for a whole group in every lane's pending buffer
          flush();
          filt = fmaxf(__shfl(thr, r), gbb);
        }
        float gm = acc[ct][v0];
#pragma unroll
        for (int v = v0 + 1; v < v0 + GS; ++v) gm = fmaxf(gm, acc[ct][v]);
        if (gm >= filt && gm > -INFINITY) {
          TKC(2, 1);
#pragma unroll
          for (int v = v0; v < v0 + GS; ++v) {
            const float sc = acc[ct][v];
            if (sc >= filt && sc > -INFINITY) {                 // ties with the root are sorted out by the owner (index order)
              pq[pc * 64 + lane] = tk_key(sc, g0 + ct * 32 + (v & 3) + 8 * (v >> 2) + 4 * h);
              ++pc;
            }
          }
        }
      }
    if (__any(pc > TK_PC - 2 * GS) && lane == 0) wflag[tile & 1] = 1;
#ifdef TK_COUNT
    TKC(7, tk_now() - ts1_);
#endif
  }
  flush();
  TKC_FLUSH;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the trailing all-zero DMAs must land before the workgroup leaves
  if (h == 0 && q < Nq) {
    float* ov = pval + ((int64_t)blockIdx.y * Nq_cap + q) * k;
    int32_t* oi = pidx + ((int64_t)blockIdx.y * Nq_cap + q) * k;
    for (int s = 0; s < k; ++s) {
      const unsigned long long e = hp[s * 32 + r];
      ov[s] = tk_key_score(e);
      oi[s] = tk_key_index(e);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The COLLECT path (round 5; D <= 512, galleries of >= 32k rows): threshold, then collect, then select.
//   A per-query streaming top-k pays for its list: at k = 50 every (query, slice) pair makes ~75 heap insertions, each a dependent chain
//   of LDS round trips during which the wave's matrix pipe idles (one wave per SIMD) — 9.2 ms at k = 50 against 7.9 at k = 1.
//   Here the similarity pass keeps NO list:
//   1. topk_partial_qreg on a strided SAMPLE of the gallery (~3 % of it in 3 slices, the M = 12 best of each: tiny heaps) ->
//      topk_thresholds: tau_q = the M-th best of the pooled sample — about 6 k + 100 gallery rows are expected to reach it;
//   2. topk_collect_qreg: the k = 1 kernel's matrix loop over the WHOLE gallery; a score >= tau_q is appended to the lane's pending
//      column in LDS (one compare per group of four scores, as before) and the columns go to the query's candidate buffer in memory
//      (one atomic add per lane and drain: a reservation, not a list) — no heap, no bound to publish, no drain stalls;
//   3. topk_select: one wave per query sorts out the k best of its n candidates (n ~ 400: k rounds of a wave-wide maximum over 64-bit
//      keys), ascending distance, ties -> lower index.  Exactness: every gallery row whose score reaches tau_q is a candidate, so the k best
//      candidates ARE the k best rows whenever n >= k.  A query with n < k (tau_q too high: a < 1e-6 event per query) or n > capacity
//      (heavy ties / duplicated rows) is put on a list, and
//   4. the queries on that list — normally none: the launches then exit at once — go through the streaming kernels above (query map +
//      device-side count), whose results are exact for any data.
// ------------------------------------------------------------------------------------------
#define TKC_CAP 2048          // most candidate slots per query (keys of 8 bytes); a call's own count: TopkCollectPlan::cap

__global__ __launch_bounds__(256) void topk_thresholds(const float* __restrict__ pval /* [slices][Nq][ms]: the ms best of every sample slice */,
                                                        int slices, int Nq, int ms, int M, float* __restrict__ tau, int* __restrict__ cnt,
                                                        int* __restrict__ nfail) {
  // one WAVE per query: the M-th best of the pooled lists (slices * ms <= 128 entries, two per lane).  Every entry counts the entries that
  // come before it in (score descending, position ascending) order — the pooled values are broadcast from LDS — and the one with M - 1
  // predecessors is the threshold.  (One THREAD per query walking M rounds over the lists took 74 us for 10k queries.)
  __shared__ float pool[4][128];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + w;
  if (blockIdx.x == 0 && threadIdx.x == 0) *nfail = 0;
  if (q >= Nq) return;                                         // (whole waves leave: no barrier below)
  const int E = slices * ms;
  float mine[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = lane + 64 * u;
    mine[u] = e < E ? pval[((int64_t)(e / ms) * Nq + q) * ms + (e % ms)] : -INFINITY;
    pool[w][e] = mine[u];
  }
  __builtin_amdgcn_wave_barrier();
  int before[2] = {0, 0};
  for (int j = 0; j < E; ++j) {
    const float x = pool[w][j];
#pragma unroll
    for (int u = 0; u < 2; ++u) before[u] += (x > mine[u] || (x == mine[u] && j < lane + 64 * u)) ? 1 : 0;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (lane + 64 * u < E && before[u] == M - 1) {
      // -inf (a sample that never filled its lists) would admit masked scores too: the lowest finite threshold instead
      tau[q] = fmaxf(mine[u], -3.0e38f);
      cnt[q] = 0;
    }
}

template <int NK, int GS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void topk_collect_qreg(
    const float* __restrict__ Q, int Nq, const float* __restrict__ G, int Ng, int D, int self_mask, int g_per_slice, int TK_PC,
    const float* __restrict__ tau, int* __restrict__ cnt, unsigned long long* __restrict__ cand /* [Nq][cap] keys */, int cap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(NK % 4 == 0, "a gallery tile is a whole number of ring turns");
  constexpr int STAGE_FLOATS = TK_BG * TK_BK;                  // gallery rows only
  unsigned long long* pend = (unsigned long long*)(lds + 4 * STAGE_FLOATS);      // [4 waves][TK_PC][64] keys, one column per LANE
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * TK_BQ;
  const int gbeg = blockIdx.y * g_per_slice;
  const int gend = min(gbeg + g_per_slice, Ng);
  unsigned long long* pq = pend + wave * TK_PC * 64;
  int pc = 0;
  const int q = q0 + 32 * wave + r;
  // both halves of a lane pair serve query r and test against its threshold; slots past Nq never see a candidate
  const float filt = q < Nq ? tau[q] : INFINITY;
  f32x4 qr[NK][4];
  {
    const float* qrow = Q + (int64_t)(q < Nq ? q : Nq - 1) * D;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = 32 * kt + 8 * qd + 4 * h;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        qr[kt][qd] = c < D ? *(const f32x4*)(qrow + c) : z;
      }
  }
  const int srow = tid >> 3;
  const int cq = (tid & 7) ^ ((srow >> 1) & 7);              // SOURCE chunk of this lane (LDS slot = tid & 7)
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(G + (int64_t)gbeg * D), 0, (int)((int64_t)(gend - gbeg) * D * 4), 0x00020000);     // rows past the slice: zeros
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned goff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) goff[i] = ((unsigned)(srow + 32 * i) * (unsigned)D + cq * 4) * 4u;
  const int klim = D - cq * 4;
  const int ntile = (gend - gbeg + TK_BG - 1) / TK_BG;
  auto issue = [&](int tile, int kt, int stage) {
    float* Gs = lds + stage * STAGE_FLOATS;
    const bool kin = kt * TK_BK < klim && tile < ntile;
    const unsigned kb = (unsigned)kt * (TK_BK * 4u);
    const unsigned gb = (unsigned)tile * (unsigned)(TK_BG * D * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (__attribute__((address_space(3))) void*)(Gs + (8 * wave + 32 * i) * TK_BK),
                                               16, (int)(kin ? gb + goff[i] + kb : OOB), 0, 0, 0);
  };
  // a lane's pending keys -> the query's candidate buffer: ONE atomic add reserves the lane's slots (its return is the only round trip
  // to memory; a drain happens at the end of the slice and when a column runs full — every ~100 tiles at 400 candidates per query)
  auto drain = [&]() {
    if (pc > 0) {
      const int base = atomicAdd(cnt + q, pc);
      for (int j = 0; j < pc; ++j)
        if (base + j < cap) {
          const unsigned long long raw = pq[j * 64 + lane];     // {score bits, gallery index}
          cand[(int64_t)q * cap + base + j] = tk_key(__uint_as_float((unsigned)raw), (int)(raw >> 32));
        }
    }
    pc = 0;
  };
  f32x16 acc[4];
  f32x4 a[2][4];
  issue(0, 0, 0);
  issue(NK > 1 ? 0 : 1, NK > 1 ? 1 : 0, 1);
  issue(NK > 2 ? 0 : 1, NK > 2 ? 2 : 0, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // step 0 has landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) a[0][ct] = *(const f32x4*)&lds[tk_off(32 * ct + r, h)];
  for (int tile = 0; tile < ntile; ++tile) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[ct][v] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      // (a drain's stores and atomic may sit among the outstanding DMAs: they only make this wait stricter)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // step s + 1 has landed (this wave's part of it)
      __builtin_amdgcn_s_barrier();                            // ... everybody's; and stage (kt + 3) & 3 has been read by all
      {
        const int kn = kt + 3;
        issue(kn >= NK ? tile + 1 : tile, kn >= NK ? kn - NK : kn, kn & 3);
      }
      const float* Gs = lds + (kt & 3) * STAGE_FLOATS;
      const float* Gn = lds + ((kt + 1) & 3) * STAGE_FLOATS;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int cur = qd & 1, nxt = cur ^ 1;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          a[nxt][ct] = qd < 3 ? *(const f32x4*)&Gs[tk_off(32 * ct + r, 2 * (qd + 1) + h)]
                              : *(const f32x4*)&Gn[tk_off(32 * ct + r, h)];       // first fragments of the next step
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][ct][t], qr[kt][qd][t], acc[ct], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    const int g0 = gbeg + tile * TK_BG;
    const bool ragged = g0 + TK_BG > gend;
    const bool selfhit = self_mask && g0 < q0 + 32 * wave + 32 && g0 + TK_BG > q0 + 32 * wave;
    if (ragged || selfhit) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int gi = g0 + ct * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
          if (gi >= gend || (self_mask && gi == q)) acc[ct][v] = -INFINITY;      // never >= a finite threshold
        }
    }
    const int gl = g0 + 4 * h;                                 // this lane's rows of the tile: gl + an immediate
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      if (__any(pc > TK_PC - 16)) drain();                      // room for the sixteen scores of this accumulator in every lane's column
#pragma unroll
      for (int v0 = 0; v0 < 16; v0 += GS) {
        float gm = acc[ct][v0];
#pragma unroll
        for (int v = v0 + 1; v < v0 + GS; ++v) gm = fmaxf(gm, acc[ct][v]);
        if (gm >= filt) {
#pragma unroll
          for (int v = v0; v < v0 + GS; ++v) {
            const float sc = acc[ct][v];
            if (sc >= filt) {                                  // the raw pair: the key is built when the column is drained
              pq[pc * 64 + lane] = ((unsigned long long)(unsigned)(gl + (ct * 32 + (v & 3) + 8 * (v >> 2))) << 32) | __float_as_uint(sc);
              ++pc;
            }
          }
        }
      }
    }
  }
  drain();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the trailing all-zero DMAs must land before the workgroup leaves
}

// one WAVE per query: the k best of its n = cnt[q] candidate keys, best first.  n <= TKS_NP keys are sorted in LDS by the wave — a
// bitonic network, 64 compare-exchanges per instruction, LDS operations of one wave execute in order so no workgroup barrier is
// involved; 45 stages for the usual ~400 candidates (padded to 512) — and the first k leave in one coalesced store.  (k rounds of a
// wave-wide maximum over keys in registers took 135-154 us for 10k queries at k = 50: twelve dependent ds_bpermute round trips per round.)
// More than TKS_NP keys (rare): k rounds over the keys in memory.  n < k or n > cap: the query goes on the fallback list and its output
// row is left to that pass.
#define TKS_NP 1024
#ifndef TKS_WAVES
#define TKS_WAVES 1           // queries (waves) per workgroup (4 / 2 / 1 measured 141 / 122 / 119 us for 10k queries: scripts/r5/ab_select.sh)
#endif
__global__ __launch_bounds__(64 * TKS_WAVES) void topk_select(const unsigned long long* __restrict__ cand, const int* __restrict__ cnt, int Nq, int k, int cap,
                                                   int32_t* __restrict__ out_idx, float* __restrict__ out_dist, int* __restrict__ failq,
                                                   int* __restrict__ nfail) {
  __shared__ unsigned long long sk[TKS_WAVES][TKS_NP];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * TKS_WAVES + w;
  if (q >= Nq) return;                                         // (whole waves leave; nothing below needs the workgroup)
  const int n = cnt[q];
  if (n < k || n > cap) {
    if (lane == 0) failq[atomicAdd(nfail, 1)] = q;
    return;
  }
  const unsigned long long* c = cand + (int64_t)q * cap;
  if (n <= TKS_NP) {
    unsigned long long* key = sk[w];
    int NP = 64;
    while (NP < n) NP <<= 1;                                   // wave-uniform
    for (int i = lane; i < NP; i += 64) key[i] = i < n ? c[i] : 0ull;      // 0 sorts last (no real key is 0: its low word would be ~(-1))
    for (int size = 2; size <= NP; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < (NP >> 1); t += 64) {
          const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
          const bool desc = (lo & size) == 0;                  // blocks alternate direction; the last merge (size == NP) is all descending
          const unsigned long long a = key[lo], b = key[hi];
          if ((a < b) == desc) { key[lo] = b; key[hi] = a; }
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int o = lane; o < k; o += 64) {
      const unsigned long long e = key[o];
      out_idx[(int64_t)q * k + o] = tk_key_index(e);
      out_dist[(int64_t)q * k + o] = fminf(fmaxf(1.0f - tk_key_score(e), 0.f), 2.f);
    }
  } else {
    unsigned long long last = ~0ull;                           // every key still in play is < last
    for (int o = 0; o < k; ++o) {
      unsigned long long b = 0ull;
      for (int e = lane; e < n; e += 64) {
        const unsigned long long x = c[e];
        if (x < last && x > b) b = x;
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)b, d), hi = __shfl_xor((unsigned)(b >> 32), d);
        const unsigned long long o2 = ((unsigned long long)hi << 32) | lo;
        b = o2 > b ? o2 : b;
      }
      if (lane == 0) {
        out_idx[(int64_t)q * k + o] = tk_key_index(b);
        out_dist[(int64_t)q * k + o] = fminf(fmaxf(1.0f - tk_key_score(b), 0.f), 2.f);
      }
      last = b;
    }
  }
}

// one WAVE per query: the slices' lists (slices * k entries, unsorted) are spread over the lanes' registers (up to
// TKM_PER per lane, else the tail is re-read from memory), then k rounds of: lane-local best -> wave arg-best by
// shuffles -> the owner retires that entry.  Order: larger s first, ties -> lower gallery index.
#define TKM_PER 32
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                                         int slices, int Nq, int k, int32_t* __restrict__ out_idx,
                                                         float* __restrict__ out_dist, const int* __restrict__ qmap /* output row of list q (NULL: q) */,
                                                         const int* __restrict__ nq_dev /* lists in use, read on the device (NULL: Nq) */) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= (nq_dev ? *nq_dev : Nq)) return;
  const int64_t qo = qmap ? qmap[q] : q;
  const int lane = threadIdx.x & 63;
  const int tot = slices * k;
  float v[TKM_PER];
  int ix[TKM_PER];
#pragma unroll
  for (int u = 0; u < TKM_PER; ++u) {
    const int e = lane + 64 * u;                         // entry e = (slice e / k, slot e % k)
    if (e < tot) {
      const int sl = e / k, t = e - sl * k;
      v[u] = pval[((int64_t)sl * Nq + q) * k + t];
      ix[u] = pidx[((int64_t)sl * Nq + q) * k + t];
    } else { v[u] = -INFINITY; ix[u] = INT_MAX; }
  }
  for (int o = 0; o < k; ++o) {
    float bs = -INFINITY; int bi = INT_MAX, bu = 0;
#pragma unroll
    for (int u = 0; u < TKM_PER; ++u)
      if (tk_better(v[u], ix[u], bs, bi)) { bs = v[u]; bi = ix[u]; bu = u; }
    float ws = bs; int wi = bi;
    for (int d = 32; d > 0; d >>= 1) {
      const float os = __shfl_xor(ws, d);
      const int oi = __shfl_xor(wi, d);
      if (tk_better(os, oi, ws, wi)) { ws = os; wi = oi; }
    }
    if (wi == bi && ws == bs && bi != INT_MAX) {         // the owner retires the winner (indices are unique per query)
#pragma unroll
      for (int u = 0; u < TKM_PER; ++u) if (u == bu) { v[u] = -INFINITY; ix[u] = INT_MAX; }
    }
    if (lane == 0) {
      out_idx[qo * k + o] = wi == INT_MAX ? -1 : wi;
      float d = 1.0f - ws;
      d = fminf(fmaxf(d, 0.f), 2.f);
      out_dist[qo * k + o] = wi == INT_MAX ? INFINITY : d;
    }
  }
}

// merge W per-shard result lists [W][Nq][k] (distance ascending within a list, GLOBAL gallery indices) into the
// k best per query: smaller distance first, ties -> lower index.  One thread per query (lists are tiny).
__global__ void topk_merge_dist_kernel(const float* __restrict__ pd, const int32_t* __restrict__ pi, int W, int Nq,
                                       int k, int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Nq) return;
  float last = -INFINITY;
  int last_i = -1;
  for (int o = 0; o < k; ++o) {
    float bd = INFINITY;
    int bi = INT_MAX;
    for (int w = 0; w < W; ++w) {
      const float* v = pd + ((int64_t)w * Nq + q) * k;
      const int32_t* ix = pi + ((int64_t)w * Nq + q) * k;
      for (int t = 0; t < k; ++t) {
        const float d = v[t];
        const int i = ix[t];
        if (i < 0) continue;
        const bool after_last = d > last || (d == last && i > last_i);
        if (after_last && (d < bd || (d == bd && i < bi))) { bd = d; bi = i; }
      }
    }
    out_idx[(int64_t)q * k + o] = bi == INT_MAX ? -1 : bi;
    out_dist[(int64_t)q * k + o] = bd;
    last = bd; last_i = bi;
  }
}

// sklearn.preprocessing.normalize(X) (l2): rows with zero norm are left as they are
__global__ void normalize_rows_sklearn(const float* __restrict__ X, int64_t N, int D, int ldx,
                                       float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  const int lane = threadIdx.x & 63;
  const float* x = X + row * ldx;
  double s = 0.0;
  for (int k = lane; k < D; k += 64) { const double v = (double)x[k]; s += v * v; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  float nrm = (float)sqrt(s);
  if (nrm == 0.f) nrm = 1.f;
  for (int k = lane; k < D; k += 64) out[row * D + k] = x[k] / nrm;
}

// euclidean_distances(X, Y) (evaluate.py:216): one thread per (i, j), direct (x - y)^2 sum
__global__ void pairwise_euclid_kernel(const float* __restrict__ X, int Nx, const float* __restrict__ Y, int Ny,
                                       int D, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)Nx * Ny) return;
  const int i = (int)(e / Ny), j = (int)(e % Ny);
  const float* x = X + (int64_t)i * D;
  const float* y = Y + (int64_t)j * D;
  float a = 0.f;
  for (int k = 0; k < D; ++k) { const float d = x[k] - y[k]; a = fmaf(d, d, a); }
  out[e] = sqrtf(a);
}

static inline hipStream_t S_(void* s) { return (hipStream_t)s; }

extern "C" int slic_normalize_rows(const float* X, int64_t N, int D, int ldx, float* out, void* stream) {
  SLIC_REQUIRE(X && out && N > 0 && D > 0 && ldx >= D, "slic_normalize_rows: bad args");
  normalize_rows_sklearn<<<dim3((unsigned)slic_cdiv(N, 4)), dim3(256), 0, S_(stream)>>>(X, N, D, ldx, out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

static int topk_slices(int Nq, int Ng, int k) {
  // The LDS lists allow ONE workgroup per CU, so the grid (query blocks x gallery slices) should be a whole number of
  // 256-workgroup rounds: among the slice counts that give 3+ rounds (if the gallery allows), take the one that fills its
  // last round best — e.g. 79 query blocks: 13 slices = 1027 workgroups = 4 rounds + 3 stragglers (80 %), 16 slices =
  // 1264 = 4.94 rounds (99 %).  k <= 0: no lists to merge afterwards (the collect pass), so no cap on the slice count.
  const int qb = (int)slic_cdiv(Nq, TK_BQ);
  int maxs = (int)slic_cdiv(Ng, 4 * TK_BG);               // at least 4 gallery tiles per slice
  const int cap = k > 0 ? (64 * TKM_PER) / k : maxs;      // the merge kernel holds slices * k entries in one wave's registers
  if (maxs > cap) maxs = cap;
  if (maxs < 1) maxs = 1;
  int lo = (int)slic_cdiv(3 * 256, qb);
  if (lo > maxs) lo = maxs;
  int hi = 2 * lo + 1 < maxs ? 2 * lo + 1 : maxs;
  int best = lo;
  double beff = -1.0;
  for (int s = lo; s <= hi; ++s) {
    const int64_t blocks = (int64_t)qb * s;
    const double eff = (double)blocks / (double)(slic_cdiv(blocks, 256) * 256);
    if (eff > beff + 0.01) { beff = eff; best = s; }
  }
  return best;
}

// The collect path's sample (see the comment above topk_thresholds): S1 slices of per1 strided rows each, the M best of every slice;
// tau_q = the M-th best of the POOLED sample (it is among the slices' M best), which a fraction ~ M / (S1 per1) of the gallery is expected
// to reach.  Aim: 6 k + 100 candidates per query from a sample of <= ~4 % of the gallery, M = 4 .. 16: both failure events (fewer than k
// candidates, more than TKC_CAP) stay below 1e-6 per query for any continuous score distribution (scripts/r5/topk_threshold_sim.py).
struct TopkCollectPlan { bool on; int S1, per1, m1, gstep, cap, ms; };
static TopkCollectPlan topk_collect_plan(int Nq, int Ng, int k) {
  TopkCollectPlan c = {false, 0, 0, 0, 1, TKC_CAP, 0};
  // SLIC_TOPK_COLLECT: 0 = never, 1 = wherever the shape allows, unset = where it is faster: k >= 16 (10k x 100k x 512 on one MI355X, ms,
  // collect / streaming: k = 1 8.4 / 8.0, k = 10 8.3 / 8.0, k = 50 8.6 / 9.1, k = 88 8.8 / 12.4 — a short list costs the streaming kernels
  // almost nothing, while the collect pass pays ~0.3 ms for its sample and tests ~0.4 % of the scores one by one whatever k is)
  const char* e = getenv("SLIC_TOPK_COLLECT");
  const bool force = e && e[0] == '1';
  if ((e && e[0] == '0') || Ng < 32768 || k > TK_KMAX || (k < 16 && !force)) return c;
  const int qb = (int)slic_cdiv(Nq, TK_BQ);
  const double target = 6.0 * k + 100.0;
  int M = (int)(target * 0.04 + 0.5);
  M = M < 4 ? 4 : (M > 12 ? 12 : M);
  const double ns = M * (double)Ng / target;
  int S1 = 256 / qb;                                            // one dispatch round of sample workgroups
  S1 = S1 < 1 ? 1 : (S1 > 8 ? 8 : S1);
  if (ns / S1 < 2 * TK_BG) { S1 = (int)(ns / (2 * TK_BG)); if (S1 < 1) S1 = 1; }
  int per1 = (int)slic_cdiv((int64_t)(ns / S1), TK_BG) * TK_BG;
  if (per1 < 2 * TK_BG) per1 = 2 * TK_BG;
  M = (int)(target * S1 * per1 / Ng + 0.5);
  M = M < 4 ? 4 : (M > 16 ? 16 : M);
  if ((int64_t)S1 * per1 * 8 > Ng) return c;
  c.on = true; c.S1 = S1; c.per1 = per1; c.m1 = M; c.gstep = Ng / (S1 * per1);
  // entries kept per sample slice: the pooled M best are spread over the slices ~ evenly, so M / S1 + 3 per slice hold them all but
  // rarely — and when they do not, tau comes out one or two order statistics lower: a few more candidates, never a wrong result
  c.ms = S1 == 1 ? M : (M < (M + S1 - 1) / S1 + 3 ? M : (M + S1 - 1) / S1 + 3);
  // candidate slots: the count is ~ Gamma(M) / M x its mean — 9 x the mean for M <= 6, 6 x up to 9, 4 x beyond keep an overflow below 1e-9
  const double room = target * (M <= 6 ? 9.0 : (M <= 9 ? 6.0 : 4.0));
  int cap = 512;
  while (cap < room && cap < TKC_CAP) cap <<= 1;
  c.cap = cap;
  return c;
}

extern "C" size_t slic_cosine_topk_workspace_bytes(int Nq, int Ng, int k) {
  size_t b = 2 * slic_align_up((size_t)topk_slices(Nq, Ng, k) * Nq * k * 4, 256) + slic_align_up((size_t)Nq * 4, 256);
  const TopkCollectPlan c = topk_collect_plan(Nq, Ng, k);
  if (c.on)
    b += 2 * slic_align_up((size_t)c.S1 * Nq * c.ms * 4, 256) + 3 * slic_align_up((size_t)Nq * 4, 256) + 256 +
         slic_align_up((size_t)Nq * c.cap * 8, 256);
  return b;
}

// the streaming path: per-(query, slice) lists in LDS + merge.  qmap / nq_dev (device memory; both or neither): query slot q of [0, *nq_dev)
// is row qmap[q] of Qn and its result goes to output row qmap[q] — the exact fallback of the collect path; Nq then bounds the slot count.
// The streaming launch (topk_stream) and the collect path's sample pass (topk_collect) launch the SAME topk_partial_qreg instantiations with
// different dynamic-LDS sizes.  Two per-call-site "largest size set so far" caches could lower the attribute under each other's feet (the
// smaller call site overwriting the larger one's value: ADVICE round 5); the attribute is therefore set ONCE, for every instantiation, to
// the device maximum (160 KB) — every launch size of either site fits below it.
static int topk_qreg_lds_attr() {
  static bool done = false;
  if (done) return SLIC_OK;
  constexpr int LDS_MAX = 160 * 1024;
  SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_qreg<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_qreg<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_qreg<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_qreg<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_qreg<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_qreg<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
  done = true;
  return SLIC_OK;
}

static int topk_stream(const float* Qn, int Nq, const float* Gn, int Ng, int D, int k, int self_mask, int32_t* out_idx, float* out_dist,
                       SlicCarver& w, hipStream_t st, const int* qmap, const int* nq_dev) {
  const int slices = topk_slices(Nq, Ng, k);
  int per = (int)slic_cdiv(Ng, slices);
  per = (int)slic_cdiv(per, TK_BG) * TK_BG;
  const int S = (int)slic_cdiv(Ng, per);
  float* pval = w.take<float>((size_t)slices * Nq * k);
  int32_t* pidx = w.take<int32_t>((size_t)slices * Nq * k);
  // LDS: 2 ring stages + heaps [4][kh][32] keys (kh = k rounded up to 4 m + 1) + pending [4][pcap][64] keys, one column per LANE
  const int kh = 1 + ((k + 2) / 4) * 4;
  const size_t fixed = (size_t)2 * 2 * TK_BQ * TK_BK * sizeof(float) + (size_t)4 * kh * 32 * 8;
  int pcap = (int)((160 * 1024 - 16 - fixed) / (4 * 64 * 8));
  pcap = pcap > TK_PC_MAX ? TK_PC_MAX : pcap;
  SLIC_REQUIRE(pcap >= 2, "slic_cosine_topk: k = %d leaves no LDS for the pending buffers", k);
  const size_t lds = fixed + (size_t)4 * pcap * 64 * 8 + 16;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  dim3 grid((unsigned)slic_cdiv(Nq, TK_BQ), (unsigned)S);
  const bool dma_ok = (int64_t)per * D * 4 < (1ll << 31) && (int64_t)TK_BQ * D * 4 < (1ll << 31);
  SLIC_REQUIRE(!qmap || (dma_ok && D <= 512), "slic_cosine_topk: internal: the query map needs the register-operand kernel");
  // LDS-DMA kernel unless a gallery slice exceeds the 32-bit byte range of one buffer resource (> 2 GiB: then the register-staged one)
  if (dma_ok) {
    static size_t lds_set2 = 0;
    if (lds > lds_set2) {
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_partial_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      lds_set2 = lds;
    }
    int* gthr = w.take<int>((size_t)Nq);
    SLIC_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)gthr, (int)0x807FFFFF, (size_t)Nq, st));   // the image of -inf
#ifndef TK_NO_QREG
    if (D <= 512) {
      { const int rc_ = topk_qreg_lds_attr(); if (rc_) return rc_; }
#define TK_LAUNCH_QREG(NK, GS) topk_partial_qreg<NK, GS><<<grid, dim3(256), lds, st>>>(Qn, Nq, Gn, Ng, D, k, self_mask, per, pcap, pval, pidx, gthr, 1, qmap, nq_dev)
      if (pcap >= 12) {                                       // pending columns long enough for groups of four scores
        if (D > 256) TK_LAUNCH_QREG(16, 4); else if (D > 128) TK_LAUNCH_QREG(8, 4); else TK_LAUNCH_QREG(4, 4);
      } else {
        if (D > 256) TK_LAUNCH_QREG(16, 2); else if (D > 128) TK_LAUNCH_QREG(8, 2); else TK_LAUNCH_QREG(4, 2);
      }
#undef TK_LAUNCH_QREG
    } else
#endif
    if (pcap >= 12) topk_partial_dma<4><<<grid, dim3(256), lds, st>>>(Qn, Nq, Gn, Ng, D, k, self_mask, per, pcap, pval, pidx, gthr);
    else topk_partial_dma<2><<<grid, dim3(256), lds, st>>>(Qn, Nq, Gn, Ng, D, k, self_mask, per, pcap, pval, pidx, gthr);
  } else
  topk_partial_kernel<<<grid, dim3(256), lds, st>>>(Qn, Nq, Gn, Ng, D, k, self_mask, per, (2 * pcap) & ~1, pval, pidx);
  SLIC_LAUNCH_CHECK();
  SLIC_REQUIRE((int64_t)S * k <= 64 * TKM_PER, "slic_cosine_topk: slices * k = %d exceeds the merge kernel's %d entries", S * k, 64 * TKM_PER);
  topk_merge_kernel<<<dim3((unsigned)slic_cdiv(Nq, 4)), dim3(256), 0, st>>>(pval, pidx, S, Nq, k, out_idx, out_dist, qmap, nq_dev);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

// threshold -> collect -> select -> exact fallback for the queries whose candidate count fell outside [k, cap]
static int topk_collect(const TopkCollectPlan& c, const float* Qn, int Nq, const float* Gn, int Ng, int D, int k, int self_mask,
                        int32_t* out_idx, float* out_dist, SlicCarver& w, hipStream_t st) {
  float* pval1 = w.take<float>((size_t)c.S1 * Nq * c.ms);
  int32_t* pidx1 = w.take<int32_t>((size_t)c.S1 * Nq * c.ms);
  float* tau = w.take<float>((size_t)Nq);
  int* cnt = w.take<int>((size_t)Nq);
  int* failq = w.take<int>((size_t)Nq);
  int* nfail = w.take<int>(64);
  unsigned long long* cand = w.take<unsigned long long>((size_t)Nq * c.cap);
  // ---- 1. the sample: S1 slices of per1 rows, every gstep-th row of the gallery, the m1 best of each (tiny heaps: LDS is no constraint)
  {
    const int kh = 1 + ((c.ms + 2) / 4) * 4;
    const int pcap = TK_PC_MAX;
    const size_t lds = (size_t)2 * 2 * TK_BQ * TK_BK * sizeof(float) + (size_t)4 * kh * 32 * 8 + (size_t)4 * pcap * 64 * 8 + 16;
    { const int rc_ = topk_qreg_lds_attr(); if (rc_) return rc_; }
    dim3 grid((unsigned)slic_cdiv(Nq, TK_BQ), (unsigned)c.S1);
    const int ns = c.S1 * c.per1;
#define TK_LAUNCH_S(NK) topk_partial_qreg<NK, 4><<<grid, dim3(256), lds, st>>>(Qn, Nq, Gn, ns, D, c.ms, 0, c.per1, pcap, pval1, pidx1, nullptr, c.gstep, nullptr, nullptr)
    if (D > 256) TK_LAUNCH_S(16); else if (D > 128) TK_LAUNCH_S(8); else TK_LAUNCH_S(4);
#undef TK_LAUNCH_S
    SLIC_LAUNCH_CHECK();
    topk_thresholds<<<dim3((unsigned)slic_cdiv(Nq, 4)), dim3(256), 0, st>>>(pval1, c.S1, Nq, c.ms, c.m1, tau, cnt, nfail);
    SLIC_LAUNCH_CHECK();
  }
  // ---- 2. the whole gallery against the thresholds
  {
    const int slices = topk_slices(Nq, Ng, 0);
    int per = (int)slic_cdiv(Ng, slices);
    per = (int)slic_cdiv(per, TK_BG) * TK_BG;
    const int S = (int)slic_cdiv(Ng, per);
    const int pcap = TK_PC_MAX;
    const size_t lds = (size_t)4 * TK_BG * TK_BK * sizeof(float) + (size_t)4 * pcap * 64 * 8;
    static bool attr_set = false;
    if (!attr_set) {
#define TK_ATTR(NK, GS) SLIC_HIP_CHECK(hipFuncSetAttribute((const void*)topk_collect_qreg<NK, GS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
      TK_ATTR(16, 4); TK_ATTR(8, 4); TK_ATTR(4, 4); TK_ATTR(16, 2); TK_ATTR(8, 2); TK_ATTR(4, 2); TK_ATTR(16, 1); TK_ATTR(8, 1); TK_ATTR(4, 1);
#undef TK_ATTR
      attr_set = true;
    }
    dim3 grid((unsigned)slic_cdiv(Nq, TK_BQ), (unsigned)S);
    // scores tested per wave-wide branch: with ~0.4 % of the scores passing, SOME lane of the wave passes in most groups of four, so the
    // group maximum of the streaming kernels buys little here; SLIC_TOPK_GS = 1 / 2 / 4 selects (experiments; default below)
    const char* ge = getenv("SLIC_TOPK_GS");
    const int gs = ge ? atoi(ge) : 4;
#define TK_LAUNCH_C(NK, GS) topk_collect_qreg<NK, GS><<<grid, dim3(256), lds, st>>>(Qn, Nq, Gn, Ng, D, self_mask, per, pcap, tau, cnt, cand, c.cap)
#define TK_LAUNCH_CG(NK) do { if (gs == 1) TK_LAUNCH_C(NK, 1); else if (gs == 2) TK_LAUNCH_C(NK, 2); else TK_LAUNCH_C(NK, 4); } while (0)
    if (D > 256) TK_LAUNCH_CG(16); else if (D > 128) TK_LAUNCH_CG(8); else TK_LAUNCH_CG(4);
#undef TK_LAUNCH_CG
#undef TK_LAUNCH_C
    SLIC_LAUNCH_CHECK();
  }
  // ---- 3. the k best candidates of every query; 4. whoever fell outside [k, TKC_CAP] through the streaming path (normally nobody)
  topk_select<<<dim3((unsigned)slic_cdiv(Nq, TKS_WAVES)), dim3(64 * TKS_WAVES), 0, st>>>(cand, cnt, Nq, k, c.cap, out_idx, out_dist, failq, nfail);
  SLIC_LAUNCH_CHECK();
  return topk_stream(Qn, Nq, Gn, Ng, D, k, self_mask, out_idx, out_dist, w, st, failq, nfail);
}

extern "C" int slic_cosine_topk_plan(int Nq, int Ng, int D, int k, int* out) {
  SLIC_REQUIRE(out && Nq > 0 && Ng > 0 && D > 0 && k >= 1, "slic_cosine_topk_plan: bad args");
  const TopkCollectPlan c = topk_collect_plan(Nq, Ng, k);
  const bool on = c.on && D <= 512 && D % 8 == 0 && (int64_t)Ng * D * 4 < (1ll << 31);
  out[0] = on ? 1 : 0; out[1] = on ? c.S1 : 0; out[2] = on ? c.per1 : 0; out[3] = on ? c.m1 : 0; out[4] = on ? c.gstep : 0; out[5] = on ? c.cap : 0;
  return SLIC_OK;
}

// Qn, Gn: L2-normalised rows (slic_normalize_rows).  out_idx / out_dist: [Nq, k], ascending distance.
extern "C" int slic_cosine_topk(const float* Qn, int Nq, const float* Gn, int Ng, int D, int k, int self_mask,
                                int32_t* out_idx, float* out_dist, void* workspace, void* stream) {
  SLIC_REQUIRE(Qn && Gn && out_idx && out_dist && workspace, "slic_cosine_topk: null pointer");
  SLIC_REQUIRE(Nq > 0 && Ng > 0 && D > 0 && D % 8 == 0 && k >= 1 && k <= TK_KMAX && k <= Ng,
               "slic_cosine_topk: need D %% 8 == 0, 1 <= k <= min(%d, Ng) (Nq=%d Ng=%d D=%d k=%d)", TK_KMAX, Nq, Ng, D, k);
  SLIC_REQUIRE(((uintptr_t)Qn % 16) == 0 && ((uintptr_t)Gn % 16) == 0, "slic_cosine_topk: unaligned");
  hipStream_t st = S_(stream);
  SlicCarver w(workspace);
  const TopkCollectPlan c = topk_collect_plan(Nq, Ng, k);
  if (c.on && D <= 512 && (int64_t)Ng * D * 4 < (1ll << 31))
    return topk_collect(c, Qn, Nq, Gn, Ng, D, k, self_mask, out_idx, out_dist, w, st);
  return topk_stream(Qn, Nq, Gn, Ng, D, k, self_mask, out_idx, out_dist, w, st, nullptr, nullptr);
}

extern "C" int slic_topk_merge_lists(const float* pdist, const int32_t* pidx, int W, int Nq, int k, int32_t* out_idx,
                                     float* out_dist, void* stream) {
  SLIC_REQUIRE(pdist && pidx && out_idx && out_dist && W > 0 && Nq > 0 && k > 0, "slic_topk_merge_lists: bad args");
  topk_merge_dist_kernel<<<dim3((unsigned)slic_cdiv(Nq, 64)), dim3(64), 0, S_(stream)>>>(pdist, pidx, W, Nq, k, out_idx, out_dist);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}

extern "C" int slic_pairwise_euclidean(const float* X, int Nx, const float* Y, int Ny, int D, float* out,
                                       void* stream) {
  SLIC_REQUIRE(X && Y && out && Nx > 0 && Ny > 0 && D > 0, "slic_pairwise_euclidean: bad args");
  const int64_t tot = (int64_t)Nx * Ny;
  pairwise_euclid_kernel<<<dim3((unsigned)slic_cdiv(tot, 256)), dim3(256), 0, S_(stream)>>>(X, Nx, Y, Ny, D, out);
  SLIC_LAUNCH_CHECK();
  return SLIC_OK;
}
