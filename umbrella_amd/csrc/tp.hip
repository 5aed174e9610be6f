// Tensor-parallel verify: the all-reduce behind the two row-split GEMMs of a layer as direct peer reads over xGMI
// (SURVEY 8(f)1 "fused all-reduce over xGMI"; :248 "prefer direct ... across the 7 links"; the reference has no
// distributed code).
//
// A [T, H] fp32 tile of a tree verify is 426 KB at T = 13: latency bound, 160 of them per 70B forward.  A ring
// collective pays 2 (P - 1) hops of that latency; xGMI is point to point, every GPU has a direct link to every other,
// so each rank can simply READ the P - 1 other tiles itself.  One all-reduce becomes two launches in the rank's own
// stream -- no library call, no ring, no extra kernel boundary next to the single-GPU schedule:
//
//   publish  (replaces umb_sum_splits): sums the S split-K slabs of this rank's partial tile into its exchange slot
//            (written through), the last block to finish releases at SYSTEM scope and stores the call's epoch into this
//            rank's flag word;
//   reduce   (replaces umb_reduce_residual_norm): one lane per peer polls that peer's flag for the epoch (relaxed
//            system-scope loads, one system-scope acquire after the last), then every row block reads the P tiles through
//            the peer-mapped pointers (system-coherent loads), sums them in RANK order -- every rank computes the same
//            bits -- and continues with the residual add and the RMSNorm exactly as the single-GPU kernel does.
//
// Exchange slots are double buffered by epoch parity: a rank that has passed reduce(c) may publish c + 1 into the other
// slot while a slower peer still reads slot c; it cannot reach publish(c + 2) before that peer has published c + 1, i.e.
// finished reduce(c) (stream order on the peer).  Epochs are device resident and only ever grow, so a captured iteration
// graph replays correctly.  The buffers are allocated fine-grained where the runtime allows it and mapped into the peers
// with hipIpc* (dmabuf); every spin is bounded and leaves a give-up code in the status word instead of hanging.
// Large tiles (prompt chunks, wide dynamic trees: a one-shot reduce moves P - 1 tiles per rank where a ring moves
// 2 (P - 1) / P) stay on the RCCL hook: model.hip picks per call from the tile size alone, the same on every rank.
#include "../../include/umbrella_hip.h"
#include "common.h"
#include <cstring>

static_assert(UMB_TP_MAX_RANKS == 16, "header / kernel agree on the rank limit");

struct TpPtrs {
  const float* slot[UMB_TP_MAX_RANKS];
  const unsigned* flag[UMB_TP_MAX_RANKS];
};

// ---- publish: slot[epoch & 1][i] = sum_s partial[s][i] ; flag = ++epoch (after a system-scope release)
__global__ __launch_bounds__(256) void tp_publish_kernel(const float* __restrict__ partial, int S, long n4, float* slot0,
                                                        long cap, unsigned* __restrict__ epoch_p,
                                                        unsigned* __restrict__ arrive, unsigned* flag) {
  const unsigned e = *epoch_p;                                   // calls completed so far: this one is e + 1
  f32x4* dst = reinterpret_cast<f32x4*>(slot0 + (long)((e + 1) & 1u) * cap);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 a = reinterpret_cast<const f32x4*>(partial)[i];
    for (int s = 1; s < S; ++s) a += reinterpret_cast<const f32x4*>(partial)[(long)s * n4 + i];   // split order
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rs, (int)(i * 16), 0, 17);   // sc0 sc1: write through
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                 // system scope: the tile is visible to the peers ...
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {                                    // ... before the last block publishes the epoch
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
      __hip_atomic_store(flag, e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(epoch_p, e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- reduce: h <- round(round(sum_r tile_r) + h) ; xn <- rmsnorm(h) * w   (umb_reduce_residual_norm over peer tiles)
// W: the loads of a row are unrolled to the smallest of 2 / 4 / 8 / 16 that covers `world` (round 4 always issued 16
// predicated loads per element).  ONE block polls the peers' flags over the link -- block 0, one lane per peer -- and
// releases the other T - 1 through a LOCAL word (`seen`, 64 bytes behind the status word): round 4 had T x (world - 1)
// lanes spinning on remote memory.
template <typename P, int W>
__global__ __launch_bounds__(1024) void tp_reduce_residual_norm_kernel(TpPtrs pp, int world, long cap,
                                                                       const unsigned* __restrict__ epoch_p, int T, int N,
                                                                       const u16* residual, u16* h_out,
                                                                       u16* __restrict__ xn_out,
                                                                       const u16* __restrict__ w, float eps,
                                                                       unsigned* __restrict__ status, long spin_limit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);            // 16 floats
  u16* row = reinterpret_cast<u16*>(smem + 64);           // N x u16
  __shared__ int gave_up;
  const int t = blockIdx.x;
  const unsigned e = *epoch_p;                             // the publish launch just ahead of this one made it e
  unsigned* seen = status + 16;                            // local release word (zero-initialised with the status line)
  if (threadIdx.x == 0) gave_up = 0;
  __syncthreads();
  if (t == 0) {
    if (threadIdx.x < world) {
      long spins = 0;
      while (__hip_atomic_load(pp.flag[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e > 0x7fffffffu) {   // flag < e (wrap safe)
        __builtin_amdgcn_s_sleep(2);
        if (++spins > spin_limit) { gave_up = 1; break; }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (gave_up) atomicExch(status, 0xDEAD0000u | (unsigned)t);      // a peer never arrived: say so (the host raises) ...
      __hip_atomic_store(seen, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... and let the other rows go either way
    }
  } else if (threadIdx.x == 0) {
    long spins = 0;
    while (__hip_atomic_load(seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - e > 0x7fffffffu) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > spin_limit) { gave_up = 1; break; }
    }
  }
  if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: nothing cached from the peers' tiles survives
  __syncthreads();
  if (gave_up) {
    if (threadIdx.x == 0 && t != 0) atomicExch(status, 0xDEAD0000u | (unsigned)t);
    return;
  }
  const long slot_off = (long)(e & 1u) * cap + (long)t * N;
  float ss = 0.f;
  for (int i = threadIdx.x * 4; i < N; i += 1024 * 4) {
    f32x4 v[W];
#pragma unroll
    for (int r = 0; r < W; ++r) {
      if (r < world) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pp.slot[r] + slot_off), 0, (unsigned)N * 4u, 0x00020000);
        v[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, i * 4, 0, 17));   // sc0 sc1: system coherent
      }
    }
    f32x4 a = v[0];
#pragma unroll
    for (int r = 1; r < W; ++r)
      if (r < world) a += v[r];                                 // rank order: the same bits on every rank
    float x0 = rnd<P>(a[0]), x1 = rnd<P>(a[1]), x2 = rnd<P>(a[2]), x3 = rnd<P>(a[3]);
    if (residual) {
      const uint2 r = *reinterpret_cast<const uint2*>(residual + (long)t * N + i);
      x0 += lo_f<P>(r.x); x1 += hi_f<P>(r.x); x2 += lo_f<P>(r.y); x3 += hi_f<P>(r.y);
    }
    uint2 o;
    o.x = pack2<P>(x0, x1); o.y = pack2<P>(x2, x3);
    if (h_out) *reinterpret_cast<uint2*>(h_out + (long)t * N + i) = o;
    *reinterpret_cast<uint2*>(row + i) = o;
    x0 = lo_f<P>(o.x); x1 = hi_f<P>(o.x); x2 = lo_f<P>(o.y); x3 = hi_f<P>(o.y);
    ss += x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
  }
  if (!xn_out) return;
  ss = block_sum<1024>(ss, red);
  const float inv = rsqrtf(ss / (float)N + eps);
  for (int i = threadIdx.x * 4; i < N; i += 1024 * 4) {
    const uint2 v = *reinterpret_cast<const uint2*>(row + i);
    const uint2 g = *reinterpret_cast<const uint2*>(w + i);
    uint2 o;
    o.x = pack2<P>(lo_f<P>(v.x) * inv * lo_f<P>(g.x), hi_f<P>(v.x) * inv * hi_f<P>(g.x));
    o.y = pack2<P>(lo_f<P>(v.y) * inv * lo_f<P>(g.y), hi_f<P>(v.y) * inv * hi_f<P>(g.y));
    *reinterpret_cast<uint2*>(xn_out + (long)t * N + i) = o;
  }
}

static inline bool peer_ok(const UmbTPPeer* p) {
  if (!p || p->world < 2 || p->world > UMB_TP_MAX_RANKS || p->rank < 0 || p->rank >= p->world || !p->epoch || !p->arrive ||
      !p->status || p->cap < 4)
    return false;
  for (int r = 0; r < p->world; ++r)
    if (!p->slot[r] || !p->flag[r]) return false;
  return true;
}

extern "C" int umb_tp_publish(const UmbTPPeer* p, const float* partial, int S, int64_t n, hipStream_t st) {
  if (!peer_ok(p) || !partial || S < 1 || n < 4 || (n & 3) || n > p->cap) return UMB_EINVAL;
  const long n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256);
  hipLaunchKernelGGL(tp_publish_kernel, dim3(grid), dim3(256), 0, st, partial, S, n4, p->slot[p->rank], (long)p->cap, p->epoch,
                     p->arrive, p->flag[p->rank]);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

extern "C" int umb_tp_reduce_residual_norm(const UmbTPPeer* p, int T, int N, const void* residual, void* h_out,
                                           void* xn_out, const void* w, float eps, int dtype, hipStream_t st) {
  if (!peer_ok(p) || T < 1 || N % 4 || (long)T * N > p->cap || (xn_out && !w)) return UMB_EINVAL;
  TpPtrs pp = {};
  for (int r = 0; r < p->world; ++r) { pp.slot[r] = p->slot[r]; pp.flag[r] = p->flag[r]; }
  const size_t sm = 64 + (size_t)N * 2;
  // x ~0.13 us per s_sleep 2 round: ~70 s by default.  A collective would simply wait; the bound only keeps a dead peer from
  // hanging the device for ever, and the engines read the status word with every iteration's accept result (round 5).
  const long spin = p->spin_limit > 0 ? p->spin_limit : (1l << 29);
#define TP_GO(WV) hipLaunchKernelGGL((tp_reduce_residual_norm_kernel<P, WV>), dim3(T), dim3(1024), sm, st, pp, p->world,       \
                                     (long)p->cap, p->epoch, T, N, (const u16*)residual, (u16*)h_out, (u16*)xn_out,             \
                                     (const u16*)w, eps, p->status, spin)
  DISPATCH_DTYPE(dtype, {
    if (p->world <= 2) TP_GO(2);
    else if (p->world <= 4) TP_GO(4);
    else if (p->world <= 8) TP_GO(8);
    else TP_GO(16);
  })
#undef TP_GO
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// ---- exchange memory: device allocation (fine-grained where the runtime allows it) + interprocess handles
extern "C" int umb_tp_xchg_alloc(size_t bytes, void** ptr, void* handle64, int* fine_grained) {
  if (!ptr || !handle64 || bytes == 0) return UMB_EINVAL;
  void* p = nullptr;
  int fg = 1;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) {
    (void)hipGetLastError();
    fg = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) return UMB_EHIP;
  }
  if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return UMB_EHIP; }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle travels as 64 bytes");
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipGetLastError();
    if (!fg) { (void)hipFree(p); return UMB_EHIP; }
    // some runtimes refuse interprocess handles for fine-grained allocations: fall back to an ordinary one
    (void)hipFree(p);
    fg = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) return UMB_EHIP;
    if (hipMemset(p, 0, bytes) != hipSuccess || hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipFree(p); return UMB_EHIP; }
  }
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return UMB_EHIP; }
  memcpy(handle64, &h, 64);
  *ptr = p;
  if (fine_grained) *fine_grained = fg;
  return UMB_OK;
}

extern "C" int umb_tp_xchg_open(const void* handle64, void** ptr) {
  if (!handle64 || !ptr) return UMB_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return UMB_EHIP; }
  *ptr = p;
  return UMB_OK;
}

extern "C" int umb_tp_xchg_close(void* ptr) { return (ptr && hipIpcCloseMemHandle(ptr) != hipSuccess) ? UMB_EHIP : UMB_OK; }
extern "C" int umb_tp_xchg_free(void* ptr) { return (ptr && hipFree(ptr) != hipSuccess) ? UMB_EHIP : UMB_OK; }
