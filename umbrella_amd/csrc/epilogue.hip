// Split-K reduce epilogues fused with the decoder layer's elementwise work.
// Every kernel reads fp32 partials [S][T][N] written by skinny_gemm_kernel and
// reduces them in split order 0..S-1 (fixed order -> deterministic).
//
// Reference ops fused here (umbrella/models/llama.py:75-114):
//   residual add + RMSNorm (flashinfer.rmsnorm, model_utils.py:54-64)
//   SiLU(gate) * up
//   q/k/v view + RoPE at tree positions (model_utils.py:17-52) + KV append (attn/cache.py:53-65)
//   embedding gather (F.embedding, llama.py:124)
#include "common.h"
// split-K partials are read exactly once, by a kernel on other XCDs than their writers: non-temporal loads (70B layer at T = 13:
// -0.7 us over its two residual reduces; profiles/r05_handoff_stores.txt)
#define LDP(p) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))

// ---- plain RMSNorm over rows of 16-bit x (also the standalone umb_rmsnorm op)
template <typename P>
__global__ __launch_bounds__(256) void rmsnorm_kernel(u16* __restrict__ out, const u16* __restrict__ x,
                                                      const u16* __restrict__ w, float eps, int H, int fm_tt) {
  // fm_tt > 0: out in FM order with fm_tt token tiles (the split schedule's GEMMs read coalesced KiB fragments, round 4)
  __shared__ float red[4];
  const int t = blockIdx.x;
  const u16* xr = x + (long)t * H;
  float ss = 0.f;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float a = lo_f<P>(v[e]), b = hi_f<P>(v[e]); ss += a * a + b * b; }
  }
  ss = block_sum<256>(ss, red);
  const float inv = rsqrtf(ss / (float)H + eps);
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + i);
    const u32x4 g = *reinterpret_cast<const u32x4*>(w + i);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack2<P>(lo_f<P>(v[e]) * inv * lo_f<P>(g[e]), hi_f<P>(v[e]) * inv * hi_f<P>(g[e]));
    *reinterpret_cast<u32x4*>(out + (fm_tt ? fm_off(t, i, fm_tt) : (long)t * H + i)) = o;
  }
}

// ---- h = residual + sum_s partial ; xn = rmsnorm(h) * w   (one 1024-thread block per token row)
// h_out may alias residual.  xn_out / w may be null (no norm), residual may be null.
// The split loop is unrolled 4-wide so the partial loads are independent (latency, not bandwidth, bound).
__device__ __forceinline__ f32x4 sum_splits(const float* __restrict__ p, int S, long sstride) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  int s = 1;
  for (; s + 3 < S; s += 4) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p + (long)s * sstride);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p + (long)(s + 1) * sstride);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(p + (long)(s + 2) * sstride);
    const f32x4 b3 = *reinterpret_cast<const f32x4*>(p + (long)(s + 3) * sstride);
    a += b0; a += b1; a += b2; a += b3;          // fixed order 0..S-1
  }
  for (; s < S; ++s) a += *reinterpret_cast<const f32x4*>(p + (long)s * sstride);
  return a;
}

template <typename P>
__global__ __launch_bounds__(1024) void reduce_residual_norm_kernel(const float* __restrict__ part, int S, int T, int N,
                                                                    const u16* residual, u16* h_out,
                                                                    u16* __restrict__ xn_out,
                                                                    const u16* __restrict__ w, float eps, int xn_fm_tt) {
  // xn_fm_tt > 0: xn_out in FM order with that many token tiles (h_out stays row-major)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);            // 16 floats
  u16* row = reinterpret_cast<u16*>(smem + 64);           // N x u16
  const int t = blockIdx.x;
  const long sstride = (long)T * N;
  float ss = 0.f;
  const bool deep = N <= 4096;                               // one column group per thread: room for 16 splits
  if ((S <= 8 && N <= 8192) || (S <= 16 && deep)) {
    // Whole row in registers (<= 2 column groups per thread, or 1 group of <= 16 splits): every load of the kernel --
    // partials, residual, norm weight -- is issued before the first add, and the normalised row is produced from
    // registers (no LDS row).
    const int i0 = threadIdx.x * 4, i1 = i0 + 4096;
    const bool ok0 = i0 < N, ok1 = i1 < N;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v0[8], v1[8];                                        // deep: v1 = splits 8..15 of group 0
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      v0[s2] = (ok0 && s2 < S) ? LDP(part + (long)t * N + i0 + (long)s2 * sstride) : z;
      const bool use1 = deep ? (ok0 && s2 + 8 < S) : (ok1 && s2 < S);
      const long off1 = deep ? (long)t * N + i0 + (long)(s2 + 8) * sstride : (long)t * N + i1 + (long)s2 * sstride;
      v1[s2] = use1 ? LDP(part + off1) : z;
    }
    const uint2 z2 = {0u, 0u};
    const uint2 r0 = (residual && ok0) ? *reinterpret_cast<const uint2*>(residual + (long)t * N + i0) : z2;
    const uint2 r1 = (residual && ok1) ? *reinterpret_cast<const uint2*>(residual + (long)t * N + i1) : z2;
    const uint2 g0 = (xn_out && ok0) ? *reinterpret_cast<const uint2*>(w + i0) : z2;
    const uint2 g1 = (xn_out && ok1) ? *reinterpret_cast<const uint2*>(w + i1) : z2;
    auto finish = [&](f32x4 a, const uint2& r, bool ok, int i) -> uint2 {
      float x0 = rnd<P>(a[0]), x1 = rnd<P>(a[1]), x2 = rnd<P>(a[2]), x3 = rnd<P>(a[3]);
      if (residual) { x0 += lo_f<P>(r.x); x1 += hi_f<P>(r.x); x2 += lo_f<P>(r.y); x3 += hi_f<P>(r.y); }
      uint2 o;
      o.x = pack2<P>(x0, x1); o.y = pack2<P>(x2, x3);
      if (ok) {
        if (h_out) *reinterpret_cast<uint2*>(h_out + (long)t * N + i) = o;
        x0 = lo_f<P>(o.x); x1 = hi_f<P>(o.x); x2 = lo_f<P>(o.y); x3 = hi_f<P>(o.y);
        ss += x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
      }
      return o;
    };
    f32x4 a0 = v0[0], a1 = v1[0];
#pragma unroll
    for (int s2 = 1; s2 < 8; ++s2)
      if (s2 < S) a0 += v0[s2];                                  // fixed order 0..S-1
    if (deep) {
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2)
        if (s2 + 8 < S) a0 += v1[s2];
    } else {
#pragma unroll
      for (int s2 = 1; s2 < 8; ++s2)
        if (s2 < S) a1 += v1[s2];
    }
    const uint2 o0 = finish(a0, r0, ok0, i0);
    const uint2 o1 = finish(a1, r1, ok1 && !deep, i1);
    if (!xn_out) return;
    ss = block_sum<1024>(ss, red);
    const float inv = rsqrtf(ss / (float)N + eps);
    auto norm = [&](const uint2& v, const uint2& g, int i) {
      uint2 o;
      o.x = pack2<P>(lo_f<P>(v.x) * inv * lo_f<P>(g.x), hi_f<P>(v.x) * inv * hi_f<P>(g.x));
      o.y = pack2<P>(lo_f<P>(v.y) * inv * lo_f<P>(g.y), hi_f<P>(v.y) * inv * hi_f<P>(g.y));
      *reinterpret_cast<uint2*>(xn_out + (xn_fm_tt ? fm_off(t, i, xn_fm_tt) : (long)t * N + i)) = o;
    };
    if (ok0) norm(o0, g0, i0);
    if (ok1) norm(o1, g1, i1);
    return;
  }
  // S <= 8 (every planned layer GEMM): all partial loads of a pair of column groups are issued before the first add.
  // The kernel is one block per row and latency bound -- with the loads chained behind the adds a row of 8 splits
  // cost five dependent round trips per group (7.3 us at N = 8192); now it is one.
  const bool wide_issue = S <= 8;
  for (int i = threadIdx.x * 4; i < N; i += 1024 * 4) {
    f32x4 a;
    if (wide_issue) {
      f32x4 v[8];
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2)
        v[s2] = s2 < S ? LDP(part + (long)t * N + i + (long)s2 * sstride) : f32x4{0.f, 0.f, 0.f, 0.f};
      a = v[0];
#pragma unroll
      for (int s2 = 1; s2 < 8; ++s2)
        if (s2 < S) a += v[s2];                       // fixed order 0..S-1
    } else {
      a = sum_splits(part + (long)t * N + i, S, sstride);
    }
    // GEMM output is rounded to the model dtype before the residual add (F.linear returns dtype)
    float v0 = rnd<P>(a[0]), v1 = rnd<P>(a[1]), v2 = rnd<P>(a[2]), v3 = rnd<P>(a[3]);
    if (residual) {
      const uint2 r = *reinterpret_cast<const uint2*>(residual + (long)t * N + i);
      v0 += lo_f<P>(r.x); v1 += hi_f<P>(r.x); v2 += lo_f<P>(r.y); v3 += hi_f<P>(r.y);
    }
    uint2 o;
    o.x = pack2<P>(v0, v1); o.y = pack2<P>(v2, v3);
    if (h_out) *reinterpret_cast<uint2*>(h_out + (long)t * N + i) = o;
    *reinterpret_cast<uint2*>(row + i) = o;
    v0 = lo_f<P>(o.x); v1 = hi_f<P>(o.x); v2 = lo_f<P>(o.y); v3 = hi_f<P>(o.y);
    ss += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
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
    *reinterpret_cast<uint2*>(xn_out + (xn_fm_tt ? fm_off(t, i, xn_fm_tt) : (long)t * N + i)) = o;
  }
}

// ---- act[t][i] = silu(gate) * up ; partial rows hold [gate(0..I) | up(I..2I)]
template <typename P>
__global__ __launch_bounds__(256) void reduce_silu_mul_kernel(const float* __restrict__ part, int S, int T, int I,
                                                              u16* __restrict__ act) {
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (gid >= (long)T * I) return;
  const int t = (int)(gid / I), i = (int)(gid % I);
  const int N = 2 * I;
  f32x4 g = *reinterpret_cast<const f32x4*>(part + (long)t * N + i);
  f32x4 u = *reinterpret_cast<const f32x4*>(part + (long)t * N + I + i);
  for (int s = 1; s < S; ++s) {
    g += *reinterpret_cast<const f32x4*>(part + ((long)s * T + t) * N + i);
    u += *reinterpret_cast<const f32x4*>(part + ((long)s * T + t) * N + I + i);
  }
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gg = rnd<P>(g[e]), uu = rnd<P>(u[e]);     // both GEMM outputs live in the model dtype
    const float sg = rnd<P>(gg / (1.f + __expf(-gg)));    // F.silu in dtype
    o[e] = sg * uu;
  }
  uint2 pk;
  pk.x = pack2<P>(o[0], o[1]); pk.y = pack2<P>(o[2], o[3]);
  *reinterpret_cast<uint2*>(act + (long)t * I + i) = pk;
}

// ---- split-K reduce + residual with the RMSNorm DEFERRED (round 6): many blocks per token row.
// reduce_residual_norm_kernel above is one block per row because the norm needs the row's sum of squares: at T = 13 that is
// 13 CUs each pulling S x N fp32 partials (128-256 KB at the 70B widths) at one CU's ~60-100 GB/s -- 2 of the launch's ~5.9 us.
// Here a block owns 512 columns of one row: h <- round(round(sum_s partial) + h), hw <- round(h * w_next) (the next norm's
// WEIGHT folded in, as the GEMM epilogue EPI_RESID and the low-latency schedule do), ssq[t][block] <- sum h^2 over its
// columns; the consumer GEMM / q-k-v reduce applies 1/rms = rsqrt(sum_b ssq[t][b] / N + eps) to its outputs, where it
// commutes with the matmul.  T x N/512 blocks, no cross-block step.  Reference: layer_norm + residual add,
// umbrella/models/llama.py:95-96,112-113 and model_utils.py:54-64.
template <typename P>
__global__ __launch_bounds__(128) void reduce_residual_hw_kernel(const float* __restrict__ part, int S, int T, int N,
                                                                 const u16* residual, u16* h_out, u16* __restrict__ hw_out,
                                                                 const u16* __restrict__ w, float* __restrict__ ssq_out,
                                                                 int ssq_stride, int hw_fm_tt) {
  __shared__ float red[2];
  const int b = blockIdx.x, t = blockIdx.y;
  const int i = b * 512 + threadIdx.x * 4;
  const long sstride = (long)T * N;
  const float* p = part + (long)t * N + i;
  f32x4 v[16];
#pragma unroll
  for (int s2 = 0; s2 < 16; ++s2) v[s2] = s2 < S ? LDP(p + (long)s2 * sstride) : f32x4{0.f, 0.f, 0.f, 0.f};
  const uint2 r = residual ? *reinterpret_cast<const uint2*>(residual + (long)t * N + i) : uint2{0u, 0u};
  const uint2 g = hw_out ? *reinterpret_cast<const uint2*>(w + i) : uint2{0u, 0u};
  f32x4 a = v[0];
#pragma unroll
  for (int s2 = 1; s2 < 16; ++s2)
    if (s2 < S) a += v[s2];                                        // fixed order 0 .. S-1
  for (int s2 = 16; s2 < S; ++s2) a += LDP(p + (long)s2 * sstride);
  float x0 = rnd<P>(a[0]), x1 = rnd<P>(a[1]), x2 = rnd<P>(a[2]), x3 = rnd<P>(a[3]);
  if (residual) { x0 += lo_f<P>(r.x); x1 += hi_f<P>(r.x); x2 += lo_f<P>(r.y); x3 += hi_f<P>(r.y); }
  uint2 o;
  o.x = pack2<P>(x0, x1); o.y = pack2<P>(x2, x3);
  if (h_out) *reinterpret_cast<uint2*>(h_out + (long)t * N + i) = o;
  x0 = lo_f<P>(o.x); x1 = hi_f<P>(o.x); x2 = lo_f<P>(o.y); x3 = hi_f<P>(o.y);
  if (hw_out) {
    uint2 ow;
    ow.x = pack2<P>(x0 * lo_f<P>(g.x), x1 * hi_f<P>(g.x)); ow.y = pack2<P>(x2 * lo_f<P>(g.y), x3 * hi_f<P>(g.y));
    *reinterpret_cast<uint2*>(hw_out + (hw_fm_tt ? fm_off(t, i, hw_fm_tt) : (long)t * N + i)) = ow;
  }
  if (!ssq_out) return;
  float ss = wave_sum(x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) ssq_out[(long)t * ssq_stride + b] = red[0] + red[1];
}

// ---- QKV reduce + RoPE + KV append.  One block per (token, head) with D/2 active pairs.
// partial row layout: [q (Hq*D) | k (Hkv*D) | v (Hkv*D)]
// K cache / V^T cache: per kv head a slab of Lmax D / D (Lmax + UMB_VT_PAD) elements in MFMA fragment order (common.h kc_off / vt_off)
template <typename P>
__global__ __launch_bounds__(64) void reduce_qkv_rope_kernel(const float* __restrict__ part, int S, int T, int Hq,
                                                             int Hkv, int D, int Lmax, const int* __restrict__ pos,
                                                             const int* __restrict__ slot,
                                                             const u16* __restrict__ cosT, const u16* __restrict__ sinT,
                                                             u16* __restrict__ q_out, u16* __restrict__ kc,
                                                             u16* __restrict__ vt, int paired,
                                                             const u16* __restrict__ bias,
                                                             const float* __restrict__ ssq_in, int ssq_groups,
                                                             int ssq_stride, float ssq_dim, float eps) {
  const int t = blockIdx.x, head = blockIdx.y;             // head in [0, Hq + 2*Hkv)
  const int N = (Hq + 2 * Hkv) * D;
  const int half = D / 2;
  const int p = pos[t], sl = slot[t];
  // low-latency schedule: the GEMM ran on h * w (norm weight folded by the producer); the per-token 1/rms from the
  // producer's sums of squares is applied to the reduced outputs here (it commutes with the matmul)
  float inv = 1.f;
  if (ssq_in) {
    float s = 0.f;
    for (int q = threadIdx.x; q < ssq_groups; q += 64) s += ssq_in[(long)t * ssq_stride + q];
    inv = rsqrtf(wave_sum(s) / ssq_dim + eps);
  }
  const float* base = part + (long)t * N + head * D;
  const long sstride = (long)T * N;
  // paired: the q/k rows were packed as RoPE partner pairs (repack mode 2): columns (2d, 2d+1) <-> (d, d + D/2)
  const bool pr = paired && head < Hq + Hkv;
  for (int d = threadIdx.x; d < half; d += 64) {
    const int ca = pr ? 2 * d : d, cb = pr ? 2 * d + 1 : d + half;
    float a, b;
    if (S <= 8) {                                    // all partial loads in flight before the first add (latency bound)
      float va[8], vb[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        va[s] = s < S ? base[s * sstride + ca] : 0.f;
        vb[s] = s < S ? base[s * sstride + cb] : 0.f;
      }
      a = va[0]; b = vb[0];
#pragma unroll
      for (int s = 1; s < 8; ++s)
        if (s < S) { a += va[s]; b += vb[s]; }       // fixed order 0..S-1
    } else {
      a = base[ca]; b = base[cb];
      for (int s = 1; s < S; ++s) { a += base[s * sstride + ca]; b += base[s * sstride + cb]; }
    }
    a *= inv; b *= inv;
    if (bias) {        // F.linear(x, W, b) (qwen.py:94-96): bias joins the fp32 accumulator, one rounding; HF feature order
      a += P::to_f(bias[head * D + d]); b += P::to_f(bias[head * D + d + half]);
    }
    a = rnd<P>(a); b = rnd<P>(b);
    if (head < Hq + Hkv) {
      // rotate-half RoPE in the model dtype (each product and the sum are rounded, as eager torch does)
      const float c0 = P::to_f(cosT[(long)p * D + d]), c1 = P::to_f(cosT[(long)p * D + d + half]);
      const float s0 = P::to_f(sinT[(long)p * D + d]), s1 = P::to_f(sinT[(long)p * D + d + half]);
      const float o0 = rnd<P>(mul_rnd<P>(a, c0) + mul_rnd<P>(-b, s0));
      const float o1 = rnd<P>(mul_rnd<P>(b, c1) + mul_rnd<P>(a, s1));
      if (head < Hq) {
        u16* qo = q_out + ((long)t * Hq + head) * D;
        qo[d] = P::from_f(o0); qo[d + half] = P::from_f(o1);
      } else {
        u16* ko = kc + (long)(head - Hq) * Lmax * D;            // fragment order inside the head's slab (common.h)
        ko[kc_off(sl, d, D)] = P::from_f(o0); ko[kc_off(sl, d + half, D)] = P::from_f(o1);
      }
    } else {
      u16* vo = vt + (long)(head - Hq - Hkv) * D * VT_LD(Lmax);
      vo[vt_off(d, sl, D)] = P::from_f(a);
      vo[vt_off(d + half, sl, D)] = P::from_f(b);
    }
  }
}

// ---- embedding gather + per-forward index prep.
// TREE mode  (tokens_all != null): token i = tokens_all[n + off + i], pos = n + depth[off+i], slot = n + off + i
// EXPLICIT   : tokens/pos/slot given; they are copied into the workspace arrays the later kernels read.
// Fused-layer extras (optional): hw = h * norm_w (the first RMSNorm's weight folded into the activations) and
// ssq[t][g] = sum of h^2 over columns [64g, 64g+64) -- what the GEMM epilogues turn into 1/rms.
template <typename P>
__global__ __launch_bounds__(256) void embed_prep_kernel(u16* __restrict__ x, const u16* __restrict__ table, int H,
                                                         const int* __restrict__ tok_in, const int* __restrict__ pos_in,
                                                         const int* __restrict__ slot_in,
                                                         const int* __restrict__ prefix_in,
                                                         const int* __restrict__ tokens_all,
                                                         const int* __restrict__ n_ptr, int off,
                                                         const int* __restrict__ depth, int* __restrict__ pos_out,
                                                         int* __restrict__ slot_out, int* __restrict__ prefix_out,
                                                         u16* __restrict__ hw, const u16* __restrict__ norm_w,
                                                         float* __restrict__ ssq, int ssq_stride, int hw_fm_tt) {
  const int i = blockIdx.x;
  int tok, p, s, pre;
  if (tokens_all) {
    const int n = *n_ptr;
    tok = table ? tokens_all[n + off + i] : 0; p = n + depth[off + i]; s = n + off + i; pre = n;
  } else {
    tok = table ? tok_in[i] : 0; p = pos_in[i]; s = slot_in[i]; pre = *prefix_in;
  }
  if (threadIdx.x == 0) { pos_out[i] = p; slot_out[i] = s; if (i == 0) *prefix_out = pre; }
  // pipeline stages > 0 (table == NULL): x already holds the activations of the previous stage
  const u32x4* src = reinterpret_cast<const u32x4*>(table ? table + (long)tok * H : x + (long)i * H);
  u32x4* dst = reinterpret_cast<u32x4*>(x + (long)i * H);
  for (int k = threadIdx.x; k < H / 8; k += 256) {
    const u32x4 v = src[k];
    if (table) dst[k] = v;
    if (hw) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(norm_w + k * 8);
      u32x4 o;
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = lo_f<P>(v[e]), b = hi_f<P>(v[e]);
        o[e] = pack2<P>(a * lo_f<P>(w[e]), b * hi_f<P>(w[e]));
        sq += a * a + b * b;
      }
      *reinterpret_cast<u32x4*>(hw + (hw_fm_tt ? fm_off(i, k * 8, hw_fm_tt) : (long)i * H + k * 8)) = o;
      sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);   // 8 threads = 64 columns
      if ((threadIdx.x & 7) == 0) ssq[(long)i * ssq_stride + (k >> 3)] = sq;
    }
  }
}

// ---- stand-alone 16-bit RoPE / KV append (per-call-site integration: the reference applies them to q / k / v tensors
// in the model dtype, model_utils.py:17-52 and attn/cache.py:53-65; the fused forms above take fp32 split-K partials)
// q [T][Hq][D] / k [T][Hkv][D] (layout 0, NHD: what apply_rotary_pos_emb sees as [1, T, H, D]) or [H][T][D] (layout 1),
// rotated in place at pos[t]; one 64-lane block per (token, head), lane d handles the pair (d, d + D/2).
template <typename P>
__global__ __launch_bounds__(64) void rope_inplace_kernel(u16* __restrict__ q, u16* __restrict__ k,
                                                          const u16* __restrict__ cosT, const u16* __restrict__ sinT,
                                                          const int* __restrict__ pos, int T, int Hq, int Hkv, int D,
                                                          int layout) {
  const int t = blockIdx.x, head = blockIdx.y, half = D / 2;
  const bool isq = head < Hq;
  const int hh = isq ? head : head - Hq, H = isq ? Hq : Hkv;
  u16* base = (isq ? q : k) + (layout == 0 ? ((long)t * H + hh) * D : ((long)hh * T + t) * D);
  const long cb = (long)pos[t] * D;
  for (int d = threadIdx.x; d < half; d += 64) {
    const float a = P::to_f(base[d]), b = P::to_f(base[d + half]);
    const float c0 = P::to_f(cosT[cb + d]), c1 = P::to_f(cosT[cb + d + half]);
    const float s0 = P::to_f(sinT[cb + d]), s1 = P::to_f(sinT[cb + d + half]);
    // x * cos + rotate_half(x) * sin in the model dtype: every product and the sum are rounded, as eager torch does
    base[d] = P::from_f(mul_rnd<P>(a, c0) + mul_rnd<P>(-b, s0));
    base[d + half] = P::from_f(mul_rnd<P>(b, c1) + mul_rnd<P>(a, s1));
  }
}

// k / v [T][Hkv][D] -> key slot[t] of the K cache and of the V^T cache (fragment order, common.h)
__global__ __launch_bounds__(64) void kv_append_kernel(u16* __restrict__ kc, u16* __restrict__ vt,
                                                       const u16* __restrict__ k, const u16* __restrict__ v,
                                                       const int* __restrict__ slot, int Hkv, int D, int Lmax) {
  const int t = blockIdx.x, h = blockIdx.y;
  const int sl = slot[t];
  if (sl < 0 || sl >= Lmax) return;                               // a slot outside the cache is dropped, never written
  const u16* ks = k + ((long)t * Hkv + h) * D;
  const u16* vs = v + ((long)t * Hkv + h) * D;
  u16* kd = kc + (long)h * Lmax * D;                              // fragment order inside the head's slab (common.h)
  u16* vd = vt + (long)h * D * VT_LD(Lmax);
  for (int d = threadIdx.x; d < D; d += 64) {
    kd[kc_off(sl, d, D)] = ks[d];
    vd[vt_off(d, sl, D)] = vs[d];
  }
}

// ---- partial[0] += partial[1] + ... + partial[S-1], split order (tensor-parallel wide forwards: one slab per all-reduce)
__global__ __launch_bounds__(256) void sum_splits_kernel(float* __restrict__ part, int S, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 a = reinterpret_cast<const f32x4*>(part)[i];
    for (int s = 1; s < S; ++s) a += reinterpret_cast<const f32x4*>(part)[(long)s * n4 + i];
    reinterpret_cast<f32x4*>(part)[i] = a;
  }
}

// ------------------------------------------------------------------ C entry points
// fm_tt: 0 = row-major out [rows][H]; 1 / 2 / 4 = out in FM order with that many 16-token tiles (rows <= 16 fm_tt, H % 32 == 0)
extern "C" int umb_rmsnorm_fm(void* out, const void* x, const void* w, float eps, int rows, int H, int fm_tt, int dtype,
                              hipStream_t st) {
  if (H % 8 || rows < 1 || fm_tt < 0 || (fm_tt && (rows > 16 * fm_tt || H % 32))) return UMB_EINVAL;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((rmsnorm_kernel<P>), dim3(rows), dim3(256), 0, st, (u16*)out, (const u16*)x, (const u16*)w, eps, H, fm_tt);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

extern "C" int umb_rmsnorm(void* out, const void* x, const void* w, float eps, int rows, int H, int dtype,
                           hipStream_t st) {
  return umb_rmsnorm_fm(out, x, w, eps, rows, H, 0, dtype, st);
}

// xn_fm_tt: as fm_tt above, for xn_out only
extern "C" int umb_reduce_residual_norm_fm(const void* partial, int S, int T, int N, const void* residual, void* h_out,
                                           void* xn_out, const void* w, float eps, int xn_fm_tt, int dtype, hipStream_t st) {
  if (N % 4 || T < 1 || xn_fm_tt < 0 || (xn_fm_tt && (T > 16 * xn_fm_tt || N % 32))) return UMB_EINVAL;
  const size_t sm = 64 + (size_t)N * 2;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((reduce_residual_norm_kernel<P>), dim3(T), dim3(1024), sm, st, (const float*)partial, S, T, N,
                       (const u16*)residual, (u16*)h_out, (u16*)xn_out, (const u16*)w, eps, xn_fm_tt);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

extern "C" int umb_reduce_residual_norm(const void* partial, int S, int T, int N, const void* residual, void* h_out,
                                        void* xn_out, const void* w, float eps, int dtype, hipStream_t st) {
  return umb_reduce_residual_norm_fm(partial, S, T, N, residual, h_out, xn_out, w, eps, 0, dtype, st);
}

// Many-blocks-per-row reduce with the norm deferred (reduce_residual_hw_kernel): h_out <- round(round(sum_s partial) + residual),
// hw_out (may be NULL) <- round(h * w) row-major or FM (hw_fm_tt token tiles), ssq_out[t][0 .. N/512) <- per-block sums of h^2.
extern "C" int umb_reduce_residual_hw(const void* partial, int S, int T, int N, const void* residual, void* h_out, void* hw_out,
                                      const void* w, float* ssq_out, int ssq_stride, int hw_fm_tt, int dtype, hipStream_t st) {
  if (N % 512 || T < 1 || S < 1 || hw_fm_tt < 0 || (hw_fm_tt && T > 16 * hw_fm_tt) || (hw_out && !w) ||
      (ssq_out && ssq_stride < N / 512))
    return UMB_EINVAL;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((reduce_residual_hw_kernel<P>), dim3(N / 512, T), dim3(128), 0, st, (const float*)partial, S, T, N,
                       (const u16*)residual, (u16*)h_out, (u16*)hw_out, (const u16*)w, ssq_out, ssq_stride, hw_fm_tt);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

extern "C" int umb_reduce_silu_mul(const void* partial, int S, int T, int I, void* act, int dtype, hipStream_t st) {
  if (I % 4) return UMB_EINVAL;
  const long n4 = (long)T * I / 4;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((reduce_silu_mul_kernel<P>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       (const float*)partial, S, T, I, (u16*)act);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

extern "C" int umb_reduce_qkv_rope2(const void* partial, int S, int T, int Hq, int Hkv, int D, int Lmax,
                                    const int* pos, const int* slot, const void* cosT, const void* sinT, void* q_out,
                                    void* k_cache, void* vt_cache, int paired, const void* bias, const float* ssq_in,
                                    int ssq_groups, int ssq_stride, float ssq_dim, float eps, int dtype, hipStream_t st) {
  if (D % 32 || Lmax % 32 || (ssq_in && (ssq_groups < 1 || ssq_stride < ssq_groups || ssq_dim <= 0.f))) return UMB_EINVAL;   // 32-key x 32-feature cache tiles
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((reduce_qkv_rope_kernel<P>), dim3(T, Hq + 2 * Hkv), dim3(64), 0, st, (const float*)partial, S, T,
                       Hq, Hkv, D, Lmax, pos, slot, (const u16*)cosT, (const u16*)sinT, (u16*)q_out, (u16*)k_cache,
                       (u16*)vt_cache, paired, (const u16*)bias, ssq_in, ssq_groups, ssq_stride, ssq_dim, eps);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

extern "C" int umb_reduce_qkv_rope(const void* partial, int S, int T, int Hq, int Hkv, int D, int Lmax,
                                   const int* pos, const int* slot, const void* cosT, const void* sinT, void* q_out,
                                   void* k_cache, void* vt_cache, int paired, const void* bias, int dtype,
                                   hipStream_t st) {
  return umb_reduce_qkv_rope2(partial, S, T, Hq, Hkv, D, Lmax, pos, slot, cosT, sinT, q_out, k_cache, vt_cache, paired,
                              bias, nullptr, 0, 0, 0.f, 0.f, dtype, st);
}

extern "C" int umb_embed_prep_fm(void* x, const void* table, int H, int T, const int* tok, const int* pos, const int* slot,
                                 const int* prefix, const int* tokens_all, const int* n_ptr, int off, const int* depth,
                                 int* pos_out, int* slot_out, int* prefix_out, void* hw, const void* norm_w, float* ssq,
                                 int ssq_stride, int hw_fm_tt, int dtype, hipStream_t st) {
  if (H % 64 || T < 1 || (hw && (!norm_w || !ssq || ssq_stride < H / 64))) return UMB_EINVAL;
  if (hw_fm_tt < 0 || (hw_fm_tt && (!hw || T > 16 * hw_fm_tt))) return UMB_EINVAL;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((embed_prep_kernel<P>), dim3(T), dim3(256), 0, st, (u16*)x, (const u16*)table, H, tok, pos, slot,
                       prefix, tokens_all, n_ptr, off, depth, pos_out, slot_out, prefix_out, (u16*)hw,
                       (const u16*)norm_w, ssq, ssq_stride, hw_fm_tt);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}
extern "C" int umb_embed_prep(void* x, const void* table, int H, int T, const int* tok, const int* pos, const int* slot,
                              const int* prefix, const int* tokens_all, const int* n_ptr, int off, const int* depth,
                              int* pos_out, int* slot_out, int* prefix_out, void* hw, const void* norm_w, float* ssq,
                              int ssq_stride, int dtype, hipStream_t st) {
  return umb_embed_prep_fm(x, table, H, T, tok, pos, slot, prefix, tokens_all, n_ptr, off, depth, pos_out, slot_out, prefix_out,
                           hw, norm_w, ssq, ssq_stride, 0, dtype, st);
}

extern "C" int umb_rope_inplace(void* q, void* k, const void* cosT, const void* sinT, const int* pos, int T, int Hq,
                                int Hkv, int D, int layout, int dtype, hipStream_t st) {
  if (T < 1 || D % 2 || Hq < 0 || Hkv < 0 || Hq + Hkv < 1 || (layout != 0 && layout != 1) || (Hq && !q) || (Hkv && !k))
    return UMB_EINVAL;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((rope_inplace_kernel<P>), dim3(T, Hq + Hkv), dim3(64), 0, st, (u16*)q, (u16*)k, (const u16*)cosT,
                       (const u16*)sinT, pos, T, Hq, Hkv, D, layout);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

extern "C" int umb_kv_append(void* k_cache, void* vt_cache, const void* k, const void* v, const int* slot, int T,
                             int Hkv, int D, int Lmax, int dtype, hipStream_t st) {
  if (T < 1 || Hkv < 1 || D < 32 || D % 32 || Lmax < 32 || Lmax % 32 || (dtype != UMB_F16 && dtype != UMB_BF16)) return UMB_EINVAL;
  hipLaunchKernelGGL(kv_append_kernel, dim3(T, Hkv), dim3(64), 0, st, (u16*)k_cache, (u16*)vt_cache, (const u16*)k,
                     (const u16*)v, slot, Hkv, D, Lmax);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// one layer slab, pinned host -> device, on the copy stream; event ordered against the compute stream: the slab may be
// overwritten only after ev_free (recorded by the caller behind the kernels that read it), and ev_copied marks arrival
extern "C" int umb_h2d_layer(void* dst, const void* src_pinned, size_t bytes, hipStream_t copy_stream, void* ev_free,
                             void* ev_copied) {
  if (!dst || !src_pinned || bytes == 0) return UMB_EINVAL;
  if (ev_free && hipStreamWaitEvent(copy_stream, (hipEvent_t)ev_free, 0) != hipSuccess) return UMB_EHIP;
  if (hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, copy_stream) != hipSuccess) return UMB_EHIP;
  if (ev_copied && hipEventRecord((hipEvent_t)ev_copied, copy_stream) != hipSuccess) return UMB_EHIP;
  return UMB_OK;
}

extern "C" int umb_sum_splits(float* partial, int S, int64_t n, hipStream_t st) {
  if (!partial || S < 1 || n < 0 || (n & 3)) return UMB_EINVAL;
  if (S == 1 || n == 0) return UMB_OK;
  const long n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
  hipLaunchKernelGGL(sum_splits_kernel, dim3(grid), dim3(256), 0, st, partial, S, n4);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}
