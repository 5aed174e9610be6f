// Row-streaming GEMV family for <= 4 token rows (the draft model's tree levels: 1-3 rows per forward).
//
// Replaces, for those forwards, the low-latency MFMA kernels of lowlat.hip on the same reference lines (llama.py:75-114:
// q/k/v projection + RoPE + KV append, o-projection + residual, gate/up + SiLU, down-projection + residual).  Why a third
// family: a launch this small is a latency chain, not a bandwidth problem.  Here
//  * every load of the kernel -- the wave's weight rows AND the activation pieces its lanes need -- is issued at kernel
//    start: no LDS staging, no barrier in front of the first FMA (probe: scripts/probe/gemv_probe2.hip);
//  * a wave owns whole rows of a plain row-major copy of the weights ([N][K], rows in the packed layouts' order: RoPE
//    partner pairs / (gate, up) pairs adjacent), so N = 2048 gives 256 workgroups of 8 rows where 16-row MFMA tiles give 128;
//  * v_dot2c accumulates in fp32, a 64-lane butterfly finishes a row; K = 8192 is cut in four 2048-element slices summed
//    through LDS in slice order (deterministic; a token's result does not depend on T).
// Activations are row-major [T][K] 16-bit; the residual stream's RMSNorm is split as in the low-latency family: producers
// write hw = h * w and per-workgroup sums of squares, consumers scale their outputs by rsqrt(mean + eps).
#include "../../include/umbrella_hip.h"
#include "common.h"
#include <type_traits>

typedef _Float16 gv_h2 __attribute__((ext_vector_type(2)));
typedef __bf16 gv_b2 __attribute__((ext_vector_type(2)));
template <typename P> __device__ __forceinline__ float gv_dot2(unsigned a, unsigned b, float c) {
  if constexpr (std::is_same<P, BF16>::value)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gv_b2, a), __builtin_bit_cast(gv_b2, b), c, false);
  else
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(gv_h2, a), __builtin_bit_cast(gv_h2, b), c, false);
}

enum { GV_SILU = 2, GV_QKV = 3, GV_RESID = 4 };

struct GvArgs {
  void* out;                                   // GV_SILU: act [T][N/2]
  const float* ssq_in; int ssq_groups, ssq_in_stride; float ssq_dim, eps;
  u16* h; u16* hw; const u16* norm_w; float* ssq_out; int ssq_out_stride;      // GV_RESID
  const int* pos; const int* slot; const u16* cosT; const u16* sinT; u16* q_out; u16* kc; u16* vt; const u16* bias;
  int Hq, Hkv, D, Lmax, rope_heads;
};

__device__ __forceinline__ int gv_rowmap_qkv(int n, int D, int rope_heads) {      // packed row -> HF feature (bias lookup)
  const int head = n / D, dp = n % D;
  if (head >= rope_heads) return n;
  return head * D + ((dp & 1) ? (dp >> 1) + D / 2 : (dp >> 1));
}

// RW rows per wave and pass, KS k-slices of 2048 elements per row (K = KS * 2048), 8 waves = 8 / KS row-group slots.
// Argument order: the leading scalars are preloaded into SGPRs (-amdgpu-kernarg-preload-count), the streams start at once.
template <typename P, int RW, int KS, int EPI>
__global__ __launch_bounds__(512) void gv_kernel(const u32x4* __restrict__ w, const u16* __restrict__ x, int T, int N, int K,
                                                 int RB, GvArgs a) {
  __shared__ float red[8][RW][4];              // k-slice partials
  __shared__ float sred[8][4];                 // sums of squares per wave (GV_RESID)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NSLOT = 8 / KS;
  const int ks = KS == 1 ? 0 : (wv & (KS - 1));
  const int slot_w = KS == 1 ? wv : (wv / KS);
  const int ngrp = RB / RW;
  const int kbase = ks * 2048;
  const int tl = lane & 3;                     // the token this lane finishes (lanes 0..3 do; the rest mirror them, unused)
  const bool fin = lane < 4 && lane < T;
  const int OOB = (int)0x80000000;
  // ---- every load of the kernel is an unconditional buffer load issued here, smallest first (loads return in order): the
  // producer's sums of squares, positions / slots, the activations, then the first pass's weights and epilogue operands.
  // Nothing waits behind a branch, so the compiler's vmcnt bookkeeping stays exact and the epilogue adds no memory round
  // trip of its own (a load behind a loop in front of the FMAs cost the q/k/v launch 4 us: 9.0 vs 4.9).
  float sqv[4][4];
  if constexpr (EPI != GV_RESID) {
    const auto rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ssq_in), 0,
                                                       a.ssq_in ? (unsigned)((T - 1) * a.ssq_in_stride + a.ssq_groups) * 4u : 0u, 0x00020000);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {              // <= 256 groups: 4 per lane; a group past the count reads as zero
        const int gq = lane + 64 * q;
        sqv[t][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            rsq, (t < T && gq < a.ssq_groups) ? (t * a.ssq_in_stride + gq) * 4 : OOB, 0, 0));
      }
  }
  int pos_t = 0, slot_t = 0;
  if constexpr (EPI == GV_QKV) {
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.pos), 0, (unsigned)T * 4u, 0x00020000);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.slot), 0, (unsigned)T * 4u, 0x00020000);
    pos_t = (int)__builtin_amdgcn_raw_buffer_load_b32(rp, fin ? tl * 4 : OOB, 0, 0);
    slot_t = (int)__builtin_amdgcn_raw_buffer_load_b32(rs, fin ? tl * 4 : OOB, 0, 0);
  }
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(w), 0, 0xffffffffu, 0x00020000);
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(x), 0, (unsigned)(T * K * 2), 0x00020000);
  u32x4 xr[4][4], wr[2][RW][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)               // rows >= T are out of range of the descriptor: zeros, no fetch
      xr[t][kc] = __builtin_amdgcn_raw_buffer_load_b128(rx, (t * K + kbase + kc * 512 + lane * 8) * 2, 0, 0);
  // epilogue operands of a pass (lane = token): residual + next norm weight (GV_RESID), cos / sin pairs and bias (GV_QKV)
  struct Epi { unsigned short hv[RW], nw[RW], cl[RW / 2 + 1], ch[RW / 2 + 1], sl[RW / 2 + 1], sh[RW / 2 + 1], bs[RW]; };
  Epi er[2];
  const auto rh = __builtin_amdgcn_make_buffer_rsrc(a.h, 0, EPI == GV_RESID ? (unsigned)(T * N * 2) : 0u, 0x00020000);
  const auto rnw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.norm_w), 0,
                                                     (EPI == GV_RESID && a.norm_w) ? (unsigned)N * 2u : 0u, 0x00020000);
  const auto rcos = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.cosT), 0, EPI == GV_QKV ? 0x7fffffffu : 0u, 0x00020000);
  const auto rsin = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.sinT), 0, EPI == GV_QKV ? 0x7fffffffu : 0u, 0x00020000);
  const auto rbs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.bias), 0,
                                                     (EPI == GV_QKV && a.bias) ? (unsigned)N * 2u : 0u, 0x00020000);
  auto issue = [&](auto bc, int g) {
    constexpr int B = decltype(bc)::value;
    const int row0 = blockIdx.x * RB + g * RW;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        wr[B][r][kc] = __builtin_amdgcn_raw_buffer_load_b128(rw, (kbase + kc * 512 + lane * 8) * 2, (row0 + r) * K * 2, 2);
    if constexpr (EPI == GV_RESID) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        er[B].hv[r] = __builtin_amdgcn_raw_buffer_load_b16(rh, fin ? (tl * N + row0 + r) * 2 : OOB, 0, 0);
        er[B].nw[r] = __builtin_amdgcn_raw_buffer_load_b16(rnw, (row0 + r) * 2, 0, 0);
      }
    }
    if constexpr (EPI == GV_QKV) {
      const int D = a.D, half = D / 2;
#pragma unroll
      for (int r = 0; r < RW; r += 2) {
        const int n = row0 + r, m = (n % D) >> 1;
        const int cb = fin ? (pos_t * D + m) * 2 : OOB;            // waits for pos only: it was requested first
        er[B].cl[r / 2] = __builtin_amdgcn_raw_buffer_load_b16(rcos, cb, 0, 0);
        er[B].ch[r / 2] = __builtin_amdgcn_raw_buffer_load_b16(rcos, fin ? cb + half * 2 : OOB, 0, 0);
        er[B].sl[r / 2] = __builtin_amdgcn_raw_buffer_load_b16(rsin, cb, 0, 0);
        er[B].sh[r / 2] = __builtin_amdgcn_raw_buffer_load_b16(rsin, fin ? cb + half * 2 : OOB, 0, 0);
        er[B].bs[r] = __builtin_amdgcn_raw_buffer_load_b16(rbs, gv_rowmap_qkv(n, D, a.rope_heads) * 2, 0, 0);
        er[B].bs[r + 1] = __builtin_amdgcn_raw_buffer_load_b16(rbs, gv_rowmap_qkv(n + 1, D, a.rope_heads) * 2, 0, 0);
      }
    }
  };
  int g = slot_w;
  if (g < ngrp) issue(std::integral_constant<int, 0>{}, g);
  // 1/rms per token from the producer's partial sums of squares (fixed order: lane-strided, then a 64-lane butterfly)
  float inv[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (EPI != GV_RESID) {
    if (a.ssq_in) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float s = ((sqv[t][0] + sqv[t][1]) + sqv[t][2]) + sqv[t][3];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        inv[t] = rsqrtf(s / a.ssq_dim + a.eps);
      }
    }
  }
  float sq_acc = 0.f;                          // GV_RESID: this lane's (token = lane) share of the block's sums of squares

  auto finish = [&](auto bc, int gg) {
    constexpr int B = decltype(bc)::value;
    float acc[RW][4];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
          for (int e = 0; e < 4; ++e) v = gv_dot2<P>(wr[B][r][kc][e], xr[t][kc][e], v);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        acc[r][t] = v;
      }
    if constexpr (KS > 1) {
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int t = 0; t < 4; ++t) red[wv][r][t] = acc[r][t];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float v = red[wv][r][t];
            for (int s = 1; s < KS; ++s) v += red[wv + s][r][t];          // slice order
            acc[r][t] = v;
          }
      }
      __syncthreads();                          // red is reused by the next pass
      if (ks != 0) return;
    }
    // lane t (< T) finishes token t of the wave's RW rows
    if (!fin) return;
    const int t = tl;
    float val[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      float v = acc[r][0];
#pragma unroll
      for (int tt = 1; tt < 4; ++tt) v = (tt == t) ? acc[r][tt] : v;
      val[r] = v;
    }
    const float iv = (t == 0) ? inv[0] : (t == 1) ? inv[1] : (t == 2) ? inv[2] : inv[3];
    const int n0 = blockIdx.x * RB + gg * RW;    // first (packed-order) output row of this pass
    if constexpr (EPI == GV_RESID) {
      // h <- round(round(gemm) + h) ; hw <- h * w_next ; sums of squares (llama.py:104,112 + the next norm's weight)
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const long off = (long)t * N + n0 + r;
        const float h0 = rnd<P>(rnd<P>(val[r]) + P::to_f(er[B].hv[r]));
        a.h[off] = P::from_f(h0);
        if (a.norm_w) a.hw[off] = P::from_f(h0 * P::to_f(er[B].nw[r]));
        sq_acc = __builtin_fmaf(h0, h0, sq_acc);      // pinned (chain.hip reproduces this order bit for bit)
      }
    } else if constexpr (EPI == GV_SILU) {
      // rows (2m, 2m+1) = (gate_m, up_m); every step rounded to the model dtype as eager torch does (llama.py:107-110)
      u16* act = reinterpret_cast<u16*>(a.out);
#pragma unroll
      for (int r = 0; r < RW; r += 2) {
        const float g0 = rnd_prod<P>(val[r], iv), u0 = rnd_prod<P>(val[r + 1], iv);
        act[(long)t * (N / 2) + ((n0 + r) >> 1)] = P::from_f(rnd<P>(g0 / (1.f + __expf(-g0))) * u0);
      }
    } else {
      // GV_QKV: rows (2m, 2m+1) of a q / k head are RoPE partners (m, m + D/2) (model_utils.py:17-52, cache.py:53-65)
      const int D = a.D, half = D / 2;
#pragma unroll
      for (int r = 0; r < RW; r += 2) {
        const int n = n0 + r;
        const int head = n / D, dp = n % D, m = dp >> 1;
        float va = val[r] * iv, vb = val[r + 1] * iv;
        asm volatile("" : "+v"(va), "+v"(vb));       // fp32 products of their own (common.h: rnd_prod)
        if (a.bias) { va += P::to_f(er[B].bs[r]); vb += P::to_f(er[B].bs[r + 1]); }
        const float a0 = rnd<P>(va), b0 = rnd<P>(vb);
        if (head < a.Hq + a.Hkv) {
          const float cl = P::to_f(er[B].cl[r / 2]), ch = P::to_f(er[B].ch[r / 2]);
          const float sl_ = P::to_f(er[B].sl[r / 2]), sh = P::to_f(er[B].sh[r / 2]);
          const float lo0 = rnd<P>(mul_rnd<P>(a0, cl) + mul_rnd<P>(-b0, sl_));
          const float hi0 = rnd<P>(mul_rnd<P>(b0, ch) + mul_rnd<P>(a0, sh));
          // slot_t is trusted here (the model runtime derives it from the engine's guarded state: HipEngine.step refuses
          // a tree that would pass max_length); a per-store range check costs this kernel 24 B of scratch per lane
          // (tests/test_abi.py guards that).  The stand-alone umb_kv_append is the range-checked form.
          // q rows are row-major; the K / V^T caches are in fragment order inside a head's slab (common.h)
          if (head < a.Hq) {
            u16* dst = a.q_out + ((long)t * a.Hq + head) * D;
            dst[m] = P::from_f(lo0);
            dst[m + half] = P::from_f(hi0);
          } else {
            u16* dst = a.kc + (long)(head - a.Hq) * a.Lmax * D;
            dst[kc_off(slot_t, m, D)] = P::from_f(lo0);
            dst[kc_off(slot_t, m + half, D)] = P::from_f(hi0);
          }
        } else {
          u16* dst = a.vt + (long)(head - a.Hq - a.Hkv) * D * VT_LD(a.Lmax);
          dst[vt_off(dp, slot_t, D)] = P::from_f(a0);
          dst[vt_off(dp + 1, slot_t, D)] = P::from_f(b0);
        }
      }
    }
  };

  // Passes over this wave's row groups, the next group's loads in flight while the current one is finished.  With KS > 1
  // the slice waves of a row group run the same trip count (they share slot_w), so the barriers inside finish() match; a
  // slice wave that is not slice 0 leaves finish() early but still takes part in the barriers of later passes.
  for (; g < ngrp; g += 2 * NSLOT) {
    if (g + NSLOT < ngrp) issue(std::integral_constant<int, 1>{}, g + NSLOT);
    finish(std::integral_constant<int, 0>{}, g);
    if (g + NSLOT < ngrp) {
      if (g + 2 * NSLOT < ngrp) issue(std::integral_constant<int, 0>{}, g + 2 * NSLOT);
      finish(std::integral_constant<int, 1>{}, g + NSLOT);
    }
  }
  if constexpr (EPI == GV_RESID) {
    // the block's sum of squares per token: waves in slot order (fixed order: deterministic)
    if (lane < 4) sred[wv][lane] = sq_acc;
    __syncthreads();
    if (wv == 0 && lane < T && lane < 4 && a.ssq_out) {
      float s = sred[0][lane];
      for (int q = 1; q < 8; ++q) s += sred[q][lane];
      a.ssq_out[(long)lane * a.ssq_out_stride + blockIdx.x] = s;
    }
  }
}

// rows of the packed order (gemm.hip rowmap(): mode 1 (gate, up) pairs, mode 2 RoPE partner pairs), plain row-major
template <int dummy>
__global__ void repack_rows_kernel(const u32x4* __restrict__ W, u32x4* __restrict__ out, int N, int K8, int mode, int D,
                                   int rope_heads) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)N * K8) return;
  const int n = (int)(gid / K8), c = (int)(gid % K8);
  int src = n;
  if (mode == 1) src = (n & 1) ? N / 2 + (n >> 1) : (n >> 1);
  if (mode == 2) {
    const int head = n / D, dp = n % D;
    if (head < rope_heads) src = head * D + ((dp & 1) ? (dp >> 1) + D / 2 : (dp >> 1));
  }
  out[gid] = W[(long)src * K8 + c];
}

extern "C" int umb_repack_rows(void* out, const void* w, int N, int K, int mode, int D, int rope_heads, hipStream_t st) {
  if (N < 1 || K % 8 || mode < 0 || mode > 2 || (mode == 2 && (D < 2 || D % 2))) return UMB_EINVAL;
  const long total = (long)N * (K / 8);
  hipLaunchKernelGGL((repack_rows_kernel<0>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const u32x4*)w,
                     (u32x4*)out, N, K / 8, mode, D, rope_heads);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// rows per workgroup: whole passes of the 8 / KS row-group slots, as close to 256 workgroups as the shape allows
static int gv_rows_per_block(int N, int RW, int KS) {
  const int step = KS > 1 ? RW * (8 / KS) : RW;        // K-sliced row groups share barriers: whole passes only
  int best = 0;
  for (int rb = step; rb <= N; rb += step) {
    if (N % rb) continue;
    const int blocks = N / rb;
    if (blocks < 256 && best) break;
    best = rb;
    if (blocks <= 256) break;
  }
  return best;
}

// rows per wave and pass: K = 8192 runs as four k-slices of 4 rows; K = 2048 one row per pass for the residual epilogue,
// a (partner) pair otherwise
static int gv_rw(int T, int K, int epi) { (void)T; return K == 8192 ? 4 : (epi == GV_RESID ? 1 : 2); }

extern "C" int umb_gemv_ok(int T, int N, int K, int epi) {
  static const bool off = getenv("UMB_NO_GEMV") != nullptr;
  // <= 4 rows.  An 8-slot instantiation was built and measured: correct, but 1.30 vs 0.78 ms per 1B forward at T = 5 -- the
  // 64-lane butterflies (rows x token slots values per pass) and 32 activation loads per lane outweigh the MFMA kernels.
  // (Its generic form also showed a trap: per-token arrays filled in loops were promoted to LDS by hipcc, 8 KiB per block,
  // and the q/k/v launch went from 6.2 to 19.4 us -- the kernel below keeps its four token slots in named registers.)
  if (off || T < 1 || T > 4 || (K != 2048 && K != 8192)) return 0;
  if (epi != GV_SILU && epi != GV_QKV && epi != GV_RESID) return 0;
  if (K == 8192 && epi != GV_RESID) return 0;
  const int RW = gv_rw(T, K, epi);
  return N % RW == 0 && gv_rows_per_block(N, RW, K / 2048) > 0;
}

extern "C" int umb_gemv(void* out, const void* x, const void* w_rows, int T, int N, int K, int epi, const UmbGemmLL* fx,
                        int dtype, hipStream_t st) {
  if (!fx || !x || !w_rows || !umb_gemv_ok(T, N, K, epi)) return UMB_EINVAL;
  GvArgs a = {};
  a.out = out;
  a.ssq_in = fx->ssq_in; a.ssq_groups = fx->ssq_groups; a.ssq_in_stride = fx->ssq_in_stride ? fx->ssq_in_stride : fx->ssq_groups;
  a.ssq_dim = fx->ssq_dim; a.eps = fx->eps;
  a.h = (u16*)fx->h; a.hw = (u16*)fx->hw; a.norm_w = (const u16*)fx->norm_w; a.ssq_out = fx->ssq_out;
  a.ssq_out_stride = fx->ssq_out_stride;
  a.pos = fx->pos; a.slot = fx->slot; a.cosT = (const u16*)fx->cosT; a.sinT = (const u16*)fx->sinT; a.q_out = (u16*)fx->q_out;
  a.kc = (u16*)fx->k_cache; a.vt = (u16*)fx->vt_cache; a.bias = (const u16*)fx->bias;
  a.Hq = fx->Hq; a.Hkv = fx->Hkv; a.D = fx->D; a.Lmax = fx->Lmax; a.rope_heads = fx->Hq + fx->Hkv;
  if (epi == GV_RESID && (!a.h || (a.norm_w && !a.hw))) return UMB_EINVAL;
  if (epi == GV_SILU && (!out || N % 2)) return UMB_EINVAL;
  if (epi == GV_QKV && (!a.pos || !a.slot || !a.cosT || !a.sinT || !a.q_out || !a.kc || !a.vt || a.D % 2 || N % 2)) return UMB_EINVAL;
  if (epi == GV_QKV && (a.D % 32 || a.Lmax % 32 || a.Lmax < 32)) return UMB_EINVAL;   // fragment-ordered caches (common.h): whole 32-key / 32-feature tiles
  // the kernel reads the producer's sums of squares as <= 4 values per lane: 256 groups at most
  if (a.ssq_in && (a.ssq_groups < 1 || a.ssq_groups > 256 || a.ssq_dim <= 0.f)) return UMB_EINVAL;
  const int KS = K / 2048;
  const int RW = gv_rw(T, K, epi);
  const int RB = gv_rows_per_block(N, RW, KS);
  if (epi == GV_RESID && a.ssq_out && a.ssq_out_stride < N / RB) return UMB_EINVAL;
#define GV_GO(RWV, KSV, EPIV)                                                                                     \
  hipLaunchKernelGGL((gv_kernel<P, RWV, KSV, EPIV>), dim3((unsigned)(N / RB)), dim3(512), 0, st, (const u32x4*)w_rows,  \
                     (const u16*)x, T, N, K, RB, a)
  DISPATCH_DTYPE(dtype, {
    if (T <= 4) {
      if (K == 2048) {
        if (epi == GV_RESID) GV_GO(1, 1, GV_RESID);
        else if (epi == GV_SILU) GV_GO(2, 1, GV_SILU);
        else GV_GO(2, 1, GV_QKV);
      } else {
        GV_GO(4, 4, GV_RESID);
      }
    } else {
      return UMB_EINVAL;
    }
  })
#undef GV_GO
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// number of sums-of-squares groups a GV_RESID launch of this shape and row count writes per token (= its workgroups)
extern "C" int umb_gemv_groups(int T, int N, int K) {
  const int RB = gv_rows_per_block(N, gv_rw(T, K, GV_RESID), K / 2048);
  return RB > 0 ? N / RB : 0;
}
