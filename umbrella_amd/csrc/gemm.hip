// Skinny GEMM  out[t][n] = sum_k x[t][k] * W[n][k]   for T = 1..64 tokens per launch.
//
// Replaces on the hot path (reference call sites):
//   F.linear(hidden, w)                       umbrella/models/llama.py:89-91,103,107-111,133
//   AwqLinear.apply -> awq_ext.gemm_forward_cuda  umbrella/quantization/awq_utils.py:63-86
//
// Regime: at T <= 64 every weight byte is read once and used for <= 64 MACs per
// element -> HBM-bandwidth bound.  Design for gfx950:
//   * weights are re-packed once at load into MFMA-fragment tile order, so one
//     wave-instruction (64 lanes x 16 B) streams one fully coalesced 1 KiB tile
//     that is directly the A operand of v_mfma_f32_16x16x32_{bf16,f16}
//       dense : tile = 16 output rows x 32 k      (1 MFMA per tile)
//       AWQ   : tile = 16 output rows x 128 k int4 (one AWQ group; 4 MFMAs per tile)
//   * weight rows are the MFMA M dimension, tokens the N dimension (<=16 per
//     fragment): the matrix pipe does the MACs, VALU only unpacks nibbles.
//   * AWQ: W = (q - z) * s is folded out of the MFMA.  The MFMA runs on the raw
//     codes (q | magic -> bf16(128+q) / fp16(1024+q), 2 VALU ops per 2 codes);
//     per 128-k group out += s * (acc - (z + magic) * sum_k x[k]) where sum_k x
//     comes from one extra MFMA against a constant all-ones fragment.
//   * activations (<= 64 x K, L2 resident) are read straight into B fragments.
//   * split-K writes fp32 partials [S][T][N]; the consumer kernel (epilogue.hip)
//     reduces them in a fixed order -> results are bit-identical for any T
//     (batch-invariant), which is what makes greedy spec == greedy AR exact.
#include "common.h"

// ------------------------------------------------------------------ repack (load time)
template <typename P>
__global__ void repack_dense_kernel(const u16* __restrict__ W, u32x4* __restrict__ out, int N, int K) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int KT = K / 32;
  const long total = (long)(N / 16) * KT * 64;
  if (gid >= total) return;
  const int lane = gid & 63;
  const long tile = gid >> 6;
  const int kt = tile % KT, nt = tile / KT;
  const int i = lane & 15, g = lane >> 4;
  out[gid] = *reinterpret_cast<const u32x4*>(W + (long)(nt * 16 + i) * K + kt * 32 + g * 8);
}

// AutoAWQ GEMM format -> tile order.  qweight [K][N/8] int32, nibble idx of word c
// holds column 8c + ORDER[idx], ORDER = {0,2,4,6,1,3,5,7}  (inverse: INV below).
__global__ void repack_awq_kernel(const unsigned* __restrict__ qweight, const unsigned* __restrict__ qzeros,
                                  const u16* __restrict__ scales, u32x4* __restrict__ outw,
                                  unsigned char* __restrict__ meta, int N, int K) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int KG = K / 128;
  const long total = (long)(N / 16) * KG * 64;
  if (gid >= total) return;
  const int lane = gid & 63;
  const long tile = gid >> 6;
  const int kg = tile % KG, nt = tile / KG;
  const int i = lane & 15, g = lane >> 4;
  const int n = nt * 16 + i;
  const int INV[8] = {0, 4, 1, 5, 2, 6, 3, 7};
  const int sh = 4 * INV[n & 7];
  const int NW = N / 8;
  unsigned w[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    unsigned d = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kg * 128 + s * 32 + g * 8 + e;
      const unsigned q = (qweight[(long)k * NW + (n >> 3)] >> sh) & 0xFu;
      d |= q << ((e & 1) ? (16 + 4 * (e >> 1)) : (4 * (e >> 1)));
    }
    w[s] = d;
  }
  u32x4 v = {w[0], w[1], w[2], w[3]};
  outw[gid] = v;
  if (g == 0) {   // per (tile, column): fp16 scale + u8 zero  -> meta[tile][16 x fp16 | 16 x u8]
    unsigned char* m = meta + tile * 48;
    reinterpret_cast<u16*>(m)[i] = scales[(long)kg * N + n];
    m[32 + i] = (unsigned char)((qzeros[(long)kg * NW + (n >> 3)] >> sh) & 0xFu);
  }
}

// ------------------------------------------------------------------ main kernel
template <typename P, int AWQ, int TT, int R> struct Stage {
  u32x4 a[R][AWQ ? 1 : 4];
  u32x4 b[TT][4];
  uint2 sc[AWQ ? R : 1];
  unsigned zz[AWQ ? R : 1];
};

template <typename P, int AWQ, int TT, int R>
__device__ __forceinline__ void stage_load(Stage<P, AWQ, TT, R>& st, const u32x4* __restrict__ wp,
                                           const unsigned char* __restrict__ meta, const u16* __restrict__ x,
                                           int ldx, int T, int nt0, int KU, int kb, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (AWQ) {
      const long tile = (long)(nt0 + r) * KU + kb;
      st.a[r][0] = __builtin_nontemporal_load(wp + tile * 64 + lane);
      const unsigned char* m = meta + tile * 48;
      st.sc[r] = *reinterpret_cast<const uint2*>(m + g * 8);
      st.zz[r] = *reinterpret_cast<const unsigned*>(m + 32 + g * 4);
    } else {
      const u32x4* p = wp + ((long)(nt0 + r) * KU + kb) * 256 + lane;   // 4 tiles (128 k) contiguous
#pragma unroll
      for (int s = 0; s < 4; ++s) st.a[r][s] = __builtin_nontemporal_load(p + s * 64);
    }
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int tok = tt * 16 + j;
    const u16* xp = x + (long)tok * ldx + kb * 128 + g * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 z = {0u, 0u, 0u, 0u};
      st.b[tt][s] = (tok < T) ? *reinterpret_cast<const u32x4*>(xp + s * 32) : z;
    }
  }
}

template <typename P, int AWQ, int TT, int R>
__device__ __forceinline__ void stage_compute(const Stage<P, AWQ, TT, R>& st, f32x4 (&acc)[R][TT]) {
  if (AWQ) {
    const u32x4 ones = {P::ONE2, P::ONE2, P::ONE2, P::ONE2};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 xs[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      xs[tt] = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) xs[tt] = P::mfma(ones, st.b[tt][s], xs[tt]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      f32x4 ga[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) ga[tt] = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned w = st.a[r][0][s];
        u32x4 f;
        f[0] = (w & 0x000F000Fu) | P::MAGIC;
        f[1] = ((w >> 4) & 0x000F000Fu) | P::MAGIC;
        f[2] = ((w >> 8) & 0x000F000Fu) | P::MAGIC;
        f[3] = ((w >> 12) & 0x000F000Fu) | P::MAGIC;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) ga[tt] = P::mfma(f, st.b[tt][s], ga[tt]);
      }
      const float s0 = F16::to_f((u16)(st.sc[r].x & 0xffffu)), s1 = F16::to_f((u16)(st.sc[r].x >> 16));
      const float s2 = F16::to_f((u16)(st.sc[r].y & 0xffffu)), s3 = F16::to_f((u16)(st.sc[r].y >> 16));
      const unsigned z = st.zz[r];
      const float z0 = (float)(z & 0xffu) + P::MAGIC_OFF, z1 = (float)((z >> 8) & 0xffu) + P::MAGIC_OFF;
      const float z2 = (float)((z >> 16) & 0xffu) + P::MAGIC_OFF, z3 = (float)(z >> 24) + P::MAGIC_OFF;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float sx = xs[tt][0];
        acc[r][tt][0] += s0 * (ga[tt][0] - z0 * sx);
        acc[r][tt][1] += s1 * (ga[tt][1] - z1 * sx);
        acc[r][tt][2] += s2 * (ga[tt][2] - z2 * sx);
        acc[r][tt][3] += s3 * (ga[tt][3] - z3 * sx);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[r][tt] = P::mfma(st.a[r][s], st.b[tt][s], acc[r][tt]);
  }
}

template <typename P, int AWQ, int TT, int R>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const u32x4* __restrict__ wp,
                                                          const unsigned char* __restrict__ meta,
                                                          const u16* __restrict__ x, int ldx,
                                                          float* __restrict__ out, int T, int Ttot, int N, int K,
                                                          int S, int round_out) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NT = N / 16;
  const int ngroups = NT / R;
  const long task = (long)blockIdx.x * 4 + wv;
  if (task >= (long)ngroups * S) return;
  const int sp = (int)(task / ngroups);
  const int nt0 = (int)(task % ngroups) * R;
  const int KB = K / 128;                                  // 128-k blocks
  const int KU = AWQ ? KB : KB;                            // tile stride unit is the 128-k block in both formats
  const int per = (KB + S - 1) / S;
  const int kb0 = sp * per;
  const int kb1 = min(KB, kb0 + per);

  f32x4 acc[R][TT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[r][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stage<P, AWQ, TT, R> s0, s1;
  if (kb0 < kb1) stage_load<P, AWQ, TT, R>(s0, wp, meta, x, ldx, T, nt0, KU, kb0, lane);
  int kb = kb0;
  for (; kb + 1 < kb1; kb += 2) {
    stage_load<P, AWQ, TT, R>(s1, wp, meta, x, ldx, T, nt0, KU, kb + 1, lane);
    stage_compute<P, AWQ, TT, R>(s0, acc);
    if (kb + 2 < kb1) stage_load<P, AWQ, TT, R>(s0, wp, meta, x, ldx, T, nt0, KU, kb + 2, lane);
    stage_compute<P, AWQ, TT, R>(s1, acc);
  }
  if (kb < kb1) stage_compute<P, AWQ, TT, R>(s0, acc);

  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int tok = tt * 16 + j;
    if (tok < T) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f32x4 v = acc[r][tt];
        if (round_out) { v[0] = rnd<P>(v[0]); v[1] = rnd<P>(v[1]); v[2] = rnd<P>(v[2]); v[3] = rnd<P>(v[3]); }
        *reinterpret_cast<f32x4*>(out + ((long)sp * Ttot + tok) * N + (nt0 + r) * 16 + g * 4) = v;
      }
    }
  }
}

// ------------------------------------------------------------------ host side
// (R, S) depend on (N, K, format) only -- never on T -- so a token's result is
// independent of how many other tokens share the launch.
extern "C" void umb_gemm_plan(int N, int K, int awq, int force_s1, int* R_out, int* S_out) {
  const int NT = N / 16, KB = K / 128;
  int R = 1;
  const int maxR = awq ? 4 : 2;
  for (int c = maxR; c > 1; c >>= 1)
    if (NT % c == 0 && NT / c >= 1024) { R = c; break; }
  int S = 1;
  if (!force_s1) {
    const int groups = NT / R;
    S = (2048 + groups - 1) / groups;
    const int min_blocks = awq ? 4 : 2;                     // keep >= 512 / 256 k per slab
    if (S > KB / min_blocks) S = KB / min_blocks;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
  }
  *R_out = R;
  *S_out = S;
}

template <typename P, int AWQ, int TT>
static int launch_r(int R, const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int Ttot,
                    int N, int K, int S, int round_out, hipStream_t st) {
  const long tasks = (long)(N / 16 / R) * S;
  const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
#define L_(RR)                                                                                          \
  hipLaunchKernelGGL((skinny_gemm_kernel<P, AWQ, TT, RR>), grid, block, 0, st, (const u32x4*)wp,         \
                     (const unsigned char*)meta, x, ldx, out, T, Ttot, N, K, S, round_out)
  if (R == 1) L_(1);
  else if (R == 2) L_(2);
  else if (R == 4) {
    if constexpr (AWQ) L_(4);
    else return UMB_EINVAL;
  } else return UMB_EINVAL;
#undef L_
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

template <typename P, int AWQ>
static int launch_tt(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int N, int K, int R,
                     int S, int round_out, hipStream_t st) {
  // tokens beyond 64 go through further launches (weights re-read from L2/HBM)
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int tn = min(64, T - t0);
    const u16* xx = x + (long)t0 * ldx;
    float* oo = out + (long)t0 * N;            // out is [S][T][N] over the full T; split stride stays T
    int rc;
    if (tn <= 16) rc = launch_r<P, AWQ, 1>(R, wp, meta, xx, ldx, oo, tn, T, N, K, S, round_out, st);
    else if (tn <= 32) rc = launch_r<P, AWQ, 2>(R, wp, meta, xx, ldx, oo, tn, T, N, K, S, round_out, st);
    else rc = launch_r<P, AWQ, 4>(R, wp, meta, xx, ldx, oo, tn, T, N, K, S, round_out, st);
    if (rc) return rc;
  }
  return UMB_OK;
}

// out: fp32 [S][T][N] partials (S from umb_gemm_plan, or 1 when force_s1)
extern "C" int umb_gemm(void* out, const void* x, int ldx, const void* wpacked, const void* meta, int T, int N, int K,
                        int awq, int S, int R, int round_out, int dtype, hipStream_t st) {
  if (N % 16 || K % 128 || T < 1 || S < 1) return UMB_EINVAL;
  DISPATCH_DTYPE(dtype, {
    if (awq) return launch_tt<P, 1>(wpacked, meta, (const u16*)x, ldx, (float*)out, T, N, K, R, S, round_out, st);
    return launch_tt<P, 0>(wpacked, meta, (const u16*)x, ldx, (float*)out, T, N, K, R, S, round_out, st);
  })
}

extern "C" int umb_repack_dense(void* out, const void* w, int N, int K, int dtype, hipStream_t st) {
  if (N % 16 || K % 32) return UMB_EINVAL;
  const long total = (long)(N / 16) * (K / 32) * 64;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((repack_dense_kernel<P>), grid, block, 0, st, (const u16*)w, (u32x4*)out, N, K);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// outw: N*K/2 bytes; meta: (N/16)*(K/128)*48 bytes
extern "C" int umb_awq_repack(void* outw, void* meta, const void* qweight, const void* qzeros, const void* scales,
                              int N, int K, int group, hipStream_t st) {
  if (N % 16 || K % 128 || group != 128) return UMB_EINVAL;
  const long total = (long)(N / 16) * (K / 128) * 64;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipLaunchKernelGGL(repack_awq_kernel, grid, block, 0, st, (const unsigned*)qweight, (const unsigned*)qzeros,
                     (const u16*)scales, (u32x4*)outw, (unsigned char*)meta, N, K);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}
