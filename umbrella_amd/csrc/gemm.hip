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
#include <type_traits>

// ------------------------------------------------------------------ repack (load time)
template <typename P>
__global__ void repack_dense_kernel(const u16* __restrict__ W, u32x4* __restrict__ out, int N, int K, int il) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int KT = K / 32;
  const long total = (long)(N / 16) * KT * 64;
  if (gid >= total) return;
  const int lane = gid & 63;
  const long tile = gid >> 6;
  const int kt = tile % KT, nt = tile / KT;
  const int i = lane & 15, g = lane >> 4;
  const int n = nt * 16 + i;
  const int src = il ? ((n & 1) ? N / 2 + (n >> 1) : (n >> 1)) : n;   // il: rows (2m, 2m+1) <- (m, N/2 + m)
  out[gid] = *reinterpret_cast<const u32x4*>(W + (long)src * K + kt * 32 + g * 8);
}

// AutoAWQ GEMM format -> tile order.  qweight [K][N/8] int32, nibble idx of word c
// holds column 8c + ORDER[idx], ORDER = {0,2,4,6,1,3,5,7}  (inverse: INV below).
__global__ void repack_awq_kernel(const unsigned* __restrict__ qweight, const unsigned* __restrict__ qzeros,
                                  const u16* __restrict__ scales, u32x4* __restrict__ outw,
                                  unsigned char* __restrict__ meta, int N, int K, int il) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int KG = K / 128;
  const long total = (long)(N / 16) * KG * 64;
  if (gid >= total) return;
  const int lane = gid & 63;
  const long tile = gid >> 6;
  const int kg = tile % KG, nt = tile / KG;
  const int i = lane & 15, g = lane >> 4;
  const int nlog = nt * 16 + i;
  const int n = il ? ((nlog & 1) ? N / 2 + (nlog >> 1) : (nlog >> 1)) : nlog;   // source column
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
  // tile order [N/64][K/128][4]: the four n-tiles a block works on are adjacent, so each 128-k step of a
  // block is one contiguous 4 KiB HBM burst (1 KiB bursts per wave measured ~25 % slower)
  const long otile = ((long)(nt >> 2) * KG + kg) * 4 + (nt & 3);
  outw[otile * 64 + lane] = v;
  if (g == 0) {   // per (tile, column): {fp16 scale, fp16(zero)} pairs -> meta[tile][16][2] (64 B)
    u16* m = reinterpret_cast<u16*>(meta + otile * 64);
    m[2 * i] = scales[(long)kg * N + n];
    m[2 * i + 1] = F16::from_f((float)((qzeros[(long)kg * NW + (n >> 3)] >> sh) & 0xFu));
  }
}

// ------------------------------------------------------------------ main kernel
// Block = 4 waves on the same K-slab; wave w owns n-tiles [(4*nb + w)*R, +R).  The activation
// fragments of a CB x 128-k chunk are staged ONCE per block in LDS in fragment (lane-linear) order
// -> conflict-free ds_read_b128, 4x less L2->L1 traffic than per-wave loads (the int4 path reads
// 4 KiB of activations per 1 KiB weight tile, so unshared B loads were the bottleneck).
// Weight tiles go straight from HBM to VGPRs (non-temporal), double buffered across 128-k blocks.
// AWQ modes: 0 dense 16-bit weights; 1 int4, dequant folded out of the MFMA (any activation dtype);
// 2 int4 with exact fp16 dequant in registers, W = fp16((q - z) * s) -- bit-identical to what
// awq_ext.dequantize_weights_cuda produces -- via v_pk_add_f16 / v_pk_mul_f16 (fp16 activations only).
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <typename P, int AWQ, int R> struct Stage {
  u32x4 a[R][AWQ ? 1 : 4];
  u32x4 m4[AWQ == 1 ? R : 1];     // folded path: 4 x {scale, zero} for output rows g*4 .. g*4+3
  unsigned m1[AWQ == 2 ? R : 1];  // exact path : {scale, zero} of this lane's weight row
};

template <typename P, int AWQ, int R>
__device__ __forceinline__ void stage_load(Stage<P, AWQ, R>& st, const u32x4* __restrict__ wp,
                                           const unsigned char* __restrict__ meta, int nt0, int KB, int kb,
                                           int lane) {
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (AWQ) {
      const int nt = nt0 + r;
      const long tile = ((long)(nt >> 2) * KB + kb) * 4 + (nt & 3);
      st.a[r][0] = __builtin_nontemporal_load(wp + tile * 64 + lane);
      const unsigned char* m = meta + tile * 64;
      if (AWQ == 1) st.m4[r] = *reinterpret_cast<const u32x4*>(m + g * 16);
      else st.m1[r] = *reinterpret_cast<const unsigned*>(m + i * 4);
    } else {
      const u32x4* p = wp + ((long)(nt0 + r) * KB + kb) * 256 + lane;   // 4 tiles (128 k) contiguous
#pragma unroll
      for (int s = 0; s < 4; ++s) st.a[r][s] = __builtin_nontemporal_load(p + s * 64);
    }
  }
}

// xf: LDS fragments of this 128-k block, index (tt*4 + s)*64 + lane
template <typename P, int AWQ, int TT, int R>
__device__ __forceinline__ void stage_compute(const Stage<P, AWQ, R>& st, const u32x4* xf, int lane,
                                              f32x4 (&acc)[R][TT]) {
  u32x4 b[TT][4];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int s = 0; s < 4; ++s) b[tt][s] = xf[(tt * 4 + s) * 64 + lane];
  if (AWQ == 2) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned mm = st.m1[r];
      const _Float16 sc = __builtin_bit_cast(_Float16, (u16)(mm & 0xffffu));
      const _Float16 zf = __builtin_bit_cast(_Float16, (u16)(mm >> 16));
      const h2 s2 = {sc, sc};
      const _Float16 nz = -((_Float16)1024.0f + zf);          // exact: |1024 + z| <= 1039
      const _Float16 nz16 = -((_Float16)64.0f + zf);
      const h2 nz2 = {nz, nz}, nz16_2 = {nz16, nz16};
      const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned w = st.a[r][0][s];
        const unsigned w8 = w >> 8;
        u32x4 f;
        // nibbles at mantissa bits 0..3 give fp16(1024 + q); at bits 4..7 fp16(1024 + 16 q): one shift per
        // dword instead of three.  (q - z) is exact in fp16 either way, then one rounding in (q - z) * s.
        const h2 t0 = __builtin_bit_cast(h2, (w & 0x000F000Fu) | 0x64006400u);
        const h2 t1 = __builtin_bit_cast(h2, (w & 0x00F000F0u) | 0x64006400u);
        const h2 t2 = __builtin_bit_cast(h2, (w8 & 0x000F000Fu) | 0x64006400u);
        const h2 t3 = __builtin_bit_cast(h2, (w8 & 0x00F000F0u) | 0x64006400u);
#ifdef UMB_EXP_NODEQ
        f[0] = w; f[1] = w8; f[2] = w + 1; f[3] = w8 + 1;
#else
        f[0] = __builtin_bit_cast(unsigned, (t0 + nz2) * s2);
        f[1] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(t1, sixteenth, nz16_2) * s2);
        f[2] = __builtin_bit_cast(unsigned, (t2 + nz2) * s2);
        f[3] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(t3, sixteenth, nz16_2) * s2);
#endif
#ifdef UMB_EXP_NOMFMA
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) { acc[r][tt][0] += __uint_as_float(f[0] ^ b[tt][s][0]); acc[r][tt][1] += __uint_as_float(f[1] ^ f[2] ^ f[3]); }
#else
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[r][tt] = P::mfma(f, b[tt][s], acc[r][tt]);
#endif
      }
    }
  } else if (AWQ == 1) {
    const u32x4 ones = {P::ONE2, P::ONE2, P::ONE2, P::ONE2};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 xs[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      xs[tt] = zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) xs[tt] = P::mfma(ones, b[tt][s], xs[tt]);
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
        for (int tt = 0; tt < TT; ++tt) ga[tt] = P::mfma(f, b[tt][s], ga[tt]);
      }
      float sc[4], zo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = F16::to_f((u16)(st.m4[r][e] & 0xffffu));
        zo[e] = F16::to_f((u16)(st.m4[r][e] >> 16)) + P::MAGIC_OFF;
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float sx = xs[tt][0];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][tt][e] += sc[e] * (ga[tt][e] - zo[e] * sx);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[r][tt] = P::mfma(st.a[r][s], b[tt][s], acc[r][tt]);
  }
}

enum { EPI_PARTIAL = 0, EPI_ROUND = 1, EPI_SILU = 2 };

template <typename P, int AWQ, int TT, int R, int CB>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const u32x4* __restrict__ wp,
                                                          const unsigned char* __restrict__ meta,
                                                          const u16* __restrict__ x, int ldx,
                                                          float* __restrict__ out, int T, int Ttot, int N, int K,
                                                          int S, int epi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);
  constexpr int F = CB * TT * 4;           // 1 KiB fragments per chunk
  constexpr int FPW = F / 4;               // fragments staged per wave
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int NT = N / 16;
  const int nblk = (NT + 4 * R - 1) / (4 * R);
  const int sp = blockIdx.x / nblk;
  const int nt0 = ((blockIdx.x % nblk) * 4 + wv) * R;
  const bool active = nt0 < NT;            // NT % R == 0 (host guarantees)
  const int KB = K / 128;
  const int per = (KB + S - 1) / S;
  const int kb0 = sp * per;
  const int kb1 = min(KB, kb0 + per);
  const int nchunks = (kb1 - kb0 + CB - 1) / CB;

  f32x4 acc[R][TT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[r][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 xr[FPW];
  auto load_x = [&](int c) {
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const int f = i * 4 + wv;
      const int kb = kb0 + c * CB + f / (TT * 4);
      const int tok = ((f >> 2) % TT) * 16 + j;
      const u32x4 z = {0u, 0u, 0u, 0u};
      xr[i] = (tok < T && kb < kb1)
                  ? *reinterpret_cast<const u32x4*>(x + (long)tok * ldx + kb * 128 + (f & 3) * 32 + g * 8) : z;
    }
  };
  auto store_x = [&](int c) {
#ifdef UMB_EXP_NOX
    if (c > 1) return;
#endif
#pragma unroll
    for (int i = 0; i < FPW; ++i) xs[((c & 1) * F + i * 4 + wv) * 64 + lane] = xr[i];
  };

  // ring of PF = 2*CB weight stages: PF x R tiles of 1 KiB loads in flight per wave
  constexpr int PF = 2 * CB;
  Stage<P, AWQ, R> st[PF];
  load_x(0);
  if (active) {
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (kb0 + i < kb1) stage_load<P, AWQ, R>(st[i], wp, meta, nt0, KB, kb0 + i, lane);
  }
  auto chunk = [&](int c, auto half) {
    constexpr int H = decltype(half)::value;           // which half of the ring this chunk uses
    store_x(c);
#ifdef UMB_EXP_NOX
    if (c < 2) __syncthreads();
    if (c + 1 < 2) load_x(c + 1);
#else
    __syncthreads();
    if (c + 1 < nchunks) load_x(c + 1);
#endif
    if (active) {
      const u32x4* xc = xs + (c & 1) * F * 64;
#pragma unroll
      for (int kl = 0; kl < CB; ++kl) {
        const int kb = kb0 + c * CB + kl;
        if (kb < kb1) {
          stage_compute<P, AWQ, TT, R>(st[H * CB + kl], xc + kl * TT * 4 * 64, lane, acc);
          if (kb + PF < kb1) stage_load<P, AWQ, R>(st[H * CB + kl], wp, meta, nt0, KB, kb + PF, lane);
        }
      }
    }
  };
  for (int c = 0; c < nchunks; c += 2) {
    chunk(c, std::integral_constant<int, 0>{});
    if (c + 1 < nchunks) chunk(c + 1, std::integral_constant<int, 1>{});
  }
  if (!active) return;

#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int tok = tt * 16 + j;
    if (tok < T) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f32x4 v = acc[r][tt];
        if (epi == EPI_SILU) {
          // rows are interleaved (gate_m, up_m): act[tok][m] = silu(gate) * up, every step rounded to the
          // model dtype as eager torch does (umbrella/models/llama.py:107-110).  out is 16-bit [Ttot][N/2].
          const float g0 = rnd<P>(v[0]), u0 = rnd<P>(v[1]), g1 = rnd<P>(v[2]), u1 = rnd<P>(v[3]);
          const float a0 = rnd<P>(g0 / (1.f + __expf(-g0))) * u0, a1 = rnd<P>(g1 / (1.f + __expf(-g1))) * u1;
          u16* act = reinterpret_cast<u16*>(out);
          *reinterpret_cast<unsigned*>(act + (long)tok * (N / 2) + (nt0 + r) * 8 + g * 2) = pack2<P>(a0, a1);
        } else {
          if (epi == EPI_ROUND) { v[0] = rnd<P>(v[0]); v[1] = rnd<P>(v[1]); v[2] = rnd<P>(v[2]); v[3] = rnd<P>(v[3]); }
          *reinterpret_cast<f32x4*>(out + ((long)sp * Ttot + tok) * N + (nt0 + r) * 16 + g * 4) = v;
        }
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
  if (!awq && NT % 2 == 0 && NT >= 4096) R = 2;   // int4 tile order is tied to R == 1 (4 n-tiles per block)
  int S = 1;
  if (!force_s1) {
    const int nblk = (NT + 4 * R - 1) / (4 * R);
    S = (1024 + nblk - 1) / nblk;                            // aim at ~1024 blocks (16 waves / CU)
    const int min_blocks = awq ? 4 : 2;                       // keep >= 512 / 256 k per slab
    if (S > KB / min_blocks) S = KB / min_blocks;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
  }
  *R_out = R;
  *S_out = S;
}

template <typename P, int AWQ, int TT, int R, int CB>
static int launch_k(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int Ttot, int N,
                    int K, int S, int epi, hipStream_t st) {
  const int NT = N / 16;
  const int nblk = (NT + 4 * R - 1) / (4 * R);
  const size_t smem = (size_t)2 * CB * TT * 4 * 1024;
  hipLaunchKernelGGL((skinny_gemm_kernel<P, AWQ, TT, R, CB>), dim3((unsigned)(nblk * S)), dim3(256), smem, st,
                     (const u32x4*)wp, (const unsigned char*)meta, x, ldx, out, T, Ttot, N, K, S, epi);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

template <typename P, int AWQ, int TT, int CB>
static int launch_r(int R, const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int Ttot,
                    int N, int K, int S, int epi, hipStream_t st) {
  if ((N / 16) % R) return UMB_EINVAL;
  if (R == 1) return launch_k<P, AWQ, TT, 1, CB>(wp, meta, x, ldx, out, T, Ttot, N, K, S, epi, st);
  if (R == 2) return launch_k<P, AWQ, TT, 2, CB>(wp, meta, x, ldx, out, T, Ttot, N, K, S, epi, st);
  return UMB_EINVAL;
}

template <typename P, int AWQ>
static int launch_tt(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int N, int K, int R,
                     int S, int epi, hipStream_t st) {
  // tokens beyond 64 go through further launches (weights re-read from L2/HBM)
  const long ostride = (epi == EPI_SILU) ? (long)(N / 2) / 2 : (long)N;   // out rows in units of float
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int tn = min(64, T - t0);
    const u16* xx = x + (long)t0 * ldx;
    float* oo = out + (long)t0 * ostride;      // out is [S][T][N] over the full T; split stride stays T
    int rc;
    if (tn <= 16) rc = launch_r<P, AWQ, 1, 4>(R, wp, meta, xx, ldx, oo, tn, T, N, K, S, epi, st);
    else if (tn <= 32) rc = launch_r<P, AWQ, 2, 4>(R, wp, meta, xx, ldx, oo, tn, T, N, K, S, epi, st);
    else rc = launch_r<P, AWQ, 4, 2>(R, wp, meta, xx, ldx, oo, tn, T, N, K, S, epi, st);
    if (rc) return rc;
  }
  return UMB_OK;
}

// out: fp32 [S][T][N] partials (epi 0/1) or 16-bit act [T][N/2] (epi 2 = fused SiLU*up, needs S == 1 and
// gate/up rows interleaved by the repack, see umb_repack_* `interleave`)
extern "C" int umb_gemm(void* out, const void* x, int ldx, const void* wpacked, const void* meta, int T, int N, int K,
                        int awq, int S, int R, int epi, int dtype, hipStream_t st) {
  if (N % 16 || K % 128 || T < 1 || S < 1 || epi < 0 || epi > 2 || (epi == EPI_SILU && S != 1)) return UMB_EINVAL;
  if (awq && (N % 64 || R != 1)) return UMB_EINVAL;
  if (awq && dtype == UMB_F16)     // exact fp16 dequant in registers (same W as the reference's dequantize kernel)
    return launch_tt<F16, 2>(wpacked, meta, (const u16*)x, ldx, (float*)out, T, N, K, R, S, epi, st);
  DISPATCH_DTYPE(dtype, {
    if (awq) return launch_tt<P, 1>(wpacked, meta, (const u16*)x, ldx, (float*)out, T, N, K, R, S, epi, st);
    return launch_tt<P, 0>(wpacked, meta, (const u16*)x, ldx, (float*)out, T, N, K, R, S, epi, st);
  })
}

extern "C" int umb_repack_dense(void* out, const void* w, int N, int K, int interleave, int dtype, hipStream_t st) {
  if (N % 16 || K % 32) return UMB_EINVAL;
  const long total = (long)(N / 16) * (K / 32) * 64;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((repack_dense_kernel<P>), grid, block, 0, st, (const u16*)w, (u32x4*)out, N, K, interleave);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// outw: N*K/2 bytes; meta: (N/16)*(K/128)*64 bytes
extern "C" int umb_awq_repack(void* outw, void* meta, const void* qweight, const void* qzeros, const void* scales,
                              int N, int K, int group, int interleave, hipStream_t st) {
  if (N % 64 || K % 128 || group != 128) return UMB_EINVAL;
  const long total = (long)(N / 16) * (K / 128) * 64;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipLaunchKernelGGL(repack_awq_kernel, grid, block, 0, st, (const unsigned*)qweight, (const unsigned*)qzeros,
                     (const u16*)scales, (u32x4*)outw, (unsigned char*)meta, N, K, interleave);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}
