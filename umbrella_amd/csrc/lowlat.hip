// Low-latency decoder-layer GEMMs for T <= 64 tokens (draft expand, tree verify): 5 launches per layer.
//
// Replaces on the hot path (reference call sites):
//   F.linear / AwqLinear.apply              umbrella/models/llama.py:89-91,103,107-111,133, awq_utils.py:63-86
//   layer_norm (flashinfer.rmsnorm)         umbrella/models/model_utils.py:54-64
//   apply_rotary_pos_emb + update_kv_cache  umbrella/models/model_utils.py:17-52, umbrella/attn/cache.py:53-65
//   residual adds, SiLU(gate) * up          umbrella/models/llama.py:104-113
//
// Why a second GEMM family.  The split-K kernels of gemm.hip spread a linear over the chip by cutting K across
// workgroups, which costs a reduce kernel (>= 4.7 us of dependent launch + an fp32 partial round trip) behind three
// of the four linears of a layer: 8 launches per layer, ~45 % of a 1B draft forward and ~20 % of a 70B verify layer
// spent between the GEMMs.  Here a workgroup owns its output rows for the WHOLE K:
//   * block = NW (8 / 16) waves = WN row groups x WK K-slices; wave (wn, wk) streams k-blocks wk, wk + WK, ... of
//     its R n-tiles (16 rows each) HBM -> VGPR -> MFMA A operand; the WK partial accumulators are summed through LDS
//     in slice order (deterministic, T-independent -> batch invariant) and the epilogue runs in the same launch;
//   * activations live in HBM in MFMA *B-fragment order* ("FM" layout: [K/32][TT][64 lanes][8 halfs], token tile
//     tt = t / 16, lane = (k % 32 / 8) * 16 + t % 16): one B fragment = one contiguous, fully coalesced 1 KiB load
//     that every wave takes straight from L2 -- no LDS staging, no barrier in the main loop, and waves of a block
//     need not share a K slab (which is what forced the cross-block split);
//   * RMSNorm is split as in the fused schedule of gemm.hip: the producer of the residual stream writes
//     hw = h * w_next (FM) and per-block sums of squares; the consumer multiplies its OUTPUTS by
//     rsqrt(mean(h^2) + eps) (a per-token factor commutes with the matmul);
//   * int4 (AWQ) weights: zero-point and scale are folded out of the MFMA.  The A operand is the raw code with a
//     magic exponent -- v_and_or_b32 gives fp16 (1024 + q) for nibbles at mantissa bits 0..3 and (64 + q) for bits
//     4..7, 5 VALU per 8 weights instead of 13 for the exact in-register dequant -- and per 128-k group
//         out += s * ( sum_k (c_k + q_k) x_k  -  sum_k c_k x_k  -  z * sum_k x_k )
//     where the two x-only sums come from MFMAs against constant fragments (shared by the R tiles of the wave; the
//     first is the C-in of the code MFMAs).  All products are exact in fp32; the result is s * sum (q - z) x with no
//     fp16 rounding of (q - z) * s, i.e. closer to the real-valued model than awq_ext's dequantised weights
//     (tolerance stated in tests/test_lowlat.py).
// Epilogues (run by wave (wn, 0) on its R x TT accumulator fragments; lane = (token j, row group g)):
//   LOGITS  fp32 out[T][N], optionally * 1/rms and rounded to the model dtype (lm_head, llama.py:133)
//   SILU    rows interleaved (gate_m, up_m): act = SiLU(gate) * up -> FM layout for the down projection
//   QKV     (+bias) 1/rms, RoPE at pos[t], q -> [T][Hq][D], K / V^T appended at slot[t]
//   RESID   h += round(out); hw = h * w_next (FM); ssq_out[t][row group] = sum h^2
#include "../../include/umbrella_hip.h"
#include "common.h"
#include <cstdlib>
#include <type_traits>

// fm_off(t, f, TT): element offset of (token t, feature f) in an FM-layout activation -- csrc/common.h

enum { LL_LOGITS = 0, LL_SILU = 2, LL_QKV = 3, LL_RESID = 4 };

struct LLArgs {
  const u32x4* w; const unsigned char* meta; const u32x4* x; void* out;
  int T, N, K, WN, WK, epi, row_from, round_out;
  const float* ssq_in; int ssq_groups, ssq_in_stride; float ssq_dim, eps;
  u16* h; u16* hw; const u16* norm_w; float* ssq_out; int ssq_out_stride;
  const int* pos; const int* slot; const u16* cosT; const u16* sinT; u16* q_out; u16* kc; u16* vt; const u16* bias;
  int Hq, Hkv, D, Lmax, rope_heads;
};

template <int AWQ, int R> struct LStage {           // weights of one k-block (R n-tiles)
  u32x4 a[R][AWQ ? 1 : 4];
};
template <int R> struct LMeta {                       // int4: 4 x {fp16 scale, fp16 zero} of output rows 4g .. 4g+3, per n-tile
  u32x4 m4[R];
};

// (a & mask) | magic.  Plain C on purpose: the result feeds an MFMA operand directly, and the VALU-write -> MFMA-read
// wait states are only inserted for instructions the compiler can see (an inline-asm v_and_or_b32 here produced wrong
// columns at some ring depths).  With `magic` pinned in a VGPR the pattern still selects to one v_and_or_b32.
__device__ __forceinline__ unsigned and_or_b32(unsigned a, unsigned mask, unsigned magic_v) { return (a & mask) | magic_v; }

// packed source row of HF feature order (inverse use: bias lookup) -- same map as gemm.hip rowmap()
__device__ __forceinline__ int ll_rowmap_qkv(int n, int D, int rope_heads) {
  const int head = n / D, dp = n % D;
  if (head >= rope_heads) return n;
  return head * D + ((dp & 1) ? (dp >> 1) + D / 2 : (dp >> 1));
}

// 8 waves per block, up to 256 registers (two waves per SIMD): the operand rings want the registers more than the
// kernel wants a second resident block -- bytes in flight per CU come from the ring depth.
// PF = k-blocks of weights in flight per wave; the host guarantees that every wave owns the same number nk = KB / WK
// of k-blocks and that nk is a multiple of PF, so the main loop has no predicate at all.
template <typename P, int AWQ, int TT, int R, int PF>
__global__ __launch_bounds__(512, 2) void ll_gemm_kernel(const u32x4* __restrict__ a_w, const unsigned char* __restrict__ a_meta,
                                                       const u32x4* __restrict__ a_x, int a_T, int a_N, int a_K, int a_WN,
                                                       int a_WK, int a_epi, int a_row_from, int a_round_out,
                                                       const LLArgs a) {
  // the leading scalars (14 dwords) are preloaded into SGPRs at wave launch (-amdgpu-kernarg-preload-count): the
  // operand streams start without waiting for a kernarg fetch; `a` carries the epilogue's arguments
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NF = R * TT;                 // accumulator fragments per wave
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NW = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int WK = a_WK;
  const int wn = __builtin_amdgcn_readfirstlane(wv / WK), wk = wv - wn * WK;     // scalar (the division runs on the VALU)
  const int NT = a_N / 16, KB = a_K / 128;
  const int grp = blockIdx.x * a_WN + wn;
  const int nt0 = grp * R;
  const bool active = nt0 < NT;
  const int nk = active ? KB / WK : 0;                       // this wave's k-blocks: kb = wk + i * WK (KB % WK == 0, nk % PF == 0)

  f32x4 acc[R][TT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[r][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Streams: buffer loads with one descriptor per array, a 32-bit lane offset (VGPR) and a wave-uniform 32-bit byte
  // offset (SGPR) -- no per-load 64-bit VALU address arithmetic, no address registers.  (Arrays are < 4 GiB.)
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a_w), 0, 0xffffffffu, 0x00020000);
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a_x), 0, (unsigned)(TT * 16 * 2) * (unsigned)a_K, 0x00020000);   // exact size: see load_x
  const auto rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a_meta), 0, 0xffffffffu, 0x00020000);
  const int voff = lane * 16, voff_m = g * 16;
  unsigned wofs[R];                                                    // byte offset of k-block 0 of n-tile r
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const unsigned nt = active ? nt0 + r : 0;
    wofs[r] = AWQ ? (((nt >> 2) * (unsigned)KB) * 4u + (nt & 3u)) * 1024u     // int4 tile order [N/64][K/128][4], 1 KiB tiles
                  : (nt * (unsigned)KB) * 4096u;                              // dense [N/16][K/32] tiles, 4 per k-block
  }
  constexpr int NT_AUX = 2;                                            // non-temporal: weights are read once
  auto load_meta = [&](LMeta<R>& mt, int kb) {
    if constexpr (AWQ) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        mt.m4[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_m, voff_m, (int)((wofs[r] + (unsigned)kb * 4096u) >> 4), 0);   // 64 B per tile
    }
  };
  auto load_stage = [&](LStage<AWQ, R>& st, int kb) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned o = wofs[r] + (unsigned)kb * 4096u;               // both formats advance 4 KiB per k-block
      if (AWQ) {
        st.a[r][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff, (int)o, NT_AUX);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          st.a[r][AWQ ? 0 : s] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff, (int)(o + s * 1024u), NT_AUX);
      }
    }
  };
  // The B fragment's lane (g, j) carries token tt * 16 + j.  Lanes of tokens >= T get an offset beyond the buffer
  // descriptor's num_records: the load returns 0 for them without touching memory (their MFMA columns are never
  // stored) -- a 3-row draft level moves 256 B per fragment instead of 1 KiB through the CU's vector-memory path, which
  // with 8 private K-slices per block is what bounds the small-N linears (one CU sustains ~50 GB/s).  No branch, no
  // exec juggling: the bounds check of the buffer instruction does it.
  int voff_x[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) voff_x[tt] = (tt * 16 + j < a_T) ? voff : (int)0x80000000;
  auto load_x = [&](u32x4 (&xb)[TT][4], int kb) {                      // the 4 x TT B fragments of k-block kb (from L2)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        xb[tt][s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_x[tt], (int)(((unsigned)(kb * 4 + s) * TT + tt) * 1024u), 0);
  };
  unsigned magic_lo = P::MAGIC;                          // fp16 1024 / bf16 128: nibble at mantissa bits 0..3
  unsigned magic_hi = 0x54005400u;                       // fp16 64: nibble at mantissa bits 4..7 (fp16 only)
  if (AWQ) { asm volatile("" : "+v"(magic_lo)); asm volatile("" : "+v"(magic_hi)); }
  // One k-block (k32-step major).
  auto compute = [&](const LStage<AWQ, R>& st, const LMeta<R>& mt, const u32x4 (&xb)[TT][4]) {
    if constexpr (!AWQ) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][tt] = P::mfma(st.a[r][s], xb[tt][s], acc[r][tt]);
      }
    } else {
      constexpr bool HALF = std::is_same<P, F16>::value;
      // per token: -sum_k c_k x_k and sum_k x_k from MFMAs against constant A fragments (every row the same, so both
      // sums land in every accumulator register)
      const unsigned nlo = HALF ? 0xE400E400u : 0xC300C300u;           // -1024 | -128
      const unsigned nhi = HALF ? 0xD400D400u : 0xC300C300u;           // -64   | -128
      const u32x4 negc = {nlo, nhi, nlo, nhi};
      const u32x4 ones = {P::ONE2, P::ONE2, P::ONE2, P::ONE2};
      f32x4 np[TT], sx[TT], ga[R][TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        np[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        sx[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) ga[r][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const unsigned w = st.a[r][0][s];
          u32x4 f;
          if constexpr (HALF) {
            const unsigned w8 = w >> 8;
            f[0] = and_or_b32(w, 0x000F000Fu, magic_lo);
            f[1] = and_or_b32(w, 0x00F000F0u, magic_hi);
            f[2] = and_or_b32(w8, 0x000F000Fu, magic_lo);
            f[3] = and_or_b32(w8, 0x00F000F0u, magic_hi);
          } else {
            f[0] = and_or_b32(w, 0x000F000Fu, magic_lo);
            f[1] = and_or_b32(w >> 4, 0x000F000Fu, magic_lo);
            f[2] = and_or_b32(w >> 8, 0x000F000Fu, magic_lo);
            f[3] = and_or_b32(w >> 12, 0x000F000Fu, magic_lo);
          }
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) ga[r][tt] = P::mfma(f, xb[tt][s], ga[r][tt]);
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          np[tt] = P::mfma(negc, xb[tt][s], np[tt]);
          sx[tt] = P::mfma(ones, xb[tt][s], sx[tt]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned m = mt.m4[r][e];
          const float sc = (float)__builtin_bit_cast(_Float16, (u16)(m & 0xffffu));
          const float nz = -(float)__builtin_bit_cast(_Float16, (u16)(m >> 16));
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
            acc[r][tt][e] = fmaf(sc, fmaf(nz, sx[tt][e], ga[r][tt][e] + np[tt][e]), acc[r][tt][e]);
        }
      }
    }
  };

  // ---- main loop: PF-deep weight ring; activations reloaded step by step inside compute().  Straight-line rings:
  // no branch inside, so the compiler's vmcnt bookkeeping is exact (at a control-flow join it waits for everything).
  // Vector-memory loads return in issue order (one vmcnt counter): an activation fragment requested AFTER a deep weight
  // prefetch cannot be consumed before that prefetch has landed.  So the activations ride in a ring of the SAME depth as
  // the weights and are requested together with them -- a k-block's operands arrive together, PF k-blocks after their
  // request, and nothing younger is ever waited for.  (First version: single activation buffer reloaded one k-block
  // ahead -> every k-block paid the HBM latency of the weight stage issued just before it: 55 % of the wave cycles
  // parked in s_waitcnt, profiles/r02_pmc_ll_gemm.txt.)
  // The int4 metadata (16 B per lane and tile) rides in its own, shallower ring: PM k-blocks ahead are enough for
  // 64-byte reads that follow the weights of the same k-block in the queue.
  constexpr int PM = PF < 2 ? PF : 2;
  LStage<AWQ, R> st[PF];
  LMeta<R> mt[PM];
  u32x4 xr[PF][TT][4];
  // this wave's share of the producer's sums of squares (summed below, fixed order)
  float ssq_part[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    float s = 0.f;
    const int t = tt * 16 + j;
    if (a.ssq_in && t < a_T) {
      const float* sq = a.ssq_in + (long)t * a.ssq_in_stride;
      for (int q = wk * 4 + g; q < a.ssq_groups; q += 4 * WK) s += sq[q];
    }
    ssq_part[tt] = s;
  }
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      load_x(xr[i], wk + i * WK);
      load_stage(st[i], wk + i * WK);
      if (i < PM) load_meta(mt[i], wk + i * WK);
    }
    int i0 = 0;
    for (; i0 + 2 * PF <= nk; i0 += PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const int i = i0 + s;
        compute(st[s], mt[s % PM], xr[s]);
        load_x(xr[s], wk + (i + PF) * WK);
        load_stage(st[s], wk + (i + PF) * WK);
        load_meta(mt[s % PM], wk + (i + PM) * WK);
        __builtin_amdgcn_sched_barrier(0);       // one k-block per scheduling region: bounds the live ranges
      }
    }
#pragma unroll
    for (int s = 0; s < PF; ++s) {               // last ring: nothing left to refill but the metadata
      compute(st[s], mt[s % PM], xr[s]);
      if (s + PM < PF) load_meta(mt[s % PM], wk + (i0 + s + PM) * WK);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- in-block reduction over the WK slices (slice order), epilogue by wave (wn, 0)
  f32x4* red = reinterpret_cast<f32x4*>(smem);                           // [NW][NF][64]
  float* ssq_l = reinterpret_cast<float*>(smem + (size_t)NW * NF * 1024);  // [NW][TT * 16]
  if (wk > 0 && active) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) red[(wv * NF + r * TT + tt) * 64 + lane] = acc[r][tt];
  }
  if (a.ssq_in) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      float s = ssq_part[tt];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) ssq_l[wv * (TT * 16) + tt * 16 + j] = s;
    }
  }
  __syncthreads();
  if (wk != 0 || !active) return;
  for (int k2 = 1; k2 < WK; ++k2) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[r][tt] += red[((wv + k2) * NF + r * TT + tt) * 64 + lane];
  }
  float inv[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    inv[tt] = 1.f;
    if (a.ssq_in) {
      float s = 0.f;
      for (int k2 = 0; k2 < WK; ++k2) s += ssq_l[(wv + k2) * (TT * 16) + tt * 16 + j];
      inv[tt] = rsqrtf(s / a.ssq_dim + a.eps);
    }
  }

  const int N = a_N, T = a_T, epi = a_epi;
  if (epi == LL_LOGITS) {
    float* out = reinterpret_cast<float*>(a.out);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = tt * 16 + j;
      if (t >= T || t < a_row_from) continue;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        f32x4 v = acc[r][tt] * inv[tt];
        if (a_round_out) { v[0] = rnd<P>(v[0]); v[1] = rnd<P>(v[1]); v[2] = rnd<P>(v[2]); v[3] = rnd<P>(v[3]); }
        *reinterpret_cast<f32x4*>(out + (long)(t - a_row_from) * N + (nt0 + r) * 16 + g * 4) = v;
      }
    }
    return;
  }
  if (epi == LL_SILU) {
    u16* act = reinterpret_cast<u16*>(a.out);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = tt * 16 + j;
      if (t >= T) continue;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const f32x4 v = acc[r][tt] * inv[tt];
        // rows are interleaved (gate_m, up_m); every step rounded to the model dtype as eager torch does (llama.py:107-110)
        const float g0 = rnd<P>(v[0]), u0 = rnd<P>(v[1]), g1 = rnd<P>(v[2]), u1 = rnd<P>(v[3]);
        const float a0 = rnd<P>(g0 / (1.f + __expf(-g0))) * u0, a1 = rnd<P>(g1 / (1.f + __expf(-g1))) * u1;
        *reinterpret_cast<unsigned*>(act + fm_off(t, (nt0 + r) * 8 + g * 2, TT)) = pack2<P>(a0, a1);
      }
    }
    return;
  }
  if (epi == LL_RESID) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = tt * 16 + j;
      float sq = 0.f;
      if (t < T) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int n = (nt0 + r) * 16 + g * 4;
          const long off = (long)t * N + n;
          const uint2 hr = *reinterpret_cast<const uint2*>(a.h + off);
          const f32x4 v = acc[r][tt];
          const float h0 = rnd<P>(rnd<P>(v[0]) + lo_f<P>(hr.x)), h1 = rnd<P>(rnd<P>(v[1]) + hi_f<P>(hr.x));
          const float h2v = rnd<P>(rnd<P>(v[2]) + lo_f<P>(hr.y)), h3 = rnd<P>(rnd<P>(v[3]) + hi_f<P>(hr.y));
          uint2 o;
          o.x = pack2<P>(h0, h1); o.y = pack2<P>(h2v, h3);
          *reinterpret_cast<uint2*>(a.h + off) = o;
          if (a.norm_w) {
            const uint2 w = *reinterpret_cast<const uint2*>(a.norm_w + n);
            uint2 ow;
            ow.x = pack2<P>(h0 * lo_f<P>(w.x), h1 * hi_f<P>(w.x)); ow.y = pack2<P>(h2v * lo_f<P>(w.y), h3 * hi_f<P>(w.y));
            *reinterpret_cast<uint2*>(a.hw + fm_off(t, n, TT)) = ow;
          }
          sq += h0 * h0 + h1 * h1 + h2v * h2v + h3 * h3;
        }
      }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (a.ssq_out && g == 0 && t < T) a.ssq_out[(long)t * a.ssq_out_stride + grp] = sq;
    }
    return;
  }
  // LL_QKV: rows were permuted at load so (2m, 2m+1) of a q / k head are RoPE partners (m, m + D/2)
  {
    const int D = a.D, half = D / 2;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int t = tt * 16 + j;
      if (t >= T) continue;
      const int sl = a.slot[t];
      const long cb = (long)a.pos[t] * D;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = (nt0 + r) * 16 + g * 4;
        const int head = n / D, dp = n % D, m = dp >> 1;
        f32x4 v = acc[r][tt] * inv[tt];
        if (a.bias) {      // F.linear(x, W, b) (qwen.py:94-96): bias joins the fp32 accumulator, one rounding
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += P::to_f(a.bias[ll_rowmap_qkv(n + e, D, a.rope_heads)]);
        }
        const float a0 = rnd<P>(v[0]), b0 = rnd<P>(v[1]), a1 = rnd<P>(v[2]), b1 = rnd<P>(v[3]);
        if (head < a.Hq + a.Hkv) {
          const unsigned cl = *reinterpret_cast<const unsigned*>(a.cosT + cb + m);
          const unsigned ch = *reinterpret_cast<const unsigned*>(a.cosT + cb + m + half);
          const unsigned sl_ = *reinterpret_cast<const unsigned*>(a.sinT + cb + m);
          const unsigned sh = *reinterpret_cast<const unsigned*>(a.sinT + cb + m + half);
          // rotate-half in the model dtype: every product and the sum are rounded (eager torch, model_utils.py:50-51)
          const float lo0 = rnd<P>(mul_rnd<P>(a0, lo_f<P>(cl)) + mul_rnd<P>(-b0, lo_f<P>(sl_)));
          const float lo1 = rnd<P>(mul_rnd<P>(a1, hi_f<P>(cl)) + mul_rnd<P>(-b1, hi_f<P>(sl_)));
          const float hi0 = rnd<P>(mul_rnd<P>(b0, lo_f<P>(ch)) + mul_rnd<P>(a0, lo_f<P>(sh)));
          const float hi1 = rnd<P>(mul_rnd<P>(b1, hi_f<P>(ch)) + mul_rnd<P>(a1, hi_f<P>(sh)));
          // q rows are row-major; the K / V^T caches are in fragment order inside a head's slab (common.h; m is even)
          u16* dlo = (head < a.Hq) ? a.q_out + ((long)t * a.Hq + head) * D + m
                                   : a.kc + (long)(head - a.Hq) * a.Lmax * D + kc_off(sl, m, D);
          u16* dhi = (head < a.Hq) ? dlo + half : a.kc + (long)(head - a.Hq) * a.Lmax * D + kc_off(sl, m + half, D);
          *reinterpret_cast<unsigned*>(dlo) = pack2<P>(lo0, lo1);
          *reinterpret_cast<unsigned*>(dhi) = pack2<P>(hi0, hi1);
        } else {
          u16* vb = a.vt + (long)(head - a.Hq - a.Hkv) * D * VT_LD(a.Lmax);
          vb[vt_off(dp, sl, D)] = P::from_f(a0); vb[vt_off(dp + 1, sl, D)] = P::from_f(b0);
          vb[vt_off(dp + 2, sl, D)] = P::from_f(a1); vb[vt_off(dp + 3, sl, D)] = P::from_f(b1);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ plan: (R, WN, WK, NW) from the layer shape only
// R n-tiles per wave, WN row groups x WK K-slices = NW waves per block.  Scored for an MI355X (256 CUs, 16 waves per
// CU at <= 128 registers): whole rounds of blocks over the CUs, >= 8 waves per CU, >= 2 k-blocks per wave.
static int ll_env_plan(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

extern "C" void umb_ll_plan(int N, int K, int awq, int* R_out, int* WN_out, int* WK_out, int* NW_out) {
  const int NT = N / 16, KB = K / 128;
  int R = (awq && NT % 2 == 0 && NT >= 512) ? 2 : 1;
  const int groups = NT / R;
  double best = -1.0;
  int bWN = 1, bWK = 1, bNW = 8;
  for (int NW = 8; NW <= 8; NW *= 2) {
    for (int WK = 1; WK <= NW; WK *= 2) {
      if (WK > KB || KB % WK) continue;
      const int WN = NW / WK;
      const int blocks = (groups + WN - 1) / WN;
      const int cap = 256 * (16 / NW);                       // resident blocks per round
      const int rounds = (blocks + cap - 1) / cap;
      double eff = (double)blocks / ((double)rounds * cap);   // tail / idle-CU loss
      if (blocks < 256) eff = (double)blocks / 256.0 * (NW == 16 ? 1.0 : 0.75);   // few blocks: deeper blocks pull more per CU
      const double waves = (double)groups * WK;
      const double fill = waves >= 2048.0 ? 1.0 : waves / 2048.0;
      const double per_wave = (double)KB / WK;
      const double depth = per_wave >= 2.0 ? 1.0 : 0.7;
      const double waste = (double)groups / ((double)blocks * WN);    // idle row groups in the last block
      const double score = eff * (0.5 + 0.5 * fill) * depth * waste * (1.0 - 0.004 * WK);
      if (score > best) { best = score; bWN = WN; bWK = WK; bNW = NW; }
    }
  }
  // experiment knobs: UMB_LL_NW (waves per block: 4 or 8), UMB_LL_WK (K-slices per block)
  static const int nw_env = ll_env_plan("UMB_LL_NW", 0), wk_env = ll_env_plan("UMB_LL_WK", 0);
  if (nw_env == 4 || nw_env == 8) {
    bNW = nw_env;
    if (bWK > bNW) bWK = bNW;
    bWN = bNW / bWK;
  }
  if (wk_env > 0 && wk_env <= bNW && (bNW % wk_env) == 0 && KB % wk_env == 0) { bWK = wk_env; bWN = bNW / bWK; }
  *R_out = R; *WN_out = bWN; *WK_out = bWK; *NW_out = bNW;
}

template <typename P, int AWQ, int TT, int R, int PF>
static int ll_launch_pf(const LLArgs& a, int NW, hipStream_t st) {
  const int groups = a.N / 16 / R;
  const int blocks = (groups + a.WN - 1) / a.WN;
  const size_t smem = (size_t)NW * R * TT * 1024 + (size_t)NW * TT * 16 * 4;
  if (smem > 64 * 1024) {
    static bool once = false;
    if (!once) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ll_gemm_kernel<P, AWQ, TT, R, PF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) return UMB_EHIP;
      once = true;
    }
  }
  hipLaunchKernelGGL((ll_gemm_kernel<P, AWQ, TT, R, PF>), dim3((unsigned)blocks), dim3(64 * NW), smem, st, a.w, a.meta, a.x, a.T, a.N, a.K,
                     a.WN, a.WK, a.epi, a.row_from, a.round_out, a);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// Ring depth from the k-blocks per wave: the deepest of {4, 2, 1} that divides it and fits the 256 registers of the
// variant (weights and activations are both PF deep).  UMB_LL_PF caps it (experiments).
static int ll_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

template <typename P, int AWQ, int TT, int R>
static int ll_launch(const LLArgs& a, int NW, hipStream_t st) {
  const int nk = (a.K / 128) / a.WK;
  static const int pf_cap = ll_env("UMB_LL_PF", 8);
  constexpr int PFMAX = TT == 1 ? 4 : (TT == 4 && AWQ) ? 1 : 2;
  if (PFMAX >= 4 && pf_cap >= 4 && nk % 4 == 0) return ll_launch_pf<P, AWQ, TT, R, PFMAX >= 4 ? 4 : 1>(a, NW, st);
  if (PFMAX >= 2 && pf_cap >= 2 && nk % 2 == 0) return ll_launch_pf<P, AWQ, TT, R, PFMAX >= 2 ? 2 : 1>(a, NW, st);
  return ll_launch_pf<P, AWQ, TT, R, 1>(a, NW, st);
}

// activation rows in FM layout are padded to a whole number of 16-token tiles: TT = 1, 2 or 4 (T = 33..48 uses 4)
static inline int ll_tt(int T) { const int t = (T + 15) / 16; return t == 3 ? 4 : t; }

extern "C" int umb_ll_token_tiles(int T) { return (T >= 1 && T <= 64) ? ll_tt(T) : 0; }

extern "C" int umb_gemm_ll(void* out, const void* x_fm, const void* wpacked, const void* meta, int T, int N, int K,
                           int awq, int epi, const UmbGemmLL* fx, int dtype, hipStream_t st) {
  if (N % 16 || K % 128 || T < 1 || T > 64 || !x_fm || !wpacked) return UMB_EINVAL;
  if (epi != LL_LOGITS && epi != LL_SILU && epi != LL_QKV && epi != LL_RESID) return UMB_EINVAL;
  if (awq && (N % 64 || !meta)) return UMB_EINVAL;
  LLArgs a = {};
  a.w = (const u32x4*)wpacked; a.meta = (const unsigned char*)meta; a.x = (const u32x4*)x_fm; a.out = out;
  a.T = T; a.N = N; a.K = K; a.epi = epi;
  int R, NW;
  umb_ll_plan(N, K, awq, &R, &a.WN, &a.WK, &NW);
  if (fx) {
    a.row_from = fx->row_from; a.round_out = fx->round_out;
    a.ssq_in = fx->ssq_in; a.ssq_groups = fx->ssq_groups; a.ssq_in_stride = fx->ssq_in_stride;
    a.ssq_dim = fx->ssq_dim; a.eps = fx->eps;
    a.h = (u16*)fx->h; a.hw = (u16*)fx->hw; a.norm_w = (const u16*)fx->norm_w; a.ssq_out = fx->ssq_out;
    a.ssq_out_stride = fx->ssq_out_stride;
    a.pos = fx->pos; a.slot = fx->slot; a.cosT = (const u16*)fx->cosT; a.sinT = (const u16*)fx->sinT;
    a.q_out = (u16*)fx->q_out; a.kc = (u16*)fx->k_cache; a.vt = (u16*)fx->vt_cache; a.bias = (const u16*)fx->bias;
    a.Hq = fx->Hq; a.Hkv = fx->Hkv; a.D = fx->D; a.Lmax = fx->Lmax; a.rope_heads = fx->Hq + fx->Hkv;
  }
  if (a.row_from < 0 || a.row_from >= T) return UMB_EINVAL;
  if (a.ssq_in && (a.ssq_groups < 1 || a.ssq_in_stride < a.ssq_groups || a.ssq_dim <= 0.f)) return UMB_EINVAL;
  if (epi == LL_RESID && (!a.h || (a.norm_w && !a.hw) || (a.ssq_out && a.ssq_out_stride < N / 16 / R))) return UMB_EINVAL;
  if (epi == LL_QKV && (!a.pos || !a.slot || !a.q_out || !a.kc || !a.vt || !a.cosT || !a.sinT || a.D % 4 ||
                        N != (a.Hq + 2 * a.Hkv) * a.D)) return UMB_EINVAL;
  if (epi == LL_QKV && (a.D % 32 || a.Lmax % 32 || a.Lmax < 32)) return UMB_EINVAL;   // fragment-ordered caches: whole tiles
  if (epi == LL_SILU && N % 32) return UMB_EINVAL;
  const int TT = ll_tt(T);
  auto run = [&](auto tag) -> int {
    using PP = decltype(tag);
    if (awq) {
      if (R == 1) return TT == 1 ? ll_launch<PP, 1, 1, 1>(a, NW, st) : TT == 2 ? ll_launch<PP, 1, 2, 1>(a, NW, st) : ll_launch<PP, 1, 4, 1>(a, NW, st);
      return TT == 1 ? ll_launch<PP, 1, 1, 2>(a, NW, st) : TT == 2 ? ll_launch<PP, 1, 2, 2>(a, NW, st) : ll_launch<PP, 1, 4, 2>(a, NW, st);
    }
    if (R == 1) return TT == 1 ? ll_launch<PP, 0, 1, 1>(a, NW, st) : TT == 2 ? ll_launch<PP, 0, 2, 1>(a, NW, st) : ll_launch<PP, 0, 4, 1>(a, NW, st);
    return UMB_EINVAL;
  };
  if (dtype == UMB_BF16) return run(BF16{});
  if (dtype == UMB_F16) return run(F16{});
  return UMB_EINVAL;
}

// ------------------------------------------------------------------ FM layout helpers
// row-major [T][K] 16-bit -> FM (op-level tests, pipeline stage hand-off); optional hw / ssq as the embed kernel
template <typename P>
__global__ __launch_bounds__(256) void to_fm_kernel(u16* __restrict__ out, const u16* __restrict__ x, int T, int K, int TT) {
  const int t = blockIdx.x;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8)
    *reinterpret_cast<u32x4*>(out + fm_off(t, k, TT)) = *reinterpret_cast<const u32x4*>(x + (long)t * K + k);
}
template <typename P>
__global__ __launch_bounds__(256) void from_fm_kernel(u16* __restrict__ out, const u16* __restrict__ x, int T, int K, int TT) {
  const int t = blockIdx.x;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8)
    *reinterpret_cast<u32x4*>(out + (long)t * K + k) = *reinterpret_cast<const u32x4*>(x + fm_off(t, k, TT));
}

extern "C" int umb_to_fm(void* out_fm, const void* x, int T, int K, int dtype, hipStream_t st) {
  if (T < 1 || T > 64 || K % 32) return UMB_EINVAL;
  DISPATCH_DTYPE(dtype, { hipLaunchKernelGGL((to_fm_kernel<P>), dim3(T), dim3(256), 0, st, (u16*)out_fm, (const u16*)x, T, K, ll_tt(T)); })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}
extern "C" int umb_from_fm(void* out, const void* x_fm, int T, int K, int dtype, hipStream_t st) {
  if (T < 1 || T > 64 || K % 32) return UMB_EINVAL;
  DISPATCH_DTYPE(dtype, { hipLaunchKernelGGL((from_fm_kernel<P>), dim3(T), dim3(256), 0, st, (u16*)out, (const u16*)x_fm, T, K, ll_tt(T)); })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// ---- measurement probe: the read-only streaming rate this device delivers (what every weight-streaming kernel here is
// priced against besides the 8 TB/s spec sheet: boxes of the same part differed by 20 % on it).  16-byte non-temporal
// loads, 8 in flight per lane, block-contiguous 32 KiB spans; the xor sink defeats dead-code elimination.
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4* __restrict__ p, long n16, unsigned* __restrict__ sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const long stride = (long)gridDim.x * 2048;
  for (long base = (long)blockIdx.x * 2048 + threadIdx.x; base < n16; base += stride) {
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long q = base + i * 256;
      v[i] = q < n16 ? __builtin_nontemporal_load(p + q) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= v[i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) *sink = 1u;
}

extern "C" int umb_stream_read(const void* p, size_t bytes, void* sink, hipStream_t st) {
  if (bytes % 16 || !sink) return UMB_EINVAL;
  hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 8), dim3(256), 0, st, (const u32x4*)p, (long)(bytes / 16), (unsigned*)sink);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// ---- embedding gather + per-forward index prep for the low-latency schedule (F.embedding, llama.py:124):
// h (row-major), hw = h * norm_w (FM), ssq[t][0..4) = per-wave sums of h^2; positions / slots / prefix resolved as in
// embed_prep_kernel (epilogue.hip) and clamped into the caches.
template <typename P>
__global__ __launch_bounds__(256) void embed_ll_kernel(u16* __restrict__ h, const u16* __restrict__ table, int H, int V,
                                                       int Lmax, const int* __restrict__ tok_in,
                                                       const int* __restrict__ pos_in, const int* __restrict__ slot_in,
                                                       const int* __restrict__ prefix_in,
                                                       const int* __restrict__ tokens_all, const int* __restrict__ n_ptr,
                                                       int off, const int* __restrict__ depth, int* __restrict__ pos_out,
                                                       int* __restrict__ slot_out, int* __restrict__ prefix_out,
                                                       u16* __restrict__ hw, const u16* __restrict__ norm_w,
                                                       float* __restrict__ ssq, int ssq_stride, int TT) {
  const int i = blockIdx.x;
  int tok, p, s, pre;
  if (tokens_all) {
    const int n = *n_ptr;
    tok = table ? tokens_all[n + off + i] : 0; p = n + depth[off + i]; s = n + off + i; pre = n;
  } else {
    tok = table ? tok_in[i] : 0; p = pos_in[i]; s = slot_in[i]; pre = *prefix_in;
  }
  tok = min(max(tok, 0), V - 1); p = min(max(p, 0), Lmax - 1); s = min(max(s, 0), Lmax - 1);
  if (threadIdx.x == 0) { pos_out[i] = p; slot_out[i] = s; if (i == 0) *prefix_out = pre; }
  const u32x4* src = reinterpret_cast<const u32x4*>(table ? table + (long)tok * H : h + (long)i * H);
  u32x4* dst = reinterpret_cast<u32x4*>(h + (long)i * H);
  float sq = 0.f;
  for (int k = threadIdx.x; k < H / 8; k += 256) {
    const u32x4 v = src[k];
    if (table) dst[k] = v;
    const u32x4 w = *reinterpret_cast<const u32x4*>(norm_w + k * 8);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = lo_f<P>(v[e]), b = hi_f<P>(v[e]);
      o[e] = pack2<P>(a * lo_f<P>(w[e]), b * hi_f<P>(w[e]));
      sq += a * a + b * b;
    }
    *reinterpret_cast<u32x4*>(hw + fm_off(i, k * 8, TT)) = o;
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) ssq[(long)i * ssq_stride + (threadIdx.x >> 6)] = sq;
}

extern "C" int umb_embed_ll(void* h, const void* table, int H, int V, int Lmax, int T, const int* tok, const int* pos,
                            const int* slot, const int* prefix, const int* tokens_all, const int* n_ptr, int off,
                            const int* depth, int* pos_out, int* slot_out, int* prefix_out, void* hw_fm,
                            const void* norm_w, float* ssq, int ssq_stride, int dtype, hipStream_t st) {
  if (H % 64 || T < 1 || T > 64 || !hw_fm || !norm_w || !ssq || ssq_stride < 4) return UMB_EINVAL;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((embed_ll_kernel<P>), dim3(T), dim3(256), 0, st, (u16*)h, (const u16*)table, H, V, Lmax, tok, pos,
                       slot, prefix, tokens_all, n_ptr, off, depth, pos_out, slot_out, prefix_out, (u16*)hw_fm,
                       (const u16*)norm_w, ssq, ssq_stride, ll_tt(T));
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}
