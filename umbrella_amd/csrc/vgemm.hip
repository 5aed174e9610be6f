// Wide verify / prefill GEMM (round 6):  out[t][n] = sum_k x[t][k] * W[n][k],  W = AWQ int4, T = 65 .. 1024 tokens.
//
// Replaces on the hot path (reference call sites):
//   AwqLinear.apply -> awq_ext.gemm_forward_cuda(x, qweight, scales, qzeros, 8)   umbrella/quantization/awq_utils.py:63-86
//   (ONE kernel tuned for T <= 16 reused by the reference at the tree verify, T = 257 ... 769, and at every prompt chunk)
//
// Regime: matrix-pipe bound, and -- measured, profiles/r06_vgemm_clock_power_trace.txt -- at the chip's POWER cap on random
// data (1.39 kW at 1.88-1.98 GHz; 1.30 kW at 2.39 GHz on zero activations).  What costs time is therefore work per flop,
// not only issue slots.  The shape of this kernel follows from that:
//   * a wave owns 32 weight rows x ALL tokens of its chunk (up to 18 token tiles = 288 tokens; 2 x 18 accumulators).  Its
//     int4 tiles are dequantised in its own registers (exact fp16, W = fp16((q - z) * s), the reference's arithmetic) ONCE
//     per 288 tokens: 0.72 VALU ops per MFMA instead of the 1.63 of the 64-row x 128-token work items of
//     verify_gemm_pp_kernel, no trip of the weights through LDS, weights read from HBM once per chunk of 288 tokens
//     instead of once per 128 / 144;
//   * T = w d + 1 is cut into chunks of WHOLE token tiles (257 -> one chunk of 17 tiles, 769 -> 17 + 16 + 16): the
//     144-token items of the previous kernel multiplied 288 / 864 rows for 257 / 769;
//   * activations never touch registers on their way in: every wave issues LDS-DMA pieces (global_load_lds_dwordx4, 8 token
//     rows x 128 B = full lines; the 16-byte column of a row is XOR-swizzled on the SOURCE side so that the B-fragment
//     ds_read_b128 is bank-conflict free) into a 3-slot ring of 64-k steps, two steps ahead of the MFMAs that read them;
//     the wave's own int4 tiles and {scale, zero} pairs ride the same queue into a wave-private 2-stage LDS buffer.  No
//     VGPR-destination load exists in the loop, so every s_waitcnt vmcnt is counted by hand and never 0;
//   * ONE raw s_barrier per 64-k step (ring hand-over), 8 waves = 2 per SIMD, 256 rows per workgroup, one workgroup per CU.
// Split-K partials and epilogues are those of gemm.hip (fp32 [S][T][N], or fused SiLU(gate) * up with S = 1).
#include "common.h"
#include "gemm_fused.h"
#include <cstdlib>
#include <type_traits>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) unsigned lds_u32;

__device__ __forceinline__ unsigned vg_and_or(unsigned a, unsigned mask, unsigned magic_v) {
  unsigned r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(magic_v));
  return r;
}

#ifdef UMB_VGW_WNT
#define VGW_WNT " nt"
#else
#define VGW_WNT ""
#endif
// one LDS-DMA piece: 64 lanes x 16 B, global address = sbase + voff (per lane), LDS address = ldsaddr + 16 * lane
__device__ __forceinline__ void dma16(unsigned voff, const void* sbase, unsigned ldsaddr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory");
}
// two pieces 1 KiB apart in global memory AND in LDS (the immediate offset moves both)
__device__ __forceinline__ void dma16x2(unsigned voff, const void* sbase, unsigned ldsaddr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2" VGW_WNT "\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:1024" VGW_WNT "\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory");
}
// 64 lanes x 4 B
__device__ __forceinline__ void dma4(unsigned voff, const void* sbase, unsigned ldsaddr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dword %1, %2\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsaddr) : "memory");
}

#ifndef UMB_VGW_BD
#define UMB_VGW_BD 4       // B fragments requested this many MFMA pairs ahead of their use
#endif
#ifndef UMB_VGW_SPREAD
#define UMB_VGW_SPREAD 1
#endif
#define VGW_WSTAGE 2304    // wave-private weight stage: 2 int4 tiles (2 KiB) + their {scale, zero} pairs (128 B, padded to 256)

template <int N> __device__ __forceinline__ void vg_wait() {
  static_assert(N >= 0 && N < 64, "vmcnt is six bits");
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// TT = token tiles (of 16) per work item; a work item = 256 rows x TT tiles x one K slab.
// (A 4-slot ROLLING variant -- the hand-over barrier BD pairs before the end of a step, B-fragment prefetch running across the
// step boundary -- was built and measured: bit-identical, +1.8 % on zero activations, +-1 % on random ones; not kept.
// profiles/r06_vgemm_w_experiments.txt)
template <int TT>
__global__ __launch_bounds__(512) void vgemm_w_kernel(const u32x4* __restrict__ wp, const unsigned char* __restrict__ meta,
                                                      const u16* __restrict__ x, int ldx, int T, int Tv, int N, int K, int S,
                                                      int epi, int nchunk, float* __restrict__ out, GemmFused fx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef F16 P;
  constexpr int NP = TT * 2;                       // 1-KiB pieces (8 token rows x 128 B) per 64-k step
  constexpr int SLOT = NP * 1024;
  constexpr int NPW = (NP + 7) / 8;                // pieces per wave and step: NPW for waves < NFULL, NPW - 1 for the rest
  constexpr int NFULL = NP - 8 * (NPW - 1);        // (8 when NP is a multiple of 8)
  constexpr int BD = UMB_VGW_BD;
  const int lane = threadIdx.x & 63;
  const int wv8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int nrb = N / 256;
  const int nb = blockIdx.x % nrb;                 // row block fastest: with nrb % 8 == 0 every chunk / split of a row block
  const int tc = (blockIdx.x / nrb) % nchunk;      // runs on the same XCD (its L2 serves the re-reads of the weights)
  const int sp = blockIdx.x / (nrb * nchunk);
  // token tiles of this chunk: the NT tiles are dealt out as evenly as whole tiles allow
  const int NT = (Tv + 15) >> 4;
  const int tbase = NT / nchunk, trem = NT % nchunk;
  const int tile0 = tc * tbase + min(tc, trem), ntl = tbase + (tc < trem ? 1 : 0);
  const int t0 = tile0 * 16, tend = min(Tv, (tile0 + ntl) * 16);
  const int KB = K / 128;
  const int per = (KB + S - 1) / S;
  const int kb0 = sp * per, kb1 = min(KB, sp * per + per);

  const unsigned lds0 = (unsigned)(size_t)(lds_char*)smem;
  constexpr unsigned NSLOT = 3u, WSTAGES = 2u;
  const unsigned wst0 = lds0 + NSLOT * SLOT + (unsigned)wv8 * (WSTAGES * VGW_WSTAGE);

  f32x4 acc[2][TT];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (kb0 >= kb1) return;                          // (never: the host keeps S <= KB)

  // ---- sources.  Weights: tile order [N/64][K/128][4] (gemm.hip repack_awq_kernel); this wave's two tiles are adjacent.
  const long wtile0 = ((long)(nb * 4 + (wv8 >> 1)) * KB) * 4 + (wv8 & 1) * 2;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(wp) + wtile0 * 1024;
  const unsigned char* msrc = meta + wtile0 * 64;
  const unsigned vw = (unsigned)lane * 16u, vm = (unsigned)(lane & 31) * 4u;
  // Activations: piece p = 8 rows; lane l lands in 16-byte unit l of the piece = row l / 8, unit (c ^ row) -> it fetches
  // column c = (l % 8) ^ (l / 8) of that row (64 k = 8 columns of 16 B).  Rows past the launch's last token re-read it.
  unsigned vx[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = (i == NPW - 1 && wv8 >= NFULL) ? wv8 : wv8 + 8 * i, r = lane >> 3, c = (lane & 7) ^ r;
    const int tok = min(t0 + 8 * p + r, Tv - 1);
    vx[i] = (unsigned)(((long)tok * ldx + c * 8) * 2);
  }
  // Every wave issues NPW pieces per step, so the counted waits are the same constants in every wave and the loop has no
  // branch: a wave without an NPW-th piece of its own (NP is not always a multiple of 8) fetches its first piece again
  // into a dump KiB behind the ring.
  const bool full = wv8 < NFULL;
  const unsigned dump = lds0 + NSLOT * SLOT + 8u * WSTAGES * VGW_WSTAGE;
  // piece i of this wave for 64-k step s -> ring slot pos
  auto issue_b1 = [&](int s, int pos, int i) {
#ifdef UMB_VGW_ABL_XHOT       // ablation (wrong results): every step re-reads the first 64 k of x (cache-hot source)
    const u16* xs = x + (long)(min(s, 2 * kb1 - 1) & 1) * 64;
#else
    const u16* xs = x + (long)min(s, 2 * kb1 - 1) * 64;
#endif
    const unsigned dst = lds0 + (unsigned)pos * SLOT + (unsigned)wv8 * 1024u + (unsigned)i * 8192u;
#if defined(UMB_VGW_ABL_NODMA) || defined(UMB_VGW_ABL_NOX)     // ablation (wrong results): the prologue's loads only
    if (s > 2 * kb0 + 1) return;
#endif
    dma16(vx[i], xs, (i == NPW - 1 && !full) ? dump : dst);
  };
  auto issue_b = [&](int s, int pos) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) issue_b1(s, pos, i);
  };
  // int4 tiles + metadata of 128-k block kb -> stage kb & 1   (3 pieces: part 0 = the two tiles, part 1 = metadata)
  auto issue_w1 = [&](int kb, int part) {
    const int kk = min(kb, kb1 - 1);
    const unsigned dst = wst0 + (unsigned)(kb & 1) * VGW_WSTAGE;
#if defined(UMB_VGW_ABL_NODMA) || defined(UMB_VGW_ABL_NOW)
    if (kb > kb0 + 1) return;
#endif
    if (part == 0) dma16x2(vw, wsrc + (long)kk * 4096, dst);
    else dma4(vm, msrc + (long)kk * 256, dst + 2048u);
  };
  auto issue_w = [&](int kb) { issue_w1(kb, 0); issue_w1(kb, 1); };
  // counted waits: the step's own group may stay in flight
#define VGW_WAIT(extra) vg_wait<NPW + (extra)>()

  // ---- operand state
  u32x4 raw[2];                                    // the wave's two int4 tiles of the current 128-k block (lane: row i, k-group g)
  h2 s2[2], nz2[2], nz16_2[2];
  u32x4 wcur[2][2], wnext[2][2];                   // dequantised A fragments [n-tile][k32 half] of this / the next 64-k step
  unsigned magic = 0x64006400u;
  asm volatile("" : "+v"(magic));
  const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  auto read_raw = [&](int kb) {
    const unsigned st = wst0 + (unsigned)(kb & 1) * VGW_WSTAGE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      raw[q] = *reinterpret_cast<const lds_u32x4*>((size_t)(st + (unsigned)q * 1024u + (unsigned)lane * 16u));
      const unsigned mm = *reinterpret_cast<const lds_u32*>((size_t)(st + 2048u + (unsigned)q * 64u + (unsigned)j * 4u));
      const _Float16 sc = __builtin_bit_cast(_Float16, (u16)(mm & 0xffffu));
      const _Float16 zf = __builtin_bit_cast(_Float16, (u16)(mm >> 16));
      s2[q] = h2{sc, sc};
      const _Float16 nz = -((_Float16)1024.0f + zf), nz16 = -((_Float16)64.0f + zf);
      nz2[q] = h2{nz, nz};
      nz16_2[q] = h2{nz16, nz16};
    }
  };
  // one dword (two weights) of fragment (q, sx) of k64-half hf: exact fp16 dequant, W = fp16((code - zero) * scale)
  auto deq1 = [&](u32x4 (&dst)[2][2], int hf, int idx) {
    const int q = idx >> 3, sx = (idx >> 2) & 1, d = idx & 3;
    const unsigned w = raw[q][hf * 2 + sx];
    const unsigned ws = (d & 2) ? (w >> 8) : w;
#ifdef UMB_VGW_ABL_NODEQ      // ablation (wrong results): raw bits as the operand
    dst[q][sx][d] = ws;
    return;
#endif
    if (d & 1) {
      const h2 c = __builtin_bit_cast(h2, vg_and_or(ws, 0x00F000F0u, magic));
      dst[q][sx][d] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(c, sixteenth, nz16_2[q]) * s2[q]);
    } else {
      const h2 c = __builtin_bit_cast(h2, vg_and_or(ws, 0x000F000Fu, magic));
      dst[q][sx][d] = __builtin_bit_cast(unsigned, (c + nz2[q]) * s2[q]);
    }
  };

  // B fragment (token tile tt, k32 half sx) of a ring slot: lane (j, g) reads row 16 tt + j, column 4 sx + g
  unsigned boff[2];
#pragma unroll
  for (int sx = 0; sx < 2; ++sx)
    boff[sx] = (unsigned)((j >> 3) * 1024 + ((j & 7) * 8 + ((sx * 4 + g) ^ (j & 7))) * 16);

  // ---- prologue: group A = {W(kb0), B(s0)}, group B = {B(s0 + 1), W(kb0 + 1)}
  const int s0 = 2 * kb0;
  issue_w(kb0);
  issue_b(s0, 0);
  issue_b(s0 + 1, 1);
  issue_w(kb0 + 1);
  VGW_WAIT(3);
  read_raw(kb0);
#pragma unroll
  for (int i = 0; i < 16; ++i) deq1(wcur, 0, i);

  int pos = 0;                                     // ring slot of the step about to be computed
  auto step = [&](auto hc, int kb) {
    constexpr int HF = decltype(hc)::value;
    const int s = 2 * kb + HF;
    // this step's activations (and, on odd steps, the next block's weights) have landed; the newest group stays in flight
    if (HF == 0) VGW_WAIT(3); else VGW_WAIT(0);
#ifndef UMB_VGW_ABL_NOBAR
    __builtin_amdgcn_s_barrier();
#endif
    const int p2 = pos == 0 ? 2 : pos - 1;         // (pos + 2) % 3: the slot every wave finished reading before this barrier
#if !UMB_VGW_SPREAD
    issue_b(s + 2, p2);
    if (HF == 1) issue_w(kb + 2);
#endif
    if (HF == 1) read_raw(kb + 1);                 // odd step: it dequantises the first half of the NEXT block
    const unsigned sb = lds0 + (unsigned)pos * SLOT;
    const lds_u32x4* bp[2] = {reinterpret_cast<const lds_u32x4*>((size_t)(sb + boff[0])),
                              reinterpret_cast<const lds_u32x4*>((size_t)(sb + boff[1]))};
    // group i (of 2 TT): fragment (tt, sx) in the order (tt0, 0), (tt0 + 1, 0), (tt0, 1), (tt0 + 1, 1): the two MFMAs on an
    // accumulator are four MFMAs apart
    auto frag_of = [&](int i, int& tt, int& sx) {
      const int pr = i >> 2, k = i & 3;
      if (2 * pr + 1 < TT) { tt = 2 * pr + (k & 1); sx = k >> 1; }
      else { tt = 2 * pr; sx = i & 1; }            // odd TT: the last tile alone
    };
    u32x4 bq[BD + 1];
#pragma unroll
    for (int i = 0; i < BD; ++i) { int tt, sx; frag_of(i, tt, sx); bq[i] = bp[sx][tt * 128]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      int tt, sx;
      frag_of(i, tt, sx);
#ifdef UMB_VGW_ABL_NOLDS      // ablation (wrong results): the first BD fragments only
      if (false)
#endif
      if (i + BD < NP) { int t2, s2x; frag_of(i + BD, t2, s2x); bq[(i + BD) % (BD + 1)] = bp[s2x][t2 * 128]; }
#ifdef UMB_VGW_ABL_NOMFMA     // ablation (wrong results)
      asm volatile("" :: "v"(bq[i % (BD + 1)]), "v"(wcur[0][sx]), "v"(wcur[1][sx]));
#else
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[q][tt] = P::mfma(wcur[q][sx], bq[i % (BD + 1)], acc[q][tt]);
#endif
      // the next step's weight fragments (16 dwords of two weights each), spread evenly over the step's MFMA pairs
#pragma unroll
      for (int k = (i * 16) / NP; k < ((i + 1) * 16) / NP; ++k) deq1(wnext, HF ^ 1, k);
#if UMB_VGW_SPREAD
      // the loads of step s + 2 (and of block kb + 2) go out between the MFMA pairs, one instruction at a time: issued in a
      // burst behind the barrier, all eight waves' pieces queue on the CU's address path while its matrix pipes wait
      {
        constexpr int NI = NPW + (HF == 1 ? 2 : 0), GAP = (NP - 2) / NI;
        if (i >= 1 && (i - 1) % GAP == 0 && (i - 1) / GAP < NI) {
          const int k = (i - 1) / GAP;
          if (k < NPW) issue_b1(s + 2, p2, k);
          else issue_w1(kb + 2, k - NPW);
        }
      }
#endif
#ifndef UMB_VGW_NOFENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) wcur[q][sx] = wnext[q][sx];
    pos = pos == 2 ? 0 : pos + 1;
  };

  for (int kb = kb0; kb < kb1; ++kb) {
    step(std::integral_constant<int, 0>{}, kb);
    step(std::integral_constant<int, 1>{}, kb);
  }
  vg_wait<0>();                                    // nothing of this wave may still be writing LDS when the workgroup retires
#undef VGW_WAIT

  // ---- epilogue (gemm.hip's): fp32 partials [S][T][N], or SiLU(gate) * up in the model dtype (S = 1, rows interleaved)
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int tok = t0 + t * 16 + j;
    if (tok >= tend) continue;
    float inv = 1.f;
    if (fx.ssq_in) {
      const float* sq = fx.ssq_in + (long)tok * (fx.ssq_stride ? fx.ssq_stride : fx.ssq_groups);
      float a = 0.f;
      for (int q = 0; q < fx.ssq_groups; q += 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(sq + q); a += v[0]; a += v[1]; a += v[2]; a += v[3]; }
      inv = rsqrtf(a / fx.ssq_dim + fx.eps);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 v = acc[q][t];
      const int ntile = nb * 16 + wv8 * 2 + q;
      if (epi == EPI_SILU) {
        v *= inv;
        const float g0 = rnd<P>(v[0]), u0 = rnd<P>(v[1]), g1 = rnd<P>(v[2]), u1 = rnd<P>(v[3]);
        const float a0 = rnd<P>(g0 / (1.f + __expf(-g0))) * u0, a1 = rnd<P>(g1 / (1.f + __expf(-g1))) * u1;
        u16* act = reinterpret_cast<u16*>(out);
        *reinterpret_cast<unsigned*>(act + (long)tok * (N / 2) + ntile * 8 + g * 2) = pack2<P>(a0, a1);
      } else {
        if (epi == EPI_ROUND) { v *= inv; v[0] = rnd<P>(v[0]); v[1] = rnd<P>(v[1]); v[2] = rnd<P>(v[2]); v[3] = rnd<P>(v[3]); }
        *reinterpret_cast<f32x4*>(out + ((long)sp * T + tok) * N + ntile * 16 + g * 4) = v;
      }
    }
  }
}

// ------------------------------------------------------------------ host side
// Work split of a wide launch: NT = ceil(T / 16) token tiles in `nchunk` chunks of TT = ceil(NT / nchunk) <= 18 tiles.
static inline int vgw_ver() {
  static const int v = getenv("UMB_VGW") ? atoi(getenv("UMB_VGW")) : 1;
  return v;
}
// The chunk count also sets how well the grid of (N / 256) x nchunk x S workgroups fills the 256 CUs (one workgroup per CU): one
// chunk more than the minimum is taken when it turns a ragged last round into whole rounds (T = 769, o / down at S = 2:
// 32 x 3 x 2 = 192 workgroups = 75 % of the chip, 32 x 4 x 2 = 256 = all of it), at the price of one more pass over the weights
// and shorter dequant amortisation (UMB_VGW_FILL=0: always the minimum).  A function of (T, N, S) only.
static inline void vgw_shape(int T, int N, int S, int* nchunk, int* tt) {
  const int NT = (T + 15) / 16;
  const int nc0 = (NT + 17) / 18;
  static const bool fill = getenv("UMB_VGW_FILL") == nullptr || atoi(getenv("UMB_VGW_FILL")) != 0;
  int best = nc0;
  if (fill) {
    double best_cost = 1e30;
    for (int nc = nc0; nc <= nc0 + 1 && nc <= NT; ++nc) {
      const long wgs = (long)(N / 256) * nc * S;
      const long rounds = (wgs + 255) / 256;
      const int ttc = (NT + nc - 1) / nc;
      if (nc > nc0 && ttc < 5) break;                // the kernel is instantiated for 5 ... 18 token tiles
      // time ~ rounds x (tiles per chunk + a fixed per-workgroup cost worth ~1.5 tiles: prologue, epilogue, weight re-read)
      const double cost = (double)rounds * (ttc + 1.5);
      if (cost < best_cost - 1e-9) { best_cost = cost; best = nc; }
    }
  }
  *nchunk = best;
  *tt = (NT + best - 1) / best;
}

extern "C" int umb_vgemm_w_ok(int T, int N, int K, int S, int epi) {
  if (vgw_ver() == 0 || T <= 64 || N % 256 || K % 128 || epi > EPI_SILU || S < 1 || S > K / 128) return 0;
  int nc, tt;
  vgw_shape(T, N, S, &nc, &tt);
  if (tt < 5) return 0;
  // small layers keep the finer-grained kernels of gemm.hip (UMB_VGW_MIN_WGS, read per call: tests drive every instantiation of
  // this kernel on small matrices)
  const char* mw = getenv("UMB_VGW_MIN_WGS");
  return (N / 256) * nc * S >= (mw ? atoi(mw) : 128);
}

template <int TT>
static int vgw_launch(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int N, int K, int S, int epi,
                      int nchunk, const GemmFused& fx, hipStream_t st) {
  constexpr size_t smem = (size_t)3 * TT * 2048 + 8 * 2 * VGW_WSTAGE + 1024;
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_done[dev]) {
    if (hipFuncSetAttribute((const void*)vgemm_w_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return UMB_EHIP;
    attr_done[dev] = true;
  }
  const unsigned grid = (unsigned)((N / 256) * nchunk * S);
  hipLaunchKernelGGL((vgemm_w_kernel<TT>), dim3(grid), dim3(512), smem, st, (const u32x4*)wp, (const unsigned char*)meta, x, ldx,
                     T, T, N, K, S, epi, nchunk, out, fx);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// fp16 activations, AWQ int4 weights in gemm.hip's tile order.  Caller (gemm.hip launch_tt) checked umb_vgemm_w_ok.
int umb_vgemm_w(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int N, int K, int S, int epi,
                const GemmFused& fx, hipStream_t st) {
  int nc, tt;
  vgw_shape(T, N, S, &nc, &tt);
  switch (tt) {
#define VGW_CASE(n) case n: return vgw_launch<n>(wp, meta, x, ldx, out, T, N, K, S, epi, nc, fx, st)
    VGW_CASE(5); VGW_CASE(6); VGW_CASE(7); VGW_CASE(8); VGW_CASE(9); VGW_CASE(10); VGW_CASE(11); VGW_CASE(12); VGW_CASE(13);
    VGW_CASE(14); VGW_CASE(15); VGW_CASE(16); VGW_CASE(17); VGW_CASE(18);
#undef VGW_CASE
  }
  return UMB_EINVAL;
}
