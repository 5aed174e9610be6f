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
#include <cstring>
#include <cstdio>
#include <type_traits>
#include <cstdlib>
#ifndef UMB_CB1
#define UMB_CB1 2        // k-blocks per LDS chunk at <= 16 tokens: 2 -> 4-stage weight ring, ~2x fewer registers, 3-4 waves per SIMD
#endif

// ------------------------------------------------------------------ repack (load time)
// packed row n <- source row rowmap(n):
//   mode 1 ([gate; up] stack): rows (2m, 2m+1) <- (m, N/2 + m)            -> SiLU(gate)*up in the GEMM epilogue
//   mode 2 ([q | k | v] stack, rope_heads = Hq + Hkv leading heads of size D): inside every q / k head
//          rows (2m, 2m+1) <- (m, m + D/2)                                  -> rotate-half RoPE in the GEMM epilogue
__host__ __device__ __forceinline__ int rowmap(int n, int N, int mode, int D, int rope_heads) {
  if (mode == 1) return (n & 1) ? N / 2 + (n >> 1) : (n >> 1);
  if (mode == 2) {
    const int head = n / D, dp = n % D;
    if (head >= rope_heads) return n;
    return head * D + ((dp & 1) ? (dp >> 1) + D / 2 : (dp >> 1));
  }
  return n;
}

template <typename P>
__global__ void repack_dense_kernel(const u16* __restrict__ W, u32x4* __restrict__ out, int N, int K, int il, int D,
                                    int rope_heads) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int KT = K / 32;
  const long total = (long)(N / 16) * KT * 64;
  if (gid >= total) return;
  const int lane = gid & 63;
  const long tile = gid >> 6;
  const int kt = tile % KT, nt = tile / KT;
  const int i = lane & 15, g = lane >> 4;
  const int src = rowmap(nt * 16 + i, N, il, D, rope_heads);
  out[gid] = *reinterpret_cast<const u32x4*>(W + (long)src * K + kt * 32 + g * 8);
}

// AutoAWQ GEMM format -> tile order.  qweight [K][N/8] int32, nibble idx of word c
// holds column 8c + ORDER[idx], ORDER = {0,2,4,6,1,3,5,7}  (inverse: INV below).
__global__ void repack_awq_kernel(const unsigned* __restrict__ qweight, const unsigned* __restrict__ qzeros,
                                  const u16* __restrict__ scales, u32x4* __restrict__ outw,
                                  unsigned char* __restrict__ meta, int N, int K, int il, int D, int rope_heads) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int KG = K / 128;
  const long total = (long)(N / 16) * KG * 64;
  if (gid >= total) return;
  const int lane = gid & 63;
  const long tile = gid >> 6;
  const int kg = tile % KG, nt = tile / KG;
  const int i = lane & 15, g = lane >> 4;
  const int nlog = nt * 16 + i;
  const int n = rowmap(nlog, N, il, D, rope_heads);   // source column
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
// Weight tiles go straight from HBM to VGPRs (non-temporal buffer loads, a ring of 2 chunks); int4 metadata rides one
// half-wave load per chunk through a wave-private LDS slot.  No load of the main loop is predicated (see the kernel).
// AWQ modes: 0 dense 16-bit weights; 1 int4, dequant folded out of the MFMA (any activation dtype; the default for
// launches of <= 64 rows); 2 int4 with exact fp16 dequant in registers, W = fp16((q - z) * s) -- bit-identical to what
// awq_ext.dequantize_weights_cuda produces -- via v_pk_add_f16 / v_pk_mul_f16 (fp16 activations; wide launches).
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// (a & mask) | magic in ONE VALU op.  gfx9 VOP3 may read a single SGPR/constant, so hipcc splits the pattern into
// v_and + v_or when both constants live in SGPRs; keeping `magic` in a VGPR makes v_and_or_b32 encodable.
__device__ __forceinline__ unsigned and_or(unsigned a, unsigned mask, unsigned magic_v) {
  unsigned r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(magic_v));
  return r;
}

template <typename P, int AWQ, int R> struct Stage {
  u32x4 a[R][AWQ ? 1 : 4];
  u32x4 m4[AWQ == 1 ? R : 1];     // folded path: 4 x {scale, zero} for output rows g*4 .. g*4+3
  unsigned m1[AWQ == 2 ? R : 1];  // exact path : {scale, zero} of this lane's weight row
};

// xf: LDS fragments of this 128-k block, index (tt*4 + s)*64 + lane.
// The weight fragments are built once per 128-k block (dequant for int4), then applied to every token tile:
// at TT = 16 (256 tokens per launch, the multi-token verify) that is 64 MFMAs per KiB of weights -> matrix-pipe
// bound, the weights are read once per 256 tokens.
template <typename P, int AWQ, int TT, int R>
__device__ __forceinline__ void stage_compute(const Stage<P, AWQ, R>& st, const u32x4* xf, int lane,
                                              f32x4 (&acc)[R][TT]) {
  if (AWQ == 1) {
    // folded dequant (any activation dtype): the MFMA runs on the raw codes under ONE magic exponent,
    //   out += s * ( sum_k (C + q_k) x_k - (C + z) * sum_k x_k )        per 128-k group,
    // sum_k x_k from one MFMA chain against a constant fragment of ones (shared by the R tiles).  fp16: every nibble is
    // moved to mantissa bits 4..7 under 0x5400 (C = 64: 3 shifts + 4 v_and_or per 8 weights); bf16 (7 mantissa bits):
    // bits 0..3 under 0x4300 (C = 128).  On gfx950 the packed-fp16 and three-operand integer VALU ops issue at HALF the
    // rate of plain VOP2 ops (scripts/probe/valu_probe.hip: 1.9 vs 1.0 ns per wave instruction), so the exact
    // in-register dequant of AWQ == 2 costs 95 ns of VALU per KiB tile against 160 ns of HBM time -- too much to hide;
    // this form costs ~60 ns (unpack 4 x 10.6 + the fp32 step 8 mixed-precision FMAs).
    constexpr bool HALF = std::is_same<P, F16>::value;
    constexpr float COFF = HALF ? 64.f : 128.f;
    u32x4 b[TT][4];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[tt][s] = xf[(tt * 4 + s) * 64 + lane];
    const u32x4 ones = {P::ONE2, P::ONE2, P::ONE2, P::ONE2};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    unsigned magic = HALF ? 0x54005400u : P::MAGIC;
    asm volatile("" : "+v"(magic));                                  // pinned in a VGPR: (w & mask) | magic selects v_and_or_b32
    f32x4 xs[TT], xc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      xs[tt] = zero;
#ifdef UMB_ABL_NOXS       // ablation (wrong results): no sum-of-x MFMA chain
      xs[tt] = __builtin_bit_cast(f32x4, b[tt][0]);
#else
#pragma unroll
      for (int s = 0; s < 4; ++s) xs[tt] = P::mfma(ones, b[tt][s], xs[tt]);
#endif
      xc[tt] = xs[tt] * (-COFF);                                      // C-in of the code chains: - C sum_k x_k
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      f32x4 ga[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) ga[tt] = xc[tt];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned w = st.a[r][0][s];
        u32x4 f;
#ifdef UMB_ABL_NOUNPACK   // ablation (wrong results): raw dwords as the operand
        f[0] = w; f[1] = w ^ magic; f[2] = w + magic; f[3] = w | magic;
        if (false)
#endif
        if constexpr (HALF) {
          f[0] = ((w << 4) & 0x00F000F0u) | magic;
          f[1] = (w & 0x00F000F0u) | magic;
          f[2] = ((w >> 4) & 0x00F000F0u) | magic;
          f[3] = ((w >> 8) & 0x00F000F0u) | magic;
        } else {
          f[0] = (w & 0x000F000Fu) | magic;
          f[1] = ((w >> 4) & 0x000F000Fu) | magic;
          f[2] = ((w >> 8) & 0x000F000Fu) | magic;
          f[3] = ((w >> 12) & 0x000F000Fu) | magic;
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) ga[tt] = P::mfma(f, b[tt][s], ga[tt]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // ga = sum_k q_k x_k.  {scale, zero} are fp16 in the metadata whatever the activation dtype: two mixed-precision
        // FMAs per output (v_fma_mix_f32 reads the halves in place); the empty asm keeps the SLP vectoriser from pairing
        // them into v_pk_fma_f32 behind eight conversions
        const _Float16 sc = __builtin_bit_cast(_Float16, (u16)(st.m4[r][e] & 0xffffu));
        const _Float16 zf = __builtin_bit_cast(_Float16, (u16)(st.m4[r][e] >> 16));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
#ifdef UMB_ABL_NOFIX      // ablation (wrong results): no fp32 scale / zero step
          acc[r][tt][e] += ga[tt][e];
          continue;
#endif
          float t = __builtin_fmaf(-(float)zf, xs[tt][e], ga[tt][e]);
          asm("" : "+v"(t));
          acc[r][tt][e] = __builtin_fmaf((float)sc, t, acc[r][tt][e]);
        }
      }
    }
    return;
  }
  // ---- weight fragments wf[r][s]
  u32x4 wf[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (AWQ == 2) {
      const unsigned mm = st.m1[r];
      const _Float16 sc = __builtin_bit_cast(_Float16, (u16)(mm & 0xffffu));
      const _Float16 zf = __builtin_bit_cast(_Float16, (u16)(mm >> 16));
      const h2 s2 = {sc, sc};
      const _Float16 nz = -((_Float16)1024.0f + zf);          // exact: |1024 + z| <= 1039
      const _Float16 nz16 = -((_Float16)64.0f + zf);
      const h2 nz2 = {nz, nz}, nz16_2 = {nz16, nz16};
      const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
      unsigned magic = 0x64006400u;
      asm volatile("" : "+v"(magic));                       // pin the magic constant in a VGPR
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned w = st.a[r][0][s];
        const unsigned w8 = w >> 8;
        // nibbles at mantissa bits 0..3 give fp16(1024 + q); at bits 4..7 fp16(1024 + 16 q): one shift per
        // dword instead of three.  (q - z) is exact in fp16 either way, then one rounding in (q - z) * s.
        const h2 t0 = __builtin_bit_cast(h2, and_or(w, 0x000F000Fu, magic));
        const h2 t1 = __builtin_bit_cast(h2, and_or(w, 0x00F000F0u, magic));
        const h2 t2 = __builtin_bit_cast(h2, and_or(w8, 0x000F000Fu, magic));
        const h2 t3 = __builtin_bit_cast(h2, and_or(w8, 0x00F000F0u, magic));
        wf[r][s][0] = __builtin_bit_cast(unsigned, (t0 + nz2) * s2);
        wf[r][s][1] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(t1, sixteenth, nz16_2) * s2);
        wf[r][s][2] = __builtin_bit_cast(unsigned, (t2 + nz2) * s2);
        wf[r][s][3] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(t3, sixteenth, nz16_2) * s2);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) wf[r][s] = st.a[r][s];
    }
  }
  // ---- token tiles: 4 conflict-free ds_read_b128 + 4 R MFMAs each
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    u32x4 b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      b[s] = xf[(tt * 4 + s) * 64 + lane];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        acc[r][tt] = P::mfma(wf[r][s], b[s], acc[r][tt]);
      }
  }
}

#include "gemm_fused.h"

// -DUMB_GEMM_TRACE (scripts/r3/gemm_trace.py builds a second library with it): wave 0 of every block stamps the constant
// 100 MHz clock at its phase boundaries into fx.counters (16 x u64 per block; unused by the direct epilogues): 0 entry,
// 1 prologue loads issued, 2 first x chunk staged, 3 first k-block computed, 4 main loop left, 5 stores drained,
// 6.. chunk c entered (c < 10).
#ifdef UMB_GEMM_TRACE
#define UMB_STAMP(i) do { if (threadIdx.x == 0 && fx.counters) \
    reinterpret_cast<unsigned long long*>(fx.counters)[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define UMB_STAMP(i) do {} while (0)
#endif

template <typename P, int AWQ, int TT, int R, int CB, int NWV = 4>
__global__ __launch_bounds__(64 * NWV) void skinny_gemm_kernel(const u32x4* __restrict__ wp,
                                                          const unsigned char* __restrict__ meta,
                                                          const u16* __restrict__ x, int ldx, int T, int N, int K,
                                                          int per, int tb, int Ttot, int epi_flags,
                                                          float* __restrict__ out, int S, GemmFused fx) {
  // Argument order: the first 14 dwords are preloaded into SGPRs at wave launch (-amdgpu-kernarg-preload-count): they
  // hold everything the weight / activation streams need, so the first loads do not wait for a kernarg fetch.
  // Grid = (n-tile groups, K splits) and `per` = k-blocks per split come from the host: no integer division here (three
  // of them -- each a VALU reciprocal chain read back through v_readfirstlane -- stood in front of the first load).
  // epi_flags = epi | x in FM layout << 8 | SiLU output in FM layout << 9.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int epi = epi_flags & 0xff;
  const bool x_fm = (epi_flags & 0x100) != 0, out_fm = (epi_flags & 0x200) != 0;
  UMB_STAMP(0);
  u32x4* xs = reinterpret_cast<u32x4*>(smem);
  constexpr int F = CB * TT * 4;           // 1 KiB fragments per chunk
  static_assert(F % NWV == 0, "every wave stages the same number of activation fragments");
  constexpr int FPW = F / NWV;             // fragments staged per wave (NWV waves per block: 4, or 8 = one block per CU
                                           // whose waves share ONE staged copy of the activations instead of two)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int NT = N / 16;
  // tb n-tiles per block (4 R by default).  One CU streams ~25 GB/s whatever runs on it, so a launch is as fast as its
  // busiest CU: tb is chosen by the host so that nblk * S blocks are whole rounds of 2 per CU (70B gate/up: 3584 tiles =
  // 512 x 7, not 448 x 8 -- the 64 CUs with one block sat idle half the launch).  With tb < 4 R the last wave slots of a
  // block are empty; pairs (R = 2) start at even GLOBAL tile indices, so the two tiles of a wave stay adjacent in the
  // [N/64][K/128][4] tile order and in the metadata (a block whose first tile is odd gives wave 0 a single tile).
  const int sp = blockIdx.y;
  const int nb = blockIdx.x;
  const int first = nb * tb;
  const int odd = (R == 2) ? (first & 1) : 0;
  const int wstart = first + wv * R - odd;
  const int nt0 = max(wstart, first);                                   // this wave's first tile
  const int ntiles = max(0, min(min(wstart + R, first + tb), NT) - nt0);  // 0 .. R
  const bool active = ntiles > 0;
  const int KB = K / 128;
  const int kb0 = sp * per;
  const int kb1 = min(KB, kb0 + per);
  const int nchunks = (kb1 - kb0 + CB - 1) / CB;

  f32x4 acc[R][TT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[r][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkb = kb1 - kb0;
  constexpr int PF = 2 * CB;                 // ring of PF weight stages: PF x R KiB-tiles in flight per wave
  // ---- streams.  Every load of the main loop is an unconditional buffer load: a k-block past the end of the slab, a
  // token row past T or the whole stream of a wave without tiles is OUT OF RANGE of its descriptor -> zeros, no fetch.
  // No load sits behind a branch or an exec mask, so the compiler's vmcnt bookkeeping stays exact and the ring really
  // stays PF stages deep (with predicated loads it drained the counter to zero at every chunk: the weight stream ran
  // one chunk deep, 4.1 TB/s where the same access pattern without compute streams 6.4 TB/s).
  __amdgpu_buffer_rsrc_t rw[R], rm[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int nt = r < ntiles ? nt0 + r : 0;
    const bool live = r < ntiles && nkb > 0;
    if (AWQ) {
      const long tile0 = ((long)(nt >> 2) * KB) * 4 + (nt & 3);        // tile order [N/64][K/128][4]
      rw[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wp + tile0 * 64), 0,
                                                live ? (unsigned)(kb1 - 1) * 4096u + 1024u : 0u, 0x00020000);
      rm[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(meta + tile0 * 64), 0,
                                                live ? (unsigned)(kb1 - 1) * 256u + 64u : 0u, 0x00020000);
    } else {
      rw[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wp + ((long)nt * KB) * 256), 0,
                                                live ? (unsigned)kb1 * 4096u : 0u, 0x00020000);
      rm[r] = rw[r];
    }
  }
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u16*>(x), 0, x_fm ? (unsigned)(TT * 16 * 2) * (unsigned)K : (unsigned)(((long)(T - 1) * ldx + K) * 2),
      0x00020000);
  int voffx[FPW];
#pragma unroll
  for (int i = 0; i < FPW; ++i) {
    const int tok = (((i * NWV + wv) >> 2) % TT) * 16 + j;              // token tile of fragment f = i * NWV + wv
    voffx[i] = x_fm ? lane * 16 : (tok < T ? (int)(((long)tok * ldx + g * 8) * 2) : (int)0x80000000);
  }
  // int4 metadata ({scale, zero} per row per k-block, 64 B per tile): ONE half-wave load per chunk brings the wave's
  // R tiles x CB k-blocks (16 B per lane), staged through a wave-private LDS slot and read back per k-block --
  // instead of R x CB separate 64-byte loads (they cost 3.5 us of a 55 us gate/up launch: 6 % of the bytes).
  constexpr int XD = (TT == 1) ? 2 : 1;              // x register ring: chunks ahead (TT > 1: registers)
  constexpr int MSLOT = 1024;                        // staged metadata per wave per ring half (64 lanes x 16 B)
  unsigned char* ms = smem + (size_t)2 * F * 1024 + wv * 2 * MSLOT;
  __amdgpu_buffer_rsrc_t rmeta = rm[0];
  int voff_m = (int)0x80000000;
  if (AWQ) {
    const int nt = active ? nt0 : 0;
    const long tile0 = ((long)(nt >> 2) * KB) * 4 + (nt & 3);
    rmeta = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(meta + tile0 * 64), 0,
                                              (active && nkb > 0) ? (unsigned)(kb1 - 1) * 256u + ntiles * 64u : 0u, 0x00020000);
    if (lane < R * CB * 4) voff_m = (lane / (R * 4)) * 256 + (lane % (R * 4)) * 16;
  }
  u32x4 xr[XD][FPW];
  u32x4 mr[2];
  Stage<P, AWQ, R> st[PF];
  auto sload = [&](Stage<P, AWQ, R>& sg, int kb) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (AWQ) {
        // the k offset rides in the VGPR offset: that is the part of the address the range check certainly covers
        sg.a[r][0] = __builtin_amdgcn_raw_buffer_load_b128(rw[r], lane * 16 + kb * 4096, 0, 2);   // non-temporal
      } else {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
          sg.a[r][AWQ ? 0 : s2] = __builtin_amdgcn_raw_buffer_load_b128(rw[r], lane * 16 + kb * 4096 + s2 * 1024, 0, 2);
      }
    }
  };
  auto load_m = [&](int c) { return __builtin_amdgcn_raw_buffer_load_b128(rmeta, voff_m + (kb0 + c * CB) * 256, 0, 0); };
  auto load_x = [&](u32x4 (&xq)[FPW], int c) {
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const int f = i * NWV + wv;
      const int kb = kb0 + c * CB + f / (TT * 4);
      // FM: the B fragment of (k32-step, token tile) is one contiguous 1 KiB; row-major: 16 bytes of row tok at k
      const int soff = x_fm ? ((kb * 4 + (f & 3)) * TT + ((f >> 2) % TT)) * 1024 : (kb * 128 + (f & 3) * 32) * 2;
      xq[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, voffx[i] + soff, 0, 0);
    }
  };
  auto store_x = [&](const u32x4 (&xq)[FPW], int c) {
#pragma unroll
    for (int i = 0; i < FPW; ++i) xs[((c & 1) * F + i * NWV + wv) * 64 + lane] = xq[i];
  };
  const int nfull = nkb > 0 ? nkb / CB : 0;  // chunks whose k-blocks all lie inside the slab
  auto chunk = [&](int c, auto half, auto guard) {
    constexpr int H = decltype(half)::value;           // which half of the ring this chunk uses
    constexpr bool G = decltype(guard)::value;         // ragged tail: skip the k-blocks past the slab (compute only)
    constexpr int XH = XD == 2 ? H : 0;
    // EARLY (<= 16 rows: x rides two chunks ahead): chunk c's activations and metadata were written to LDS in the middle
    // of chunk c - 1 (below), so nothing but the barrier stands between two chunks -- the ds_write -> lgkmcnt(0) latency
    // sat on the critical path of every chunk of every wave of the block (64 chunks in a 70B gate/up launch)
#ifdef UMB_NO_EARLY
    constexpr bool EARLY = false;
#else
    constexpr bool EARLY = XD == 2;
#endif
    if constexpr (!EARLY) {
      store_x(xr[XH], c);
      if (AWQ) *reinterpret_cast<u32x4*>(ms + H * MSLOT + lane * 16) = mr[H];
    }
    __syncthreads();
    if (c == 0) UMB_STAMP(2);
#ifdef UMB_GEMM_TRACE
    if (c < 10) UMB_STAMP(6 + c);
#endif
    // everything loaded from here on is for chunk c + 2 (x of TT > 1: c + 1), in the order it will be consumed
    load_x(xr[XH], c + XD);
    if (AWQ) mr[H] = load_m(c + 2);
    const u32x4* xc = xs + (c & 1) * F * 64;
#pragma unroll
    for (int kl = 0; kl < CB; ++kl) {
      const int kb = kb0 + c * CB + kl;
      Stage<P, AWQ, R>& sg = st[H * CB + kl];
      if constexpr (EARLY) {
        if (kl == CB - 1) {          // next chunk's operands: buffers (c + 1) & 1 were last read before this chunk's barrier
          store_x(xr[1 - XH], c + 1);
          if (AWQ) *reinterpret_cast<u32x4*>(ms + (1 - H) * MSLOT + lane * 16) = mr[1 - H];
        }
      }
      if (!G || kb < kb1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const unsigned char* mrec = ms + H * MSLOT + (kl * R + r) * 64;
          if (AWQ == 1) sg.m4[r] = *reinterpret_cast<const u32x4*>(mrec + g * 16);
          if (AWQ == 2) sg.m1[r] = *reinterpret_cast<const unsigned*>(mrec + j * 4);
        }
        stage_compute<P, AWQ, TT, R>(sg, xc + kl * TT * 4 * 64, lane, acc);
#ifdef UMB_GEMM_TRACE
        if (c == 0 && kl == 0) { asm volatile("" :: "v"(acc[0][0][0])); UMB_STAMP(3); }
#endif
      }
      sload(sg, kb + PF);
    }
  };
  // prologue, in consumption order: chunk 0 (x, metadata, weights), then chunk 1
  load_x(xr[0], 0);
  if (AWQ) mr[0] = load_m(0);
#pragma unroll
  for (int i = 0; i < CB; ++i) sload(st[i], kb0 + i);
  if (XD == 2) load_x(xr[XD - 1], 1);
  if (AWQ) mr[1] = load_m(1);
#pragma unroll
  for (int i = CB; i < PF; ++i) sload(st[i], kb0 + i);
  UMB_STAMP(1);
  // per-token 1/rms of the producer's residual stream (its loads overlap the weight stream)
  float inv[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    inv[tt] = 1.f;
    const int tok = tt * 16 + j;
    if (fx.ssq_in && tok < T) {
      const float* sq = fx.ssq_in + (long)tok * (fx.ssq_stride ? fx.ssq_stride : fx.ssq_groups);
      float a = 0.f;
      int q = 0;
      for (; q + 16 <= fx.ssq_groups; q += 16) {            // four loads per round trip (the GEMV schedule leaves 256 groups);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sq + q), v1 = *reinterpret_cast<const f32x4*>(sq + q + 4);   // same order
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(sq + q + 8), v3 = *reinterpret_cast<const f32x4*>(sq + q + 12);
        a += v0[0]; a += v0[1]; a += v0[2]; a += v0[3]; a += v1[0]; a += v1[1]; a += v1[2]; a += v1[3];
        a += v2[0]; a += v2[1]; a += v2[2]; a += v2[3]; a += v3[0]; a += v3[1]; a += v3[2]; a += v3[3];
      }
      for (; q < fx.ssq_groups; q += 4) {                    // groups are a multiple of 4 or padded with zeros
        const f32x4 v = *reinterpret_cast<const f32x4*>(sq + q);
        a += v[0]; a += v[1]; a += v[2]; a += v[3];
      }
      inv[tt] = rsqrtf(a / fx.ssq_dim + fx.eps);
    }
  }
#ifndef UMB_NO_EARLY
  if constexpr (XD == 2) {                             // EARLY: chunk 0's operands go to LDS here
    store_x(xr[0], 0);
    if (AWQ) *reinterpret_cast<u32x4*>(ms + lane * 16) = mr[0];
  }
#endif
  {
    int c = 0;
    for (; c + 2 <= nfull; c += 2) {                   // steady state: no branch, no predicate
      chunk(c, std::integral_constant<int, 0>{}, std::false_type{});
      chunk(c + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }
    for (; c < nchunks; c += 2) {                      // ragged tail of the slab
      chunk(c, std::integral_constant<int, 0>{}, std::true_type{});
      if (c + 1 < nchunks) chunk(c + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
  }

  // ------------------------------------------------------------------ direct epilogues (no cross-block step)
  UMB_STAMP(4);
  if (epi <= EPI_SILU) {
    if (!active) return;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int tok = tt * 16 + j;
      if (tok < T) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r >= ntiles) continue;
          f32x4 v = acc[r][tt];
          if (epi == EPI_SILU) {
            // rows are interleaved (gate_m, up_m): act[tok][m] = silu(gate) * up, every step rounded to the
            // model dtype as eager torch does (umbrella/models/llama.py:107-110).  out is 16-bit [Ttot][N/2].
            v *= inv[tt];
            const float g0 = rnd<P>(v[0]), u0 = rnd<P>(v[1]), g1 = rnd<P>(v[2]), u1 = rnd<P>(v[3]);
            const float a0 = rnd<P>(g0 / (1.f + __expf(-g0))) * u0, a1 = rnd<P>(g1 / (1.f + __expf(-g1))) * u1;
            u16* act = reinterpret_cast<u16*>(out);
            const int fa = (nt0 + r) * 8 + g * 2;
            const long off = out_fm ? ((((long)(fa >> 5) * TT + (tok >> 4)) * 64 + ((fa >> 3) & 3) * 16 + (tok & 15)) << 3) + (fa & 7)
                                       : (long)tok * (N / 2) + fa;
            *reinterpret_cast<unsigned*>(act + off) = pack2<P>(a0, a1);
          } else {
            if (epi == EPI_ROUND) {
              v *= inv[tt];
              v[0] = rnd<P>(v[0]); v[1] = rnd<P>(v[1]); v[2] = rnd<P>(v[2]); v[3] = rnd<P>(v[3]);
            }
{   // write-through (sc1): the partials are read by the NEXT kernel on other XCDs -- nothing is left for the end-of-kernel
              // write-back to drain (70B layer at T = 13: -1.2 us, profiles/r05_handoff_stores.txt); same bits, another cache policy
              const auto rso = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rso, (int)((((long)sp * Ttot + tok) * N + (nt0 + r) * 16 + g * 4) * 4), 0, UMB_HANDOFF_AUX);
            }
          }
        }
      }
    }
#ifdef UMB_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UMB_STAMP(5);
#endif
    return;
  }

  // ------------------------------------------------------------------ split epilogues (R == 1): publish, last arriver finishes
  if (S > 1) {
    if (active) {
      // write-through (sc1) partial stores: no release fence / L2 write-back needed before the ticket
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int tok = tt * 16 + j;
        if (tok < T) {
          const long off = (((long)sp * Ttot + tok) * N + nt0 * 16 + g * 4) * 4;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[0][tt]), rsrc, (int)off, 0, 16);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);                  // LDS staging buffers are dead by now
    if (threadIdx.x == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(fx.counters + nb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == (unsigned)(S - 1);
      if (last) {
        __hip_atomic_store(fx.counters + nb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-reset for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    if (active) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int tok = tt * 16 + j;
        if (tok < T) {
          const float* p = out + (long)tok * N + nt0 * 16 + g * 4;
          f32x4 a = *reinterpret_cast<const f32x4*>(p);
          for (int s2 = 1; s2 < S; ++s2) a += *reinterpret_cast<const f32x4*>(p + (long)s2 * Ttot * N);   // fixed order 0..S-1
          acc[0][tt] = a;
        }
      }
    }
    __syncthreads();                                             // flag word is reused below as reduction scratch
  }

  if (epi == EPI_RESID) {
    // h <- round(round(gemm) + h) ; hw <- h * w_next (the next RMSNorm's weight, folded) ; ssq_out[t][nb] <- sum h^2
    float* red = reinterpret_cast<float*>(smem);                 // [4 waves][TT*16]
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int tok = tt * 16 + j;
      float sq = 0.f;
      if (active && tok < T) {
        const long off = (long)tok * N + nt0 * 16 + g * 4;
        const uint2 hr = *reinterpret_cast<const uint2*>(fx.h + off);
        const f32x4 v = acc[0][tt];
        const float h0 = rnd<P>(rnd<P>(v[0]) + lo_f<P>(hr.x)), h1 = rnd<P>(rnd<P>(v[1]) + hi_f<P>(hr.x));
        const float h2v = rnd<P>(rnd<P>(v[2]) + lo_f<P>(hr.y)), h3 = rnd<P>(rnd<P>(v[3]) + hi_f<P>(hr.y));
        uint2 o;
        o.x = pack2<P>(h0, h1); o.y = pack2<P>(h2v, h3);
        *reinterpret_cast<uint2*>(fx.h + off) = o;
        if (fx.norm_w) {
          const uint2 w = *reinterpret_cast<const uint2*>(fx.norm_w + nt0 * 16 + g * 4);
          uint2 ow;
          ow.x = pack2<P>(h0 * lo_f<P>(w.x), h1 * hi_f<P>(w.x)); ow.y = pack2<P>(h2v * lo_f<P>(w.y), h3 * hi_f<P>(w.y));
          *reinterpret_cast<uint2*>(fx.hw + off) = ow;
        }
        sq = h0 * h0 + h1 * h1 + h2v * h2v + h3 * h3;
      }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (g == 0) red[wv * (TT * 16) + tt * 16 + j] = sq;
    }
    __syncthreads();
    if (fx.ssq_out && threadIdx.x < TT * 16 && (int)threadIdx.x < T) {
      const int t = threadIdx.x;
      const float tot = ((red[t] + red[TT * 16 + t]) + red[2 * TT * 16 + t]) + red[3 * TT * 16 + t];
      fx.ssq_out[(long)t * fx.ssq_out_stride + nb] = tot;
    }
    return;
  }

  // EPI_QKV: rows were permuted at load so (2m, 2m+1) of a q/k head are RoPE partners (m, m + D/2)
  if (!active) return;
  {
    const int D = fx.D, half = D / 2;
    const int n = nt0 * 16 + g * 4;
    const int head = n / D, dp = n % D, m = dp >> 1;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int tok = tt * 16 + j;
      if (tok >= T) continue;
      f32x4 v = acc[0][tt];
      v *= inv[tt];
      const float a0 = rnd<P>(v[0]), b0 = rnd<P>(v[1]), a1 = rnd<P>(v[2]), b1 = rnd<P>(v[3]);
      const int sl = fx.slot[tok];
      if (head < fx.Hq + fx.Hkv) {
        const long cb = (long)fx.pos[tok] * D;
        const unsigned cl = *reinterpret_cast<const unsigned*>(fx.cosT + cb + m);
        const unsigned ch = *reinterpret_cast<const unsigned*>(fx.cosT + cb + m + half);
        const unsigned sl_ = *reinterpret_cast<const unsigned*>(fx.sinT + cb + m);
        const unsigned sh = *reinterpret_cast<const unsigned*>(fx.sinT + cb + m + half);
        // rotate-half in the model dtype: every product and the sum are rounded (eager torch, model_utils.py:50-51)
        const float lo0 = rnd<P>(mul_rnd<P>(a0, lo_f<P>(cl)) + mul_rnd<P>(-b0, lo_f<P>(sl_)));
        const float lo1 = rnd<P>(mul_rnd<P>(a1, hi_f<P>(cl)) + mul_rnd<P>(-b1, hi_f<P>(sl_)));
        const float hi0 = rnd<P>(mul_rnd<P>(b0, lo_f<P>(ch)) + mul_rnd<P>(a0, lo_f<P>(sh)));
        const float hi1 = rnd<P>(mul_rnd<P>(b1, hi_f<P>(ch)) + mul_rnd<P>(a1, hi_f<P>(sh)));
        // q rows are row-major; the K / V^T caches are in fragment order inside a head's slab (common.h: m is even, so the
        // pair (m, m + 1) stays one 4-byte store)
        u16* dlo = (head < fx.Hq) ? fx.q_out + ((long)tok * fx.Hq + head) * D + m
                                  : fx.kc + (long)(head - fx.Hq) * fx.Lmax * D + kc_off(sl, m, D);
        u16* dhi = (head < fx.Hq) ? dlo + half : fx.kc + (long)(head - fx.Hq) * fx.Lmax * D + kc_off(sl, m + half, D);
        *reinterpret_cast<unsigned*>(dlo) = pack2<P>(lo0, lo1);
        *reinterpret_cast<unsigned*>(dhi) = pack2<P>(hi0, hi1);
      } else {
        u16* vb = fx.vt + (long)(head - fx.Hq - fx.Hkv) * D * VT_LD(fx.Lmax);
        vb[vt_off(dp, sl, D)] = P::from_f(a0); vb[vt_off(dp + 1, sl, D)] = P::from_f(b0);
        vb[vt_off(dp + 2, sl, D)] = P::from_f(a1); vb[vt_off(dp + 3, sl, D)] = P::from_f(b1);
      }
    }
  }
}

// ------------------------------------------------------------------ verify GEMM (T > 64): matrix-pipe bound
// The multi-token verify of the dynamic (SpecExec) trees, T = 257 ... 769, is a true dense contraction: every
// weight is used by >= 257 tokens.  Block = 128 output rows x 128 tokens, 4 waves as 2 (rows) x 2 (tokens), each
// wave a 64 x 64 register tile = 4 x 4 MFMA 16x16x32 accumulators, so every LDS fragment read feeds 4 MFMAs
// (32 MFMAs per 16 ds_read_b128 per 64-k step).  Weight tiles are dequantised ONCE per block per step (wave w
// unpacks n-tiles 2w, 2w+1) and shared through LDS in fragment order; activations are staged the same way; both
// double buffered, one barrier per step.  Weights are streamed once per 128 tokens (later token chunks hit the
// 256 MB Infinity Cache: a 70B linear is 33-235 MB).
template <typename P, int AWQ>
__global__ __launch_bounds__(256) void verify_gemm_kernel(const u32x4* __restrict__ wp,
                                                          const unsigned char* __restrict__ meta,
                                                          const u16* __restrict__ x, int ldx, float* __restrict__ out,
                                                          int T, int Tv, int N, int K, int S, int epi, GemmFused fx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* sA = reinterpret_cast<u32x4*>(smem);                        // [2 buf][8 nt][2 s][64]
  u32x4* sB = sA + 2 * 8 * 2 * 64;                                   // [2 buf][8 tt][2 s][64]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wv >> 1, wt = wv & 1;                               // wave position in the 2 x 2 grid
  const int j = lane & 15, g = lane >> 4;
  const int nblk = N / 128;
  const int nchunk = (Tv + 127) / 128;                               // Tv: tokens handled here (T: row stride of out)
  const int nb = blockIdx.x % nblk;
  const int tc = (blockIdx.x / nblk) % nchunk;
  const int sp = blockIdx.x / (nblk * nchunk);
  const int KB = K / 128;
  const int per = (KB + S - 1) / S;
  const int ks0 = 2 * sp * per, ks1 = 2 * min(KB, sp * per + per);   // 64-k steps
  const int t0 = tc * 128;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave stages n-tiles nb*8 + 2*wv + {0,1} and activation fragments 4*wv .. 4*wv+3 of each step
  const u32x4* wbase[2];
  const unsigned char* mbase[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int nt = nb * 8 + 2 * wv + q;
    wbase[q] = AWQ ? wp + (((long)(nt >> 2) * KB) * 4 + (nt & 3)) * 64 : wp + ((long)nt * KB) * 256;
    mbase[q] = AWQ ? meta + (((long)(nt >> 2) * KB) * 4 + (nt & 3)) * 64 : nullptr;
  }
  u32x4 ra[2][AWQ ? 1 : 2];   // staged weights per n-tile: dense 2 tiles; AWQ {dword, dword, meta, -}
  u32x4 rb[4];                // staged activation fragments
  auto gload = [&](int ks) {
    const int kb = ks >> 1, hf = ks & 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (AWQ) {
        const unsigned* tile = reinterpret_cast<const unsigned*>(wbase[q] + (long)kb * 256 + lane);
        const uint2 w2 = *reinterpret_cast<const uint2*>(tile + hf * 2);
        ra[q][0][0] = w2.x; ra[q][0][1] = w2.y;
        ra[q][0][2] = *reinterpret_cast<const unsigned*>(mbase[q] + (long)kb * 256 + (lane & 15) * 4);
      } else {
        const u32x4* tile = wbase[q] + (long)kb * 256 + hf * 128 + lane;
        ra[q][0] = __builtin_nontemporal_load(tile);
        ra[q][AWQ ? 0 : 1] = __builtin_nontemporal_load(tile + 64);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = wv * 4 + i, tt = f >> 1, sx = f & 1;
      const int tok = t0 + tt * 16 + j;
      const u32x4 z = {0u, 0u, 0u, 0u};
      rb[i] = (tok < Tv) ? *reinterpret_cast<const u32x4*>(x + (long)tok * ldx + ks * 64 + sx * 32 + g * 8) : z;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      u32x4 f0, f1;
      if (AWQ) {
        const unsigned mm = ra[q][0][2];
        const _Float16 sc = __builtin_bit_cast(_Float16, (u16)(mm & 0xffffu));
        const _Float16 zf = __builtin_bit_cast(_Float16, (u16)(mm >> 16));
        const h2 s2 = {sc, sc};
        const _Float16 nz = -((_Float16)1024.0f + zf), nz16 = -((_Float16)64.0f + zf);
        const h2 nz2 = {nz, nz}, nz16_2 = {nz16, nz16};
        const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
        unsigned magic = 0x64006400u;
        asm volatile("" : "+v"(magic));
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const unsigned w = ra[q][0][sx], w8 = w >> 8;
          const h2 q0 = __builtin_bit_cast(h2, and_or(w, 0x000F000Fu, magic));
          const h2 q1 = __builtin_bit_cast(h2, and_or(w, 0x00F000F0u, magic));
          const h2 q2 = __builtin_bit_cast(h2, and_or(w8, 0x000F000Fu, magic));
          const h2 q3 = __builtin_bit_cast(h2, and_or(w8, 0x00F000F0u, magic));
          u32x4& f = sx ? f1 : f0;
          f[0] = __builtin_bit_cast(unsigned, (q0 + nz2) * s2);
          f[1] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(q1, sixteenth, nz16_2) * s2);
          f[2] = __builtin_bit_cast(unsigned, (q2 + nz2) * s2);
          f[3] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(q3, sixteenth, nz16_2) * s2);
        }
      } else {
        f0 = ra[q][0]; f1 = ra[q][AWQ ? 0 : 1];
      }
      sA[((buf * 8 + 2 * wv + q) * 2 + 0) * 64 + lane] = f0;
      sA[((buf * 8 + 2 * wv + q) * 2 + 1) * 64 + lane] = f1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sB[(buf * 16 + wv * 4 + i) * 64 + lane] = rb[i];
  };

  if (ks0 < ks1) {
    gload(ks0);
    sstore(0);
  }
  __syncthreads();
  for (int ks = ks0; ks < ks1; ++ks) {
    const int buf = (ks - ks0) & 1;
    if (ks + 1 < ks1) gload(ks + 1);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      u32x4 a[4], b[4];
#pragma unroll
      for (int n4 = 0; n4 < 4; ++n4) a[n4] = sA[((buf * 8 + wn * 4 + n4) * 2 + sx) * 64 + lane];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) b[t4] = sB[(buf * 16 + (wt * 4 + t4) * 2 + sx) * 64 + lane];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) acc[n4][t4] = P::mfma(a[n4], b[t4], acc[n4][t4]);
    }
    if (ks + 1 < ks1) sstore(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    const int tok = t0 + (wt * 4 + t4) * 16 + j;
    if (tok >= Tv) continue;
    float inv = 1.f;
    if (fx.ssq_in) {                                                 // per-token 1/rms (fused RMSNorm consumers only)
      const float* sq = fx.ssq_in + (long)tok * (fx.ssq_stride ? fx.ssq_stride : fx.ssq_groups);
      float a = 0.f;
      for (int q = 0; q < fx.ssq_groups; q += 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(sq + q); a += v[0]; a += v[1]; a += v[2]; a += v[3]; }
      inv = rsqrtf(a / fx.ssq_dim + fx.eps);
    }
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) {
      f32x4 v = acc[n4][t4];
      const int ntile = nb * 8 + wn * 4 + n4;
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

// ------------------------------------------------------------------ verify GEMM, weights never in LDS (int4 x fp16, dense 16-bit)
// LDS delivers ~64 B / clock / CU for ds_read_b128 on this part, and the 2 x 2 kernel above moves 0.75 KiB of LDS data
// per MFMA (fragment reads + dequantised-weight and activation stores): a third of the matrix rate at best, which is
// what every LDS-sharing variant measured.  Here block = 256 output rows x 128 tokens, 4 waves stacked along the
// rows: wave w owns 64 rows (4 n-tiles) x all 8 token tiles = 4 x 8 accumulators.  Its int4 tiles go HBM ->
// registers -> exact fp16 dequant -> MFMA A operand, never through LDS; only the activation fragments are shared
// (16 ds_read_b128 per 64 MFMAs = 0.31 KiB per MFMA with the stores).  <= 256 registers so two blocks share a CU:
// one wave's dequant and LDS waits hide behind the other's MFMAs.
// TT = token tiles per block: 8 (128 tokens) or 9 (144 tokens: T = w d + 1 = 257, 385, 769 split into whole chunks and
// the one-token tail launch, a full extra pass over the weights, disappears).
template <typename P, int AWQ, int TT>
__global__ __launch_bounds__(256, 2) void verify_gemm_r_kernel(const u32x4* __restrict__ wp,
                                                               const unsigned char* __restrict__ meta,
                                                               const u16* __restrict__ x, int ldx,
                                                               float* __restrict__ out, int T, int Tv, int N, int K,
                                                               int S, int epi, GemmFused fx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* sB = reinterpret_cast<u32x4*>(smem);                        // [2 buf][TT][2 s][64]   (64-k steps)
  constexpr int NF = TT * 2;                                         // activation fragments per step
  constexpr int FPW = (NF + 3) / 4;                                  // staged per wave
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int nblk = N / 256;
  const int nchunk = (Tv + TT * 16 - 1) / (TT * 16);
  const int nb = blockIdx.x % nblk;
  const int tc = (blockIdx.x / nblk) % nchunk;
  const int sp = blockIdx.x / (nblk * nchunk);
  const int KB = K / 128;
  const int per = (KB + S - 1) / S;
  const int ks0 = 2 * sp * per, ks1 = 2 * min(KB, sp * per + per);   // 64-k steps
  const int t0 = tc * TT * 16;

  f32x4 acc[4][TT];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long tile_base = ((long)(nb * 4 + wv) * KB) * 4;             // int4 tile order [N/64][K/128][4]
  const unsigned* wbase = reinterpret_cast<const unsigned*>(wp + tile_base * 64 + lane);
  const unsigned char* mbase = AWQ ? meta + tile_base * 64 + j * 4 : nullptr;
  const u32x4* dbase = wp + ((long)(nb * 4 + wv) * 4 * KB) * 256 + lane;      // dense tile order [N/16][K/32]
  uint2 ra[4];                                                       // int4: the two dwords of this 64-k half, per n-tile
  unsigned rm[4];
  u32x4 rd[AWQ ? 1 : 4][2];                                          // dense: the two 16 x 32 tiles of this step, per n-tile
  u32x4 rb[FPW];
  auto gload = [&](int ks) {
    const int kb = ks >> 1, hf = ks & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (AWQ) {
        ra[q] = *reinterpret_cast<const uint2*>(wbase + ((long)kb * 4 + q) * 256 + hf * 2);
        rm[q] = *reinterpret_cast<const unsigned*>(mbase + ((long)kb * 4 + q) * 64);
      } else {
        const u32x4* tile = dbase + ((long)q * KB) * 256 + (long)ks * 128;
        rd[AWQ ? 0 : q][0] = __builtin_nontemporal_load(tile);
        rd[AWQ ? 0 : q][1] = __builtin_nontemporal_load(tile + 64);
      }
    }
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const int f = i * 4 + wv, tt = f >> 1, sx = f & 1;
      const int tok = t0 + tt * 16 + j;
      const u32x4 z = {0u, 0u, 0u, 0u};
      rb[i] = (tok < Tv && f < NF) ? *reinterpret_cast<const u32x4*>(x + (long)tok * ldx + ks * 64 + sx * 32 + g * 8) : z;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < FPW; ++i)
      if (NF % 4 == 0 || i * 4 + wv < NF) sB[(buf * NF + i * 4 + wv) * 64 + lane] = rb[i];
  };

  if (ks0 < ks1) {
    gload(ks0);
    sstore(0);
  }
  __syncthreads();
  for (int ks = ks0; ks < ks1; ++ks) {
    const int buf = (ks - ks0) & 1;
    // dequantise this step's two fragments per n-tile from the staged registers, then refill them
    u32x4 wf[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!AWQ) { wf[q][0] = rd[AWQ ? 0 : q][0]; wf[q][1] = rd[AWQ ? 0 : q][1]; continue; }
      const unsigned mm = rm[q];
      const _Float16 sc = __builtin_bit_cast(_Float16, (u16)(mm & 0xffffu));
      const _Float16 zf = __builtin_bit_cast(_Float16, (u16)(mm >> 16));
      const h2 s2 = {sc, sc};
      const _Float16 nz = -((_Float16)1024.0f + zf), nz16 = -((_Float16)64.0f + zf);
      const h2 nz2 = {nz, nz}, nz16_2 = {nz16, nz16};
      const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
      unsigned magic = 0x64006400u;
      asm volatile("" : "+v"(magic));
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        const unsigned w = sx ? ra[q].y : ra[q].x, w8 = w >> 8;
        const h2 q0 = __builtin_bit_cast(h2, and_or(w, 0x000F000Fu, magic));
        const h2 q1 = __builtin_bit_cast(h2, and_or(w, 0x00F000F0u, magic));
        const h2 q2 = __builtin_bit_cast(h2, and_or(w8, 0x000F000Fu, magic));
        const h2 q3 = __builtin_bit_cast(h2, and_or(w8, 0x00F000F0u, magic));
#ifdef UMB_VG_NODEQ       // ablation (wrong results): raw bits as the operand, no dequant arithmetic
        wf[q][sx][0] = w ^ mm; wf[q][sx][1] = w8; wf[q][sx][2] = w; wf[q][sx][3] = w8 ^ mm;
        (void)q0; (void)q1; (void)q2; (void)q3; (void)s2; (void)nz2; (void)nz16_2; (void)sixteenth;
#else
        wf[q][sx][0] = __builtin_bit_cast(unsigned, (q0 + nz2) * s2);
        wf[q][sx][1] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(q1, sixteenth, nz16_2) * s2);
        wf[q][sx][2] = __builtin_bit_cast(unsigned, (q2 + nz2) * s2);
        wf[q][sx][3] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(q3, sixteenth, nz16_2) * s2);
#endif
      }
    }
#ifndef UMB_VG_NOGLOAD      // ablation (wrong results): no global loads inside the loop
    if (ks + 1 < ks1) gload(ks + 1);
#endif
    // activation fragments: the read for fragment i+1 is in flight while fragment i feeds 4 MFMAs
    const u32x4* sb = sB + (buf * NF) * 64 + lane;
    u32x4 bcur = sb[0];
    __builtin_amdgcn_s_setprio(2);   // the SIMD's other wave (other block) gets the issue slots only when this one waits:
                                     // keeps the two out of phase, one in its MFMA run while the other dequantises (-6.5 %)
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int t = i >> 1, sx = i & 1;
      u32x4 bnext = bcur;
#ifndef UMB_VG_NOLDS        // ablation (wrong results): one fragment read per step
      if (i + 1 < NF) bnext = sb[(i + 1) * 64];
#endif
#ifdef UMB_VG_NOMFMA        // ablation (wrong results): one MFMA per fragment instead of four
      acc[0][t] = P::mfma(wf[0][sx] ^ wf[1][sx] ^ wf[2][sx] ^ wf[3][sx], bcur, acc[0][t]);
#else
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q][t] = P::mfma(wf[q][sx], bcur, acc[q][t]);
#endif
      bcur = bnext;
    }
    __builtin_amdgcn_s_setprio(0);
#ifndef UMB_VG_NOSTORE      // ablation (wrong results)
    if (ks + 1 < ks1) sstore(buf ^ 1);
#endif
#ifndef UMB_VG_NOBAR        // ablation (wrong results)
    __syncthreads();
#endif
  }

#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int tok = t0 + t * 16 + j;
    if (tok >= Tv) continue;
    float inv = 1.f;
    if (fx.ssq_in) {
      const float* sq = fx.ssq_in + (long)tok * (fx.ssq_stride ? fx.ssq_stride : fx.ssq_groups);
      float a = 0.f;
      for (int q = 0; q < fx.ssq_groups; q += 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(sq + q); a += v[0]; a += v[1]; a += v[2]; a += v[3]; }
      inv = rsqrtf(a / fx.ssq_dim + fx.eps);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = acc[q][t];
      const int ntile = (nb * 4 + wv) * 4 + q;
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

// ------------------------------------------------------------------ verify GEMM, explicit two-phase (ping-pong) schedule
// The register-resident kernel above leaves the overlap of one wave's MFMA run with the other wave's dequant / loads /
// staging to the wave scheduler (two independent blocks per CU + s_setprio).  Here the two live in ONE 8-wave workgroup as
// two groups of four waves, each group a 256-row x TT*16-token work item of its own, and every step is cut into an X phase
// (activations of this step -> LDS, dequant, next step's loads) and an M phase (fragment reads + 4 TT x 2 MFMAs) with a
// workgroup barrier at every phase boundary; group 1 starts one barrier late, so on every SIMD one wave is in its M
// phase while the other is in its X phase, in lockstep.  The groups of a workgroup take the same weight rows and
// neighbouring token chunks when the chunk count is even (the second group's weight loads hit in L1 / L2 half a step
// later: one HBM fetch per two chunks), otherwise neighbouring row blocks.  All loads unconditional (buffer descriptors).
#ifndef UMB_PP_LM
#define UMB_PP_LM 7
#endif
template <typename P, int AWQ, int TT, int SHX = 0>
__global__ __launch_bounds__(512) void verify_gemm_pp_kernel(const u32x4* __restrict__ wp,
                                                             const unsigned char* __restrict__ meta,
                                                             const u16* __restrict__ x, int ldx, int T, int Tv, int N, int K,
                                                             int S, int epi, int pair_tc, float* __restrict__ out, GemmFused fx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NF = TT * 2;                                         // activation fragments per 64-k step
  constexpr int FPW = (NF + 3) / 4;                                  // staged per wave
  // SHX (round 4): the two work items of the workgroup are neighbouring ROW blocks of the SAME token chunk and stage its
  // activations ONCE -- two buffers by step parity, each group loading and storing half of a step's fragments (group 1 one step
  // ahead of its own compute, so that both halves are in place when group 0 reads them).  The activations' trip registers -> LDS
  // is 8 % of this kernel's wall time (profiles/r04_vgemm_pp32_negative.txt, 6): half the ds_write_b128 per MFMA.
  constexpr int FH = SHX ? (FPW + 1) / 2 : FPW;                      // fragments this wave loads / stores per step
  const int lane = threadIdx.x & 63;
  const int wv8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wv8 >> 2, wv = wv8 & 3;
  u32x4* sB = reinterpret_cast<u32x4*>(smem) + (SHX ? 0 : grp * NF * 64);   // one buffer per group: [TT][2 s][64] (SHX: per step parity)
  const int ibase = SHX ? grp * (FPW / 2) : 0;                       // SHX: group 0 stages fragment rows i = 0 .. FPW/2 - 1, group 1 the rest
  const int j = lane & 15, g = lane >> 4;
  const int nblk = N / 256;
  const int nchunk = (Tv + TT * 16 - 1) / (TT * 16);
  int nb, tc, sp;
  if (!SHX && pair_tc) {                                             // groups: token chunks 2 q, 2 q + 1 of the same rows
    const int hc = nchunk / 2;
    nb = blockIdx.x % nblk; tc = 2 * ((blockIdx.x / nblk) % hc) + grp; sp = blockIdx.x / (nblk * hc);
  } else {                                                           // groups: row blocks 2 q, 2 q + 1
    const int hb = nblk / 2;
    nb = 2 * (blockIdx.x % hb) + grp; tc = (blockIdx.x / hb) % nchunk; sp = blockIdx.x / (hb * nchunk);
  }
  const int KB = K / 128;
  const int per = (KB + S - 1) / S;
  const int ks0 = 2 * sp * per, ks1 = 2 * min(KB, sp * per + per);   // 64-k steps (same for both groups: same sp)
  const int t0 = tc * TT * 16;

  f32x4 acc[4][TT];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long tile_base = ((long)(nb * 4 + wv) * KB) * 4;             // int4 tile order [N/64][K/128][4]
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(AWQ ? wp + tile_base * 64 : wp + ((long)(nb * 4 + wv) * 4 * KB) * 256), 0,
      AWQ ? (unsigned)KB * 4096u : (unsigned)KB * 16384u, 0x00020000);
  const auto rmt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(AWQ ? meta + tile_base * 64 : (const unsigned char*)wp), 0,
                                                     AWQ ? (unsigned)KB * 256u : 0u, 0x00020000);
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(x), 0, (unsigned)(((long)(Tv - 1) * ldx + K) * 2), 0x00020000);
  int voffx[FH], soff[FH];
#pragma unroll
  for (int i = 0; i < FH; ++i) {
    const int f = (ibase + i) * 4 + wv, tok = t0 + (f >> 1) * 16 + j;
    const bool mine = f < NF && (!SHX || grp == 1 || i < FPW / 2);   // SHX, odd FPW: the extra row is group 1's
    voffx[i] = (mine && tok < Tv) ? (int)(((long)tok * ldx + (f & 1) * 32 + g * 8) * 2) : (int)0x80000000;
    soff[i] = (mine ? f * 64 : 2 * NF * 64) + lane;                  // a fragment that is not this wave's goes to a dump slot
  }
  // Loads are the X phase's long pole (phase trace, scripts/r3/vg_trace.py: 13 buffer loads per step cost ~940 cycles of
  // issue with four waves of the CU in their X phase at once -- the texture-address path takes a wave64 load every ~16
  // cycles whatever its width), so a 128-k block's int4 tile is ONE 16-byte load per n-tile and its metadata one load,
  // shared by the block's two 64-k steps: 16 loads per two steps instead of 26.
  u32x4 ra[4];                                                       // int4: the lane's four dwords (4 x k32) of this 128-k block, per n-tile
  unsigned rm[4];
  u32x4 rd[AWQ ? 1 : 4][AWQ ? 1 : 4];                                // dense: the four 16 x 32 tiles of this block, per n-tile
  u32x4 rb[FH];
  const int kb0 = ks0 >> 1, kb1 = ks1 >> 1;
  // single loads, so that the M phase can issue them one at a time between its MFMA groups
  auto load_w1 = [&](int kb_, int q) {                               // int4 tile q of block kb_ (dense: its four k32 tiles)
#ifdef UMB_PP_NOWLOAD      // ablation (wrong results): the weights of the first block only
    if (kb_ > kb0) return;
#endif
    const int kb = min(kb_, kb1 - 1);                                // past the slab: a harmless reload of its last block
    if (AWQ) {
      ra[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, (kb * 4 + q) * 1024, 0);
    } else {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        rd[AWQ ? 0 : q][AWQ ? 0 : s4] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, (q * KB * 4 + kb * 4 + s4) * 1024, 2);
    }
  };
  auto load_m1 = [&](int kb_, int q) {
#ifdef UMB_PP_NOWLOAD
    if (kb_ > kb0) return;
#endif
    const int kb = min(kb_, kb1 - 1);
    if (AWQ) rm[q] = __builtin_amdgcn_raw_buffer_load_b32(rmt, j * 4, (kb * 4 + q) * 64, 0);
  };
  auto load_x1 = [&](int ks_, int i) {
#ifdef UMB_PP_NOXLOAD      // ablation (wrong results): the activations of the first step only
    if (ks_ > ks0) return;
#endif
    const int ks = min(ks_, ks1 - 1);
    rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, voffx[i], ks * 128, 0);
  };
  auto gload_w = [&](int kb_) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { load_w1(kb_, q); load_m1(kb_, q); }
  };
  auto gload_x = [&](int ks_) {
#pragma unroll
    for (int i = 0; i < FH; ++i) load_x1(ks_, i);
  };
  auto sstore = [&](int par_store) {
#ifdef UMB_PP_NOSTORE      // ablation (wrong results): no activation stores to LDS
    return;
#endif
    if constexpr (SHX) {
      // group 0 stores this step's half into the buffer of this step's parity, group 1 the NEXT step's half into the other one
#pragma unroll
      for (int i = 0; i < FH; ++i) sB[par_store * NF * 64 + soff[i]] = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < FPW; ++i)
        if (NF % 4 == 0 || i * 4 + wv < NF) sB[(i * 4 + wv) * 64 + lane] = rb[i];
    }
  };
#ifdef UMB_VG_TRACE
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define VG_T(i) do { const unsigned long long tn = __builtin_readcyclecounter(); tsum[i] += tn - tprev; tprev = tn; } while (0)
#else
#define VG_T(i) do {} while (0)
#endif
  auto step = [&](auto hc, int kb) {
    constexpr int HF = decltype(hc)::value;                          // which 64-k half of the block
    // ---- X phase: this step's activations to LDS, its weights dequantised, the next loads issued
    sstore(SHX ? (HF ^ grp) : 0);
#ifdef UMB_VG_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    VG_T(0);
    u32x4 wf[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!AWQ) { wf[q][0] = rd[AWQ ? 0 : q][AWQ ? 0 : HF * 2]; wf[q][1] = rd[AWQ ? 0 : q][AWQ ? 0 : HF * 2 + 1]; continue; }
      const unsigned mm = rm[q];
      const _Float16 sc = __builtin_bit_cast(_Float16, (u16)(mm & 0xffffu));
      const _Float16 zf = __builtin_bit_cast(_Float16, (u16)(mm >> 16));
      const h2 s2 = {sc, sc};
      const _Float16 nz = -((_Float16)1024.0f + zf), nz16 = -((_Float16)64.0f + zf);
      const h2 nz2 = {nz, nz}, nz16_2 = {nz16, nz16};
      const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
      unsigned magic = 0x64006400u;
      asm volatile("" : "+v"(magic));
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        const unsigned w = ra[q][HF * 2 + sx], w8 = w >> 8;
        const h2 q0 = __builtin_bit_cast(h2, and_or(w, 0x000F000Fu, magic));
        const h2 q1 = __builtin_bit_cast(h2, and_or(w, 0x00F000F0u, magic));
        const h2 q2 = __builtin_bit_cast(h2, and_or(w8, 0x000F000Fu, magic));
        const h2 q3 = __builtin_bit_cast(h2, and_or(w8, 0x00F000F0u, magic));
#ifdef UMB_VG_NODEQ       // ablation (wrong results): raw bits as the operand, no dequant arithmetic
        wf[q][sx][0] = w ^ mm; wf[q][sx][1] = w8; wf[q][sx][2] = w; wf[q][sx][3] = w8 ^ mm;
        (void)q0; (void)q1; (void)q2; (void)q3; (void)s2; (void)nz2; (void)nz16_2; (void)sixteenth;
#else
        wf[q][sx][0] = __builtin_bit_cast(unsigned, (q0 + nz2) * s2);
        wf[q][sx][1] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(q1, sixteenth, nz16_2) * s2);
        wf[q][sx][2] = __builtin_bit_cast(unsigned, (q2 + nz2) * s2);
        wf[q][sx][3] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(q3, sixteenth, nz16_2) * s2);
#endif
      }
    }
#ifdef UMB_VG_TRACE
    asm volatile("" :: "v"(wf[0][0][0]), "v"(wf[1][1][1]), "v"(wf[2][0][2]), "v"(wf[3][1][3]));
#endif
    VG_T(1);
    // which loads the M phase issues (bit 0 activations, 1 weights, 2 metadata); the rest go out here
    if (!(UMB_PP_LM & 1)) gload_x(2 * kb + HF + 1 + (SHX ? grp : 0));
    if (HF == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!(UMB_PP_LM & 2)) load_w1(kb + 1, q);
        if (!(UMB_PP_LM & 4)) load_m1(kb + 1, q);
      }
    }
    VG_T(2);
    __syncthreads();
    VG_T(3);
    // ---- M phase.  Fragment reads run BD groups (of 4 MFMAs = 64 cycles) ahead of their use and are fenced there: left
    // to itself the scheduler sinks every ds_read to just in front of its first MFMA
#ifdef UMB_PP_BD
    constexpr int BD = UMB_PP_BD;
#else
    constexpr int BD = 3;
#endif
    const u32x4* sb = sB + (SHX ? HF * NF * 64 : 0) + lane;
    u32x4 bq[BD + 1];
#pragma unroll
    for (int i = 0; i < BD; ++i) bq[i] = sb[i * 64];
    __builtin_amdgcn_s_setprio(1);
#ifndef UMB_PP_NOFENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int t = i >> 1, sx = i & 1;
      if (i + BD < NF) bq[(i + BD) % (BD + 1)] = sb[(i + BD) * 64];
#ifdef UMB_PP_NOMFMA       // ablation (wrong results): the phase without its matrix instructions
      asm volatile("" :: "v"(bq[i % (BD + 1)]), "v"(wf[0][sx]), "v"(wf[1][sx]), "v"(wf[2][sx]), "v"(wf[3][sx]));
#else
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q][t] = P::mfma(wf[q][sx], bq[i % (BD + 1)], acc[q][t]);
#endif
#ifdef UMB_PP_NOFENCE
#define PP_FENCE() do {} while (0)
#else
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
      // The next loads go out HERE, one per MFMA group: issued from the X phase they cost ~90 cycles each (four waves of the
      // CU queue on the texture-address path at once); between MFMAs the issue slots are free.  Registers: rb was
      // stored at the start of this step's X phase, ra / rm were consumed by its dequant.
      if ((UMB_PP_LM & 1) && (i & 1) == 1 && (i >> 1) < FH) load_x1(2 * kb + HF + 1 + (SHX ? grp : 0), i >> 1);
      if ((UMB_PP_LM & 2) && HF == 1 && i >= NF - 8 && i < NF - 4) load_w1(kb + 1, i - (NF - 8));
      if ((UMB_PP_LM & 4) && HF == 1 && i >= NF - 4) load_m1(kb + 1, i - (NF - 4));
      PP_FENCE();
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef UMB_VG_TRACE
    asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[3][TT - 1][3]));
#endif
    VG_T(4);
    __syncthreads();
    VG_T(5);
  };

  if (kb0 < kb1) {
    gload_x(ks0);
    gload_w(kb0);
  }
  if (grp == 1) {
    if constexpr (SHX) {
      // group 1's half of the FIRST step goes out before the stagger barrier (group 0 reads it in its first M phase);
      // from then on group 1 carries the activations of the step after the one it computes
      if (kb0 < kb1) { sstore(0); gload_x(ks0 + 1); }
    }
    __syncthreads();                                                 // stagger: group 1 runs one phase behind group 0
  }
  for (int kb = kb0; kb < kb1; ++kb) {
    step(std::integral_constant<int, 0>{}, kb);
    step(std::integral_constant<int, 1>{}, kb);
  }
  if (grp == 0) __syncthreads();
#ifdef UMB_VG_TRACE
  if (lane == 0 && fx.counters) {
    unsigned long long* tr = reinterpret_cast<unsigned long long*>(fx.counters) + ((size_t)blockIdx.x * 8 + wv8) * 8;
    for (int i = 0; i < 6; ++i) tr[i] = tsum[i];
    tr[6] = (unsigned long long)(ks1 - ks0);
  }
#endif

#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int tok = t0 + t * 16 + j;
    if (tok >= Tv) continue;
    float inv = 1.f;
    if (fx.ssq_in) {
      const float* sq = fx.ssq_in + (long)tok * (fx.ssq_stride ? fx.ssq_stride : fx.ssq_groups);
      float a = 0.f;
      for (int q = 0; q < fx.ssq_groups; q += 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(sq + q); a += v[0]; a += v[1]; a += v[2]; a += v[3]; }
      inv = rsqrtf(a / fx.ssq_dim + fx.eps);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = acc[q][t];
      const int ntile = (nb * 4 + wv) * 4 + q;
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

template <typename P, int AWQ>
static int launch_verify(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int Tv, int N,
                         int K, int S, int epi, const GemmFused& fx, hipStream_t st, bool tt9 = false) {
  const int nchunk = (Tv + 127) / 128;
  if constexpr (AWQ != 1) {
    static const bool plain = getenv("UMB_VGEMM_PLAIN") != nullptr;  // diagnostic: the LDS-shared 128 x 128 kernel
    // 256-row blocks: only when they still give every CU a block (small layers keep the 128-row kernel's finer grid)
    if (!plain && N % 256 == 0 && (N / 256) * nchunk * S >= 256) {
      // two-phase schedule: two work items per 8-wave workgroup (UMB_VG_PP=0: the two-blocks-per-CU kernel, A/B)
      static const bool no_pp = getenv("UMB_VG_PP") != nullptr && atoi(getenv("UMB_VG_PP")) == 0;
      const int nch = tt9 ? (Tv + 143) / 144 : nchunk;
      const int pair_tc = nch % 2 == 0;
      if constexpr (AWQ == 2) if (!no_pp && (pair_tc || (N / 256) % 2 == 0)) {      // int4 only (a dense 128-k block of tiles is 64 registers)
        const unsigned grid = (unsigned)((N / 256) * nch * S / 2);
        // shared activation staging: the two work items are neighbouring row blocks of one token chunk (UMB_PP_SHX=0: the
        // round-3 pairing -- same rows, neighbouring token chunks, each group staging its own activations)
        static const bool shx = getenv("UMB_PP_SHX") == nullptr || atoi(getenv("UMB_PP_SHX")) != 0;
        if (shx && (N / 256) % 2 == 0) {
          if (tt9) {
            hipLaunchKernelGGL((verify_gemm_pp_kernel<P, AWQ, 9, 1>), dim3(grid), dim3(512), (size_t)(2 * 18 + 1) * 64 * 16, st,
                               (const u32x4*)wp, (const unsigned char*)meta, x, ldx, T, Tv, N, K, S, epi, 0, out, fx);
          } else {
            hipLaunchKernelGGL((verify_gemm_pp_kernel<P, AWQ, 8, 1>), dim3(grid), dim3(512), (size_t)(2 * 16 + 1) * 64 * 16, st,
                               (const u32x4*)wp, (const unsigned char*)meta, x, ldx, T, Tv, N, K, S, epi, 0, out, fx);
          }
          UMB_LAUNCH_CHECK();
          return UMB_OK;
        }
        if (tt9) {
          hipLaunchKernelGGL((verify_gemm_pp_kernel<P, AWQ, 9>), dim3(grid), dim3(512), (size_t)2 * 18 * 64 * 16, st,
                             (const u32x4*)wp, (const unsigned char*)meta, x, ldx, T, Tv, N, K, S, epi, pair_tc, out, fx);
        } else {
          hipLaunchKernelGGL((verify_gemm_pp_kernel<P, AWQ, 8>), dim3(grid), dim3(512), (size_t)2 * 16 * 64 * 16, st,
                             (const u32x4*)wp, (const unsigned char*)meta, x, ldx, T, Tv, N, K, S, epi, pair_tc, out, fx);
        }
        UMB_LAUNCH_CHECK();
        return UMB_OK;
      }
      if (tt9) {
        hipLaunchKernelGGL((verify_gemm_r_kernel<P, AWQ, 9>), dim3((unsigned)((N / 256) * ((Tv + 143) / 144) * S)),
                           dim3(256), (size_t)2 * 18 * 64 * 16, st, (const u32x4*)wp, (const unsigned char*)meta, x, ldx,
                           out, T, Tv, N, K, S, epi, fx);
      } else {
        hipLaunchKernelGGL((verify_gemm_r_kernel<P, AWQ, 8>), dim3((unsigned)((N / 256) * nchunk * S)), dim3(256),
                           (size_t)2 * 16 * 64 * 16, st, (const u32x4*)wp, (const unsigned char*)meta, x, ldx, out, T, Tv,
                           N, K, S, epi, fx);
      }
      UMB_LAUNCH_CHECK();
      return UMB_OK;
    }
    if (tt9) return UMB_EINVAL;                                      // caller checks r_kernel_ok() first
  }
  const size_t smem = (size_t)(2 * 8 * 2 + 2 * 8 * 2) * 64 * 16;     // 64 KiB
  hipLaunchKernelGGL((verify_gemm_kernel<P, AWQ>), dim3((unsigned)((N / 128) * nchunk * S)), dim3(256), smem, st,
                     (const u32x4*)wp, (const unsigned char*)meta, x, ldx, out, T, Tv, N, K, S, epi, fx);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// Split count for a wide (T > 64) forward.  The register-resident verify kernel runs two 256-row x 128/144-token
// blocks per CU, i.e. 512 block slots: S is chosen so that (N/256) * chunks * S lands just under 512 (one full round;
// measured optimum on the 70B shapes: T = 257 -> 6 / 8 / 8 for qkv / o / down, T = 769 -> 2, T = 1024 -> 3 / 2 / 2).
// More splits only add fp32 partial traffic (S * T * N * 8 bytes per GEMM), fewer leave CUs with one wave per SIMD.
extern "C" int umb_gemm_wide_split(int T, int N, int S_plan) {
  if (T <= 64 || S_plan <= 1) return S_plan;
  const int rem = T % 128, nc9 = (T + 143) / 144;
  const bool tail = rem >= 1 && rem <= 64;
  const int nchunk = (tail && nc9 == T / 128) ? nc9 : tail ? T / 128 : (T + 127) / 128;
  const int b0 = max(1, (N / 256) * max(1, nchunk));
  int S = 512 / b0;
  if (S < 2) S = b0 <= 341 ? 3 : b0 <= 512 ? 2 : 1;
  return max(1, min(S, S_plan));
}

// ------------------------------------------------------------------ host side
// (R, S) depend on (N, K, format) only -- never on T -- so a token's result is
// independent of how many other tokens share the launch.
extern "C" void umb_gemm_plan(int N, int K, int awq, int force_s1, int* R_out, int* S_out) {
  const int NT = N / 16, KB = K / 128;
  int R = 1;
  if (!awq && NT % 2 == 0 && NT >= 4096) R = 2;
  if (awq && NT % 8 == 0 && NT >= 512) R = 2;      // two independent dequant chains per wave: +6-8 % measured
  int S = 1;
  if (!force_s1) {
    const int nblk = (NT + 4 * R - 1) / (4 * R);
    S = (1024 / R + nblk - 1) / nblk;                        // aim at ~4096 n-tile streams (16 / CU)
    const int min_blocks = awq ? 4 : 2;                       // keep >= 512 / 256 k per slab
    if (S > KB / min_blocks) S = KB / min_blocks;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
  }
  *R_out = R;
  *S_out = S;
}

// The full plan (see the header).  One CU streams ~25 GB/s whatever runs on it and one 4-wave block alone ~17 GB/s
// (scripts/r3/gemm_trace.py: per-block phase stamps), so a launch wants whole rounds of 2 blocks per CU.  70B gate/up
// (3584 n-tiles): 448 blocks of 8 tiles left 64 CUs with one block for the whole launch -> 512 blocks of 7 tiles, capped
// at two per CU (launch_k): 46.1 -> 44.9 us inside the iteration graph (48.4 -> 44.8 us stand-alone).
// S_row stays 0 (the runtime's rule): running the 70B o-projection with one tile per wave and its full 8 splits (1024
// blocks instead of 256) was 2 us faster stand-alone and 3 us SLOWER inside the graph (GEMM 11.1 -> 12.1 us, the
// row reduce reads 8 partials instead of 4) -- measured, not kept.
extern "C" void umb_gemm_plan2(int N, int K, int awq, int force_s1, int* R_out, int* S_out, int* tb_out, int* S_row_out) {
  int R, S;
  umb_gemm_plan(N, K, awq, force_s1, &R, &S);
  const int NT = N / 16;
  int tb = 0;
  static const bool off = getenv("UMB_PLAN2_OFF") != nullptr;                  // A/B: the round-2 plan
  static const bool no_w8 = getenv("UMB_NO_W8") != nullptr;                    // A/B: 4-wave blocks only
  if (!off && R == 2) {
    const int b8 = ((NT + 7) / 8) * S, b7 = ((NT + 6) / 7) * S, b14 = ((NT + 13) / 14) * S;
    if (b8 % 256 != 0 && b7 % 256 == 0) tb = 7;
    // exactly one 8-wave block of 14 tiles per CU: the CU's waves share ONE staged copy of the activations (two 4-wave
    // blocks each load their own: 6.6 KB of x per 14 KiB of weights and chunk); flag 0x80 = 8 waves per block
    if (!no_w8 && awq && b8 % 256 != 0 && b14 == 256) tb = 14 | 0x80;

  }
  // Round 4: a linear that the rules above leave with at most ONE 4-wave block per CU (8B-AWQ gate/up: 256 blocks of 2 + 2 + 2 + 1
  // tiles) or with a ragged round (8B dense gate/up: 448 four-tile blocks), and whose n-tiles are exactly 256 groups of 5 ... 8, runs as
  // one 8-wave block per CU with one tile per wave (1792 tiles = 256 x 7).  Twice the waves per CU on the same bytes, one staged copy of
  // the activations; same per-output summation order, so the bits do not change.  8B forward, dense at 5 / 31 rows 3.52 -> 3.46 /
  // 3.90 -> 3.86 ms, int4 at 1 / 16 / 32 rows -2 / -1 / -1 %.
  if (!off && !no_w8 && !(tb & 0x80)) {
    const int per = tb ? tb : 4 * R;
    const int cur = ((NT + per - 1) / per) * S;
    if (cur <= 256 || (S == 1 && cur % 256 != 0))             // the ragged case only where it was measured (unsplit gate/up)
      for (int t8 = 8; t8 >= 5; --t8)
        if (NT % t8 == 0 && (NT / t8) * S == 256) { R = 1; tb = t8 | 0x80; break; }
  }
  int S_row = 0;
  // experiments: UMB_PLAN_OVR="N,K:R,S,tb,Srow;..." replaces the plan of the named shapes (tb may carry the 0x80 flag)
  static const char* ovr = getenv("UMB_PLAN_OVR");
  if (ovr) {
    for (const char* p = ovr; p && *p; ) {
      int n = 0, k = 0, r = 0, sp = 0, t = 0, sr = 0;
      if (sscanf(p, "%d,%d:%d,%d,%d,%d", &n, &k, &r, &sp, &t, &sr) == 6 && n == N && k == K) { R = r; S = sp; tb = t; S_row = sr; }
      p = strchr(p, ';');
      if (p) ++p;
    }
  }
  *R_out = R; *S_out = S; *tb_out = tb; *S_row_out = S_row;
}

template <typename P, int AWQ, int TT, int R, int CB, int NWV = 4>
static int launch_k(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int Ttot, int N,
                    int K, int S, int epi, const GemmFused& fx, hipStream_t st, int tb = 0) {
  const int NT = N / 16;
  if (tb <= 0 || tb > NWV * R) tb = NWV * R;
  const int nblk = (NT + tb - 1) / tb;
  size_t smem = (size_t)2 * CB * TT * 4 * 1024 + (AWQ ? NWV * 2 * 1024 : 0);   // x chunks (double buffered) + staged int4 metadata
  // experiment knob UMB_LDS_KB: ask for at least that much dynamic LDS per block, which caps the blocks a CU admits
  // (160 KiB per CU: 56 -> 2 blocks, 84 -> 1)
  static const int lds_kb = getenv("UMB_LDS_KB") ? atoi(getenv("UMB_LDS_KB")) : 0;
  if (lds_kb > 0 && smem < (size_t)lds_kb * 1024) smem = (size_t)lds_kb * 1024;
  // a launch of exactly two blocks per CU: ask for 56 KiB of LDS so that no CU admits a third (the registers would
  // allow it) and leaves another with one -- every CU then streams the same bytes
  if (lds_kb == 0 && NWV == 4 && nblk * S == 512 && smem < 56 * 1024) smem = 56 * 1024;
  if (smem > 64 * 1024) {
    static bool once = false;      // one-time opt-in to > 64 KiB of dynamic LDS for this instantiation
    if (!once) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_kernel<P, AWQ, TT, R, CB, NWV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return UMB_EHIP;
      once = true;
    }
  }
  const int per = (K / 128 + S - 1) / S;
  hipLaunchKernelGGL((skinny_gemm_kernel<P, AWQ, TT, R, CB, NWV>), dim3((unsigned)nblk, (unsigned)S), dim3(64 * NWV), smem, st,
                     (const u32x4*)wp, (const unsigned char*)meta, x, ldx, T, N, K, per, tb, Ttot,
                     epi | (fx.x_fm ? 0x100 : 0) | (fx.out_fm ? 0x200 : 0), out, S, fx);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

template <typename P, int AWQ, int TT, int CB>
static int launch_r(int R, const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int Ttot,
                    int N, int K, int S, int epi, const GemmFused& fx, hipStream_t st, int tb = 0) {
  if ((N / 16) % R) return UMB_EINVAL;
  if (epi > EPI_SILU) tb = 0;                                          // the in-kernel split epilogues index counters by 4-tile block
  const bool w8 = (tb & 0x80) != 0;                                    // plan: 8 waves per block (one block per CU)
  tb &= 0x7f;
  if (R == 1) {
    // 8 waves x 1 tile (round 4): a linear whose tiles do not fill the chip in whole rounds of 4-wave blocks but do as ONE block
    // of 5 ... 8 tiles per CU (8B gate/up: 1792 tiles = 256 x 7; 448 four-tile blocks leave 64 CUs with half the work)
    if constexpr ((CB * TT * 4) % 8 == 0) {
      if (w8 && epi <= EPI_SILU) return launch_k<P, AWQ, TT, 1, CB, 8>(wp, meta, x, ldx, out, T, Ttot, N, K, S, epi, fx, st, tb);
    }
    return launch_k<P, AWQ, TT, 1, CB>(wp, meta, x, ldx, out, T, Ttot, N, K, S, epi, fx, st, w8 ? 0 : tb);
  }
  if (R == 2 && epi <= EPI_SILU) {
    if constexpr ((CB * TT * 4) % 8 == 0 && AWQ == 1 && TT == 1) {     // instantiated where the plan uses it: int4, <= 16 rows
      if (w8) return launch_k<P, AWQ, TT, 2, CB, 8>(wp, meta, x, ldx, out, T, Ttot, N, K, S, epi, fx, st, tb);
    }
    if (w8) tb = (tb + 1) / 2;                                         // the same balance with 4-wave blocks (14 -> 7 tiles)
    return launch_k<P, AWQ, TT, 2, CB>(wp, meta, x, ldx, out, T, Ttot, N, K, S, epi, fx, st, tb);
  }
  return UMB_EINVAL;
}

template <typename P, int AWQ>
static int launch_tt(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int N, int K, int R,
                     int S, int epi, const GemmFused& fx0, hipStream_t st, int tb = 0) {
  // tokens beyond 64 go through further launches (weights re-read from L2/HBM)
  const long ostride = (epi == EPI_SILU) ? (long)(N / 2) / 2 : (long)N;   // out rows in units of float
  int tdone = 0;
  if constexpr (AWQ != 1) {
    static const bool no_vgemm = getenv("UMB_NO_VGEMM") != nullptr;     // diagnostic: 64-token chunks only
    if (T > 64 && N % 128 == 0 && epi <= EPI_SILU && !no_vgemm) {
      // round 6: 32-row x whole-chunk waves, LDS-DMA activations (vgemm.hip); UMB_VGW=0: the kernels below
      if constexpr (AWQ == 2 && std::is_same<P, F16>::value)
        if (umb_vgemm_w_ok(T, N, K, S, epi)) return umb_vgemm_w(wp, meta, x, ldx, out, T, N, K, S, epi, fx0, st);
      const int rem = T % 128;
      // T = w d + 1 (257, 385, 769 ...): 144-token blocks swallow the short tail in the same number of chunks
      const int nc9 = (T + 143) / 144;
      if (rem >= 1 && rem <= 64 && nc9 == T / 128 && N % 256 == 0 && (N / 256) * nc9 * S >= 256 &&
          getenv("UMB_VGEMM_PLAIN") == nullptr)
        return launch_verify<P, AWQ>(wp, meta, x, ldx, out, T, T, N, K, S, epi, fx0, st, /*tt9=*/true);
      tdone = (rem == 0 || rem > 64) ? T : T - rem;                  // a tail of <= 64 tokens is HBM-bound: skinny kernel
      const int rc = launch_verify<P, AWQ>(wp, meta, x, ldx, out, T, tdone, N, K, S, epi, fx0, st);
      if (rc || tdone == T) return rc;
    }
  }
  // token chunking: 64 per launch.  A 256-token variant of this kernel (TT = 16, one n-tile per wave) was measured
  // SLOWER per token (1600 vs 4 x 314 us per 70B layer): every MFMA pulls its 1 KiB B fragment from LDS, which
  // pins the matrix pipe to the LDS rate with one wave per SIMD.  The large-T verify needs a register-tiled
  // (R x TT per wave) kernel -- next round; until then weights are re-read once per 64 tokens.
  const int step = 64;
  for (int t0 = tdone; t0 < T; t0 += step) {
    const int tn = min(step, T - t0);
    const u16* xx = x + (long)t0 * ldx;
    float* oo = out + (long)t0 * ostride;      // out is [S][T][N] over the full T; split stride stays T
    GemmFused fx = fx0;                        // per-chunk views of the token-indexed side buffers
    if (fx.ssq_in) fx.ssq_in += (long)t0 * (fx.ssq_stride ? fx.ssq_stride : fx.ssq_groups);
    if (fx.ssq_out) fx.ssq_out += (long)t0 * fx.ssq_out_stride;
    if (fx.h) fx.h += (long)t0 * N;
    if (fx.hw) fx.hw += (long)t0 * N;
    if (fx.pos) fx.pos += t0;
    if (fx.slot) fx.slot += t0;
    if (fx.q_out) fx.q_out += (long)t0 * fx.Hq * fx.D;
    int rc;
    // k-blocks per LDS chunk (= half the weight ring) by token tiles.  Fewer = fewer registers = more waves per SIMD;
    // measured optima below (UMB_CB / UMB_CB2 / UMB_CB4 override: experiments)
    static const int cb1 = getenv("UMB_CB") ? atoi(getenv("UMB_CB")) : UMB_CB1;
    // 16 layers of the 70B at T = 31: 3.50 (CB 4) / 2.79 / 2.78 ms.  Round 4: int4 takes 2 -- the 8B-AWQ draft at 32 rows (one 4-wave
    // block per CU on gate/up and o: the ring depth IS the bytes in flight) 2.63 -> 2.54 ms per forward, 21 rows 2.49 -> 2.43; 4: 2.86
    static const int cb2 = getenv("UMB_CB2") ? atoi(getenv("UMB_CB2")) : (AWQ ? 2 : 1);
    static const int cb4 = getenv("UMB_CB4") ? atoi(getenv("UMB_CB4")) : 1;      // T = 64: 4.96 (CB 2) / 4.60 ms
#define UMB_LR(TTV, CBV) rc = launch_r<P, AWQ, TTV, CBV>(R, wp, meta, xx, ldx, oo, tn, T, N, K, S, epi, fx, st, tb)
    if (tn <= 16) { if (cb1 == 1) UMB_LR(1, 1); else if (cb1 == 2) UMB_LR(1, 2); else UMB_LR(1, 4); }
    else if (tn <= 32) { if (cb2 == 1) UMB_LR(2, 1); else if (cb2 == 2) UMB_LR(2, 2); else UMB_LR(2, 4); }
    else { if (cb4 == 1) UMB_LR(4, 1); else UMB_LR(4, 2); }
#undef UMB_LR
    if (rc) return rc;
  }
  return UMB_OK;
}

// C mirror of GemmFused (include/umbrella_hip.h: UmbGemmFused)
struct UmbGemmFusedC {
  const float* ssq_in; int ssq_groups; float ssq_dim; float eps; int pad0;
  unsigned* counters;
  void* h; void* hw; const void* norm_w; float* ssq_out; int ssq_out_stride; int pad1;
  const int* pos; const int* slot; const void* cosT; const void* sinT;
  void* q_out; void* k_cache; void* vt_cache; int Hq, Hkv, D, Lmax;
};

// out: fp32 [S][T][N] partials (epi 0/1) or 16-bit act [T][N/2] (epi 2 = fused SiLU*up, needs S == 1 and
// gate/up rows interleaved by the repack); epi 3 / 4: see UmbGemmFused in the header.
extern "C" int umb_gemm_fused(void* out, const void* x, int ldx, const void* wpacked, const void* meta, int T, int N,
                              int K, int awq, int S, int R, int epi, const UmbGemmFusedC* fxc, int dtype,
                              hipStream_t st) {
  // R: low byte = n-tiles per wave; bits 8..15 = n-tiles per block (0: 4 R), the plan's balance knob (umb_gemm_plan2)
  int tb = (R >> 8) & 0xff;
  R &= 0xff;
  static const int tb_env = getenv("UMB_TB") ? atoi(getenv("UMB_TB")) : 0;     // experiments
  if (tb_env > 0) tb = tb_env;
  if (N % 16 || K % 128 || T < 1 || S < 1 || epi < 0 || epi > 4 || (epi == EPI_SILU && S != 1)) return UMB_EINVAL;
  if ((long)S * T * N * 4 >= (1L << 31)) return UMB_EINVAL;              // fp32 partials are stored through a 32-bit buffer offset
  if (awq && (N % 64 || (R != 1 && (N / 16) % 2))) return UMB_EINVAL;
  if ((tb & 0x7f) > ((tb & 0x80) ? 8 : 4) * R) tb = 0;
  GemmFused fx = {};
  if (fxc) {
    fx.ssq_stride = fxc->pad0; fx.x_fm = fxc->pad1 & 1; fx.out_fm = (fxc->pad1 >> 1) & 1;
    fx.ssq_in = fxc->ssq_in; fx.ssq_groups = fxc->ssq_groups; fx.ssq_dim = fxc->ssq_dim; fx.eps = fxc->eps;
    fx.counters = fxc->counters; fx.h = (u16*)fxc->h; fx.hw = (u16*)fxc->hw; fx.norm_w = (const u16*)fxc->norm_w;
    fx.ssq_out = fxc->ssq_out; fx.ssq_out_stride = fxc->ssq_out_stride; fx.pos = fxc->pos; fx.slot = fxc->slot; fx.cosT = (const u16*)fxc->cosT;
    fx.sinT = (const u16*)fxc->sinT; fx.q_out = (u16*)fxc->q_out; fx.kc = (u16*)fxc->k_cache; fx.vt = (u16*)fxc->vt_cache;
    fx.Hq = fxc->Hq; fx.Hkv = fxc->Hkv; fx.D = fxc->D; fx.Lmax = fxc->Lmax;
  }
  if (epi >= EPI_QKV) {
    if (R != 1 || N % 64 || (S > 1 && !fx.counters)) return UMB_EINVAL;
    if (epi == EPI_RESID && (!fx.h || (fx.ssq_out && fx.ssq_out_stride < N / 64))) return UMB_EINVAL;
    if (epi == EPI_QKV && (!fx.pos || !fx.slot || !fx.q_out || !fx.kc || !fx.vt || fx.D % 4 || (fx.ssq_in && fx.ssq_groups % 4)))
      return UMB_EINVAL;
    if (epi == EPI_QKV && (fx.D % 32 || fx.Lmax % 32 || fx.Lmax < 32)) return UMB_EINVAL;   // fragment-ordered caches: whole tiles
  }
  if (fx.ssq_in && (fx.ssq_groups % 4 || fx.ssq_stride % 4)) return UMB_EINVAL;
  if ((fx.x_fm || fx.out_fm) && T > 64) return UMB_EINVAL;      // FM buffers hold one launch of <= 64 tokens
  // fp16 int4, T <= 64 (HBM-bound launches): the FOLDED form s * (sum q x - z sum x) in fp32 -- 63 ns of VALU per KiB
  // tile instead of 101 for the exact in-register dequant (gfx950 issues packed-fp16 and 3-operand integer VALU ops at half
  // rate, scripts/probe/valu_probe.hip), 70B gate/up 47.9 vs 52.6 us on the same box.  It computes the real-valued
  // (q - z) * s model without rounding each weight to fp16 first: within 2^-11 relative per weight of the reference's
  // awq_ext-dequantised weights (tests: 2e-3 of the row scale), and what the low-latency family always did.
  // T > 64 (matrix-pipe bound verify GEMMs) and UMB_DEQUANT=exact (or the older UMB_AWQ_EXACT=1): exact dequant,
  // W = fp16((q - z) * s) bit for bit, for EVERY row count -- one int4 arithmetic from the T = 1 row to the widest verify,
  // as the reference's single awq_ext kernel has (awq_utils.py:67-77).  Read per call (launch arguments are frozen into a
  // captured graph anyway), so a process can hold both forms.  The folded form exists for <= 64 rows only.
  const char* dq = getenv("UMB_DEQUANT");
  const bool awq_exact = (dq && dq[0] == 'e') || getenv("UMB_AWQ_EXACT") != nullptr;
  if (awq && dtype == UMB_F16 && (awq_exact || T > 64))
    return launch_tt<F16, 2>(wpacked, meta, (const u16*)x, ldx, (float*)out, T, N, K, R, S, epi, fx, st, tb);
  DISPATCH_DTYPE(dtype, {
    if (awq) return launch_tt<P, 1>(wpacked, meta, (const u16*)x, ldx, (float*)out, T, N, K, R, S, epi, fx, st, tb);
    return launch_tt<P, 0>(wpacked, meta, (const u16*)x, ldx, (float*)out, T, N, K, R, S, epi, fx, st, tb);
  })
}

extern "C" int umb_gemm(void* out, const void* x, int ldx, const void* wpacked, const void* meta, int T, int N, int K,
                        int awq, int S, int R, int epi, int dtype, hipStream_t st) {
  if (epi > EPI_SILU) return UMB_EINVAL;
  return umb_gemm_fused(out, x, ldx, wpacked, meta, T, N, K, awq, S, R, epi, nullptr, dtype, st);
}

extern "C" int umb_repack_dense(void* out, const void* w, int N, int K, int mode, int D, int rope_heads, int dtype,
                                hipStream_t st) {
  const int interleave = mode;
  if (N % 16 || K % 32) return UMB_EINVAL;
  const long total = (long)(N / 16) * (K / 32) * 64;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL((repack_dense_kernel<P>), grid, block, 0, st, (const u16*)w, (u32x4*)out, N, K, interleave, D, rope_heads);
  })
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// outw: N*K/2 bytes; meta: (N/16)*(K/128)*64 bytes
extern "C" int umb_awq_repack(void* outw, void* meta, const void* qweight, const void* qzeros, const void* scales,
                              int N, int K, int group, int mode, int D, int rope_heads, hipStream_t st) {
  const int interleave = mode;
  if (N % 64 || K % 128 || group != 128) return UMB_EINVAL;
  const long total = (long)(N / 16) * (K / 128) * 64;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipLaunchKernelGGL(repack_awq_kernel, grid, block, 0, st, (const unsigned*)qweight, (const unsigned*)qzeros,
                     (const u16*)scales, (u32x4*)outw, (unsigned char*)meta, N, K, interleave, D, rope_heads);
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}
