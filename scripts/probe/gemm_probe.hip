// Lab kernel 2: the int4 x fp16 skinny GEMM inner loop rebuilt feature by feature on top of the pure streaming loop
// (stream_probe.hip, 6.4 TB/s on the gate/up footprint), to find which ingredient breaks the overlap with the
// weight stream.  Results are garbage by construction (random bytes); only the timing matters.
//   FEAT bit 0: MFMAs (4 per tile per k-block, TT = 1)        bit 1: exact fp16 dequant (13 VALU per dword)
//        bit 2: B fragments read from LDS (else constant regs)  bit 3: x staged per chunk: global -> regs -> LDS + barrier
//        bit 4: metadata as per-k-block 64-byte loads           bit 5: metadata as one half-wave load per chunk via LDS
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int FEAT, int R, int CB>
__global__ __launch_bounds__(256) void gprobe(const u32x4* __restrict__ wp, const unsigned char* __restrict__ meta,
                                              const u32x4* __restrict__ x, int KB, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);                       // [2][CB*4 fragments][64]
  constexpr int F = CB * 4, FPW = F / 4, PF = 2 * CB;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int nt0 = (blockIdx.x * 4 + wv) * R;
  unsigned char* ms = smem + 2 * F * 1024 + wv * 2048;
  __amdgpu_buffer_rsrc_t rw[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int nt = nt0 + r;
    rw[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wp + (((long)(nt >> 2) * KB) * 4 + (nt & 3)) * 64), 0,
                                              (unsigned)(KB - 1) * 4096u + 1024u, 0x00020000);
  }
  const long tile0 = ((long)(nt0 >> 2) * KB) * 4 + (nt0 & 3);
  const auto rmeta = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(meta + tile0 * 64), 0,
                                                       (unsigned)(KB - 1) * 256u + R * 64u, 0x00020000);
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(x), 0, (unsigned)KB * 4096u, 0x00020000);
  const int voff_m = lane < R * CB * 4 ? (lane / (R * 4)) * 256 + (lane % (R * 4)) * 16 : (int)0x80000000;
  f32x4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[PF][R];
  unsigned m1[PF][R];
  u32x4 xr[2][FPW], mr[2];
  u32x4 bc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) { bc[s] = u32x4{0x3c003c00u + lane, 0x3c003c00u, 0x38003800u, 0x3c003c00u + s}; asm volatile("" : "+v"(bc[s])); }
  auto sload = [&](int slot, int kb) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      a[slot][r] = __builtin_amdgcn_raw_buffer_load_b128(rw[r], lane * 16 + kb * 4096, 0, 2);
      if (FEAT & 16) m1[slot][r] = __builtin_amdgcn_raw_buffer_load_b32(rmeta, r * 64 + j * 4 + kb * 256, 0, 0);
    }
  };
  auto load_x = [&](u32x4 (&xq)[FPW], int c) {
#pragma unroll
    for (int i = 0; i < FPW; ++i) xq[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, lane * 16 + ((c * CB * 4) + i * 4 + wv) * 1024, 0, 0);
  };
  auto compute = [&](int slot, const u32x4* xc, int H, int kl) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      unsigned mm = 0x3c002c00u;
      if (FEAT & 16) mm = m1[slot][r];
      if (FEAT & 32) mm = *reinterpret_cast<const unsigned*>(ms + H * 1024 + (kl * R + r) * 64 + j * 4);
      u32x4 wf[4];
      if (FEAT & 64) {
        // folded: codes with magic exponents straight into the MFMA; C-in = -(1024 S_lo + 64 S_hi) of this k-block and
        // token (LDS, computed once per block); fp32 epilogue out += s * (A - z * S) per 128-k group
        const u32x4 m4 = *reinterpret_cast<const u32x4*>(ms + H * 1024 + (kl * R + r) * 64 + (lane >> 4) * 16);
        const float* sums = reinterpret_cast<const float*>(smem + 2 * F * 1024 + 4 * 2048) + ((H * CB + kl) * 2) * 16;
        const float np = sums[j], sx = sums[16 + j];
        f32x4 ga = {np, np, np, np};
        unsigned magic_lo = 0x64006400u, magic_hi = 0x54005400u;
        asm volatile("" : "+v"(magic_lo), "+v"(magic_hi));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const unsigned w = a[slot][r][s], w8 = w >> 8;
          u32x4 f;
          f[0] = (w & 0x000F000Fu) | magic_lo; f[1] = (w & 0x00F000F0u) | magic_hi;
          f[2] = (w8 & 0x000F000Fu) | magic_lo; f[3] = (w8 & 0x00F000F0u) | magic_hi;
          const u32x4 b = (FEAT & 4) ? xc[(kl * 4 + s) * 64 + lane] : bc[s];
          ga = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, f), __builtin_bit_cast(h8, b), ga, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 sc = __builtin_bit_cast(_Float16, (unsigned short)(m4[e] & 0xffffu));
          const _Float16 zf = __builtin_bit_cast(_Float16, (unsigned short)(m4[e] >> 16));
          const float t = __builtin_fmaf(-(float)zf, sx, ga[e]);
          acc[r][e] = __builtin_fmaf((float)sc, t, acc[r][e]);
        }
        continue;
      }
      if (FEAT & 2) {
        const _Float16 sc = __builtin_bit_cast(_Float16, (unsigned short)(mm & 0xffffu));
        const _Float16 zf = __builtin_bit_cast(_Float16, (unsigned short)(mm >> 16));
        const h2 s2 = {sc, sc};
        const _Float16 nz = -((_Float16)1024.0f + zf), nz16 = -((_Float16)64.0f + zf);
        const h2 nz2 = {nz, nz}, nz16_2 = {nz16, nz16};
        const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
        unsigned magic = 0x64006400u;
        asm volatile("" : "+v"(magic));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const unsigned w = a[slot][r][s], w8 = w >> 8;
          const h2 t0 = __builtin_bit_cast(h2, (w & 0x000F000Fu) | magic);
          const h2 t1 = __builtin_bit_cast(h2, (w & 0x00F000F0u) | magic);
          const h2 t2 = __builtin_bit_cast(h2, (w8 & 0x000F000Fu) | magic);
          const h2 t3 = __builtin_bit_cast(h2, (w8 & 0x00F000F0u) | magic);
          wf[s][0] = __builtin_bit_cast(unsigned, (t0 + nz2) * s2);
          wf[s][1] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(t1, sixteenth, nz16_2) * s2);
          wf[s][2] = __builtin_bit_cast(unsigned, (t2 + nz2) * s2);
          wf[s][3] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(t3, sixteenth, nz16_2) * s2);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) { const unsigned w = a[slot][r][s]; wf[s] = u32x4{w, w >> 8, w ^ mm, w}; }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4 b = (FEAT & 4) ? xc[(kl * 4 + s) * 64 + lane] : bc[s];
        if (FEAT & 1) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, wf[s]), __builtin_bit_cast(h8, b), acc[r], 0, 0, 0);
        else acc[r][0] += __builtin_bit_cast(float, wf[s][0] ^ wf[s][1] ^ wf[s][2] ^ wf[s][3] ^ b[0]);
      }
    }
  };
  auto chunk = [&](int c, auto half) {
    constexpr int H = decltype(half)::value;
    if (FEAT & 8) {
#pragma unroll
      for (int i = 0; i < FPW; ++i) xs[((c & 1) * F + i * 4 + wv) * 64 + lane] = xr[H][i];
    }
    if (FEAT & 32) *reinterpret_cast<u32x4*>(ms + H * 1024 + lane * 16) = mr[H];
    if (FEAT & (8 | 32)) __syncthreads();
    if (FEAT & 8) load_x(xr[H], c + 2);
    if (FEAT & 32) mr[H] = __builtin_amdgcn_raw_buffer_load_b128(rmeta, voff_m + ((c + 2) * CB) * 256, 0, 0);
    const u32x4* xc = xs + ((FEAT & 8) ? (c & 1) * F * 64 : 0);
    if (FEAT & 128) {
      // wave w: the two x-only sums of k-block w of this chunk (CB == 4), for the whole block
      float* sums = reinterpret_cast<float*>(smem + 2 * F * 1024 + 4 * 2048) + ((H * CB + wv) * 2) * 16;
      const u32x4 ones = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
      const u32x4 negc = {0xE400E400u, 0xD400D400u, 0xE400E400u, 0xD400D400u};
      f32x4 sx = {0.f, 0.f, 0.f, 0.f}, np = {0.f, 0.f, 0.f, 0.f};
      if (wv < CB) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const u32x4 b = xc[(wv * 4 + s) * 64 + lane];
          sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ones), __builtin_bit_cast(h8, b), sx, 0, 0, 0);
          np = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, negc), __builtin_bit_cast(h8, b), np, 0, 0, 0);
        }
        if (lane < 16) { sums[lane] = np[0]; sums[16 + lane] = sx[0]; }
      }
      __syncthreads();
    }
#pragma unroll
    for (int kl = 0; kl < CB; ++kl) {
      compute(H * CB + kl, xc, H, kl);
      sload(H * CB + kl, c * CB + kl + PF);
    }
  };
  // LDS content for the static-B variant
  for (int i = threadIdx.x; i < 2 * F * 64; i += 256) xs[i] = u32x4{0x3c003c00u, 0x3c003c00u + i, 0x38003800u, 0x3c003c00u};
  __syncthreads();
  if (FEAT & 8) load_x(xr[0], 0);
  if (FEAT & 32) mr[0] = __builtin_amdgcn_raw_buffer_load_b128(rmeta, voff_m, 0, 0);
#pragma unroll
  for (int i = 0; i < CB; ++i) sload(i, i);
  if (FEAT & 8) load_x(xr[1], 1);
  if (FEAT & 32) mr[1] = __builtin_amdgcn_raw_buffer_load_b128(rmeta, voff_m + CB * 256, 0, 0);
#pragma unroll
  for (int i = CB; i < PF; ++i) sload(i, i);
  const int nchunks = KB / CB;
  for (int c = 0; c < nchunks; c += 2) {
    chunk(c, std::integral_constant<int, 0>{});
    chunk(c + 1, std::integral_constant<int, 1>{});
  }
  float v = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) v += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
  if (v == 12345.678f) out[0] = v;
}

#define GP(FEAT, R, CB)                                                                                       \
  if (feat == FEAT && r == R && cb == CB) {                                                                   \
    const int lds = 2 * CB * 4 * 1024 + 4 * 2048 + 2048;                                                             \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gprobe<FEAT, R, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((gprobe<FEAT, R, CB>), dim3(tiles / R / 4), dim3(256), lds, st, (const u32x4*)w, (const unsigned char*)meta, \
                       (const u32x4*)x, kb, (float*)out);                                                     \
    return 0;                                                                                                 \
  }
extern "C" int gprobe_launch(const void* w, const void* meta, const void* x, int tiles, int kb, int feat, int r, int cb,
                             void* out, hipStream_t st) {
  GP(0, 2, 4) GP(1, 2, 4) GP(3, 2, 4) GP(7, 2, 4) GP(15, 2, 4) GP(31, 2, 4) GP(47, 2, 4) GP(2, 2, 4) GP(5, 2, 4) GP(13, 2, 4)
  GP(109, 2, 4) GP(237, 2, 4) GP(205, 2, 4) GP(77, 2, 4) GP(237, 1, 4) GP(47, 2, 2) GP(47, 1, 4) GP(47, 1, 2) GP(15, 2, 2) GP(15, 1, 4) GP(45, 2, 4) GP(11, 2, 4)
  return 1;
}
