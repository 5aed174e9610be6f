// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of umbrella_amd.
#pragma once
// V^T cache rows are padded: a power-of-two row stride puts the 16 d-rows one MFMA fragment load touches on the
// same memory channel (measured 9-14 % on the attention kernel: T=257 55 -> 50 us, T=769 267 -> 233 us).  Same value
// as umbrella_hip.h.
#ifndef UMB_VT_PAD
#define UMB_VT_PAD 32
#endif
#define VT_LD(Lmax) ((Lmax) + UMB_VT_PAD)
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- KV cache element order inside one (layer, kv head) slab: MFMA FRAGMENT order (round 4).  The tree-attention kernels
// consume K as the A operand of S^T = K Q^T (16 keys x 32 d per instruction, lane (j, g) = key j, d 8g .. 8g+7) and V^T as the A
// operand of O^T = V^T P^T (16 d x 32 keys, lane (j, g) = d row j, keys 8g .. 8g+7).  A 32-key tile is stored as exactly those
// fragments -- K: 2 (key halves) x D/32 fragments, V^T: D/16 fragments, 1 KiB (64 lanes x 16 B) each, contiguous -- so every load
// instruction of the attention kernels reads ONE contiguous KiB instead of 16 rows x 64 B (profiles/r04_attn_frag_probe.txt:
// T = 13 verify 6.8 -> 5.2 us at 100 keys, 13.5 -> 10.1 at 1000; T = 257 24.2 -> 17.2).  Writers (q/k/v epilogues, kv append,
// compaction) address single elements through these two functions; slab strides are unchanged (K: Lmax D, V^T: D VT_LD(Lmax)
// elements per kv head, Lmax a multiple of 32).  Python mirror: umbrella_amd/attn/cache.py.
// ---- activations in FM (MFMA B-fragment) order, [K/32][TT token tiles][64 lanes][8]: element offset of (token t, feature f).
// One B fragment (16 tokens x 32 features) is one contiguous KiB (lowlat.hip header; umb_to_fm / umb_from_fm convert).
__host__ __device__ __forceinline__ long fm_off(int t, int f, int TT) {
  return ((((long)(f >> 5) * TT + (t >> 4)) * 64 + ((f >> 3) & 3) * 16 + (t & 15)) << 3) + (f & 7);
}
//   key p, feature d of K   : tile p / 32, key-in-tile kk = p % 32 -> half s = (kk / 4) % 2, lane row j = 4 (kk / 8) + kk % 4
__host__ __device__ __forceinline__ long kc_off(int p, int d, int D) {
  const int kk = p & 31;
  const int s = (kk >> 2) & 1, j = ((kk >> 3) << 2) | (kk & 3);
  return ((((long)(p >> 5) * 2 + s) * (D >> 5) + (d >> 5)) * 64 + ((d >> 3) & 3) * 16 + j) * 8 + (d & 7);
}
//   feature d, key p of V^T : tile p / 32, fragment d / 16, lane (j = d % 16, g = (p % 32) / 8), element p % 8
__host__ __device__ __forceinline__ long vt_off(int d, int p, int D) {
  return (((long)(p >> 5) * (D >> 4) + (d >> 4)) * 64 + ((p >> 3) & 3) * 16 + (d & 15)) * 8 + (p & 7);
}

// cache policy of stores whose data the NEXT kernel reads on other XCDs (buffer-store aux bits: 16 = sc1, write-through)
#ifndef UMB_HANDOFF_AUX
#define UMB_HANDOFF_AUX 16
#endif

#define UMB_OK 0
#define UMB_EINVAL (-22)
#define UMB_EHIP (-5)

#define UMB_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return UMB_EHIP;        \
  } while (0)

enum { UMB_F16 = 0, UMB_BF16 = 1 };

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- 16-bit storage types: raw bits + conversion policy -------------------
struct BF16 {
  typedef bf16x8 frag;
  static constexpr unsigned MAGIC = 0x43004300u;   // bf16(128 + q) for q < 128
  static constexpr float MAGIC_OFF = 128.0f;
  static constexpr unsigned ONE2 = 0x3F803F80u;    // packed (1.0, 1.0)
  __device__ __forceinline__ static float to_f(u16 b) { return __uint_as_float(((unsigned)b) << 16); }
  __device__ __forceinline__ static u16 from_f(float f) {   // round-to-nearest-even
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
  }
  __device__ __forceinline__ static f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
struct F16 {
  typedef f16x8 frag;
  static constexpr unsigned MAGIC = 0x64006400u;   // fp16(1024 + q)
  static constexpr float MAGIC_OFF = 1024.0f;
  static constexpr unsigned ONE2 = 0x3C003C00u;
  __device__ __forceinline__ static float to_f(u16 b) { return (float)__builtin_bit_cast(_Float16, b); }
  __device__ __forceinline__ static u16 from_f(float f) { return __builtin_bit_cast(u16, (_Float16)f); }
  __device__ __forceinline__ static f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

template <typename P> __device__ __forceinline__ float rnd(float f) { return P::to_f(P::from_f(f)); }
// x * y rounded to the model dtype and made opaque to the optimiser.  hipcc compiles device code with -ffp-contract=fast
// and folds fptrunc(fpext(a) * fpext(c)) back to half arithmetic: rnd(a * c) + rnd(b * s) on fp16 then becomes
// v_fma_f16(a, c, b * s) -- ONE rounding where eager torch (apply_rotary_pos_emb, model_utils.py:50-51) performs two.
// The empty asm keeps the rounded product a value of its own.
template <typename P> __device__ __forceinline__ float mul_rnd(float x, float y) {
  float p = rnd<P>(x * y);
  asm volatile("" : "+v"(p));
  return p;
}
// round(x * y) for fp32 operands, as TWO roundings (fp32 product, then the model dtype) whatever the surrounding code: left
// alone, hipcc picks v_fma_mixlo_f16 (one rounding) in some kernels and v_mul_f32 + v_cvt (two) in others -- the GEMV
// launches and the persistent chain (chain.hip) must agree bit for bit, and they differed at ~1 output in 8000.
template <typename P> __device__ __forceinline__ float rnd_prod(float x, float y) {
  float p = x * y;
  asm volatile("" : "+v"(p));
  return rnd<P>(p);
}
template <typename P> __device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return (unsigned)P::from_f(lo) | ((unsigned)P::from_f(hi) << 16);
}
template <typename P> __device__ __forceinline__ float lo_f(unsigned w) { return P::to_f((u16)(w & 0xffffu)); }
template <typename P> __device__ __forceinline__ float hi_f(unsigned w) { return P::to_f((u16)(w >> 16)); }

// ---- wave64 / block reductions ----------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over a block of NT threads (NT multiple of 64, <= 1024); result broadcast to all threads
template <int NT> __device__ __forceinline__ float block_sum(float v, float* red /* >= NT/64 floats of LDS */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}

#define DISPATCH_DTYPE(dt, ...)                       \
  if ((dt) == UMB_BF16) { using P = BF16; __VA_ARGS__ } \
  else if ((dt) == UMB_F16) { using P = F16; __VA_ARGS__ } \
  else return UMB_EINVAL;
