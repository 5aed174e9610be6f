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
