// Feasibility probe v2 (round 3): latency-optimal GEMV for <= 4 token rows.  Every load of the kernel -- weights AND the
// activation pieces a lane needs -- is issued at kernel start (no LDS staging, no barrier before the first FMA); a wave owns
// whole rows (K <= 2048: the full row; K = 8192: a 2048-element slice, four slices summed through LDS).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));

template <int BF> __device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
  if constexpr (BF) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a), __builtin_bit_cast(b2, b), c, false);
  else return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}

// KS = k-slices per row (1 or 4); a wave handles RW rows x 4 chunks of 512 elements of its slice, GW row groups in sequence
template <int BF, int RW, int KS>
__global__ __launch_bounds__(512) void gv2_kernel(const u32x4* __restrict__ w, const uint16_t* __restrict__ x, int T, int N,
                                                  int K, int RB, float* __restrict__ out) {
  __shared__ float red[8][4][4];                               // [wave][row][token] partials of the k-slices
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ks = KS == 1 ? 0 : (wv & (KS - 1));                // k-slice of this wave
  const int rgw = KS == 1 ? wv : (wv / KS);                    // row-group slot of this wave
  const int nslot = 8 / KS;
  const int ngrp = RB / RW;
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(w), 0, 0xffffffffu, 0x00020000);
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, (unsigned)(T * K * 2), 0x00020000);
  u32x4 xr[4][4], wr[2][RW][4];
  const int kbase = ks * 2048;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      xr[t][kc] = __builtin_amdgcn_raw_buffer_load_b128(rx, (kbase + kc * 512 + lane * 8) * 2, t * K * 2, 0);   // rows >= T: out of range -> 0
  auto issue = [&](auto bc, int g) {
    constexpr int B = decltype(bc)::value;
    const int row0 = blockIdx.x * RB + g * RW;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        wr[B][r][kc] = __builtin_amdgcn_raw_buffer_load_b128(rw, (kbase + kc * 512 + lane * 8) * 2, (row0 + r) * K * 2, 2);
  };
  auto consume = [&](auto bc, int gg) {
    constexpr int B = decltype(bc)::value;
    float acc[RW][4];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float a = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
          for (int e = 0; e < 4; ++e) a = dot2<BF>(wr[B][r][kc][e], xr[t][kc][e], a);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        acc[r][t] = a;
      }
    if constexpr (KS == 1) {
      if (lane < RW * 4) {
        const int r = lane >> 2, t = lane & 3;
        float v = 0.f;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) v = (rr == r && tt == t) ? acc[rr][tt] : v;
        if (t < T) out[(long)t * N + blockIdx.x * RB + gg * RW + r] = v;
      }
    } else {
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int t = 0; t < 4; ++t) red[wv][r][t] = acc[r][t];
      }
      __syncthreads();
      if (ks == 0 && lane < RW * 4) {
        const int r = lane >> 2, t = lane & 3;
        float v = 0.f;
        for (int s = 0; s < KS; ++s) v += red[rgw * KS + s][r][t];
        if (t < T) out[(long)t * N + blockIdx.x * RB + gg * RW + r] = v;
      }
    }
  };
  int g = rgw;
  if (g < ngrp) issue(std::integral_constant<int, 0>{}, g);
  for (; g < ngrp; g += 2 * nslot) {
    if (g + nslot < ngrp) issue(std::integral_constant<int, 1>{}, g + nslot);
    consume(std::integral_constant<int, 0>{}, g);
    if (g + nslot < ngrp) {
      if (g + 2 * nslot < ngrp) issue(std::integral_constant<int, 0>{}, g + 2 * nslot);
      consume(std::integral_constant<int, 1>{}, g + nslot);
    }
  }
}

extern "C" int gv2_probe(void* out, const void* x, const void* w, int T, int N, int K, int RB, int RW, int bf16, hipStream_t st) {
  if (N % RB || RB % RW || T < 1 || T > 4 || bf16) return 1;
#define GO(RWV, KSV) hipLaunchKernelGGL((gv2_kernel<0, RWV, KSV>), dim3(N / RB), dim3(512), 0, st, (const u32x4*)w, (const uint16_t*)x, T, N, K, RB, (float*)out)
  if (K == 2048) { if (RW == 1) GO(1, 1); else if (RW == 2) GO(2, 1); else return 1; }
  else if (K == 8192) { if (RW == 4) GO(4, 4); else if (RW == 2) GO(2, 4); else return 1; }
  else return 1;
  return hipGetLastError() != hipSuccess;
}
