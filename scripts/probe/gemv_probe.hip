// Feasibility probe (round 3): row-streaming GEMV for <= 4 token rows, dense 16-bit weights in plain [N][K] order.
// A wave owns whole rows (RW at a time), every lane 16 B of each 512-element chunk; x sits in LDS; v_dot2c accumulates
// in fp32; a 64-lane butterfly finishes a row.  Row granularity means N = 2048 fills 256 CUs (8 rows each) where the
// MFMA kernels' 16-row tiles give 128 workgroups.  Built stand-alone: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));

template <int BF> __device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
  if constexpr (BF) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a), __builtin_bit_cast(b2, b), c, false);
  else return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);
}

template <int BF, int RW, int KC>
__global__ __launch_bounds__(512) void gv_probe_kernel(const u32x4* __restrict__ w, const uint16_t* __restrict__ x, int T,
                                                       int N, int K, int RB, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* xs = reinterpret_cast<u32x4*>(smem);                    // [4][K / 8]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ngrp = RB / RW;
  const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(w), 0, 0xffffffffu, 0x00020000);
  u32x4 wr[2][RW][KC];
  auto issue = [&](auto bc, int g) {
    constexpr int B = decltype(bc)::value;
    const int row0 = blockIdx.x * RB + g * RW;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        wr[B][r][kc] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + kc * 1024, (row0 + r) * K * 2, 2);
  };
  int g = wv;
  if (g < ngrp) issue(std::integral_constant<int, 0>{}, g);
  // stage x (rows >= T are zero)
  const int K8 = K / 8;
  for (int i = threadIdx.x; i < 4 * K8; i += 512) {
    const int t = i / K8;
    xs[i] = t < T ? reinterpret_cast<const u32x4*>(x)[i] : u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  auto consume = [&](auto bc, int gg) {
    constexpr int B = decltype(bc)::value;
    float acc[RW][4];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[r][t] = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const u32x4 xv = xs[t * K8 + kc * 64 + lane];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r][t] = dot2<BF>(wr[B][r][kc][e], xv[e], acc[r][t]);
      }
      __builtin_amdgcn_sched_barrier(0);       // keep the LDS reads of later chunks from being hoisted (registers)
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float a = acc[r][t];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
        if (lane == 0 && t < T) out[(long)t * N + blockIdx.x * RB + gg * RW + r] = a;
      }
  };
  for (; g < ngrp; g += 16) {
    if (g + 8 < ngrp) issue(std::integral_constant<int, 1>{}, g + 8);
    consume(std::integral_constant<int, 0>{}, g);
    if (g + 8 < ngrp) {
      if (g + 16 < ngrp) issue(std::integral_constant<int, 0>{}, g + 16);
      consume(std::integral_constant<int, 1>{}, g + 8);
    }
  }
}

extern "C" int gv_probe(void* out, const void* x, const void* w, int T, int N, int K, int RB, int RW, int bf16, hipStream_t st) {
  const int KC = K / 512;
  if (K % 512 || N % RB || RB % RW || T < 1 || T > 4) return 1;
  const size_t smem = (size_t)4 * K * 2;
#define GO(BF, RWV, KCV)                                                                                          \
  hipLaunchKernelGGL((gv_probe_kernel<BF, RWV, KCV>), dim3(N / RB), dim3(512), smem, st, (const u32x4*)w,        \
                     (const uint16_t*)x, T, N, K, RB, (float*)out)
#define GO_R(BF, KCV) if (RW == 1) GO(BF, 1, KCV); else if (RW == 2) GO(BF, 2, KCV); else return 1
#define GO_K(BF) if (KC == 4) { GO_R(BF, 4); } else if (KC == 16) { if (RW == 1) GO(BF, 1, 16); else return 1; } else return 1
  if (bf16) { GO_K(1); } else { GO_K(0); }
  return hipGetLastError() != hipSuccess;
}
