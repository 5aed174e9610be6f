// Lab kernel: which property of the weight-streaming GEMMs costs HBM rate?  Pure loads in the GEMM's access pattern
// (AWQ tile order [N/64][K/128][4 x 1 KiB]), with the knobs of the real kernels: pieces in flight per wave (U), 1 or
// 2 KiB per piece (LPP), occupancy (LDS bytes), a block barrier per chunk of 4 pieces, extra VALU work per piece.
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U, int LPP, int BAR, int VALU>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ p, int npieces, unsigned* __restrict__ sink) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  constexpr int WPG = 4 / LPP;                               // waves per 4 KiB group row
  const u32x4* base = p + (wid / WPG) * (long)npieces * 256 + (wid % WPG) * LPP * 64 + lane;
  u32x4 acc = {0u, 0u, 0u, 0u};
  u32x4 v[U][LPP];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int l = 0; l < LPP; ++l) v[u][l] = __builtin_nontemporal_load(base + (long)u * 256 + l * 64);
  auto use = [&](u32x4 t) {
    // VALU full-rate ops per dword (the exact int4 dequant is 13 per dword: 1 shift, 4 and-or, 8 packed fp16)
#pragma unroll
    for (int k = 0; k < VALU; ++k) t = (t ^ acc) + (t >> 3);
    acc ^= t;
  };
  // predicate-free steady state (npieces % U == 0): a branch inside the ring makes the compiler drain vmcnt to 0
  for (int i = U; i < npieces; i += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (BAR && (u % 4) == 0) __syncthreads();
#pragma unroll
      for (int l = 0; l < LPP; ++l) {
        use(v[u][l]);
        v[u][l] = __builtin_nontemporal_load(base + (long)(i + u) * 256 + l * 64);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int l = 0; l < LPP; ++l) use(v[u][l]);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) *sink = smem[0];
}

#define LAUNCH(U, LPP, BAR, VALU)                                                                      \
  if (u == U && lpp == LPP && bar == BAR && valu == VALU) {                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<U, LPP, BAR, VALU>),                      \
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);                              \
    hipLaunchKernelGGL((probe<U, LPP, BAR, VALU>), dim3(blocks), dim3(256), lds, st, (const u32x4*)p, npieces, (unsigned*)sink); \
    return 0;                                                                                          \
  }

extern "C" int probe_launch(const void* p, long tiles, int npieces, int u, int lpp, int bar, int valu, int lds, void* sink,
                            hipStream_t st) {
  const int blocks = (int)(tiles / lpp / 4);
  LAUNCH(4, 1, 0, 0) LAUNCH(8, 1, 0, 0) LAUNCH(16, 1, 0, 0) LAUNCH(4, 2, 0, 0) LAUNCH(8, 2, 0, 0) LAUNCH(16, 2, 0, 0)
  LAUNCH(8, 2, 1, 0) LAUNCH(8, 1, 1, 0) LAUNCH(8, 2, 0, 4) LAUNCH(8, 2, 1, 4) LAUNCH(8, 2, 0, 7) LAUNCH(8, 2, 1, 7)
  LAUNCH(8, 1, 0, 4) LAUNCH(8, 1, 0, 7) LAUNCH(16, 1, 0, 4) LAUNCH(16, 1, 1, 7) LAUNCH(16, 1, 0, 7) LAUNCH(4, 2, 0, 7)
  return 1;
}
