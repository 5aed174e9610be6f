// LDS-DMA stream probe (round 5): what rate does a per-CU loader wave reach, and what limits it?
// One workgroup per CU; NL loader waves stream this CU's share of a big buffer HBM -> LDS ring with
// global_load_lds_dwordx4 (16 KiB slots, 4 pieces per M0 setting via the immediate offset), nothing consumes (slots are
// overwritten round robin): the pure fill cadence.  Variants: loader waves per CU (each has its own 6-bit vmcnt budget),
// slots in flight per wave, nt / default policy, CU-contiguous vs slot-interleaved placement, skewed slot order.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/ldsdma_probe.hip -o gpurun_out/ldsdma_probe && gpurun_out/ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) char lds_char;
enum { SLOT = 16384 };

template <int NT, int DEPTH>   // NT: 1 = nt policy; DEPTH: slots in flight per loader wave (2..4)
__global__ __launch_bounds__(256) void fill_kernel(const char* base, int slots_per_cu, int nl, int ring, int layout, int skew,
                                                   unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cu = blockIdx.x, ncu = gridDim.x;
  if (wv >= nl) return;
  const unsigned ring0 = (unsigned)(size_t)(lds_char*)smem;
  const unsigned voff = (unsigned)lane * 16u;
  // this wave's slots: s = wv, wv + nl, ...
  int cnt = 0;
  for (int s = wv; s < slots_per_cu; s += nl) {
    int sk = skew ? (s + cu) % slots_per_cu : s;
    // layout 0: CU-contiguous (CU c owns slots [c * n, (c + 1) * n)); 1: slot-interleaved (slot s of all CUs contiguous)
    const size_t idx = layout == 0 ? (size_t)cu * slots_per_cu + sk : (size_t)sk * ncu + cu;
    const char* sp = base + idx * SLOT;
    const unsigned dst = ring0 + (unsigned)((s % ring) * SLOT);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned keep;
      if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sp + j * 4096), "s"(dst + j * 4096) : "memory");
      else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sp + j * 4096), "s"(dst + j * 4096) : "memory");
    }
    ++cnt;
    if (cnt >= DEPTH) {
      if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && sink && cu == 100000) sink[0] = ((unsigned*)smem)[0];
}

template <int NT, int DEPTH> static float run(const char* buf, size_t bytes, int slots_per_cu, int nl, int ring, int layout, int skew) {
  static bool done = false;
  if (!done) { CK(hipFuncSetAttribute((const void*)fill_kernel<NT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); done = true; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t per = (size_t)256 * slots_per_cu * SLOT;
  const int nreg = (int)(bytes / per);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((fill_kernel<NT, DEPTH>), dim3(256), dim3(256), ring * SLOT, 0, buf + (size_t)(i % nreg) * per, slots_per_cu, nl, ring, layout, skew, nullptr);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((fill_kernel<NT, DEPTH>), dim3(256), dim3(256), ring * SLOT, 0, buf + (size_t)(i % nreg) * per, slots_per_cu, nl, ring, layout, skew, nullptr);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;   // us per launch
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)2 << 30;
  char* buf;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  printf("LDS-DMA fill probe: 256 workgroups (1 per CU), 16 KiB slots, us per launch -> TB/s, us per slot per CU\n");
  std::vector<int> slot_list = {29, 116};  // 29 = one 1B layer's chain; 116 = four layers' worth
  if (argc > 1) { slot_list.clear(); for (int i = 1; i < argc; ++i) slot_list.push_back(atoi(argv[i])); }   // e.g. 9 11 30 61: the 70B o / qkv / down / gate-up footprints
  const bool only = getenv("ONLY") != nullptr;   // ONLY=1: the shipped variant alone (under rocprofv3: its kernel durations per footprint)
  for (int slots : slot_list) {
    const double mb = 256.0 * slots * SLOT / 1e6;
    printf("-- %d slots per CU (%.1f MB per launch)\n", slots, mb);
    struct V { const char* name; int nt, depth, nl, ring, layout, skew; };
    std::vector<V> vs = {
      {"nt  depth4 1 loader  contiguous       ", 1, 4, 1, 8, 0, 0},
      {"nt  depth4 1 loader  contiguous skewed", 1, 4, 1, 8, 0, 1},
      {"nt  depth4 1 loader  interleaved      ", 1, 4, 1, 8, 1, 0},
      {"nt  depth3 1 loader  contiguous skewed", 1, 3, 1, 8, 0, 1},
      {"nt  depth2 1 loader  contiguous skewed", 1, 2, 1, 8, 0, 1},
      {"def depth4 1 loader  contiguous skewed", 0, 4, 1, 8, 0, 1},
      {"nt  depth4 2 loaders contiguous skewed", 1, 4, 2, 8, 0, 1},
      {"nt  depth4 2 loaders interleaved      ", 1, 4, 2, 8, 1, 0},
      {"nt  depth2 2 loaders contiguous skewed", 1, 2, 2, 8, 0, 1},
      {"nt  depth4 4 loaders contiguous skewed", 1, 4, 4, 8, 0, 1},
      {"nt  depth2 4 loaders contiguous skewed", 1, 2, 4, 8, 0, 1},
    };
    if (only) vs = {vs[1]};
    for (auto& v : vs) {
      float us = 0;
      if (v.nt && v.depth == 4) us = run<1, 4>(buf, bytes, slots, v.nl, v.ring, v.layout, v.skew);
      else if (v.nt && v.depth == 3) us = run<1, 3>(buf, bytes, slots, v.nl, v.ring, v.layout, v.skew);
      else if (v.nt && v.depth == 2) us = run<1, 2>(buf, bytes, slots, v.nl, v.ring, v.layout, v.skew);
      else us = run<0, 4>(buf, bytes, slots, v.nl, v.ring, v.layout, v.skew);
      printf("%s: %7.2f us  %5.2f TB/s  %5.2f us per slot\n", v.name, us, mb / us, us / slots);
    }
  }
  return 0;
}
