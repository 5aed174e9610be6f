// Epilogue selector and optional fused work of the GEMM family (gemm.hip: skinny / verify kernels; vgemm.hip: the wide
// verify kernel of round 6).  Device-side mirror of include/umbrella_hip.h UmbGemmFused.
#pragma once
#include "common.h"

enum { EPI_PARTIAL = 0, EPI_ROUND = 1, EPI_SILU = 2, EPI_QKV = 3, EPI_RESID = 4 };


// Optional fused work around the GEMM (all pointers may be null).  Passed by value.
//  * ssq_in : [T][ssq_groups] partial sums of squares of the producer's residual stream.  The RMSNorm weight is
//    already folded into x by the producer (x = h * w), the per-token factor rsqrt(mean(h^2) + eps) commutes with
//    the matmul and is applied to the outputs here (epi 1, 2, 3).
//  * epi 3 / 4 with S > 1: every split block publishes its fp32 partial tile, the LAST block to arrive on the
//    n-group's counter sums the S partials in split order (deterministic) and runs the epilogue -- no reduce kernel.
struct GemmFused {
  int ssq_stride;          // row stride of ssq_in (0: ssq_groups)
  int x_fm, out_fm;        // x / the SiLU output in FM (MFMA B-fragment) layout, lowlat.hip: the low-latency schedule's buffers
  const float* ssq_in; int ssq_groups; float ssq_dim; float eps;
  unsigned* counters;
  u16* h; u16* hw; const u16* norm_w; float* ssq_out; int ssq_out_stride;    // epi 4
  const int* pos; const int* slot; const u16* cosT; const u16* sinT;          // epi 3
  u16* q_out; u16* kc; u16* vt; int Hq, Hkv, D, Lmax;
};


// vgemm.hip: the wide (T > 64) int4 verify GEMM, fp16 activations, gemm.hip's tile order and epilogues
extern "C" int umb_vgemm_w_ok(int T, int N, int K, int S, int epi);
int umb_vgemm_w(const void* wp, const void* meta, const u16* x, int ldx, float* out, int T, int N, int K, int S, int epi,
                const GemmFused& fx, hipStream_t st);
