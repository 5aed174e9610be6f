// Whole-model forward: launches the per-layer kernel chain from native code so a
// verify / draft step costs one C call (and one hipGraph node sequence when captured).
//
// Restates Llama*.inference + layer_compute (umbrella/models/llama.py:75-134, 262-322,
// 461-502) and the offload loop (llama.py:196-219) as a kernel schedule:
//   embed+prep -> L x [ qkv gemm(+1/rms, RoPE, KV append) -> tree attention(+split merge) ->
//   o gemm(+residual, next norm weight folded, sum of squares) -> gate/up gemm(+1/rms, SiLU*up) ->
//   down gemm(+residual, ...) ] -> lm_head gemm(+1/rms, fp32 logits)          = 5 launches per layer
#include "../../include/umbrella_hip.h"
#include "common.h"
#include <cstdlib>
#include <cstdio>

// UMB_DEBUG_SYNC=1: synchronise after every launch and name the first one that fails (eager launches only)
static inline int dbg_sync(const char* what, hipStream_t st) {
  static const bool on = getenv("UMB_DEBUG_SYNC") != nullptr;
  if (!on) return 0;
  const hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) { fprintf(stderr, "[umb] %s failed: %s\n", what, hipGetErrorString(e)); return UMB_EHIP; }
  fprintf(stderr, "[umb] ok %s\n", what);
  return 0;
}
#define CK(x) do { int rc__ = (x); if (rc__) return rc__; rc__ = dbg_sync(#x, st); if (rc__) return rc__; } while (0)

// split count actually used: the plan's S (a function of the layer shape only, so T <= 64 results are batch
// invariant); the wide verify (T > 64) takes the S <= plan that fills the verify kernel's block slots once (gemm.hip)
// row_reduce: the partials are summed by reduce_residual_norm, one block per token row, which reads them at a single
// CU's ~60 GB/s (0.55 us per split at N = 8192) -- worth more than the last splits give the GEMM when K is short
// (70B o-proj, T = 13: S = 8 -> 4 costs the GEMM +0.4 us and saves the reduce 2.2 us; 70B down, K = 28672, keeps 8;
// 8B-AWQ down keeps 8 of its planned 16).  A function of the layer shape only, like the plan itself.
static inline int eff_s(const UmbLinear& l, int T, bool row_reduce = false) {
  int S = umb_gemm_wide_split(T, l.N, l.S);
  if (row_reduce && T <= 64 && l.S_row > 0) return l.S_row;       // the plan's measured choice (umb_gemm_plan2)
  if (row_reduce && T <= 64) {
    const int cap = l.K / 1792 > 4 ? l.K / 1792 : 4;
    const int nblk = l.N / (64 * (l.R > 0 ? l.R : 1));        // 64 R rows per block: never drop below one block per CU
    if (S > cap && nblk * cap >= 256) S = cap;
  }
  return S;
}

static inline int lin(const UmbLinear& l, const void* x, int ldx, void* out, int T, int dtype, hipStream_t st, int epi,
                      const UmbGemmFused* fx, bool row_reduce = false) {
  return umb_gemm_fused(out, x, ldx, l.w, l.meta, T, l.N, l.K, l.awq, eff_s(l, T, row_reduce), l.R | (l.tb << 8), epi, fx, dtype,
                        st);
}

// embedding (stage 0) / index resolution + hw = h * norm1_w and the per-64-column sums of squares of h
static int prologue(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const void* first_norm, hipStream_t st,
                    int fm_tt = 0) {
  if (s->T < 1 || s->T > ws->Tmax) return UMB_EINVAL;
  if (ws->fused != 1) {
    CK(umb_embed_prep(ws->h, s->skip_embed ? nullptr : m->embed, m->H, s->T, s->tokens, s->positions, s->slots,
                      s->prefix_len, s->tokens_all, s->n_ptr, s->tree_off, s->depth, ws->pos, ws->slot, ws->prefix,
                      nullptr, nullptr, nullptr, 0, m->dtype, st));
    return umb_rmsnorm_fm(ws->xn, ws->h, first_norm, m->eps, s->T, m->H, fm_tt, m->dtype, st);
  }
  return umb_embed_prep(ws->h, s->skip_embed ? nullptr : m->embed, m->H, s->T, s->tokens, s->positions, s->slots,
                        s->prefix_len, s->tokens_all, s->n_ptr, s->tree_off, s->depth, ws->pos, ws->slot, ws->prefix,
                        ws->hw, first_norm, ws->ssq, ws->ssq_stride, m->dtype, st);
}

// ---- tensor parallelism: this rank holds 1/P of every linear (q/k/v, gate/up column-split; o, down
// row-split), so the fp32 partial sums of the two row-split GEMMs are summed over the ranks before the residual reduce.
// The collective is the caller's (RCCL all-reduce on `st`, captured into the iteration's hipGraph like every kernel
// here; host-staged gloo in the CPU-side tests): two calls per layer.  The split-K slabs are ALWAYS summed first
// (umb_sum_splits, one ~5 us launch): the collective then carries exactly one [T, H] fp32 tile -- T H 4 bytes, 426 KB at
// T = 13 -- instead of S of them (round 3 sent the four slabs of a T = 13 tile as they were: 1.7 MB per call, 160 calls per
// 70B verify, on point-to-point links where such messages are latency- and per-link-bound).
static inline bool tp_on(const UmbTP* tp) { return tp && tp->world > 1 && (tp->allreduce || tp->peer); }
// small tiles: direct peer reads (csrc/tp.hip); the choice depends on the tile size only -- the same on every rank
static inline bool tp_peer(const UmbTP* tp, long TN) {
  return tp && tp->world > 1 && tp->peer && TN <= tp->peer_max_floats && TN <= tp->peer->cap;
}
static int tp_allreduce(const UmbTP* tp, float* partial, int* S, long TN, hipStream_t st) {
  if (!tp_on(tp)) return UMB_OK;
  if (!tp->allreduce) return UMB_EINVAL;                        // a tile too large for the peer path and no hook
  if (*S > 1) {
    CK(umb_sum_splits(partial, *S, TN, st));
    *S = 1;
  }
  return tp->allreduce(tp->ctx, partial, (int64_t)TN, st) ? UMB_EHIP : UMB_OK;
}

// Round 4: forwards of <= 64 rows keep the split schedule's 16-bit activations (normed residual, attention output, SiLU output)
// in FM order -- one MFMA B fragment = one contiguous KiB -- like the low-latency schedule's buffers: the skinny GEMMs stage them
// with coalesced loads instead of 16 rows x 64 B per instruction (70B layer at 13 rows: qkv 12.7 -> 12.4, o 10.5 -> 10.1, gate/up
// 44.6 -> 43.7, down 24.3 -> 23.5 us; 8B-AWQ at 32 rows: qkv 9.2 -> 8.1, gate/up 23.1 -> 21.9; profiles/r04_x_fm_probe.txt).  Same
// values through the same MFMAs: the bits do not change.  Returns the token-tile count (0: row-major).  UMB_SPLIT_FM=0: A/B.
static inline int split_fm_tt(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbTP* tp) {
  static const bool off = getenv("UMB_SPLIT_FM") != nullptr && getenv("UMB_SPLIT_FM")[0] == '0';
  // from 16 rows: whole forwards, FM vs row-major -- 8B-AWQ 16 / 32 rows 2.006 -> 1.969 / 2.475 -> 2.414 ms, 8B dense 31 rows 3.808 -> 3.684,
  // 70B-AWQ at 13 rows 2.088 -> 2.085 (a 13-row tile is 19 % padding; the headline iteration measured 0.05 ms slower): stays row-major
  if (off || ws->fused == 1 || tp_on(tp) || s->T > 64 || s->T < 16) return 0;
  const int tt = umb_ll_token_tiles(s->T);
  if (tt < 1 || ws->Tmax < 16 * tt || m->H % 32 || m->I % 32 || (m->Hq * m->D) % 32) return 0;
  return tt;
}

// Schedule 0 (default): one decoder layer = 8 launches; split-K partials are reduced at kernel boundaries by
// small epilogue kernels.  Measured faster on MI355X than schedule 1: an in-kernel cross-workgroup hand-off costs
// as much as a kernel boundary on the 8-XCD part (profiles/README.md), and it serialises a tail onto every GEMM.
// fm: token tiles of the FM-ordered activations (split_fm_tt; 0 = row-major); more: another layer of THIS forward follows, i.e.
// the normed residual this layer leaves is a GEMM operand (FM) and not the lm_head's / the next pipeline stage's input (row-major)
static int layer_split(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbLayer& ly, int l,
                       const void* next_norm, hipStream_t st, const UmbTP* tp = nullptr, int fm = 0, bool more = false) {
  const int T = s->T, dt = m->dtype;
  const size_t esz = 2;
  char* kc = (char*)m->k_cache + (size_t)l * m->Hkv * m->Lmax * m->D * esz;
  char* vt = (char*)m->vt_cache + (size_t)l * m->Hkv * VT_LD(m->Lmax) * m->D * esz;
  UmbGemmFused fxm = {}, fgu = {};
  fxm.pad1 = 1;                                             // x in FM order
  fgu.pad1 = 3;                                             // x and the SiLU output in FM order
  const UmbGemmFused* xf = fm ? &fxm : nullptr;
  CK(lin(ly.qkv, ws->xn, m->H, ws->partial, T, dt, st, 0, xf));
  CK(umb_reduce_qkv_rope(ws->partial, eff_s(ly.qkv, T), T, m->Hq, m->Hkv, m->D, m->Lmax, ws->pos, ws->slot, m->rope_cos,
                         m->rope_sin, ws->q, kc, vt, /*paired=*/1, ly.qkv_bias, dt, st));
  CK(umb_tree_attn2(ws->attn, ws->q, kc, vt, ws->attn_po, ws->attn_ml, ws->prefix, s->mask_bits, s->mask_words,
                    s->n_mask_keys, T, m->Hq, m->Hkv, m->D, m->Lmax, ws->attn_chunk, ws->attn_splits, m->attn_scale,
                    ws->attn_counters, fm, dt, st));
  CK(lin(ly.o, ws->attn, m->Hq * m->D, ws->partial, T, dt, st, 0, xf, true));
  int So = eff_s(ly.o, T, true);
  if (tp_peer(tp, (long)T * m->H)) {
    CK(umb_tp_publish(tp->peer, ws->partial, So, (int64_t)T * m->H, st));
    CK(umb_tp_reduce_residual_norm(tp->peer, T, m->H, ws->h, ws->h, ws->xn, ly.norm2, m->eps, dt, st));
  } else {
    CK(tp_allreduce(tp, ws->partial, &So, (long)T * m->H, st));
    CK(umb_reduce_residual_norm_fm(ws->partial, So, T, m->H, ws->h, ws->h, ws->xn, ly.norm2, m->eps, fm, dt, st));
  }
  if (ly.gu.S != 1) return UMB_EINVAL;     // gate/up rows are interleaved at load time: SiLU(gate)*up is the epilogue
  CK(lin(ly.gu, ws->xn, m->H, ws->act, T, dt, st, /*EPI_SILU*/2, fm ? &fgu : nullptr));
  CK(lin(ly.down, ws->act, m->I, ws->partial, T, dt, st, 0, xf, true));
  int Sd = eff_s(ly.down, T, true);
  if (tp_peer(tp, (long)T * m->H)) {
    CK(umb_tp_publish(tp->peer, ws->partial, Sd, (int64_t)T * m->H, st));
    CK(umb_tp_reduce_residual_norm(tp->peer, T, m->H, ws->h, ws->h, next_norm ? ws->xn : nullptr, next_norm, m->eps, dt, st));
  } else {
    CK(tp_allreduce(tp, ws->partial, &Sd, (long)T * m->H, st));
    CK(umb_reduce_residual_norm_fm(ws->partial, Sd, T, m->H, ws->h, ws->h, next_norm ? ws->xn : nullptr, next_norm, m->eps,
                                   more ? fm : 0, dt, st));
  }
  return UMB_OK;
}

// Schedule 0 with the RMSNorm deferred (ws->defer_norm; round 6): the same 8 launches, but the two residual reduces run N / 512
// blocks per token row (umb_reduce_residual_hw: h, hw = h * w_next, sums of squares per block) instead of one block per row that
// has to see the whole row before it can normalise -- at T = 13 thirteen CUs each pulled 128-256 KB of fp32 partials.  1 / rms is
// applied where it commutes with the matmul: by the q/k/v reduce, by the gate/up epilogue, by the lm_head epilogue.  *groups:
// how many sums of squares per token the producer of ws->hw left (H / 64 after the embedding kernel, H / 512
// after a residual reduce).
// Forwards of <= 64 rows only: from 65 rows on the one-block-per-row reduce already fills the chip and the wide GEMMs do not
// want a per-token factor in their epilogues (C3 / C4 iterations measured 1 % slower with it; the two regimes differ in their
// int4 arithmetic anyway, so no invariance is lost across that boundary).
static inline bool use_defer(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbTP* tp) {
  return s->T <= 64 && ws->fused != 1 && ws->defer_norm && !tp_on(tp) && m->H % 512 == 0 && ws->ssq_stride >= m->H / 64 && ws->ssq_stride % 4 == 0 &&
         (m->H / 512) % 4 == 0;
}
static int prologue_defer(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const void* first_norm, int fm, int* groups,
                          hipStream_t st) {
  if (s->T < 1 || s->T > ws->Tmax) return UMB_EINVAL;
  // one grouping of the sums of squares (per 64 columns) whatever the row count and the layout of hw: a token's 1 / rms is the
  // same bits in a 1-row and in a 64-row forward
  *groups = m->H / 64;
  return umb_embed_prep_fm(ws->h, s->skip_embed ? nullptr : m->embed, m->H, s->T, s->tokens, s->positions, s->slots,
                           s->prefix_len, s->tokens_all, s->n_ptr, s->tree_off, s->depth, ws->pos, ws->slot, ws->prefix,
                           ws->hw, first_norm, ws->ssq, ws->ssq_stride, fm, m->dtype, st);
}
static int layer_split_defer(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbLayer& ly, int l,
                             const void* next_norm, int* groups, hipStream_t st, int fm, bool more) {
  const int T = s->T, dt = m->dtype;
  const size_t esz = 2;
  char* kc = (char*)m->k_cache + (size_t)l * m->Hkv * m->Lmax * m->D * esz;
  char* vt = (char*)m->vt_cache + (size_t)l * m->Hkv * VT_LD(m->Lmax) * m->D * esz;
  UmbGemmFused fxm = {}, fgu = {};
  fxm.pad1 = 1;                                             // x in FM order
  const UmbGemmFused* xf = fm ? &fxm : nullptr;
  CK(lin(ly.qkv, ws->hw, m->H, ws->partial, T, dt, st, 0, xf));
  CK(umb_reduce_qkv_rope2(ws->partial, eff_s(ly.qkv, T), T, m->Hq, m->Hkv, m->D, m->Lmax, ws->pos, ws->slot, m->rope_cos,
                          m->rope_sin, ws->q, kc, vt, /*paired=*/1, ly.qkv_bias, ws->ssq, *groups, ws->ssq_stride, (float)m->H,
                          m->eps, dt, st));
  CK(umb_tree_attn2(ws->attn, ws->q, kc, vt, ws->attn_po, ws->attn_ml, ws->prefix, s->mask_bits, s->mask_words,
                    s->n_mask_keys, T, m->Hq, m->Hkv, m->D, m->Lmax, ws->attn_chunk, ws->attn_splits, m->attn_scale,
                    ws->attn_counters, fm, dt, st));
  CK(lin(ly.o, ws->attn, m->Hq * m->D, ws->partial, T, dt, st, 0, xf, true));
  const int g2 = m->H / 512;
  CK(umb_reduce_residual_hw(ws->partial, eff_s(ly.o, T, true), T, m->H, ws->h, ws->h, ws->hw, ly.norm2, ws->ssq, ws->ssq_stride, fm,
                            dt, st));
  if (ly.gu.S != 1) return UMB_EINVAL;     // gate/up rows are interleaved at load time: SiLU(gate)*up is the epilogue
  fgu.ssq_in = ws->ssq; fgu.ssq_groups = g2; fgu.pad0 = ws->ssq_stride; fgu.ssq_dim = (float)m->H; fgu.eps = m->eps;
  fgu.pad1 = fm ? 3 : 0;                                    // x and the SiLU output in FM order
  CK(lin(ly.gu, ws->hw, m->H, ws->act, T, dt, st, /*EPI_SILU*/2, &fgu));
  CK(lin(ly.down, ws->act, m->I, ws->partial, T, dt, st, 0, xf, true));
  CK(umb_reduce_residual_hw(ws->partial, eff_s(ly.down, T, true), T, m->H, ws->h, ws->h, next_norm ? ws->hw : nullptr, next_norm,
                            next_norm ? ws->ssq : nullptr, ws->ssq_stride, more ? fm : 0, dt, st));
  *groups = g2;
  return UMB_OK;
}
static int head_defer(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, int groups, hipStream_t st) {
  if (s->head_from >= s->T) return UMB_OK;
  const int rows = s->T - s->head_from;
  const char* x = (const char*)ws->hw + (size_t)s->head_from * m->H * 2;
  UmbGemmFused fh = {};
  fh.ssq_in = ws->ssq + (size_t)s->head_from * ws->ssq_stride; fh.ssq_groups = groups; fh.pad0 = ws->ssq_stride;
  fh.ssq_dim = (float)m->H; fh.eps = m->eps;
  return lin(m->lm_head, x, m->H, ws->logits, rows, m->dtype, st, /*EPI_ROUND*/1, &fh);
}

// Schedule 1: one decoder layer = 5 launches.  RMSNorm is split: its weight is folded into the activations by
// the producer (hw = h * w), its per-token factor is applied to the consumer GEMM's outputs from `ssq`.
static int layer_fused(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbLayer& ly, int l,
                 const void* next_norm, hipStream_t st) {
  const int T = s->T, dt = m->dtype;
  const size_t esz = 2;
  char* kc = (char*)m->k_cache + (size_t)l * m->Hkv * m->Lmax * m->D * esz;
  char* vt = (char*)m->vt_cache + (size_t)l * m->Hkv * VT_LD(m->Lmax) * m->D * esz;
  if (ly.qkv_bias) return UMB_EINVAL;          // projection bias: default schedule only
  UmbGemmFused fx = {};
  // 1. qkv GEMM -> (last split block) 1/rms, RoPE at tree positions, q out, K/V appended at their slots
  fx.ssq_in = ws->ssq; fx.ssq_groups = ws->ssq_stride; fx.ssq_dim = (float)m->H; fx.eps = m->eps;
  fx.counters = ws->counters;
  fx.pos = ws->pos; fx.slot = ws->slot; fx.cosT = m->rope_cos; fx.sinT = m->rope_sin;
  fx.q_out = ws->q; fx.k_cache = kc; fx.vt_cache = vt; fx.Hq = m->Hq; fx.Hkv = m->Hkv; fx.D = m->D; fx.Lmax = m->Lmax;
  CK(lin(ly.qkv, ws->hw, m->H, ws->partial, T, dt, st, /*EPI_QKV*/3, &fx));
  // 2. tree attention (+ fused split merge)
  CK(umb_tree_attn(ws->attn, ws->q, kc, vt, ws->attn_po, ws->attn_ml, ws->prefix, s->mask_bits, s->mask_words,
                   s->n_mask_keys, T, m->Hq, m->Hkv, m->D, m->Lmax, ws->attn_chunk, ws->attn_splits, m->attn_scale,
                   ws->attn_counters, dt, st));
  // 3. o_proj GEMM -> h += o ; hw = h * norm2_w ; ssq
  UmbGemmFused fo = {};
  fo.counters = ws->counters; fo.h = ws->h; fo.hw = ws->hw; fo.norm_w = ly.norm2; fo.ssq_out = ws->ssq;
  fo.ssq_out_stride = ws->ssq_stride;
  CK(lin(ly.o, ws->attn, m->Hq * m->D, ws->partial, T, dt, st, /*EPI_RESID*/4, &fo));
  // 4. gate/up GEMM (unsplit, interleaved rows) -> act = SiLU(gate/rms) * up/rms
  UmbGemmFused fg = {};
  fg.ssq_in = ws->ssq; fg.ssq_groups = ws->ssq_stride; fg.ssq_dim = (float)m->H; fg.eps = m->eps;
  if (ly.gu.S != 1) return UMB_EINVAL;
  CK(lin(ly.gu, ws->hw, m->H, ws->act, T, dt, st, /*EPI_SILU*/2, &fg));
  // 5. down GEMM -> h += d ; hw = h * (next layer's norm1 | final norm) ; ssq
  UmbGemmFused fd = {};
  fd.counters = ws->counters; fd.h = ws->h; fd.hw = next_norm ? ws->hw : nullptr; fd.norm_w = next_norm;
  fd.ssq_out = ws->ssq; fd.ssq_out_stride = ws->ssq_stride;
  CK(lin(ly.down, ws->act, m->I, ws->partial, T, dt, st, /*EPI_RESID*/4, &fd));
  return UMB_OK;
}

// Schedule 2 (low latency, T <= 64): 5 launches per layer, no cross-workgroup step at all.  Every GEMM workgroup owns
// its output rows for the whole K (csrc/lowlat.hip) and runs the layer's elementwise work as its epilogue;
// activations travel in FM (MFMA B-fragment) layout: ws->hw (K = H), ws->attn (K = Hq D), ws->act (K = I).
// A DRAFT-role model (its linears carry row-major copies: Llama.use_gemv) leaves the low-latency schedule from 16 rows on: there the
// split schedule with FM operands and the deferred norm measured faster (1B forward, 16 / 32 rows: 0.843 / 0.915 ms vs 0.889 / 1.099;
// 1 ... 8 rows: low-latency wins, 0.61-0.81 vs 0.78-0.83).  A draft only proposes, so its rows need not be the same bits at every row
// count; targets keep ONE schedule for all <= 64-row forwards (speculative == autoregressive, bit for bit, rests on that).
static inline bool use_ll(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s) {
  if (ws->fused != 2 || s->T > 64) return false;
  static const bool keep = getenv("UMB_LL_DRAFT_WIDE") != nullptr && getenv("UMB_LL_DRAFT_WIDE")[0] == '1';   // A/B
  if (!keep && s->T >= 16 && s->layer_begin < s->layer_end && m->layers[s->layer_begin].qkv.w_rows) return false;
  return true;
}
static inline int ll_groups(const UmbLinear& l) {
  int R, WN, WK, NW;
  umb_ll_plan(l.N, l.K, l.awq, &R, &WN, &WK, &NW);
  return l.N / 16 / R;
}

// gate/up on the LDS-shared split-K-family kernel (S = 1): when its grid of N / (64 R) four-wave blocks covers the chip.
// UMB_LL_GU = ll | shared overrides (experiments).  Shape-only, so batch invariance is kept.
static inline bool gu_on_shared_kernel(const UmbLinear& l) {
  static const char* env = getenv("UMB_LL_GU");
  if (env && env[0] == 'l') return false;
  if (env && env[0] == 's') return l.S == 1;
  return l.S == 1 && l.N / (64 * (l.R > 0 ? l.R : 1)) >= 256;
}

// qkv on the split-K kernel + reduce: when the low-latency plan leaves a ragged last round (blocks not a multiple of the
// CU count and fewer than two rounds).  UMB_LL_QKV = ll | split overrides (experiments).  Shape-only.
static inline bool qkv_on_split_kernel(const UmbLinear& l) {
  static const char* env = getenv("UMB_LL_QKV");
  if (env && env[0] == 'l') return false;
  if (env && env[0] == 's') return l.S > 1;
  int R, WN, WK, NW;
  umb_ll_plan(l.N, l.K, l.awq, &R, &WN, &WK, &NW);
  const int blocks = (l.N / 16 / R + WN - 1) / WN;
  return l.S > 1 && blocks > 256 && blocks < 512;
}

static int prologue_ll(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const void* first_norm, hipStream_t st) {
  if (s->T < 1 || s->T > ws->Tmax) return UMB_EINVAL;
  return umb_embed_ll(ws->h, s->skip_embed ? nullptr : m->embed, m->H, m->V, m->Lmax, s->T, s->tokens, s->positions,
                      s->slots, s->prefix_len, s->tokens_all, s->n_ptr, s->tree_off, s->depth, ws->pos, ws->slot,
                      ws->prefix, ws->hw, first_norm, ws->ssq, ws->ssq_stride, m->dtype, st);
}

// ssq_groups: how many partial sums of squares the producer of ws->hw left per token (4 after the embedding kernel,
// N / 16 / R after a residual GEMM)
static int layer_ll(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbLayer& ly, int l,
                    const void* next_norm, int* ssq_groups, hipStream_t st) {
  const int T = s->T, dt = m->dtype;
  const size_t esz = 2;
  char* kc = (char*)m->k_cache + (size_t)l * m->Hkv * m->Lmax * m->D * esz;
  char* vt = (char*)m->vt_cache + (size_t)l * m->Hkv * VT_LD(m->Lmax) * m->D * esz;
  const int og = ll_groups(ly.o), dg = ll_groups(ly.down);
  if (og > ws->ssq_stride || dg > ws->ssq_stride) return UMB_EINVAL;
  // 1. qkv: 1/rms, (+bias), RoPE at the tree positions, q out, K / V appended at their slots.  A linear whose row groups
  // do not tile the 256 CUs in whole rounds (70B: 320 blocks) runs split-K on the LDS-shared kernel + its reduce kernel
  // instead (16.9 + 5.0 vs 24.5 us), reading the same FM activations.
  if (qkv_on_split_kernel(ly.qkv)) {
    UmbGemmFused fs = {};
    fs.pad1 = 1;                                            // x in FM layout
    CK(lin(ly.qkv, ws->hw, m->H, ws->partial, T, dt, st, 0, &fs));
    CK(umb_reduce_qkv_rope2(ws->partial, eff_s(ly.qkv, T), T, m->Hq, m->Hkv, m->D, m->Lmax, ws->pos, ws->slot, m->rope_cos,
                            m->rope_sin, ws->q, kc, vt, /*paired=*/1, ly.qkv_bias, ws->ssq, *ssq_groups, ws->ssq_stride,
                            (float)m->H, m->eps, dt, st));
  } else {
  UmbGemmLL fq = {};
  fq.ssq_in = ws->ssq; fq.ssq_groups = *ssq_groups; fq.ssq_in_stride = ws->ssq_stride; fq.ssq_dim = (float)m->H; fq.eps = m->eps;
  fq.pos = ws->pos; fq.slot = ws->slot; fq.cosT = m->rope_cos; fq.sinT = m->rope_sin; fq.q_out = ws->q; fq.k_cache = kc;
  fq.vt_cache = vt; fq.bias = ly.qkv_bias; fq.Hq = m->Hq; fq.Hkv = m->Hkv; fq.D = m->D; fq.Lmax = m->Lmax;
  CK(umb_gemm_ll(nullptr, ws->hw, ly.qkv.w, ly.qkv.meta, T, ly.qkv.N, ly.qkv.K, ly.qkv.awq, 3, &fq, dt, st));
  }
  // 2. tree attention, output in FM layout for the o-projection
  CK(umb_tree_attn2(ws->attn, ws->q, kc, vt, ws->attn_po, ws->attn_ml, ws->prefix, s->mask_bits, s->mask_words,
                    s->n_mask_keys, T, m->Hq, m->Hkv, m->D, m->Lmax, ws->attn_chunk, ws->attn_splits, m->attn_scale,
                    ws->attn_counters, umb_ll_token_tiles(T), dt, st));
  // 3. o_proj: h += o ; hw = h * norm2_w ; sums of squares
  UmbGemmLL fo = {};
  fo.h = ws->h; fo.hw = ws->hw; fo.norm_w = ly.norm2; fo.ssq_out = ws->ssq; fo.ssq_out_stride = ws->ssq_stride;
  CK(umb_gemm_ll(nullptr, ws->attn, ly.o.w, ly.o.meta, T, ly.o.N, ly.o.K, ly.o.awq, 4, &fo, dt, st));
  // 4. gate/up (rows interleaved at load): act = SiLU(gate / rms) * up / rms, FM layout.  A linear with enough n-tiles
  // to fill the chip without any K split runs on the LDS-shared kernel of gemm.hip (4 waves share each activation
  // fragment; 70B gate/up: 53 vs 65 us), reading and writing the same FM buffers.
  if (gu_on_shared_kernel(ly.gu) && og % 4 == 0 && ws->ssq_stride % 4 == 0) {
    UmbGemmFused fs = {};
    fs.ssq_in = ws->ssq; fs.ssq_groups = og; fs.pad0 = ws->ssq_stride; fs.ssq_dim = (float)m->H; fs.eps = m->eps;
    fs.pad1 = 3;                                            // x and the SiLU output in FM layout
    CK(lin(ly.gu, ws->hw, m->H, ws->act, T, dt, st, /*EPI_SILU*/2, &fs));
  } else {
    UmbGemmLL fg = {};
    fg.ssq_in = ws->ssq; fg.ssq_groups = og; fg.ssq_in_stride = ws->ssq_stride; fg.ssq_dim = (float)m->H; fg.eps = m->eps;
    CK(umb_gemm_ll(ws->act, ws->hw, ly.gu.w, ly.gu.meta, T, ly.gu.N, ly.gu.K, ly.gu.awq, 2, &fg, dt, st));
  }
  // 5. down: h += d ; hw = h * (next layer's norm1 | final norm) ; sums of squares
  UmbGemmLL fd = {};
  fd.h = ws->h; fd.hw = ws->hw; fd.norm_w = next_norm; fd.ssq_out = ws->ssq; fd.ssq_out_stride = ws->ssq_stride;
  CK(umb_gemm_ll(nullptr, ws->act, ly.down.w, ly.down.meta, T, ly.down.N, ly.down.K, ly.down.awq, 4, &fd, dt, st));
  *ssq_groups = dg;
  return UMB_OK;
}

static int head_ll(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, int ssq_groups, hipStream_t st) {
  if (s->head_from >= s->T) return UMB_OK;
  // Every row wants logits (a dynamic or static draft level) and the head runs unsplit: the LDS-shared kernel of gemm.hip reads
  // the same FM activations and sums of squares (round 4: 1B head at 16 rows 116 -> 9x us; the whole-K kernel re-reads the
  // activations once per wave).  UMB_LL_HEAD=ll: the whole-K kernel (A/B).  Shape-only + head_from, so batch invariance is kept.
  static const char* env = getenv("UMB_LL_HEAD");
  // a draft's 5 ... 8-row levels (the reference's 5- and 6-wide Sequoia trees): the streaming engine of chain.hip reads the same
  // FM activations and sums of squares
  if (s->head_from == 0 && m->lm_head.w_rows && ssq_groups <= 256 && umb_head_stream_ok(s->T, m->lm_head.N, m->H))
    return umb_head_stream(ws->logits, ws->hw, ws->ssq, ws->ssq_stride, ssq_groups, m->eps, m->lm_head.w_rows, s->T, m->lm_head.N,
                           m->H, umb_ll_token_tiles(s->T), m->dtype, st);
  if (!(env && env[0] == 'l') && s->head_from == 0 && m->lm_head.S == 1 && !m->lm_head.awq && ssq_groups % 4 == 0 &&
      ws->ssq_stride % 4 == 0) {
    UmbGemmFused fs = {};
    fs.ssq_in = ws->ssq; fs.ssq_groups = ssq_groups; fs.pad0 = ws->ssq_stride; fs.ssq_dim = (float)m->H; fs.eps = m->eps;
    fs.pad1 = 1;                                            // x in FM layout
    return lin(m->lm_head, ws->hw, m->H, ws->logits, s->T, m->dtype, st, /*EPI_ROUND*/1, &fs);
  }
  UmbGemmLL fh = {};
  fh.row_from = s->head_from; fh.round_out = 1;
  fh.ssq_in = ws->ssq; fh.ssq_groups = ssq_groups; fh.ssq_in_stride = ws->ssq_stride; fh.ssq_dim = (float)m->H; fh.eps = m->eps;
  return umb_gemm_ll(ws->logits, ws->hw, m->lm_head.w, m->lm_head.meta, s->T, m->lm_head.N, m->lm_head.K, m->lm_head.awq,
                     0, &fh, m->dtype, st);
}

// Schedule 3 (GEMV, T <= 4 rows of a low-latency model whose linears carry row-major copies): the low-latency schedule's
// five launches with the four GEMMs on the row-streaming kernels of gemv.hip and ROW-MAJOR activations (ws->hw, ws->attn,
// ws->act); the RMSNorm is split the same way (producer: hw = h * w and per-workgroup sums of squares; consumer: 1/rms).
static inline bool use_gv(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s) {
  if (ws->fused != 2 || s->T > 4 || s->layer_begin >= s->layer_end) return false;
  const UmbLayer& ly = m->layers[s->layer_begin];
  for (int l = s->layer_begin; l < s->layer_end; ++l) {              // every layer of the range, not only its first
    const UmbLayer& x = m->layers[l];
    if (!x.qkv.w_rows || !x.o.w_rows || !x.gu.w_rows || !x.down.w_rows) return false;
  }
  if (m->H % 64) return false;
  if (!umb_gemv_ok(s->T, ly.qkv.N, ly.qkv.K, 3) || !umb_gemv_ok(s->T, ly.o.N, ly.o.K, 4) ||
      !umb_gemv_ok(s->T, ly.gu.N, ly.gu.K, 2) || !umb_gemv_ok(s->T, ly.down.N, ly.down.K, 4)) return false;
  const int og = umb_gemv_groups(s->T, ly.o.N, ly.o.K), dg = umb_gemv_groups(s->T, ly.down.N, ly.down.K);
  return og <= ws->ssq_stride && dg <= ws->ssq_stride && og % 4 == 0 && dg % 4 == 0 && (m->H / 64) % 4 == 0;
}
static int prologue_gv(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const void* first_norm, int* groups,
                       hipStream_t st) {
  if (s->T < 1 || s->T > ws->Tmax) return UMB_EINVAL;
  *groups = m->H / 64;                         // umb_embed_prep: one sum of squares per 64 columns
  return umb_embed_prep(ws->h, s->skip_embed ? nullptr : m->embed, m->H, s->T, s->tokens, s->positions, s->slots,
                        s->prefix_len, s->tokens_all, s->n_ptr, s->tree_off, s->depth, ws->pos, ws->slot, ws->prefix,
                        ws->hw, first_norm, ws->ssq, ws->ssq_stride, m->dtype, st);
}
static int layer_gv(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbLayer& ly, int l,
                    const void* next_norm, int* groups, hipStream_t st) {
  const int T = s->T, dt = m->dtype;
  const size_t esz = 2;
  char* kc = (char*)m->k_cache + (size_t)l * m->Hkv * m->Lmax * m->D * esz;
  char* vt = (char*)m->vt_cache + (size_t)l * m->Hkv * VT_LD(m->Lmax) * m->D * esz;
  UmbGemmLL fq = {};
  fq.ssq_in = ws->ssq; fq.ssq_groups = *groups; fq.ssq_in_stride = ws->ssq_stride; fq.ssq_dim = (float)m->H; fq.eps = m->eps;
  fq.pos = ws->pos; fq.slot = ws->slot; fq.cosT = m->rope_cos; fq.sinT = m->rope_sin; fq.q_out = ws->q; fq.k_cache = kc;
  fq.vt_cache = vt; fq.bias = ly.qkv_bias; fq.Hq = m->Hq; fq.Hkv = m->Hkv; fq.D = m->D; fq.Lmax = m->Lmax;
  CK(umb_gemv(nullptr, ws->hw, ly.qkv.w_rows, T, ly.qkv.N, ly.qkv.K, 3, &fq, dt, st));
  CK(umb_tree_attn(ws->attn, ws->q, kc, vt, ws->attn_po, ws->attn_ml, ws->prefix, s->mask_bits, s->mask_words,
                   s->n_mask_keys, T, m->Hq, m->Hkv, m->D, m->Lmax, ws->attn_chunk, ws->attn_splits, m->attn_scale,
                   ws->attn_counters, dt, st));
  UmbGemmLL fo = {};
  fo.h = ws->h; fo.hw = ws->hw; fo.norm_w = ly.norm2; fo.ssq_out = ws->ssq; fo.ssq_out_stride = ws->ssq_stride;
  CK(umb_gemv(nullptr, ws->attn, ly.o.w_rows, T, ly.o.N, ly.o.K, 4, &fo, dt, st));
  UmbGemmLL fg = {};
  fg.ssq_in = ws->ssq; fg.ssq_groups = umb_gemv_groups(T, ly.o.N, ly.o.K); fg.ssq_in_stride = ws->ssq_stride;
  fg.ssq_dim = (float)m->H; fg.eps = m->eps;
  CK(umb_gemv(ws->act, ws->hw, ly.gu.w_rows, T, ly.gu.N, ly.gu.K, 2, &fg, dt, st));
  UmbGemmLL fd = {};
  fd.h = ws->h; fd.hw = ws->hw; fd.norm_w = next_norm; fd.ssq_out = ws->ssq; fd.ssq_out_stride = ws->ssq_stride;
  CK(umb_gemv(nullptr, ws->act, ly.down.w_rows, T, ly.down.N, ly.down.K, 4, &fd, dt, st));
  *groups = umb_gemv_groups(T, ly.down.N, ly.down.K);
  return UMB_OK;
}
// ---- persistent chain (chain.hip): the GEMV schedule's layer as tree attention + ONE launch (o, gate/up, down, next q/k/v)
static inline bool use_chain(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s) {
  if (!ws->chain_xchg || ws->ssq_stride < 256) return false;
  for (int l = s->layer_begin; l < s->layer_end; ++l)
    if (m->layers[l].qkv_bias) return false;
  const UmbLayer& ly = m->layers[s->layer_begin];
  if (ly.o.N != m->H || ly.down.N != m->H || ly.gu.N != 2 * m->I) return false;
  return umb_chain_ok(s->T, m->H, m->I, ly.qkv.N, m->D, 0) != 0;
}
// lf: layer whose o / gate-up / down run (-1: none); lt: layer whose q/k/v runs (-1: none)
static int chain_launch(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, int lf, int lt, const void* next_norm,
                        int groups_in, hipStream_t st) {
  UmbChain c = {};
  if (lf >= 0) {
    const UmbLayer& ly = m->layers[lf];
    c.front = 1; c.w_o = ly.o.w_rows; c.w_gu = ly.gu.w_rows; c.w_down = ly.down.w_rows; c.norm2 = ly.norm2;
    c.attn = ws->attn;
  }
  if (lt >= 0) {
    const UmbLayer& ly = m->layers[lt];
    c.tail = 1; c.w_qkv = ly.qkv.w_rows;
    c.k_cache = (char*)m->k_cache + (size_t)lt * m->Hkv * m->Lmax * m->D * 2;
    c.vt_cache = (char*)m->vt_cache + (size_t)lt * m->Hkv * VT_LD(m->Lmax) * m->D * 2;
    c.pos = ws->pos; c.slot = ws->slot; c.cosT = m->rope_cos; c.sinT = m->rope_sin; c.q_out = ws->q;
  }
  c.NQKV = m->layers[lt >= 0 ? lt : lf].qkv.N;
  c.h = ws->h; c.hw = ws->hw; c.ssq = ws->ssq; c.next_norm = next_norm; c.xchg = ws->chain_xchg;
  c.T = s->T; c.Tmax = UMB_CHAIN_TMAX;       /* the exchange layout is fixed (umbrella_hip.h: UmbWorkspace.chain_xchg) */
  c.H = m->H; c.I = m->I; c.ssq_stride = ws->ssq_stride; c.ssq_groups_in = groups_in;
  c.Hq = m->Hq; c.Hkv = m->Hkv; c.D = m->D; c.Lmax = m->Lmax; c.eps = m->eps;
  return umb_draft_chain(&c, m->dtype, st);
}
static int attn_gv(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, int l, hipStream_t st) {
  char* kc = (char*)m->k_cache + (size_t)l * m->Hkv * m->Lmax * m->D * 2;
  char* vt = (char*)m->vt_cache + (size_t)l * m->Hkv * VT_LD(m->Lmax) * m->D * 2;
  return umb_tree_attn(ws->attn, ws->q, kc, vt, ws->attn_po, ws->attn_ml, ws->prefix, s->mask_bits, s->mask_words,
                       s->n_mask_keys, s->T, m->Hq, m->Hkv, m->D, m->Lmax, ws->attn_chunk, ws->attn_splits, m->attn_scale,
                       ws->attn_counters, m->dtype, st);
}
static int head_gv(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, int groups, hipStream_t st) {
  if (s->head_from >= s->T) return UMB_OK;
  const int rows = s->T - s->head_from;
  const char* x = (const char*)ws->hw + (size_t)s->head_from * m->H * 2;
  // the head's rows as a plain row-major table (a tied model's embedding table): the streaming engine of chain.hip
  if (m->lm_head.w_rows && umb_head_stream_ok(rows, m->lm_head.N, m->H))
    return umb_head_stream(ws->logits, x, ws->ssq + (size_t)s->head_from * ws->ssq_stride, ws->ssq_stride, groups, m->eps,
                           m->lm_head.w_rows, rows, m->lm_head.N, m->H, 0, m->dtype, st);
  UmbGemmFused fh = {};
  fh.ssq_in = ws->ssq + (size_t)s->head_from * ws->ssq_stride; fh.ssq_groups = groups; fh.pad0 = ws->ssq_stride;
  fh.ssq_dim = (float)m->H; fh.eps = m->eps;
  return lin(m->lm_head, x, m->H, ws->logits, rows, m->dtype, st, /*EPI_ROUND*/1, &fh);
}

static int layer(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbLayer& ly, int l,
                 const void* next_norm, hipStream_t st, int fm = 0, bool more = false) {
  return ws->fused == 1 ? layer_fused(m, ws, s, ly, l, next_norm, st) : layer_split(m, ws, s, ly, l, next_norm, st, nullptr, fm, more);
}

static int head(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, hipStream_t st) {
  if (s->head_from >= s->T) return UMB_OK;
  const int rows = s->T - s->head_from;
  if (ws->fused != 1) {
    const char* xn = (const char*)ws->xn + (size_t)s->head_from * m->H * 2;
    return lin(m->lm_head, xn, m->H, ws->logits, rows, m->dtype, st, /*EPI_ROUND*/1, nullptr);
  }
  const char* x = (const char*)ws->hw + (size_t)s->head_from * m->H * 2;
  UmbGemmFused fh = {};
  fh.ssq_in = ws->ssq + (size_t)s->head_from * ws->ssq_stride; fh.ssq_groups = ws->ssq_stride;
  fh.ssq_dim = (float)m->H; fh.eps = m->eps;
  return lin(m->lm_head, x, m->H, ws->logits, rows, m->dtype, st, /*EPI_ROUND*/1, &fh);
}

static int model_forward(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbTP* tp, hipStream_t st) {
  const int lb = s->layer_begin, le = s->layer_end;
  if (lb < 0 || le > m->L || lb >= le) return UMB_EINVAL;
  if (tp_on(tp)) {
    // row-split partial sums meet between the GEMM and its reduce kernel: the 8-launch schedule only
    if (ws->fused != 0) return UMB_EINVAL;
    CK(prologue(m, ws, s, m->layers[lb].norm1, st));
    for (int l = lb; l < le; ++l) {
      const void* nn = (l + 1 < le) ? m->layers[l + 1].norm1 : (le == m->L ? m->final_norm : nullptr);
      CK(layer_split(m, ws, s, m->layers[l], l, nn, st, tp));
    }
    if (le == m->L) CK(head(m, ws, s, st));
    return UMB_OK;
  }
  if (use_gv(m, ws, s)) {
    int groups = 0;
    CK(prologue_gv(m, ws, s, m->layers[lb].norm1, &groups, st));
    if (use_chain(m, ws, s)) {
      // 2 launches per layer: the first q/k/v alone, then per layer tree attention + the chain into the next q/k/v
      CK(chain_launch(m, ws, s, -1, lb, nullptr, groups, st));
      for (int l = lb; l < le; ++l) {
        const void* nn = (l + 1 < le) ? m->layers[l + 1].norm1 : (le == m->L ? m->final_norm : nullptr);
        CK(attn_gv(m, ws, s, l, st));
        CK(chain_launch(m, ws, s, l, l + 1 < le ? l + 1 : -1, nn, 0, st));
      }
      groups = umb_gemv_groups(s->T, m->layers[lb].down.N, m->layers[lb].down.K);
      if (le == m->L) CK(head_gv(m, ws, s, groups, st));
      return UMB_OK;
    }
    for (int l = lb; l < le; ++l) {
      const void* nn = (l + 1 < le) ? m->layers[l + 1].norm1 : (le == m->L ? m->final_norm : nullptr);
      CK(layer_gv(m, ws, s, m->layers[l], l, nn, &groups, st));
    }
    if (le == m->L) CK(head_gv(m, ws, s, groups, st));
    return UMB_OK;
  }
  const bool ll = use_ll(m, ws, s);
  const int fm = ll ? 0 : split_fm_tt(m, ws, s, nullptr);
  const bool df = !ll && use_defer(m, ws, s, nullptr);
  int sg = 4;
  CK(ll ? prologue_ll(m, ws, s, m->layers[lb].norm1, st)
        : df ? prologue_defer(m, ws, s, m->layers[lb].norm1, fm, &sg, st) : prologue(m, ws, s, m->layers[lb].norm1, st, fm));
  for (int l = lb; l < le; ++l) {
    const void* nn = (l + 1 < le) ? m->layers[l + 1].norm1 : (le == m->L ? m->final_norm : nullptr);
    CK(ll ? layer_ll(m, ws, s, m->layers[l], l, nn, &sg, st)
          : df ? layer_split_defer(m, ws, s, m->layers[l], l, nn, &sg, st, fm, l + 1 < le)
               : layer(m, ws, s, m->layers[l], l, nn, st, fm, l + 1 < le));
  }
  if (le == m->L) CK(ll ? head_ll(m, ws, s, sg, st) : df ? head_defer(m, ws, s, sg, st) : head(m, ws, s, st));
  return UMB_OK;
}

extern "C" int umb_model_forward(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, hipStream_t st) {
  return model_forward(m, ws, s, nullptr, st);
}

extern "C" int umb_model_forward_tp(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbTP* tp,
                                    hipStream_t st) {
  if (tp && (tp->world < 1 || tp->rank < 0 || tp->rank >= tp->world)) return UMB_EINVAL;
  return model_forward(m, ws, s, tp, st);
}

// ---- offload: double-buffered layer streaming, event ordered (no device-wide syncs)
static inline const void* rel(const void* off, const void* base) {
  return off ? (const void*)((const char*)base + (size_t)off - 1) : nullptr;   // offsets are stored +1 (0 == NULL)
}
// only the four linears live in the streamed slab; norm weights (2*H elements) stay device resident
static UmbLayer rebase(const UmbLayer& in, const void* base) {
  UmbLayer o = in;
  o.qkv.w = rel(in.qkv.w, base); o.qkv.meta = rel(in.qkv.meta, base);
  o.o.w = rel(in.o.w, base); o.o.meta = rel(in.o.meta, base);
  o.gu.w = rel(in.gu.w, base); o.gu.meta = rel(in.gu.meta, base);
  o.down.w = rel(in.down.w, base); o.down.meta = rel(in.down.meta, base);
  return o;
}

extern "C" int umb_model_forward_offload(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s,
                                         const UmbOffload* off, hipStream_t st) {
  const int lb = s->layer_begin, le = s->layer_end;
  if (lb < 0 || le > m->L || lb >= le) return UMB_EINVAL;
  const int NS = off->n_slabs;
  if (NS < 2 || NS > UMB_MAX_SLABS) return UMB_EINVAL;
  hipStream_t cs = (hipStream_t)off->copy_stream;
  auto issue_copy = [&](int l, int buf) -> int {
    // the slab may be overwritten only after the kernels that read it (ev_free is recorded on st)
    if (hipStreamWaitEvent(cs, (hipEvent_t)off->ev_free[buf], 0) != hipSuccess) return UMB_EHIP;
    if (hipMemcpyAsync(off->dev_slab[buf], off->host_slabs[l], off->slab_bytes, hipMemcpyHostToDevice, cs) != hipSuccess)
      return UMB_EHIP;
    if (hipEventRecord((hipEvent_t)off->ev_copied[buf], cs) != hipSuccess) return UMB_EHIP;
    return UMB_OK;
  };
  // the first NS streamed layers of this range, in order
  int first[UMB_MAX_SLABS];
  int nfirst = 0;
  for (int l = lb; l < le && nfirst < NS; ++l)
    if (off->host_slabs[l]) first[nfirst++] = l;
  for (int i = nfirst; i < UMB_MAX_SLABS; ++i) first[i] = -1;
  // Cross-forward prefetch (the reference's loop copies layer (idx + 1) % num_layers, llama.py:203-209: layer 0 of the
  // NEXT forward is in flight while lm_head, sampling and the next draft tree run).  The previous forward left its
  // epilogue copies of exactly these layers in the slabs / on the copy stream: nothing to issue, and nothing on the
  // compute stream gates them.
  int32_t* pf = off->prefetched;
  bool have = pf != nullptr && nfirst > 0;
  for (int i = 0; have && i < NS; ++i) have = pf[i] == first[i];
  int next = lb, issued = 0, used = 0;
  auto advance = [&]() { while (next < le && off->host_slabs[next] == nullptr) ++next; };
  advance();
  if (have) {
    for (int i = 0; i < NS && next < le; ++i) { ++issued; ++next; advance(); }
  } else {
    // cold start (or a different layer range): order the copy stream behind everything already queued on st that may
    // still read the slabs, then fetch the first NS layers
    for (int b = 0; b < NS; ++b)
      if (hipEventRecord((hipEvent_t)off->ev_free[b], st) != hipSuccess) return UMB_EHIP;
    for (int i = 0; i < NS && next < le; ++i) { CK(issue_copy(next, issued % NS)); ++issued; ++next; advance(); }
  }
  if (pf) for (int i = 0; i < UMB_MAX_SLABS; ++i) pf[i] = -1;

  const bool ll = use_ll(m, ws, s);
  const int fm = ll ? 0 : split_fm_tt(m, ws, s, nullptr);
  const bool df = !ll && use_defer(m, ws, s, nullptr);
  int sg = 4;
  CK(ll ? prologue_ll(m, ws, s, m->layers[lb].norm1, st)
        : df ? prologue_defer(m, ws, s, m->layers[lb].norm1, fm, &sg, st) : prologue(m, ws, s, m->layers[lb].norm1, st, fm));
  for (int l = lb; l < le; ++l) {
    UmbLayer cur = m->layers[l];
    int buf = -1;
    if (off->host_slabs[l]) {
      buf = used % NS;
      if (hipStreamWaitEvent(st, (hipEvent_t)off->ev_copied[buf], 0) != hipSuccess) return UMB_EHIP;
      cur = rebase(m->layers[l], off->dev_slab[buf]);
    }
    const void* nn = (l + 1 < le) ? m->layers[l + 1].norm1 : (le == m->L ? m->final_norm : nullptr);
    CK(ll ? layer_ll(m, ws, s, cur, l, nn, &sg, st)
          : df ? layer_split_defer(m, ws, s, cur, l, nn, &sg, st, fm, l + 1 < le) : layer(m, ws, s, cur, l, nn, st, fm, l + 1 < le));
    if (buf >= 0) {
      ++used;
      if (hipEventRecord((hipEvent_t)off->ev_free[buf], st) != hipSuccess) return UMB_EHIP;
      if (next < le) { CK(issue_copy(next, buf)); ++issued; ++next; advance(); }
    }
  }
  // Epilogue: the next forward's first NS slabs go out NOW, on the copy stream only -- each waits (on that stream) for
  // the ev_free of the last layer that read its slab, which was recorded above; the compute stream carries on with the
  // lm_head, the sampling kernels, the next draft tree and the next forward's resident layers while the link is busy.
  // Slab order restarts at 0.  (The slabs still in use by the last NS layers of THIS forward are reused in ring order,
  // each copy gated by its own ev_free.)
  if (pf && nfirst > 0) {
    // When every streamed layer of the range fits the ring at once (n_streamed <= NS) no in-loop copy overwrote a slab:
    // slab i still holds first[i], and the next forward finds its layers in place -- nothing crosses the link again
    // (the epilogue used to re-issue all of them every forward, ~444 MB per slab for the 70B-AWQ).
    int n_streamed = 0;
    for (int l = lb; l < le; ++l) n_streamed += off->host_slabs[l] != nullptr;
    if (n_streamed > NS)
      for (int i = 0; i < nfirst; ++i) CK(issue_copy(first[i], i));
    for (int i = 0; i < NS; ++i) pf[i] = first[i];
  }
  if (le == m->L) CK(ll ? head_ll(m, ws, s, sg, st) : df ? head_defer(m, ws, s, sg, st) : head(m, ws, s, st));
  return UMB_OK;
}

extern "C" const char* umb_version(void) { return "umbrella_hip 0.1 (gfx950)"; }
