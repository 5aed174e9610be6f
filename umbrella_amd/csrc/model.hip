// Whole-model forward: launches the per-layer kernel chain from native code so a
// verify / draft step costs one C call (and one hipGraph node sequence when captured).
//
// Restates Llama*.inference + layer_compute (umbrella/models/llama.py:75-134, 262-322,
// 461-502) and the offload loop (llama.py:196-219) as a kernel schedule:
//   embed+prep -> rmsnorm -> L x [ qkv gemm -> reduce+rope+kv append -> tree attention ->
//   o gemm -> reduce+residual+norm -> gate/up gemm -> reduce+silu*mul -> down gemm ->
//   reduce+residual+next norm ] -> lm_head gemm (fp32 logits)
#include "../../include/umbrella_hip.h"
#include "common.h"

extern "C" int umb_gemm(void*, const void*, int, const void*, const void*, int, int, int, int, int, int, int, int, hipStream_t);
extern "C" int umb_rmsnorm(void*, const void*, const void*, float, int, int, int, hipStream_t);
extern "C" int umb_reduce_residual_norm(const void*, int, int, int, const void*, void*, void*, const void*, float, int, hipStream_t);
extern "C" int umb_reduce_silu_mul(const void*, int, int, int, void*, int, hipStream_t);
extern "C" int umb_reduce_qkv_rope(const void*, int, int, int, int, int, int, const int*, const int*, const void*, const void*, void*, void*, void*, int, hipStream_t);
extern "C" int umb_embed_prep(void*, const void*, int, int, const int*, const int*, const int*, const int*, const int*, const int*, int, const int*, int*, int*, int*, int, hipStream_t);
extern "C" int umb_tree_attn(void*, const void*, const void*, const void*, void*, void*, const int*, const void*, int, int, int, int, int, int, int, int, int, float, int, hipStream_t);

#define CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

static inline int lin(const UmbLinear& l, const void* x, int ldx, float* out, int T, int dtype, hipStream_t st,
                      int epi = 0) {
  return umb_gemm(out, x, ldx, l.w, l.meta, T, l.N, l.K, l.awq, l.S, l.R, epi, dtype, st);
}

static int prologue(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, hipStream_t st) {
  if (s->T < 1 || s->T > ws->Tmax) return UMB_EINVAL;
  // positions / slots / prefix are resolved on every stage; the gather itself only on stage 0
  if (!s->skip_embed) {
    CK(umb_embed_prep(ws->h, m->embed, m->H, s->T, s->tokens, s->positions, s->slots, s->prefix_len, s->tokens_all,
                      s->n_ptr, s->tree_off, s->depth, ws->pos, ws->slot, ws->prefix, m->dtype, st));
  } else {
    // pipeline stage > 0: resolve indices only (table == NULL skips the gather); ws->h holds the incoming activations
    CK(umb_embed_prep(ws->xn, nullptr, m->H, s->T, s->tokens, s->positions, s->slots, s->prefix_len, s->tokens_all,
                      s->n_ptr, s->tree_off, s->depth, ws->pos, ws->slot, ws->prefix, m->dtype, st));
  }
  return UMB_OK;
}

static int layer(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, const UmbLayer& ly, int l,
                 const void* next_norm, hipStream_t st) {
  const int T = s->T, dt = m->dtype;
  const size_t esz = 2;
  char* kc = (char*)m->k_cache + (size_t)l * m->Hkv * m->Lmax * m->D * esz;
  char* vt = (char*)m->vt_cache + (size_t)l * m->Hkv * m->Lmax * m->D * esz;
  CK(lin(ly.qkv, ws->xn, m->H, ws->partial, T, dt, st));
  CK(umb_reduce_qkv_rope(ws->partial, ly.qkv.S, T, m->Hq, m->Hkv, m->D, m->Lmax, ws->pos, ws->slot, m->rope_cos,
                         m->rope_sin, ws->q, kc, vt, dt, st));
  CK(umb_tree_attn(ws->attn, ws->q, kc, vt, ws->attn_po, ws->attn_ml, ws->prefix, s->mask_bits, s->mask_words,
                   s->n_mask_keys, T, m->Hq, m->Hkv, m->D, m->Lmax, ws->attn_chunk, ws->attn_splits, m->attn_scale, dt,
                   st));
  CK(lin(ly.o, ws->attn, m->Hq * m->D, ws->partial, T, dt, st));
  CK(umb_reduce_residual_norm(ws->partial, ly.o.S, T, m->H, ws->h, ws->h, ws->xn, ly.norm2, m->eps, dt, st));
  // gate/up rows are interleaved at load time and the GEMM runs unsplit: SiLU(gate)*up is its epilogue
  if (ly.gu.S != 1) return UMB_EINVAL;
  CK(lin(ly.gu, ws->xn, m->H, (float*)ws->act, T, dt, st, /*epi=*/2));
  CK(lin(ly.down, ws->act, m->I, ws->partial, T, dt, st));
  CK(umb_reduce_residual_norm(ws->partial, ly.down.S, T, m->H, ws->h, ws->h, next_norm ? ws->xn : nullptr, next_norm,
                              m->eps, dt, st));
  return UMB_OK;
}

static int head(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, hipStream_t st) {
  if (s->head_from >= s->T) return UMB_OK;
  const int rows = s->T - s->head_from;
  const char* x = (const char*)ws->xn + (size_t)s->head_from * m->H * 2;
  return lin(m->lm_head, x, m->H, ws->logits, rows, m->dtype, st, /*epi=EPI_ROUND*/1);
}

extern "C" int umb_model_forward(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* s, hipStream_t st) {
  const int lb = s->layer_begin, le = s->layer_end;
  if (lb < 0 || le > m->L || lb >= le) return UMB_EINVAL;
  CK(prologue(m, ws, s, st));
  CK(umb_rmsnorm(ws->xn, ws->h, m->layers[lb].norm1, m->eps, s->T, m->H, m->dtype, st));
  for (int l = lb; l < le; ++l) {
    const void* nn = (l + 1 < le) ? m->layers[l + 1].norm1 : (le == m->L ? m->final_norm : nullptr);
    CK(layer(m, ws, s, m->layers[l], l, nn, st));
  }
  if (le == m->L) CK(head(m, ws, s, st));
  return UMB_OK;
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
  hipStream_t cs = (hipStream_t)off->copy_stream;
  auto issue_copy = [&](int l, int buf) -> int {
    // the slab may be overwritten only after the kernels that read it (ev_free is recorded on st)
    if (hipStreamWaitEvent(cs, (hipEvent_t)off->ev_free[buf], 0) != hipSuccess) return UMB_EHIP;
    if (hipMemcpyAsync(off->dev_slab[buf], off->host_slabs[l], off->slab_bytes, hipMemcpyHostToDevice, cs) != hipSuccess)
      return UMB_EHIP;
    if (hipEventRecord((hipEvent_t)off->ev_copied[buf], cs) != hipSuccess) return UMB_EHIP;
    return UMB_OK;
  };
  // order the copy stream behind everything already queued on st that may still read the slabs
  for (int b = 0; b < 2; ++b)
    if (hipEventRecord((hipEvent_t)off->ev_free[b], st) != hipSuccess) return UMB_EHIP;
  int next = lb, issued = 0, used = 0;
  auto advance = [&]() { while (next < le && off->host_slabs[next] == nullptr) ++next; };
  advance();
  for (int i = 0; i < 2 && next < le; ++i) { CK(issue_copy(next, issued & 1)); ++issued; ++next; advance(); }

  CK(prologue(m, ws, s, st));
  CK(umb_rmsnorm(ws->xn, ws->h, m->layers[lb].norm1, m->eps, s->T, m->H, m->dtype, st));
  for (int l = lb; l < le; ++l) {
    UmbLayer cur = m->layers[l];
    int buf = -1;
    if (off->host_slabs[l]) {
      buf = used & 1;
      if (hipStreamWaitEvent(st, (hipEvent_t)off->ev_copied[buf], 0) != hipSuccess) return UMB_EHIP;
      cur = rebase(m->layers[l], off->dev_slab[buf]);
    }
    const void* nn = (l + 1 < le) ? m->layers[l + 1].norm1 : (le == m->L ? m->final_norm : nullptr);
    CK(layer(m, ws, s, cur, l, nn, st));
    if (buf >= 0) {
      ++used;
      if (hipEventRecord((hipEvent_t)off->ev_free[buf], st) != hipSuccess) return UMB_EHIP;
      if (next < le) { CK(issue_copy(next, buf)); ++issued; ++next; advance(); }
    }
  }
  if (le == m->L) CK(head(m, ws, s, st));
  return UMB_OK;
}

extern "C" const char* umb_version(void) { return "umbrella_hip 0.1 (gfx950)"; }
