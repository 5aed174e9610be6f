/* libumbrella_hip.so -- C ABI of the MI355X (gfx950) speculative-decoding hot path.
 *
 * Drop-in boundary for Infini-AI-Lab/UMbreLLa's draft-expand / verify step.  The
 * reference (100 % Python) reaches native code only through seven third-party
 * symbols + torch; each entry point below names the reference call site(s) it
 * replaces (paths relative to the reference tree).
 *
 * Conventions: extern "C"; plain pointers and sizes (no torch types); every
 * function returns 0 on success or a negative errno-style code (-22 bad
 * argument, -5 HIP launch error); nothing throws; the caller owns every buffer;
 * the library allocates nothing; every call is asynchronous on `stream` and
 * safe to capture into a hipGraph (all run-time state -- prefix length,
 * positions, slots, accept results -- is read from device memory).
 *
 * dtype: 0 = fp16, 1 = bf16 (activations, dense weights, KV cache).
 */
#ifndef UMBRELLA_HIP_H
#define UMBRELLA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* KV cache layout (k_cache / vt_cache arguments below).  Per (layer, kv head) one slab: K of Lmax * D elements, V^T of
 * D * (Lmax + UMB_VT_PAD) elements (the first Lmax * D are used; the stride is kept from the earlier row-padded form),
 * Lmax a multiple of 32.  Inside a slab the elements are in MFMA FRAGMENT order, so that every load instruction of the
 * tree-attention kernels is one contiguous KiB:
 *   K  (key p, feature d): tile = p / 32, kk = p % 32, s = (kk / 4) % 2, j = 4 (kk / 8) + kk % 4
 *        offset = ((((tile * 2 + s) * (D / 32) + d / 32) * 64 + ((d / 8) % 4) * 16 + j) * 8 + d % 8
 *   V^T (feature d, key p): offset = (((p / 32) * (D / 16) + d / 16) * 64 + ((p / 8) % 4) * 16 + d % 16) * 8 + p % 8
 * (umbrella_amd/csrc/common.h kc_off / vt_off; Python mirror with converters: umbrella_amd/attn/cache.py).  Callers never
 * index the caches themselves: umb_reduce_qkv_rope* / umb_gemm_fused (epi 3) / umb_gemm_ll (epi 3) / umb_gemv (epi 3) /
 * umb_kv_append write keys, umb_kv_compact moves them, umb_tree_attn* reads them.  The reference's layouts
 * (umbrella/attn/cache.py:5-96 [L, Lmax, Hkv, D], :98-192 [L, Hkv, Lmax, D]) are what the converters speak. */
#ifndef UMB_VT_PAD
#define UMB_VT_PAD 32
#endif

typedef struct ihipStream_t* umb_stream_t;   /* == hipStream_t */

/* ------------------------------------------------------------------ weights (load time) */
/* dense W[N][K] row-major (HF layout; umbrella/models/llama_layer.py:25-40) -> MFMA tile order.
 * out: N*K 16-bit elements.  Row permutation `mode`: 0 none; 1 fused [gate; up] stack, packed rows (2m, 2m+1) <-
 * source rows (m, N/2 + m) so SiLU(gate)*up can be the GEMM epilogue; 2 fused [q | k | v] stack, inside each of the
 * first `rope_heads` heads of size D packed rows (2m, 2m+1) <- (m, m + D/2) so rotate-half RoPE is lane-local. */
int umb_repack_dense(void* out, const void* w, int N, int K, int mode, int D, int rope_heads, int dtype,
                     umb_stream_t stream);
/* AutoAWQ GEMM tensors (umbrella/quantization/awq_utils.py:20-27: qweight [K][N/8] i32,
 * qzeros [K/128][N/8] i32, scales [K/128][N] fp16) -> int4 tile order.
 * outw: N*K/2 bytes, meta: (N/16)*(K/128)*64 bytes (16 x {fp16 scale, fp16 zero} per tile). */
int umb_awq_repack(void* outw, void* meta, const void* qweight, const void* qzeros, const void* scales,
                   int N, int K, int group, int mode, int D, int rope_heads, umb_stream_t stream);

/* ------------------------------------------------------------------ linear layers */
/* split plan for a [N][K] linear: depends on (N, K, format) only, never on T. */
void umb_gemm_plan(int N, int K, int awq, int force_s1, int* R_out, int* S_out);
/* the full plan: R n-tiles per wave, S K-splits, tb n-tiles per 4-wave block (4 R, or one less where that makes
 * nblk * S whole rounds of 2 blocks per CU -- a launch is as fast as its busiest CU) and S_row = the split count to use
 * when the partials go to the one-block-per-row reduce kernel (0: the runtime's rule).  Shape-only. */
void umb_gemm_plan2(int N, int K, int awq, int force_s1, int* R_out, int* S_out, int* tb_out, int* S_row_out);
/* split count the model runtime uses for a T-token forward of that linear: the plan's S for T <= 64 (so a token's
 * result does not depend on its batch mates there); for wider forwards (tree verify, prompt chunks) the S <= plan
 * that fills the 2 x 256 block slots of the register-resident verify kernel about once. */
int umb_gemm_wide_split(int T, int N, int S_plan);
/* out[S][T][N] (fp32 split-K partials) = x[T][K] (row stride ldx) . W^T
 * replaces F.linear (umbrella/models/llama.py:89-91,103,107-111,133) and
 * AwqLinear.apply -> awq_ext.gemm_forward_cuda / dequantize_weights_cuda
 * (umbrella/quantization/awq_utils.py:63-86).  epi: 0 raw fp32 partials; 1 round results to `dtype`
 * (what F.linear(...).float() yields for the lm_head, llama.py:133); 2 fused SiLU(gate)*up
 * (llama.py:107-110): needs S == 1 and interleaved rows, `out` is then 16-bit act[T][N/2].
 * R: low byte = n-tiles per wave; bits 8..15 = the plan's `tb` word: n-tiles per block (0: 4 R), | 0x80 for 8-wave
 * blocks (one block per CU sharing one staged copy of the activations). */
int umb_gemm(void* out, const void* x, int ldx, const void* wpacked, const void* meta, int T, int N, int K,
             int awq, int S, int R, int epi, int dtype, umb_stream_t stream);

/* Fused work around the GEMM.  RMSNorm is split in two: the producer of the residual stream folds the norm WEIGHT
 * into the activations it hands over (hw = h * w) and leaves per-64-column sums of squares (ssq [T][stride]); the
 * consumer GEMM multiplies its OUTPUTS by rsqrt(sum(ssq[t])/ssq_dim + eps) (the per-token factor commutes with the
 * matmul).  epi 3 / 4 with S > 1: every K-split block publishes its fp32 partial tile, the last block to arrive on
 * the n-group's counter sums the S partials in split order and runs the epilogue (deterministic, no reduce kernel).
 *   epi 3: [q|k|v] rows (repack mode 2): 1/rms, RoPE at pos[t] (model_utils.py:17-52), q_out[T][Hq][D],
 *          K/V appended at slot[t] (attn/cache.py:53-65)
 *   epi 4: h <- round(round(gemm) + h); hw <- h * norm_w (optional); ssq_out[t][n/64] <- sum h^2 (llama.py:104,112) */
typedef struct UmbGemmFused {
  const float* ssq_in; int32_t ssq_groups; float ssq_dim; float eps;
  int32_t pad0;                     /* row stride of ssq_in in floats (0: ssq_groups) */
  uint32_t* counters;               /* >= N/64 zeroed words, self-resetting */
  void* h; void* hw; const void* norm_w; float* ssq_out; int32_t ssq_out_stride;
  int32_t pad1;                     /* bit 0: x is in FM layout (T <= 64); bit 1: the epi-2 output is written in FM layout */
  const int32_t* pos; const int32_t* slot; const void* cosT; const void* sinT;
  void* q_out; void* k_cache; void* vt_cache; int32_t Hq, Hkv, D, Lmax;
} UmbGemmFused;
int umb_gemm_fused(void* out, const void* x, int ldx, const void* wpacked, const void* meta, int T, int N, int K,
                   int awq, int S, int R, int epi, const UmbGemmFused* fx, int dtype, umb_stream_t stream);

/* ------------------------------------------------------------------ low-latency layer GEMMs (T <= 64 tokens)
 * One workgroup owns its output rows for the whole K (in-block K split, summed through LDS in slice order), so the
 * epilogue -- residual add, RMSNorm bookkeeping, SiLU * up, RoPE + KV append -- runs in the GEMM launch: 5 launches per
 * decoder layer, no fp32 partial round trips (csrc/lowlat.hip).  Activations are exchanged in "FM" layout = MFMA
 * B-fragment order [K/32][TT][64 lanes][8 x 16-bit], TT = umb_ll_token_tiles(T) token tiles of 16: element (t, k) at
 *   ((((k/32) * TT + t/16) * 64 + (k%32/8) * 16 + t%16) * 8 + k%8.
 * Replaces the same reference lines as umb_gemm_fused; AWQ int4 uses the folded form s * sum_k (q - z) x in fp32
 * (no fp16 rounding of (q - z) * s).  epi: 0 fp32 out[T - row_from][N] (* 1/rms, rounded to dtype if round_out);
 * 2 SiLU(gate) * up -> out = act in FM layout (K' = N/2); 3 q/k/v + RoPE + KV append; 4 residual. */
typedef struct UmbGemmLL {
  int32_t row_from, round_out;                 /* epi 0 */
  const float* ssq_in; int32_t ssq_groups, ssq_in_stride; float ssq_dim, eps;   /* 1/rms of the producer's stream; NULL: 1 */
  void* h; void* hw; const void* norm_w; float* ssq_out; int32_t ssq_out_stride, pad0;   /* epi 4: h row-major, hw FM */
  const int32_t* pos; const int32_t* slot; const void* cosT; const void* sinT;             /* epi 3 */
  void* q_out; void* k_cache; void* vt_cache; const void* bias; int32_t Hq, Hkv, D, Lmax;
} UmbGemmLL;
int umb_gemm_ll(void* out, const void* x_fm, const void* wpacked, const void* meta, int T, int N, int K, int awq,
                int epi, const UmbGemmLL* fx, int dtype, umb_stream_t stream);
/* ------------------------------------------------------------------ row-streaming GEMV (T <= 4 rows: draft tree levels)
 * out = epilogue(x @ w_rows^T) with x row-major [T][K] and w_rows a plain row-major [N][K] copy of the weights whose rows
 * are in the packed layouts' order (umb_repack_rows: mode 1 (gate, up) pairs, mode 2 RoPE partner pairs).  Every load is
 * issued at kernel start, a wave owns whole rows (8-row granularity: N = 2048 fills 256 CUs), v_dot2c accumulates in fp32.
 * Same reference lines and the same epilogue arithmetic as umb_gemm_ll (epi 2 / 3 / 4), with ROW-MAJOR act / hw outputs and
 * the sums of squares per workgroup: ssq_out[t][0 .. umb_gemv_groups(T, N, K)).  T <= 4; K must be 2048 (any epilogue) or 8192 (epi 4).
 * umb_gemv_ok: 1 if this (T, N, K, epi) is covered (0 with UMB_NO_GEMV=1).  ssq_groups <= 256 (EINVAL beyond: four per lane).
 * epi 3 trusts slot[t] in [0, Lmax) -- the fused epilogues are the model runtime's internal form; umb_kv_append is the
 * range-checked stand-alone one. */
int umb_gemv_ok(int T, int N, int K, int epi);
int umb_gemv_groups(int T, int N, int K);
int umb_gemv(void* out, const void* x, const void* w_rows, int T, int N, int K, int epi, const UmbGemmLL* fx, int dtype,
             umb_stream_t stream);
/* out[n][k] = w[rowmap(n)][k]: rows in the order of the packed layouts (mode 0 identity, 1 interleaved gate/up, 2 q/k RoPE
 * partner pairs over rope_heads heads of size D), 16-bit elements, K % 8 == 0 */
int umb_repack_rows(void* out, const void* w, int N, int K, int mode, int D, int rope_heads, umb_stream_t stream);

/* ------------------------------------------------------------------ persistent chain (csrc/chain.hip; round 5)
 * The draft model's <= 4-row layer as TWO launches: tree attention + ONE persistent launch that runs
 *   o-projection + residual -> gate/up + SiLU -> down-projection + residual -> the NEXT layer's q/k/v + RoPE + KV append
 * (umbrella/models/llama.py:461-533, LlamaCudagraph.layer_compute / graph_inference: the reference replays ~25 launches per
 * layer from one CUDA graph; umb_gemv runs 5).  One workgroup per CU: a loader wave streams the CU's rows of every op's
 * w_rows HBM -> LDS (LDS-DMA, 16 KiB slots, running ahead of the dependency edges), three consumer waves compute with
 * v_dot2; the [T][N] edges travel as 8-byte {tag, value} granules through `xchg`.  Bit-identical to the umb_gemv chain.
 * front: o / gate-up / down present (reads attn, h; writes h, hw = h * next_norm, ssq[t][0 .. 256)); tail: q/k/v present
 * (writes q_out and the K / V^T caches at slot[t]).  front = 0, tail = 1 is the forward's first q/k/v: x = hw, 1/rms from
 * ssq[t][0 .. ssq_groups_in).  xchg: umb_chain_xchg_bytes(Tmax, H, I) device bytes, zeroed once, then the epoch word at
 * its end set to 1 (umb_chain_xchg_init); a launch that gives up on a hand-off (every spin is bounded: 20 ms, or
 * UMB_CHAIN_TIMEOUT_MS) ORs 0xDEADxxxx into the status word (umb_chain_status).  Needs every workgroup resident: one
 * process per device.  umb_chain_ok: 1 where the shape is covered (H 2048, I 8192, T <= 3, no q/k/v bias, 256 CUs;
 * 0 with UMB_NO_CHAIN=1).  The exchange LAYOUT is fixed at UMB_CHAIN_TMAX rows whatever the forward's T or the workspace's
 * Tmax: allocate umb_chain_xchg_bytes(UMB_CHAIN_TMAX, H, I), initialise and query with the same constant. */
#define UMB_CHAIN_TMAX 4
typedef struct UmbChain {
  const void* w_o; const void* w_gu; const void* w_down; const void* w_qkv;   /* umb_repack_rows copies */
  const void* attn; void* h; void* hw; float* ssq;
  const void* norm2; const void* next_norm;
  const int32_t* pos; const int32_t* slot; const void* cosT; const void* sinT; void* q_out; void* k_cache; void* vt_cache;
  void* xchg;
  int32_t T, Tmax, front, tail, H, I, NQKV, ssq_stride, ssq_groups_in, Hq, Hkv, D, Lmax;
  float eps;
} UmbChain;
int umb_chain_ok(int T, int H, int I, int NQKV, int D, int has_bias);
size_t umb_chain_xchg_bytes(int Tmax, int H, int I);
int umb_chain_xchg_init(void* xchg, int Tmax, int H, int I, umb_stream_t stream);
int umb_chain_status(const void* xchg, int Tmax, int H, int I, uint32_t* status_out, umb_stream_t stream);
int umb_draft_chain(const UmbChain* c, int dtype, umb_stream_t stream);
/* lm_head of a <= 8-row forward of the same models on the same engine (`self.lm_head(hidden_states)` behind the final norm,
 * umbrella/models/llama.py:130-133): ONE workgroup per CU streams its share of the head's rows -- plain row-major [V][2048], a tied
 * model's embedding table as it is -- through an LDS ring and multiplies from there; no hand-off between workgroups.  x: [rows][2048]
 * = h * final-norm weight; ssq: the producer's sums of squares ([rows][ssq_stride], `groups` valid: 1/rms is applied to the outputs);
 * logits: fp32 [rows][V], every value rounded to the model dtype as F.linear returns it.  x_fm_tt: 0 = x row-major, > 0 = x in FM
 * order with that many token tiles (the low-latency schedule's activations).  umb_head_stream_ok: rows <= 8, H == 2048,
 * V % 4 == 0, a 256-CU device (UMB_NO_HEAD_STREAM answers 0: the MFMA kernel of umb_gemm takes the head). */
int umb_head_stream_ok(int rows, int V, int H);
int umb_head_stream(float* logits, const void* x, const float* ssq, int ssq_stride, int groups, float eps, const void* w_rows,
                    int rows, int V, int H, int x_fm_tt, int dtype, umb_stream_t stream);

/* (R n-tiles per wave, WN row groups x WK K-slices = NW waves per block) for a [N][K] linear: shape-only, so a
 * token's result never depends on its batch mates; epi 4 writes N / 16 / R sums of squares per token. */
void umb_ll_plan(int N, int K, int awq, int* R_out, int* WN_out, int* WK_out, int* NW_out);
int umb_ll_token_tiles(int T);                  /* 1, 2 or 4 for T = 1..64; 0 otherwise */
int umb_to_fm(void* out_fm, const void* x, int T, int K, int dtype, umb_stream_t stream);      /* row-major -> FM */
int umb_from_fm(void* out, const void* x_fm, int T, int K, int dtype, umb_stream_t stream);
/* measurement probe (bench.py / scripts): one read-only pass over `bytes` (multiple of 16) of device memory with
 * 16-byte non-temporal loads -- the streaming rate the device delivers to a kernel; sink: 4 writable device bytes.
 * No reference counterpart. */
int umb_stream_read(const void* p, size_t bytes, void* sink, umb_stream_t stream);
/* umb_embed_prep for the low-latency schedule: h row-major, hw = h * norm_w in FM layout, ssq[t][0..4); tokens /
 * positions / slots are clamped into [0, V) / [0, Lmax) (a tree that overruns the context cannot write past the caches) */
int umb_embed_ll(void* h, const void* table, int H, int V, int Lmax, int T, const int* tok, const int* pos,
                 const int* slot, const int* prefix, const int* tokens_all, const int* n_ptr, int off,
                 const int* depth, int* pos_out, int* slot_out, int* prefix_out, void* hw_fm, const void* norm_w,
                 float* ssq, int ssq_stride, int dtype, umb_stream_t stream);

/* ------------------------------------------------------------------ stand-alone epilogues (op-level API) */
/* flashinfer.rmsnorm (umbrella/models/model_utils.py:54-64) */
int umb_rmsnorm(void* out, const void* x, const void* w, float eps, int rows, int H, int dtype, umb_stream_t stream);
/* the same with the output in FM order (MFMA B-fragment order, [H/32][fm_tt token tiles][64 lanes][8]: what the GEMMs of a
 * <= 64-row forward read as contiguous KiB fragments); fm_tt = 0: row-major, else 1 / 2 / 4 tiles of 16 rows (umb_ll_token_tiles) */
int umb_rmsnorm_fm(void* out, const void* x, const void* w, float eps, int rows, int H, int fm_tt, int dtype, umb_stream_t stream);
/* h = residual + sum_s partial ; xn = rmsnorm(h)*w   (llama.py:104-106,112-113 + next layer's :87) */
int umb_reduce_residual_norm(const void* partial, int S, int T, int N, const void* residual, void* h_out,
                             void* xn_out, const void* w, float eps, int dtype, umb_stream_t stream);
/* the same with xn_out in FM order (xn_fm_tt as fm_tt of umb_rmsnorm_fm; h_out stays row-major) */
int umb_reduce_residual_norm_fm(const void* partial, int S, int T, int N, const void* residual, void* h_out,
                                void* xn_out, const void* w, float eps, int xn_fm_tt, int dtype, umb_stream_t stream);
/* the split-K reduce with the RMSNorm DEFERRED, N / 512 blocks per token row (round 6): h_out = round(round(sum_s partial) +
 * residual); hw_out (may be NULL) = round(h * w), row-major or FM (hw_fm_tt as above); ssq_out[t][0 .. N/512) = the blocks' sums
 * of h^2 -- the consumer applies rsqrt(sum ssq / N + eps) to its outputs (umb_gemm_fused ssq_in, umb_reduce_qkv_rope2).
 * N % 512 == 0, ssq_stride >= N / 512.  Same reference lines as umb_reduce_residual_norm. */
int umb_reduce_residual_hw(const void* partial, int S, int T, int N, const void* residual, void* h_out, void* hw_out,
                           const void* w, float* ssq_out, int ssq_stride, int hw_fm_tt, int dtype, umb_stream_t stream);
/* act = silu(gate) * up  (llama.py:107-110); partial rows are [gate | up] */
int umb_reduce_silu_mul(const void* partial, int S, int T, int I, void* act, int dtype, umb_stream_t stream);
/* q/k/v split + apply_rotary_pos_emb (umbrella/models/model_utils.py:17-52) at positions pos[t]
 * + KV_Cache.update_kv_cache (umbrella/attn/cache.py:53-65) at slots slot[t].
 * K / V^T caches: layer base pointers, one fragment-ordered slab per kv head (layout note at the top of this file).
 * paired != 0: the q/k rows of the linear were packed as RoPE partner pairs (repack mode 2).
 * bias: NULL or the fused [q|k|v] projection bias [(Hq + 2 Hkv) D] in the model dtype, HF feature order
 * (Qwen2: umbrella/models/qwen.py:94-96). */
int umb_reduce_qkv_rope(const void* partial, int S, int T, int Hq, int Hkv, int D, int Lmax, const int* pos,
                        const int* slot, const void* cosT, const void* sinT, void* q_out, void* k_cache,
                        void* vt_cache, int paired, const void* bias, int dtype, umb_stream_t stream);
/* the same with the per-token 1/rms of the low-latency schedule applied to the reduced outputs (the GEMM ran on h * w):
 * inv[t] = rsqrt(sum_{g < ssq_groups} ssq_in[t * ssq_stride + g] / ssq_dim + eps); ssq_in == NULL: umb_reduce_qkv_rope */
int umb_reduce_qkv_rope2(const void* partial, int S, int T, int Hq, int Hkv, int D, int Lmax, const int* pos,
                         const int* slot, const void* cosT, const void* sinT, void* q_out, void* k_cache,
                         void* vt_cache, int paired, const void* bias, const float* ssq_in, int ssq_groups,
                         int ssq_stride, float ssq_dim, float eps, int dtype, umb_stream_t stream);
/* Stand-alone forms on 16-bit tensors, for an integrator who replaces the reference call site by call site
 * (INTEGRATION.md option B) instead of adopting the fused layer chain:
 * apply_rotary_pos_emb (umbrella/models/model_utils.py:17-52) in place at positions pos[t]: q [T][Hq][D] and
 * k [T][Hkv][D] when layout == 0 (NHD, the reference's [1, T, H, D] tensors), [H][T][D] when layout == 1;
 * x * cos + rotate_half(x) * sin in the model dtype (each product and the sum rounded, as eager torch does).
 * cosT / sinT: [Lmax][D] model dtype (llama.py:48-60).  Either of q / k may be NULL with its head count 0. */
int umb_rope_inplace(void* q, void* k, const void* cosT, const void* sinT, const int* pos, int T, int Hq, int Hkv,
                     int D, int layout, int dtype, umb_stream_t stream);
/* KV_Cache.update_kv_cache / StaticKV_Cache index_copy_ (umbrella/attn/cache.py:53-65, 155-156): k, v [T][Hkv][D]
 * 16-bit -> key slot[t] of the K and V^T caches (fragment-ordered slabs, layout note at the top of this file)
 * (layer base pointers).  Slots outside [0, Lmax) are dropped. */
int umb_kv_append(void* k_cache, void* vt_cache, const void* k, const void* v, const int* slot, int T, int Hkv, int D,
                  int Lmax, int dtype, umb_stream_t stream);
/* LlamaAwqLayer.copy / LlamaLayer.copy (umbrella/models/llama_layer.py:244-258, 23 small copies per layer there): ONE
 * hipMemcpyAsync of a whole layer slab from pinned host memory on `copy_stream`.  ev_free (hipEvent_t or NULL): the
 * copy waits for it on copy_stream (record it on the compute stream behind the kernels that last read `dst`);
 * ev_copied (hipEvent_t or NULL): recorded behind the copy (the compute stream waits for it before using `dst`). */
int umb_h2d_layer(void* dst, const void* src_pinned, size_t bytes, umb_stream_t copy_stream, void* ev_free,
                  void* ev_copied);
/* F.embedding (llama.py:124) + per-forward position/slot/prefix resolution.
 * explicit mode: tok/pos/slot/prefix given.  tree mode (tokens_all != NULL):
 * token i = tokens_all[*n_ptr + off + i], position = *n_ptr + depth[off+i], slot = *n_ptr + off + i.
 * table == NULL: indices only (pipeline stage > 0; x already holds the activations).  hw/norm_w/ssq (optional):
 * hw = x * norm_w and ssq[t][c/64] = sum of x^2, the inputs of the fused layer chain. */
int umb_embed_prep(void* x, const void* table, int H, int T, const int* tok, const int* pos, const int* slot,
                   const int* prefix, const int* tokens_all, const int* n_ptr, int off, const int* depth,
                   int* pos_out, int* slot_out, int* prefix_out, void* hw, const void* norm_w, float* ssq,
                   int ssq_stride, int dtype, umb_stream_t stream);
/* the same with hw in FM order (hw_fm_tt token tiles, 0 = row-major).  The sums of squares keep their per-64-column grouping, so a
 * token's 1 / rms is the same bits whether its forward carries 1 row or 64 (batch invariance of the deferred-norm schedule). */
int umb_embed_prep_fm(void* x, const void* table, int H, int T, const int* tok, const int* pos, const int* slot,
                      const int* prefix, const int* tokens_all, const int* n_ptr, int off, const int* depth,
                      int* pos_out, int* slot_out, int* prefix_out, void* hw, const void* norm_w, float* ssq,
                      int ssq_stride, int hw_fm_tt, int dtype, umb_stream_t stream);

/* ------------------------------------------------------------------ attention */
/* flashinfer.single_prefill_with_kv_cache(custom_mask=...) (umbrella/attn/cache.py:77-85) and
 * StaticKV_Cache.compute_attention (cache.py:169-192).  Keys [0,*prefix_len) are visible to every
 * row; key *prefix_len + b is visible to row t iff bit b of mask_bits[t*mask_words ...] (NULL: b <= t).
 * po: [max_splits][T][Hq][D] fp32, pml: [max_splits][T][Hq][2] fp32 scratch; chunk*max_splits >= Lmax.
 * One launch: 8 waves per (kv head, 16-row query tile) split the keys and merge in LDS; contexts beyond 2048 keys
 * use further 2048-key span blocks merged by the last arriver, which needs `counters` (>= Hkv * ceil(T*(Hq/Hkv)/16)
 * zeroed words, self-resetting).  counters == NULL: one launch only if Lmax <= 2048, else key splits of `chunk`
 * keys + a combine kernel. */
int umb_tree_attn(void* out, const void* q, const void* k_cache, const void* vt_cache, void* po, void* pml,
                  const int* prefix_len, const void* mask_bits, int mask_words, int n_mask_keys, int T, int Hq,
                  int Hkv, int D, int Lmax, int chunk, int max_splits, float scale, uint32_t* counters, int dtype,
                  umb_stream_t stream);

/* the same, output optionally in FM layout (out_fm_tt = umb_ll_token_tiles(T); 0 = row-major [T][Hq][D]); FM output
 * needs the single-launch path (counters != NULL or Lmax <= 2048) */
int umb_tree_attn2(void* out, const void* q, const void* k_cache, const void* vt_cache, void* po, void* pml,
                   const int* prefix_len, const void* mask_bits, int mask_words, int n_mask_keys, int T, int Hq,
                   int Hkv, int D, int Lmax, int chunk, int max_splits, float scale, uint32_t* counters, int out_fm_tt,
                   int dtype, umb_stream_t stream);

/* ------------------------------------------------------------------ tree bookkeeping */
/* target_logits.argmax(-1) (static_speculation_engine.py:307) */
int umb_argmax_rows(int* out, const float* logits, int rows, int V, umb_stream_t stream);
/* topk per row (speculation_utils.py:57-61; dynamic_speculation_engine.py:236); optional Sequoia child
 * placement tokens_all[*n_ptr + child_start[row] + r] = idx[r], r < child_cnt[row] (static:115-123,279-281) */
int umb_topk_rows(int* out_idx, float* out_val, const float* logits, int rows, int V, int k, int* tokens_all,
                  const int* n_ptr, const int* child_start, const int* child_cnt, umb_stream_t stream);
/* the same with each row's vocabulary split over 16 blocks (a single block reads a row at ~50 GB/s); workspace =
 * 4096 bytes of zeroed row counters (self-resetting) + rows * 16 * k * 8 bytes; falls back to umb_topk_rows when the
 * workspace is NULL / too small or V < 16384. */
int umb_topk_rows_ws(int* out_idx, float* out_val, const float* logits, int rows, int V, int k, int* tokens_all,
                     const int* n_ptr, const int* child_start, const int* child_cnt, void* workspace,
                     size_t workspace_bytes, umb_stream_t stream);
/* verification sampling, one token per tree node (static_speculation_engine.py:298-310,
 * dynamic_speculation_engine.py:266-281; helpers speculation_utils.py:340-352; replaces
 * flashinfer.sampling.top_k_top_p_sampling_from_logits / top_p_renorm_prob + torch.multinomial):
 * HF repetition penalty over tokens_all[0..*n_ptr] when penalty > 1.01 (applied to `logits` in place), then
 * arg-max if temperature < 0.05, else top-k (ties at the k-th value kept) -> softmax(x / temperature) -> top-p
 * renormalisation -> one draw keyed by (*seed [u64, device], *n_ptr, row).  Optional dbg_k > 0: the first dbg_k
 * entries of the sorted filtered distribution (token or -1, renormalised probability) per row.
 * Needs topk <= 1024, topp > 0, V <= 1048576 when the penalty is on. */
int umb_sample_rows(int* sampled, float* logits, int rows, int V, const int* tokens_all, const int* n_ptr,
                    float penalty, float temperature, int topk, float topp, const void* seed, int dbg_k,
                    int* dbg_idx, float* dbg_p, umb_stream_t stream);
/* The same pipeline with the REFERENCE's draw (static_speculation_engine.py:131,310: ONE uniform_samples = rand(3, T)
 * tensor, reused by every verify): penalty -> top-k mask -> softmax(x / temperature) -> flashinfer 0.2.x
 * TopPSamplingFromProb, the rejection sampler over caller-supplied uniforms u[round * ustride + row], `rounds` rounds
 * (3 in the reference), cumulative sums in vocabulary order, the last round's token returned whether accepted or not
 * (restated in oracle/ops.py: top_k_top_p_sampling_from_logits).  Same logits + same uniforms -> the reference's token.
 * dbg_*: the top-k set (token, softmax probability) per row, sorted by probability. */
int umb_sample_rows_uniform(int* sampled, float* logits, int rows, int V, const int* tokens_all, const int* n_ptr,
                            float penalty, float temperature, int topk, float topp, const float* uniforms, int rounds,
                            int ustride, int dbg_k, int* dbg_idx, float* dbg_p, umb_stream_t stream);
/* SpecExec beam expansion of one level (dynamic_speculation_engine.py:236-248) */
int umb_beam_expand(const int* top_idx, const float* top_val, int w, int B, int W, int lvl_off, float* tree_score,
                    int* parents, int* tokens_all, const int* n_ptr, void* mask_bits, int mask_words,
                    umb_stream_t stream);
/* accept scan + token / num_nodes update + EOS search (static:313-341, dynamic:283-316).
 * out5 = {kept, bonus token, eos hit, new num_nodes, raw accept length}; path[i] = i-th accepted tree index */
int umb_accept_scan(const int* sampled, const int* parents, int* tokens_all, int* n_ptr, int T, const int* eos,
                    int n_eos, int* out5, int* path, umb_stream_t stream);
/* KV_Cache.gather_kv_incremental (umbrella/attn/cache.py:41-49) without the tail memset */
int umb_kv_compact(void* k_cache, void* vt_cache, const int* res, const int* path, int L, int Hkv, int D, int Lmax,
                   int max_path, int dtype, umb_stream_t stream);
int umb_set_int(int* p, int v, umb_stream_t stream);
/* target_logits[-1:, eos] = -inf (dynamic_speculation_engine.py:130,163) */
int umb_mask_eos(float* logits_row, const int* eos, int n_eos, umb_stream_t stream);
int umb_write_token(int* tokens_all, const int* n_ptr, const int* src, umb_stream_t stream);
/* measurement knob: tokens_all[*n_ptr + off + i] = tbl[off + i] where tbl >= 0 (controllable-acceptance draft), i < cnt <= 1024.
 * parents (tree tables, may be NULL): a sibling that already carries the forced token takes the displaced one, so the
 * children of one parent stay distinct -- as the top-k that drafted them guarantees (static_speculation_engine.py:279-281). */
int umb_apply_override(int* tokens_all, const int* n_ptr, const int* tbl, const int* parents, int off, int cnt,
                       umb_stream_t stream);

/* ------------------------------------------------------------------ whole-model forward */
typedef struct UmbLinear {
  const void* w;        /* packed tiles */
  const void* meta;     /* AWQ scale/zero tiles or NULL */
  int32_t N, K, awq, R, S;
  int32_t tb, S_row;    /* umb_gemm_plan2: n-tiles per block (0: 4 R); split count under the row reduce (0: rule) */
  int32_t pad_;
  const void* w_rows;   /* NULL, or the same weights as plain row-major [N][K] rows in packed order (umb_repack_rows): the
                           GEMV family's operand for forwards of <= 4 rows (dense 16-bit, resident layers only) */
} UmbLinear;

typedef struct UmbLayer {
  UmbLinear qkv, o, gu, down;       /* fused [q|k|v], o_proj, fused [gate|up], down_proj */
  const void* norm1;                /* input_layernorm weight [H] */
  const void* norm2;                /* post_attention_layernorm weight [H] */
  const void* qkv_bias;             /* NULL, or fused [q|k|v] bias (Qwen2), model dtype, HF feature order */
} UmbLayer;

typedef struct UmbModel {
  int32_t dtype, L, H, I, Hq, Hkv, D, V, Lmax, pad_;
  float eps, attn_scale;
  const void* embed;                /* [V][H] row-major */
  UmbLinear lm_head;
  const void* final_norm;
  const void* rope_cos;             /* [Lmax][D] model dtype (umbrella/models/llama.py:48-60) */
  const void* rope_sin;
  void* k_cache;                    /* [L][Hkv] slabs of Lmax * D elements, fragment order (note at the top) */
  void* vt_cache;                   /* [L][Hkv] slabs of D * (Lmax + UMB_VT_PAD) elements, fragment order */
  const UmbLayer* layers;           /* host array, L entries */
} UmbModel;

typedef struct UmbWorkspace {
  void* h;  void* xn;  void* q;  void* attn;  void* act;   /* [Tmax][H|H|Hq*D|Hq*D|I] 16-bit */
  float* partial;                   /* split-K partials, >= max S*Tmax*N floats */
  float* attn_po;  float* attn_ml;
  int32_t* pos;  int32_t* slot;  int32_t* prefix;
  float* logits;                    /* [Tmax][V] */
  void* hw;                         /* [Tmax][H] residual stream with the next RMSNorm weight folded in */
  float* ssq;                       /* [Tmax][ssq_stride] per-64-column sums of squares of h (zero padded) */
  uint32_t* counters;               /* >= max(N)/64 zeroed words for the split-K last-arriver epilogues */
  uint32_t* attn_counters;          /* >= Hkv * ceil(Tmax*(Hq/Hkv)/16) zeroed words */
  int32_t Tmax, attn_chunk, attn_splits, ssq_stride;
  int32_t fused, defer_norm;        /* defer_norm (schedule 0 only, no tensor parallelism, H % 512 == 0): the two residual reduces
                                       run N / 512 blocks per row and leave hw = h * w_next + sums of squares; the consumers apply
                                       1 / rms (umb_reduce_residual_hw).
                                       layer schedule: 0 = 8 launches (split-K reduced at kernel boundaries), 1 = 5 launches
                                       (in-kernel last-arriver reduces; slower), 2 = low-latency 5 launches (whole-K
                                       workgroups, FM activations; T <= 64 only, wider forwards use schedule 0) */
  void* chain_xchg;                 /* NULL, or umb_chain_xchg_bytes(UMB_CHAIN_TMAX, H, I) bytes set up by
                                       umb_chain_xchg_init(.., UMB_CHAIN_TMAX, ..) -- NOT this workspace's Tmax: the epoch and status
                                       words sit behind a 4-row layout.  Forwards of <= 3 rows of a GEMV-role model then run the
                                       persistent chain (csrc/chain.hip) */
} UmbWorkspace;

typedef struct UmbStep {
  int32_t T;
  int32_t tree_off;                 /* tree mode: offset of the first query row inside the tree */
  const int32_t* tokens;            /* explicit mode */
  const int32_t* positions;
  const int32_t* slots;
  const int32_t* prefix_len;
  const int32_t* tokens_all;        /* tree mode: engine token buffer (NULL -> explicit mode) */
  const int32_t* n_ptr;             /* device scalar: num_nodes */
  const int32_t* depth;             /* tree depth table */
  const void* mask_bits;            /* rows for the T queries (row stride mask_words u64), NULL = causal */
  int32_t mask_words, n_mask_keys;
  int32_t head_from;                /* lm_head over rows [head_from, T) -> logits rows [0, T-head_from); >= T: skip */
  int32_t layer_begin, layer_end;   /* pipeline stage [begin, end); 0, L for the whole model */
  int32_t skip_embed;               /* stage > 0: ws.h already holds the incoming activations;
                                       positions/slots/prefix are still resolved */
} UmbStep;

/* Llama*.inference (umbrella/models/llama.py:117-134, 305-322): embedding -> layers -> norm -> lm_head.
 * fp32 logits land in ws->logits. */
int umb_model_forward(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* step, umb_stream_t stream);

/* Tensor-parallel forward (no reference counterpart -- the reference has no distributed code): `m` is
 * this rank's 1/world shard of the model (q/k/v and gate/up column-split by head / MLP column, o and down row-split,
 * lm_head vocabulary-split; embedding, norms and the residual stream replicated).  After each row-split GEMM the
 * library calls `allreduce(ctx, buf, count, stream)`: sum `count` fp32 values at `buf` over all ranks, in place, ordered
 * on `stream` (RCCL ncclAllReduce on that stream -- capturable into a hipGraph like every kernel of the forward -- or
 * any other transport); non-zero return aborts the forward with -5.  Two calls per layer; world == 1 (or a NULL
 * descriptor): none, identical to umb_model_forward.  Needs ws->fused == 0 (the 8-launch schedule).  The logits of rows
 * [head_from, T) land in ws->logits as [rows][lm_head.N] = this rank's vocabulary slice. */
typedef int (*umb_allreduce_fn)(void* ctx, float* buf, int64_t count, umb_stream_t stream);
/* Direct-xGMI all-reduce of the small tiles (csrc/tp.hip; SURVEY 8(f)1 "fused all-reduce over xGMI", :248): every rank
 * owns an exchange buffer [2][cap] floats + an epoch flag word, mapped into all peers (hipIpc*, umb_tp_xchg_*);
 * umb_tp_publish sums this rank's split-K slabs into its slot and publishes the call's epoch at system scope,
 * umb_tp_reduce_residual_norm waits for every peer's epoch and reads the P tiles through the peer pointers, summing in
 * rank order (bit-identical on every rank), then residual + RMSNorm as umb_reduce_residual_norm.  No collective library,
 * no ring: two launches in the rank's own stream where the single-GPU schedule has two (sum / reduce).
 * slot[r], flag[r]: rank r's buffer / flag as mapped HERE (own entries: the local pointers).  epoch, arrive, status:
 * local zero-initialised device words (calls so far; self-resetting block counter; 0 or 0xDEADxxxx after a peer failed
 * to arrive within spin_limit polls, 0: ~70 s).  status points at a zero-initialised 128-byte line: the word 64 bytes
 * behind it is the library's local release word (block 0 polls the peers, the other row blocks wait on it). */
#define UMB_TP_MAX_RANKS 16
typedef struct UmbTPPeer {
  int32_t rank, world;
  float* slot[UMB_TP_MAX_RANKS];
  uint32_t* flag[UMB_TP_MAX_RANKS];
  uint32_t* epoch; uint32_t* arrive; uint32_t* status;
  int64_t cap;                      /* floats per slot */
  int64_t spin_limit;
} UmbTPPeer;
int umb_tp_publish(const UmbTPPeer* peer, const float* partial, int S, int64_t n, umb_stream_t stream);
int umb_tp_reduce_residual_norm(const UmbTPPeer* peer, int T, int N, const void* residual, void* h_out, void* xn_out,
                                const void* w, float eps, int dtype, umb_stream_t stream);
/* exchange memory: device allocation (fine-grained where the runtime allows; *fine_grained says which) + its 64-byte
 * interprocess handle; open / close a peer's handle; free one's own.  Host-synchronous, load time only. */
int umb_tp_xchg_alloc(size_t bytes, void** ptr, void* handle64, int* fine_grained);
int umb_tp_xchg_open(const void* handle64, void** ptr);
int umb_tp_xchg_close(void* ptr);
int umb_tp_xchg_free(void* ptr);
typedef struct UmbTP {
  int32_t rank, world;
  umb_allreduce_fn allreduce;       /* the collective hook (RCCL on the launch stream; host-staged gloo in tests) */
  void* ctx;
  const UmbTPPeer* peer;            /* NULL: every tile goes through the hook */
  int64_t peer_max_floats;          /* tiles of at most this many floats take the peer path (one-shot reads move P - 1
                                       tiles per rank, a ring 2 (P - 1) / P: large tiles stay on the hook) */
} UmbTP;
int umb_model_forward_tp(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* step, const UmbTP* tp,
                         umb_stream_t stream);
/* partial[0][i] += partial[1][i] + ... + partial[S-1][i] for i < n, in split order (the fixed order every reduce here uses) */
int umb_sum_splits(float* partial, int S, int64_t n, umb_stream_t stream);

/* LlamaOffload / LlamaAwqOffload.inference (llama.py:196-219): layers whose weights live in pinned
 * host slabs are streamed into two device slabs on copy_stream, event-ordered against compute.
 * host_slabs[l] == NULL -> layer l is device resident (its UmbLayer pointers are used as is);
 * otherwise layers[l] pointers are offsets relative to the slab base. */
#define UMB_MAX_SLABS 8
typedef struct UmbOffload {
  void* const* host_slabs;          /* L entries */
  size_t slab_bytes;
  void* dev_slab[UMB_MAX_SLABS];    /* n_slabs device slabs used as a ring (the reference has two, llama.py:160-167) */
  umb_stream_t copy_stream;
  void* ev_copied[UMB_MAX_SLABS];   /* hipEvent_t */
  void* ev_free[UMB_MAX_SLABS];
  int32_t* prefetched;              /* host int32[UMB_MAX_SLABS] owned by the caller, initialised to -1 (NULL: no
                                       cross-forward prefetch): the layers whose slabs the previous forward left in flight
                                       for this one.  The reference's (idx + 1) % num_layers copy (llama.py:203-209): the
                                       next forward's first n_slabs streamed layers move while lm_head, sampling, the next
                                       draft tree and -- with a device-resident prefix (num_cache_layers) -- the resident
                                       layers run: more than two slabs keep the link busy through that prefix. */
  int32_t n_slabs;                  /* 2 .. UMB_MAX_SLABS */
  int32_t pad_;
} UmbOffload;
int umb_model_forward_offload(const UmbModel* m, const UmbWorkspace* ws, const UmbStep* step, const UmbOffload* off,
                              umb_stream_t stream);

/* diagnostic: n dependent no-op launches of `blocks` workgroups (dispatch cadence) */
int umb_bench_launch(int n, int blocks, int* p, umb_stream_t stream);

const char* umb_version(void);

#ifdef __cplusplus
}
#endif
#endif
