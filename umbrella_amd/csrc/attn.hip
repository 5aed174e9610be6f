// Tree-masked GQA attention for the draft-expand / verify step (flash-decoding style).
//
// Replaces flashinfer.single_prefill_with_kv_cache(custom_mask=...) at
// umbrella/attn/cache.py:77-85 and the eager masked attention at cache.py:169-192.
//
// Instead of a dense [T, n+T] bool mask the kernel takes (prefix_len, bit-packed
// mask over the key slots after the prefix): key slot c < prefix_len is visible to
// every row; slot prefix_len + b is visible to row t iff bit b of mask_bits[t] is
// set (mask_bits == null: causal, b <= t).
//
// Layout (chosen for 16-byte MFMA fragment loads, wave64):
//   q   [T][Hq][D] row-major.  K / V^T caches: one slab per (layer, kv head) of Lmax D / D (Lmax + UMB_VT_PAD) elements whose
//   INSIDE is in MFMA fragment order since round 4 (common.h: kc_off / vt_off) -- a 32-key tile of K is 2 x D/32 fragments,
//   of V^T D/16 fragments, 1 KiB each, so every load instruction below reads ONE contiguous KiB.  (Rounds 1-3: row-major
//   K [Hkv][Lmax][D], V^T [Hkv][D][Lmax + UMB_VT_PAD]; the slab strides are unchanged.)  Key offsets handed to the tile
//   address (k0 >> 5) are multiples of 32: Lmax % 32 == 0, span and chunk sizes are multiples of 256 (checked at the entry points).
// Per kv head the T*g query rows (g = Hq/Hkv) form 16-row tiles.  The kernel
// computes S^T = K Q^T so each lane owns ONE query column: softmax statistics
// are lane-local (+2 cross-lane xor steps), and the exp'd scores are already in
// B-operand layout for O^T = V^T P^T -- no LDS, no transposes.
//   grid = (Hkv, query-tile groups, key splits); partial (m, l, O) per split are
//   merged by attn_combine_kernel.
#include "common.h"

#define NEG_BIG (-1.0e30f)

template <typename P, int D>
__global__ __launch_bounds__(256) void tree_attn_kernel(const u16* __restrict__ q, const u16* __restrict__ kc,
                                                        const u16* __restrict__ vt, float* __restrict__ po,
                                                        float* __restrict__ pml, const int* __restrict__ prefix_p,
                                                        const unsigned long long* __restrict__ mask_bits,
                                                        int mask_words, int n_mask_keys, int T, int Hq, int Hkv,
                                                        int Lmax, int chunk, int qtiles_per_wave, float scale) {
  constexpr int DS = D / 32;     // k-steps for Q K^T
  constexpr int DT = D / 16;     // 16-row d tiles of O^T
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, gq = lane >> 4;
  // grid = (Hkv, query-tile groups, key splits): workgroups are dealt round-robin to the 8 XCDs by linear id, so
  // kv head h stays on XCD h % 8 (its K/V is fetched into one L2) and the splits past kv_end -- which exit at
  // once -- are the slowest dimension.  (With the splits fastest, a 2-split verify ran on 2 of the 8 XCDs.)
  const int sp = blockIdx.z, h = blockIdx.x, zq = blockIdx.y;
  const int g = Hq / Hkv;
  const int nrows = T * g;
  const int prefix = *prefix_p;
  const int kv_end = prefix + n_mask_keys;
  const int k_lo = sp * chunk;
  if (k_lo >= kv_end) return;
  const int k_hi = min(kv_end, k_lo + chunk);
  const u16* kbase = kc + (long)h * Lmax * D;
  const long LV = VT_LD(Lmax);
  const u16* vbase = vt + (long)h * D * LV;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  for (int qi = 0; qi < qtiles_per_wave; ++qi) {
    const int qt = (zq * 4 + wv) * qtiles_per_wave + qi;
    if (qt * 16 >= nrows) break;
    const int row = qt * 16 + j;                 // this lane's query row (column of S^T)
    const bool row_ok = row < nrows;
    const int t = row_ok ? row / g : 0;
    const int hq = h * g + (row_ok ? row % g : 0);
    u32x4 bq[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds)
      bq[ds] = row_ok ? *reinterpret_cast<const u32x4*>(q + ((long)t * Hq + hq) * D + ds * 32 + gq * 8) : zero4;

    float m = NEG_BIG, l = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K fragments of one 32-key tile: two 16-key subtiles; row i of subtile s <-> key k0 + (i>>2)*8 + s*4 + (i&3)
    auto load_k = [&](int k0, u32x4 (&ak)[2][DS]) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int key = k0 + (j >> 2) * 8 + s * 4 + (j & 3);
#pragma unroll
        for (int ds = 0; ds < DS; ++ds)        // fragment order (common.h kc_off): one contiguous KiB per load
          ak[s][ds] = (key < k_hi) ? *reinterpret_cast<const u32x4*>(kbase + ((long)((k0 >> 5) * 2 * DS + s * DS + ds) * 64 + lane) * 8) : zero4;
      }
    };
    auto tile = [&](int k0, const u32x4 (&ak)[2][DS]) {
      // A = V^T tiles: lane (i = j -> d row, gq) holds keys k0 + gq*8 .. +7 (16 B, 8-key aligned); issued first
      u32x4 av[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        av[dt] = *reinterpret_cast<const u32x4*>(vbase + ((long)((k0 >> 5) * DT + dt) * 64 + lane) * 8);
      f32x4 st[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) acc = P::mfma(ak[s][ds], bq[ds], acc);
        st[s] = acc;
      }
      // lane (j, gq) now holds scores of query row `row` vs keys k0 + gq*8 + s*4 + r
      float pv[8];
      float tmax = NEG_BIG;
      const bool need_mask = (k0 + 32 > prefix);
      // this lane's 8 keys are contiguous: bit b = key - prefix of mask row t; they span at most two 64-bit words,
      // fetched once per tile (the per-key loads serialised into ~1 us round trips each on wide trees)
      const int bfirst = k0 + gq * 8 - prefix;
      unsigned vbits = 0xffu;                                   // bit e: key e of this lane is visible
      if (need_mask) {
        if (mask_bits) {
          const int lo = max(bfirst, 0), wi = lo >> 6;
          const unsigned long long* mrow = mask_bits + (long)t * mask_words;
          const unsigned long long w0 = mrow[min(wi, mask_words - 1)], w1 = mrow[min(wi + 1, mask_words - 1)];
          vbits = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int b = bfirst + e;
            unsigned v = 1u;                                    // keys before the prefix boundary are visible
            if (b >= 0) v = (unsigned)((((b >> 6) == wi ? w0 : w1) >> (b & 63)) & 1ull);
            vbits |= v << e;
          }
        } else {
          vbits = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) vbits |= (unsigned)(bfirst + e <= t) << e;      // causal: b <= t
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = k0 + gq * 8 + e;
        float sc = st[e >> 2][e & 3] * scale;
        const bool vis = row_ok && key < k_hi && ((vbits >> e) & 1u);
        sc = vis ? sc : -INFINITY;
        pv[e] = sc;
        tmax = fmaxf(tmax, sc);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m, tmax);        // >= NEG_BIG, finite
      const float alpha = __expf(m - m_new);
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { pv[e] = __expf(pv[e] - m_new); psum += pv[e]; }
      // probabilities are cast to the model dtype before P V (cache.py:187)
      u32x4 pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) pb[e] = pack2<P>(pv[2 * e], pv[2 * e + 1]);
      l = l * alpha + psum;
      m = m_new;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        o[dt] *= alpha;
        o[dt] = P::mfma(av[dt], pb, o[dt]);
      }
    };

    u32x4 ka[2][DS], kb2[2][DS];
    load_k(k_lo, ka);
    for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
      const bool two = k0 + 32 < k_hi;
      if (two) load_k(k0 + 32, kb2);
      tile(k0, ka);
      if (two) {
        if (k0 + 64 < k_hi) load_k(k0 + 64, ka);
        tile(k0 + 32, kb2);
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (row_ok) {
      // partial layout follows the output: row index = t*Hq + hq
      const long prow = ((long)sp * T + t) * Hq + hq;
      float* op = po + prow * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(op + dt * 16 + gq * 4) = o[dt];
      if (gq == 0) { pml[prow * 2] = m; pml[prow * 2 + 1] = l; }
    }
  }
}

// ---- single-launch variant: no combine kernel.
// Block = 8 waves on ONE 16-row query tile of one kv head and one span of KBK keys; wave w takes the 32-key tiles
// w, w+8, w+16 ... of the span and the eight (m, l, O^T) partials are merged through LDS.  While the context fits one
// span (kv_end <= KBK) that is the whole job: no partial buffers, no cross-block traffic -- one dependent launch
// (>= 4.7 us) less per layer in both models.  Longer contexts activate further spans (grid z; spans past kv_end exit
// at once): each span block publishes its merged partial with write-through stores and the last one to arrive on the
// (head, tile) counter combines them (same protocol as the split-K epilogues).  grid = (Hkv, query tiles, spans):
// kv head h stays on XCD h % 8.
template <typename P, int D, int NW, int NQ = 1>
__global__ __launch_bounds__(64 * NW, NQ > 1 ? 1 : 2) void tree_attn1_kernel(const u16* __restrict__ q, const u16* __restrict__ kc,
                                                         const u16* __restrict__ vt, const int* __restrict__ prefix_p,
                                                         int T, int Hq, int Hkv, int Lmax, int mask_words, int n_mask_keys,
                                                         const unsigned long long* __restrict__ mask_bits,
                                                         float scale, u16* __restrict__ out, int KBK,
                                                         float* __restrict__ po, float* __restrict__ pml,
                                                         unsigned* __restrict__ counters, int out_fm_tt, int single_max) {
  // argument order: the first 14 dwords (what the q / K loads need) are preloaded into SGPRs at wave launch
  // NQ (round 4): query tiles per block.  Every K / V^T tile a wave loads serves NQ 16-row query tiles instead of one --
  // the wide trees and prompt chunks (hundreds of query tiles per kv head) are bound by that L2 -> register traffic
  // (T = 769: 385 query tiles x 8 kv heads each walking ~500 keys x 512 B = 0.8 GB per layer), not by the matrix pipe.
  constexpr int DS = D / 32, DT = D / 16;
  __shared__ f32x4 so[NW][DT][64];
  __shared__ float sm[NW][16], sl[NW][16];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, gq = lane >> 4;
  // last query tiles first: they see the most keys (a node never sees a later one), so the long blocks start early
  const int h = blockIdx.x, qt = gridDim.y - 1 - blockIdx.y;
  const int g = Hq / Hkv;
  const int nrows = T * g;
  // First K / V^T tile of this wave: its addresses depend on the block and wave index only, so the loads go out before
  // the device-resident prefix length is known (one dependent round trip less on a launch that lasts ~3 of them).
  // Rows past kv_end hold zeros or stale finite values (the cache is zero-initialised) and are masked below.
  const int k_lo = blockIdx.z * KBK;
  const int kstart = k_lo + wv * 32;
  const bool spec = kstart + 32 <= Lmax;
  u32x4 ka[2][DS], kb2[2][DS], va[DT], vb[DT];
  if (spec) {
    // fragment order (common.h kc_off / vt_off): each of the 2 DS + DT loads of a tile is one contiguous KiB
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ds = 0; ds < DS; ++ds)
        ka[s][ds] = *reinterpret_cast<const u32x4*>(kc + (long)blockIdx.x * Lmax * D + ((long)((kstart >> 5) * 2 * DS + s * DS + ds) * 64 + lane) * 8);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      va[dt] = *reinterpret_cast<const u32x4*>(vt + (long)blockIdx.x * D * VT_LD(Lmax) + ((long)((kstart >> 5) * DT + dt) * 64 + lane) * 8);
  }
  const int prefix = *prefix_p;
  const int kv_end = prefix + n_mask_keys;
  // A context of <= single_max keys is ONE span whatever KBK is (span 0 walks all of it, the other spans' blocks leave):
  // the cross-block merge costs a narrow launch ~2 us, which 512-key spans only win back once there are several of them
  const bool one_span = kv_end <= single_max;
  if (one_span ? blockIdx.z > 0 : k_lo >= kv_end) return;      // whole block: span not in use (yet)
  int k_hi = one_span ? kv_end : min(kv_end, k_lo + KBK);
  const u16* kbase = kc + (long)h * Lmax * D;
  const long LV = VT_LD(Lmax);
  const u16* vbase = vt + (long)h * D * LV;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  bool row_ok[NQ];
  int t[NQ], hq[NQ];
  u32x4 bq[NQ][DS];
  float m[NQ], l[NQ];
  f32x4 o[NQ][DT];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int row = (qt * NQ + qi) * 16 + j;
    row_ok[qi] = row < nrows;
    t[qi] = row_ok[qi] ? row / g : 0;
    hq[qi] = h * g + (row_ok[qi] ? row % g : 0);
#pragma unroll
    for (int ds = 0; ds < DS; ++ds)
      bq[qi][ds] = row_ok[qi] ? *reinterpret_cast<const u32x4*>(q + ((long)t[qi] * Hq + hq[qi]) * D + ds * 32 + gq * 8) : zero4;
    m[qi] = NEG_BIG; l[qi] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[qi][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Keys past the last one any row of this block can see are never touched: in a tree (and in causal prefill) a node
  // sees no later node, so the tile of tokens [t0, t1] stops at prefix + t1 + 1 -- half of the tree keys on average.
  if (n_mask_keys > 64) {                                        // small trees: nothing worth skipping, no scan
    int top = -1;                                                // highest visible tree key of this lane's rows
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      if (row_ok[qi] && mask_bits) {
        const unsigned long long* mrow = mask_bits + (long)t[qi] * mask_words;
        for (int w = mask_words - 1; w >= 0; --w) {
          const unsigned long long x = mrow[w];
          if (x) { top = max(top, w * 64 + 63 - __clzll((long long)x)); break; }
        }
      } else if (row_ok[qi]) {
        top = max(top, t[qi]);
      }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) top = max(top, __shfl_xor(top, off, 64));
    k_hi = min(k_hi, prefix + top + 1);
  }

  auto load_k = [&](int k0, u32x4 (&ak)[2][DS]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int key = k0 + (j >> 2) * 8 + s * 4 + (j & 3);
#pragma unroll
      for (int ds = 0; ds < DS; ++ds)
        ak[s][ds] = (key < k_hi) ? *reinterpret_cast<const u32x4*>(kbase + ((long)((k0 >> 5) * 2 * DS + s * DS + ds) * 64 + lane) * 8) : zero4;
    }
  };
  // A = V^T tiles: lane (i = j -> d row, gq) holds keys k0 + gq*8 .. +7 (16 B); loaded one tile ahead like K.
  // Keys in [k_hi, k0 + 32) read stale-but-finite cache contents (rows are Lmax + 32 long) and get P = 0.
  auto load_v = [&](int k0, u32x4 (&av)[DT]) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      av[dt] = *reinterpret_cast<const u32x4*>(vbase + ((long)((k0 >> 5) * DT + dt) * 64 + lane) * 8);
  };
  auto tile = [&](int k0, const u32x4 (&ak)[2][DS], const u32x4 (&av)[DT]) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      f32x4 st[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) acc = P::mfma(ak[s][ds], bq[qi][ds], acc);
        st[s] = acc;
      }
      const int bfirst = k0 + gq * 8 - prefix;
      unsigned vbits = 0xffu;                                     // bit e: key k0 + gq*8 + e visible to this row
      if (k0 + 32 > prefix && bfirst > -8) {
        const int neg = max(-bfirst, 0);                           // leading keys that still belong to the prefix
        if (mask_bits) {
          // the 8 mask bits starting at max(bfirst, 0), funnel-shifted out of two adjacent words
          const int lo = max(bfirst, 0), wi = lo >> 6, sh = lo & 63;
          const unsigned long long* mrow = mask_bits + (long)t[qi] * mask_words;
          const unsigned long long w0 = mrow[min(wi, mask_words - 1)], w1 = mrow[min(wi + 1, mask_words - 1)];
          unsigned long long x = w0 >> sh;
          if (sh) x |= w1 << (64 - sh);
          vbits = (unsigned)x & 0xffu;
        } else {
          const int n = min(max(t[qi] - max(bfirst, 0) + 1, 0), 8);      // causal: tree keys 0 .. t
          vbits = (1u << n) - 1u;
        }
        vbits = ((vbits << neg) | ((1u << neg) - 1u)) & 0xffu;
      }
      float pv[8];
      float tmax = NEG_BIG;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = k0 + gq * 8 + e;
        float sc = st[e >> 2][e & 3] * scale;
        const bool vis = row_ok[qi] && key < k_hi && ((vbits >> e) & 1u);
        sc = vis ? sc : -INFINITY;
        pv[e] = sc;
        tmax = fmaxf(tmax, sc);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m[qi], tmax);
      const float alpha = __expf(m[qi] - m_new);
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { pv[e] = __expf(pv[e] - m_new); psum += pv[e]; }
      u32x4 pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) pb[e] = pack2<P>(pv[2 * e], pv[2 * e + 1]);
      l[qi] = l[qi] * alpha + psum;
      m[qi] = m_new;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        o[qi][dt] *= alpha;
        o[qi][dt] = P::mfma(av[dt], pb, o[qi][dt]);
      }
    }
  };

  if (kstart < k_hi && !spec) { load_k(kstart, ka); load_v(kstart, va); }
  for (int k0 = kstart; k0 < k_hi; k0 += 2 * NW * 32) {
    const int k1 = k0 + NW * 32;
    const bool two = k1 < k_hi;
    if (two) { load_k(k1, kb2); load_v(k1, vb); }
    tile(k0, ka, va);
    if (two) {
      if (k1 + NW * 32 < k_hi) { load_k(k1 + NW * 32, ka); load_v(k1 + NW * 32, va); }
      tile(k1, kb2, vb);
    }
  }
  const int nsp = one_span ? 1 : (kv_end + KBK - 1) / KBK;     // span blocks that did not exit above
  constexpr int NDT = (DT + NW - 1) / NW;
  __shared__ int s_last;
  bool last_known = false;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    float lq = l[qi];
    lq += __shfl_xor(lq, 16, 64);
    lq += __shfl_xor(lq, 32, 64);
    // ---- merge the NW wave partials through LDS; wave w finishes the d-tiles w, w + NW, ...
    if (qi > 0) __syncthreads();                                // the previous query tile's merge has read so / sm / sl
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) so[wv][dt][lane] = o[qi][dt];
    if (gq == 0) { sm[wv][j] = m[qi]; sl[wv][j] = lq; }
    __syncthreads();
    float M = NEG_BIG, L = 0.f;
    f32x4 acc[NDT];
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, sm[w][j]);
#pragma unroll
    for (int c = 0; c < NDT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float wt = __expf(sm[w][j] - M);
      L += sl[w][j] * wt;
#pragma unroll
      for (int c = 0; c < NDT; ++c)
        if (wv + c * NW < DT) acc[c] += so[w][wv + c * NW][lane] * wt;
    }
    const int tq = t[qi], hqq = hq[qi];
    const bool rok = row_ok[qi];
    const long orow = (long)tq * Hq + hqq;
    auto store_out = [&](const f32x4& a, float inv, int dt) {
      uint2 o2;
      o2.x = pack2<P>(a[0] * inv, a[1] * inv); o2.y = pack2<P>(a[2] * inv, a[3] * inv);
      if (out_fm_tt) {
        // FM layout (MFMA B-fragment order of the o-projection, csrc/lowlat.hip): element (t, f = hq * D + d)
        const int f = hqq * D + dt * 16 + gq * 4;
        const long off = ((((long)(f >> 5) * out_fm_tt + (tq >> 4)) * 64 + ((f >> 3) & 3) * 16 + (tq & 15)) << 3) + (f & 7);
        *reinterpret_cast<uint2*>(out + off) = o2;
      } else {
        *reinterpret_cast<uint2*>(out + orow * D + dt * 16 + gq * 4) = o2;
      }
    };
    if (nsp == 1) {
      if (rok) {
        const float inv = L > 0.f ? 1.f / L : 0.f;
#pragma unroll
        for (int c = 0; c < NDT; ++c)
          if (wv + c * NW < DT) store_out(acc[c], inv, wv + c * NW);
      }
      continue;
    }
    // ---- more than one span: publish (M, L, acc) write-through, last arriver combines
    const long rows_all = (long)T * Hq;
    if (rok) {
      const long prow = (long)blockIdx.z * rows_all + orow;
      const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int c = 0; c < NDT; ++c)
        if (wv + c * NW < DT)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[c]), rs_o,
                                                 (int)((prow * D + (wv + c * NW) * 16 + gq * 4) * 4), 0, 16);
      if (wv == 0 && gq == 0) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const auto rs_m = __builtin_amdgcn_make_buffer_rsrc(pml, 0, 0x7fffffff, 0x00020000);
        u32x2 ml = {__float_as_uint(M), __float_as_uint(L)};
        __builtin_amdgcn_raw_buffer_store_b64(ml, rs_m, (int)(prow * 8), 0, 16);
      }
    }
    if (qi + 1 < NQ) continue;                                   // one ticket per block, after its last query tile is out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned* cnt = counters + (long)h * gridDim.y + qt;
      const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == (unsigned)(nsp - 1);
      if (last) {
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      s_last = last;
    }
    __syncthreads();
    last_known = true;
  }
  if (!last_known || !s_last) return;
  // the last span block of this (head, query-tile group) combines every span's partials, query tile by query tile
  const long rows_all = (long)T * Hq;
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    if (!row_ok[qi]) continue;
    const int tq = t[qi], hqq = hq[qi];
    const long orow = (long)tq * Hq + hqq;
    float Mg = NEG_BIG;
    for (int s2 = 0; s2 < nsp; ++s2) Mg = fmaxf(Mg, pml[(s2 * rows_all + orow) * 2]);
    float Lg = 0.f;
    for (int s2 = 0; s2 < nsp; ++s2) {
      const long pr = s2 * rows_all + orow;
      Lg += pml[pr * 2 + 1] * __expf(pml[pr * 2] - Mg);
    }
    const float inv = Lg > 0.f ? 1.f / Lg : 0.f;
#pragma unroll
    for (int c = 0; c < NDT; ++c) {
      const int dt = wv + c * NW;
      if (dt >= DT) continue;
      f32x4 ag = {0.f, 0.f, 0.f, 0.f};
      for (int s2 = 0; s2 < nsp; ++s2) {
        const long pr = s2 * rows_all + orow;
        ag += *reinterpret_cast<const f32x4*>(po + pr * D + dt * 16 + gq * 4) * __expf(pml[pr * 2] - Mg);
      }
      uint2 o2;
      o2.x = pack2<P>(ag[0] * inv, ag[1] * inv); o2.y = pack2<P>(ag[2] * inv, ag[3] * inv);
      if (out_fm_tt) {
        const int f = hqq * D + dt * 16 + gq * 4;
        const long off = ((((long)(f >> 5) * out_fm_tt + (tq >> 4)) * 64 + ((f >> 3) & 3) * 16 + (tq & 15)) << 3) + (f & 7);
        *reinterpret_cast<uint2*>(out + off) = o2;
      } else {
        *reinterpret_cast<uint2*>(out + orow * D + dt * 16 + gq * 4) = o2;
      }
    }
  }
}

// merge the per-split partials: one wave per (t, hq) row
template <typename P, int D>
__global__ __launch_bounds__(256) void attn_combine_kernel(u16* __restrict__ out, const float* __restrict__ po,
                                                           const float* __restrict__ pml,
                                                           const int* __restrict__ prefix_p, int n_mask_keys, int chunk,
                                                           int rows /* T*Hq */) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int kv_end = *prefix_p + n_mask_keys;
  const int nsp = (kv_end + chunk - 1) / chunk;
  float M = NEG_BIG;
  for (int s = 0; s < nsp; ++s) M = fmaxf(M, pml[((long)s * rows + r) * 2]);
  float L = 0.f;
  float acc[D / 64 > 0 ? D / 64 : 1] = {0.f};
  float acc2 = 0.f;    // D == 32 / 64 / 128 handled as D/64 full lanes-strides (+ tail for D < 64)
  for (int s = 0; s < nsp; ++s) {
    const long pr = (long)s * rows + r;
    const float w = __expf(pml[pr * 2] - M);
    L += pml[pr * 2 + 1] * w;
    if (D >= 64) {
#pragma unroll
      for (int c = 0; c < D / 64; ++c) acc[c] += po[pr * D + c * 64 + lane] * w;
    } else if (lane < D) {
      acc2 += po[pr * D + lane] * w;
    }
  }
  const float inv = L > 0.f ? 1.f / L : 0.f;
  if (D >= 64) {
#pragma unroll
    for (int c = 0; c < D / 64; ++c) out[(long)r * D + c * 64 + lane] = P::from_f(acc[c] * inv);
  } else if (lane < D) {
    out[(long)r * D + lane] = P::from_f(acc2 * inv);
  }
}

// partial buffers: po  [max_splits][T][Hq][D] fp32, pml [max_splits][T][Hq][2] fp32
// counters: >= Hkv * ceil(T * (Hq/Hkv) / 16) zeroed uint32 (self-resetting) -> single-launch kernel for any Lmax;
// NULL -> single launch only if Lmax <= 2048, else key splits + a combine kernel
extern "C" int umb_tree_attn2(void* out, const void* q, const void* k_cache, const void* vt_cache, void* po, void* pml,
                              const int* prefix_len, const void* mask_bits, int mask_words, int n_mask_keys, int T,
                              int Hq, int Hkv, int D, int Lmax, int chunk, int max_splits, float scale,
                              unsigned* counters, int out_fm_tt, int dtype, hipStream_t st) {
  if (T < 1 || Hq % Hkv || chunk % 32 || Lmax % 32 || (D != 32 && D != 64 && D != 128)) return UMB_EINVAL;
  if (out_fm_tt && (out_fm_tt * 16 < T || (Hq * D) % 32)) return UMB_EINVAL;
  const int nrows = T * (Hq / Hkv);
  const int nqt = (nrows + 15) / 16;
  // Single-launch kernel: with a counters buffer for any Lmax (spans of 2048 keys, cross-block merge only once the
  // context outgrows a span); without one only when a single span covers Lmax.  The choice depends on Lmax alone
  // (kv_end lives on the device), so a captured graph stays valid as the context grows.
  static const bool no_single = getenv("UMB_ATTN_SPLIT") != nullptr;
  // Keys per span.  Wide trees fill the chip with (kv head, query tile) pairs and take 2048-key spans; a NARROW launch (a
  // draft level or a static-tree verify: 8 ... 56 blocks) walks a long context on that many CUs -- 70B, T = 13, context 1.6 k:
  // 56 blocks x 786 KB of K / V^T each -- so its spans are 512 keys (round 4): up to Lmax / 512 times the blocks once the
  // context is long, nothing changes below 512 keys (one live span, no merge; the other spans' blocks exit at once).
  // Shape-only (T, Lmax), so a captured graph stays valid as the context grows.  UMB_ATTN_KBK: experiments.
  static const int kbk_env = getenv("UMB_ATTN_KBK") ? atoi(getenv("UMB_ATTN_KBK")) : 0;
  static const int one_env = getenv("UMB_ATTN_ONE") ? atoi(getenv("UMB_ATTN_ONE")) : 0;
  int KBK = 2048, single_max = 2048;
  if (counters && Hkv * nqt < 256) {
    KBK = kbk_env >= 256 && kbk_env % 256 == 0 ? kbk_env : 512;      // multiples of 32 by construction: tile addresses are (k0 >> 5)
    static_assert(512 % 32 == 0 && 2048 % 32 == 0, "key spans must be whole 32-key tiles");
    while ((Lmax + KBK - 1) / KBK > max_splits && KBK < 2048) KBK *= 2;
    // one span up to 1024 keys.  Row-major cache (first A/B): 768 -- at 512-640 keys two spans +2 us per launch, at 1.6 k keys four spans
    // -5.8 us.  On the fragment-ordered cache a round of the walk is cheaper and the merge is not: T = 13 at 760 / 1000 keys 9.8 / 10.1 us
    // (768) vs 8.2 / 9.6 (1024), equal from 1200 keys; the headline engine at 700 / 900-token prompts 12.67 / 12.79 -> 12.56 / 12.74 ms
    single_max = one_env > 0 ? one_env : 1024;
    if (single_max < KBK) single_max = KBK;
  }
  const int spans = (Lmax + KBK - 1) / KBK;
  if ((counters || spans == 1) && nqt <= 65535 && spans <= max_splits && !no_single) {
    // Waves per (kv head, query tile): eight while the tiles are few (narrow trees: the keys of a tile are spread over
    // the waves and merged in LDS); with many tiles the chip is filled by tiles instead and a wave walks more keys,
    // so the per-block prologue and merge are paid 4x / 8x less often.
    static const int nw_env = getenv("UMB_ATTN_NW") ? atoi(getenv("UMB_ATTN_NW")) : 0;
    const int nblk = Hkv * nqt;
    const int nw = nw_env ? nw_env : nblk >= 1792 ? 1 : nblk >= 512 ? 2 : 8;   // fragment-ordered cache: T = 505 (2024 pairs) 31.8 (1) vs 35.1 (2) us, T = 385 (1544) 25.9 vs 24.4
    static const int nq_env = getenv("UMB_ATTN_NQ") ? atoi(getenv("UMB_ATTN_NQ")) : 0;
    // On the row-major cache two query tiles per loaded K / V^T tile won 11-40 % from 1536 pairs up; on the fragment-ordered cache a
    // tile costs 2 DS + DT coalesced KiB loads and the second query tile's registers (one wave per SIMD) cost more than they save:
    // T = 769 52.3 (one) vs 64.7 us (two), T = 385 24.4 vs 31.8, causal 1024-token chunk at 1024 keys 137 vs 156
    // (profiles/r04_attn_geometry_sweep.txt).  One tile per block; UMB_ATTN_NQ=2 keeps the other instantiation for experiments.
    const int nq = (nw <= 2 && nq_env >= 2) ? 2 : 1;
    const dim3 grid1(Hkv, (nqt + nq - 1) / nq, spans), block1(64 * nw);
#define ATT1N_(DD, NWV, NQV)                                                                                      \
  hipLaunchKernelGGL((tree_attn1_kernel<P, DD, NWV, NQV>), grid1, block1, 0, st, (const u16*)q, (const u16*)k_cache, \
                     (const u16*)vt_cache, prefix_len, T, Hq, Hkv, Lmax, mask_words, n_mask_keys,                   \
                     (const unsigned long long*)mask_bits, scale, (u16*)out, KBK, (float*)po, (float*)pml, counters, out_fm_tt, \
                     single_max)
#define ATT1_(DD)                                                                                                 \
  if (nw == 1 && nq == 2) { ATT1N_(DD, 1, 2); } else if (nw == 1) { ATT1N_(DD, 1, 1); }                            \
  else if (nw == 2 && nq == 2) { ATT1N_(DD, 2, 2); } else if (nw == 2) { ATT1N_(DD, 2, 1); }                       \
  else if (nw == 4) { ATT1N_(DD, 4, 1); } else { ATT1N_(DD, 8, 1); }
    DISPATCH_DTYPE(dtype, {
      if (D == 128) { ATT1_(128) }
      else if (D == 64) { ATT1_(64) }
      else { ATT1_(32) }
    })
#undef ATT1_
#undef ATT1N_
    UMB_LAUNCH_CHECK();
    return UMB_OK;
  }
  if (out_fm_tt) return UMB_EINVAL;             // FM output: single-launch kernel only
  int qpw = 1;
  while ((nqt + 4 * qpw - 1) / (4 * qpw) > 64 && qpw < 8) qpw *= 2;
  const int gz = (nqt + 4 * qpw - 1) / (4 * qpw);
  const dim3 grid(Hkv, gz, max_splits), block(256);
  const int rows = T * Hq;
#define ATT_(DD)                                                                                                  \
  hipLaunchKernelGGL((tree_attn_kernel<P, DD>), grid, block, 0, st, (const u16*)q, (const u16*)k_cache,            \
                     (const u16*)vt_cache, (float*)po, (float*)pml, prefix_len,                                    \
                     (const unsigned long long*)mask_bits, mask_words, n_mask_keys, T, Hq, Hkv, Lmax, chunk, qpw,  \
                     scale);                                                                                       \
  hipLaunchKernelGGL((attn_combine_kernel<P, DD>), dim3((rows + 3) / 4), dim3(256), 0, st, (u16*)out,              \
                     (const float*)po, (const float*)pml, prefix_len, n_mask_keys, chunk, rows)
  DISPATCH_DTYPE(dtype, {
    if (D == 128) { ATT_(128); }
    else if (D == 64) { ATT_(64); }
    else { ATT_(32); }
  })
#undef ATT_
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

extern "C" int umb_tree_attn(void* out, const void* q, const void* k_cache, const void* vt_cache, void* po, void* pml,
                             const int* prefix_len, const void* mask_bits, int mask_words, int n_mask_keys, int T,
                             int Hq, int Hkv, int D, int Lmax, int chunk, int max_splits, float scale,
                             unsigned* counters, int dtype, hipStream_t st) {
  return umb_tree_attn2(out, q, k_cache, vt_cache, po, pml, prefix_len, mask_bits, mask_words, n_mask_keys, T, Hq, Hkv, D,
                        Lmax, chunk, max_splits, scale, counters, 0, dtype, st);
}
