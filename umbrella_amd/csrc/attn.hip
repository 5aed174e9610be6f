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
//   q   [T][Hq][D]            K cache [Hkv][Lmax][D]        V cache TRANSPOSED [Hkv][D][Lmax + UMB_VT_PAD]
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
                                                        int Lmax, int chunk, int qtiles_per_wave, float scale,
                                                        unsigned* __restrict__ counters, u16* __restrict__ out) {
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
        const u16* kp = kbase + (long)key * D + gq * 8;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds)
          ak[s][ds] = (key < k_hi) ? *reinterpret_cast<const u32x4*>(kp + ds * 32) : zero4;
      }
    };
    auto tile = [&](int k0, const u32x4 (&ak)[2][DS]) {
      // A = V^T tiles: lane (i = j -> d row, gq) holds keys k0 + gq*8 .. +7 (16 B, 8-key aligned); issued first
      u32x4 av[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        av[dt] = *reinterpret_cast<const u32x4*>(vbase + (long)(dt * 16 + j) * LV + k0 + gq * 8);
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
      // partial layout follows the output: row index = t*Hq + hq.  Write-through (sc1) stores when the
      // merge is fused: the last-arriving block must see them without a release fence / L2 write-back.
      const long prow = ((long)sp * T + t) * Hq + hq;
      if (counters) {
        const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0x7fffffff, 0x00020000);
        const auto rs_m = __builtin_amdgcn_make_buffer_rsrc(pml, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[dt]), rs_o,
                                                 (int)((prow * D + dt * 16 + gq * 4) * 4), 0, 16);
        if (gq == 0) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          u32x2 ml = {__float_as_uint(m), __float_as_uint(l)};
          __builtin_amdgcn_raw_buffer_store_b64(ml, rs_m, (int)(prow * 8), 0, 16);
        }
      } else {
        float* op = po + prow * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(op + dt * 16 + gq * 4) = o[dt];
        if (gq == 0) { pml[prow * 2] = m; pml[prow * 2 + 1] = l; }
      }
    }
  }
  if (!counters) return;

  // ---- fused combine: the last key-split block to arrive for this (head, query group) merges the partials
  __shared__ int s_last;
  const int nsp = (kv_end + chunk - 1) / chunk;                // blocks past kv_end returned above and never arrive
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* cnt = counters + (long)h * gridDim.y + zq;
    const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == (unsigned)(nsp - 1);
    if (last) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  const long rows_all = (long)T * Hq;
  for (int qi = 0; qi < qtiles_per_wave; ++qi) {
    const int qt = (zq * 4 + wv) * qtiles_per_wave + qi;
    if (qt * 16 >= nrows) break;
    const int row = qt * 16 + j;
    if (row >= nrows) continue;
    const int t = row / g, hq = h * g + row % g;
    const long r = (long)t * Hq + hq;
    float M = NEG_BIG;
    for (int s2 = 0; s2 < nsp; ++s2) M = fmaxf(M, pml[(s2 * rows_all + r) * 2]);
    float L = 0.f;
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < nsp; ++s2) {
      const long pr = s2 * rows_all + r;
      const float w = __expf(pml[pr * 2] - M);
      L += pml[pr * 2 + 1] * w;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc[dt] += *reinterpret_cast<const f32x4*>(po + pr * D + dt * 16 + gq * 4) * w;
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      uint2 o2;
      o2.x = pack2<P>(acc[dt][0] * inv, acc[dt][1] * inv); o2.y = pack2<P>(acc[dt][2] * inv, acc[dt][3] * inv);
      *reinterpret_cast<uint2*>(out + r * D + dt * 16 + gq * 4) = o2;
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
// counters: NULL -> separate combine kernel; else >= Hkv * 64 zeroed uint32 (self-resetting): combine fused into the
// attention kernel (last-arriving key-split block per (kv head, query group))
extern "C" int umb_tree_attn(void* out, const void* q, const void* k_cache, const void* vt_cache, void* po, void* pml,
                             const int* prefix_len, const void* mask_bits, int mask_words, int n_mask_keys, int T,
                             int Hq, int Hkv, int D, int Lmax, int chunk, int max_splits, float scale,
                             unsigned* counters, int dtype, hipStream_t st) {
  if (T < 1 || Hq % Hkv || chunk % 32 || Lmax % 8 || (D != 32 && D != 64 && D != 128)) return UMB_EINVAL;
  const int nrows = T * (Hq / Hkv);
  const int nqt = (nrows + 15) / 16;
  int qpw = 1;
  while ((nqt + 4 * qpw - 1) / (4 * qpw) > 64 && qpw < 8) qpw *= 2;
  const int gz = (nqt + 4 * qpw - 1) / (4 * qpw);
  const dim3 grid(Hkv, gz, max_splits), block(256);
  const int rows = T * Hq;
#define ATT_(DD)                                                                                                  \
  hipLaunchKernelGGL((tree_attn_kernel<P, DD>), grid, block, 0, st, (const u16*)q, (const u16*)k_cache,            \
                     (const u16*)vt_cache, (float*)po, (float*)pml, prefix_len,                                    \
                     (const unsigned long long*)mask_bits, mask_words, n_mask_keys, T, Hq, Hkv, Lmax, chunk, qpw,  \
                     scale, counters, (u16*)out);                                                                  \
  if (!counters)                                                                                                   \
    hipLaunchKernelGGL((attn_combine_kernel<P, DD>), dim3((rows + 3) / 4), dim3(256), 0, st, (u16*)out,            \
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
