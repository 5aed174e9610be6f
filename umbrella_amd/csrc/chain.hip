// Persistent weight-stream engine for the draft model's <= 4-row forwards (round 5): ONE launch runs the chain
//     o-projection + residual  ->  gate/up + SiLU  ->  down-projection + residual  ->  NEXT layer's q/k/v + RoPE + KV append
// (reference: umbrella/models/llama.py:461-533 LlamaCudagraph.layer_compute / graph_inference -- the reference replays
// the draft forward as one CUDA graph of ~25 launches per layer; the GEMV family of gemv.hip runs it as 5; this file as 2:
// tree attention + this chain).  Why: at <= 4 rows every launch of the layer is a latency chain --
// the weights of op k+1 do not depend on the output of op k, yet each launch starts its HBM stream from zero after a
// drain.  Here the stream survives the dependency edges:
//   * one workgroup per CU (grid = CU count, 256 threads): wave 0 is a LOADER that copies this CU's share of every op's
//     weight rows HBM -> LDS with global_load_lds_dwordx4 ... nt (LDS-DMA, no registers), 16 KiB slots in a ring that runs
//     AHEAD of the consumers across op boundaries; waves 1-3 are CONSUMERS (v_dot2 from LDS, fp32 accumulation).
//   * the [T][N] activation edges between ops travel as 8-byte {tag, value} granules (write-through agent-scope stores, the
//     data is its own flag: MI355X_MICROARCH.md "handoff-1to1 / allgather", cdna_hip_programming.md Guideline 16 R2);
//     every CU's consumers sweep the whole edge into LDS and rebuild their operands from it.
//   * weights are the GEMV family's plain row-major copies (umb_repack_rows): CU c owns rows [c R, (c + 1) R) of each
//     linear, contiguous in memory, so a slot is a straight 16 KiB copy and needs no second weight layout.
// Arithmetic: every output is computed with the SAME operation order as gv_kernel (per-lane k order, 64-lane xor butterfly,
// 2048-element k-slices summed in slice order, the same epilogues and the same sums-of-squares grouping), so the chain
// is bit-identical to the five-launch GEMV schedule (tests/test_chain.py) and a token's result does not depend on T.
// Liveness: the protocol needs every workgroup resident (one per CU by LDS footprint; grid == CU count).  Every spin is
// bounded by the 100 MHz wall clock; a give-up sets status != 0, poisons nothing further and lets the launch finish.
#include "../../include/umbrella_hip.h"
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(3))) char lds_char;
// every LDS word this kernel polls or publishes is addressed through an address_space(3) pointer: a generic pointer makes
// hipcc emit FLAT loads / stores, which count in vmcnt as well -- the loader's flag polls then waited for its whole DMA
// stream (s_waitcnt vmcnt(0) in front of every slot: 0.9 instead of 0.63 us per slot)
typedef __attribute__((address_space(3))) volatile unsigned lds_vu;
typedef __attribute__((address_space(3))) unsigned lds_u;
typedef __attribute__((address_space(3))) volatile float lds_vf;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

typedef _Float16 ch_h2 __attribute__((ext_vector_type(2)));
typedef __bf16 ch_b2 __attribute__((ext_vector_type(2)));
template <typename P> __device__ __forceinline__ float ch_dot2(unsigned a, unsigned b, float c) {
  if constexpr (std::is_same<P, BF16>::value)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ch_b2, a), __builtin_bit_cast(ch_b2, b), c, false);
  else
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(ch_h2, a), __builtin_bit_cast(ch_h2, b), c, false);
}

// ---- folded 64-lane sums.  The GEMV family finishes a dot product with the xor butterfly v += shfl_xor(v, m), m = 32 .. 1
// (ds_bpermute: 6 LDS-crossbar round trips per value).  Up to four values fold into ONE register on the way down with the
// SAME pairs added in the SAME order, so the bits are the butterfly's: m = 32 is a v_permlane32_swap of two values (lower
// half: value A's pairs (i, i + 32), upper half: B's), m = 16 a v_permlane16_swap of two such registers (16-lane rows:
// A, C, B, D), m = 8, 4 row rotations and m = 2, 1 quad permutes inside a row (DPP; a rotation by 4 pairs the same values
// as xor 4 once lanes i and i ^ 8 are equal).  Result: row t (lanes 16 t .. 16 t + 15) holds value t's sum.  10 VALU
// operations for four values instead of 24 bpermutes + 24 adds, and no LDS traffic next to the DMA ring.
__device__ __forceinline__ void ch_swap32(float& a, float& b) {       // lanes 32-63 of a <-> lanes 0-31 of b
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // volatile: a plain asm is not convergent and
                                                                                   // was sunk into the `if (fin)` block (partial EXEC)
}
__device__ __forceinline__ void ch_swap16(float& a, float& b) {       // rows 1, 3 of a <-> rows 0, 2 of b
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int CTRL> __device__ __forceinline__ float ch_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// v[t], t < TT  ->  row t of the result = the butterfly sum of v[t] (rows >= TT repeat another token's sum)
template <int TT> __device__ __forceinline__ float ch_fold(const float (&v)[TT]) {
  float a0 = v[0], b0 = v[TT > 2 ? 2 : 0];                           // rows 0 / 2
  ch_swap32(a0, b0);
  float s1 = a0 + b0, s2 = s1;
  if constexpr (TT > 1) {
    float c0 = v[1], d0 = v[TT > 3 ? 3 : 1];                         // rows 1 / 3
    ch_swap32(c0, d0);
    s2 = c0 + d0;
  }
  ch_swap16(s1, s2);
  float u = s1 + s2;
  u += ch_dpp<0x128>(u);                                             // row_ror:8
  u += ch_dpp<0x124>(u);                                             // row_ror:4
  u += ch_dpp<0x4E>(u);                                              // quad_perm [2,3,0,1]
  u += ch_dpp<0xB1>(u);                                              // quad_perm [1,0,3,2]
  asm volatile("" : "+v"(u));                                        // the sum exists HERE, in uniform control flow
  return u;
}

enum { CH_SLOT = 16384, CH_MAXRING = 8 };
// flag words (u32 index into the flag area)
enum { F_LANDED = 0, F_FREED = 16, F_GATH = 32, F_SWEEP = 33 /* consumers inside a sweep */, F_GO = 34, F_PAIR = 36, F_SQCNT = 40, F_DEAD = 41,
       F_SQ = 48 /* float [8][4] */, F_HV = 80 /* u32 [8][4] */, F_WORDS = 128 };

struct ChainArgs {
  const char* w_o; const char* w_gu; const char* w_down; const char* w_qkv;      // row-major packed-order rows
  const u16* attn;                    // front: [T][H] attention output
  u16* h; u16* hw; float* ssq;        // residual stream (plain): h in (front) / out, hw + ssq out (front) or in (tail only)
  const u16* norm2; const u16* next_norm;
  const int* pos; const int* slot; const u16* cosT; const u16* sinT; u16* q_out; u16* kc; u16* vt;
  gu64* g_h1; gu64* g_act; gu64* g_h2; gu32* epoch; gu32* status;
  int T, R, front, tail, H, I, NQKV, ssq_stride, ssq_groups_in, Hq, Hkv, D, Lmax;
  float eps;
  unsigned off_sth, off_sta;
  long long timeout_ticks;
  int hint;                           // sentinel poll before the full sweeps: 2 one wave per CU (default), 1 every consumer, 0 none
  int drop_cu;                        // test hook (UMB_CHAIN_TEST_DROP_CU): this workgroup withholds its o-projection granules
  unsigned long long* trace;          // UMB_CHAIN_TRACE builds only
};

// -DUMB_CHAIN_TRACE (scripts/r5/chain_trace.py builds a second library with it): every wave stamps the 100 MHz wall clock at
// its phase boundaries into trace[cu][wave][32] (device pointer in UMB_CHAIN_TRACE_PTR); compiled out of the product.
#ifdef UMB_CHAIN_TRACE
#define CH_STAMP() do { if (a.trace && lane == 0 && tr_n < 32) a.trace[((size_t)cu * 4 + wv) * 32 + tr_n] = wall_clock64(); ++tr_n; } while (0)
#else
#define CH_STAMP() do { } while (0)
#endif
__device__ __forceinline__ void ch_sleep() { __builtin_amdgcn_s_sleep(2); }

// bounded wait on an LDS word (>= target); returns false after the deadline (status set by the caller)
__device__ __forceinline__ bool lds_wait_ge(lds_vu* w, unsigned target, long long deadline, lds_vu* dead) {
  unsigned spins = 0;                                // every value that steers the loop is made wave-uniform (SGPR control flow)
  while (__builtin_amdgcn_readfirstlane(*w) < target) {
    ch_sleep();
    if ((++spins & 1023u) == 0) {
      if (__builtin_amdgcn_readfirstlane(*dead)) return false;
      if (__builtin_amdgcn_readfirstlane((int)((long long)wall_clock64() > deadline))) { *dead = 1; return false; }
    }
  }
  return true;
}

// Argument order: what the LOADER needs -- the four weight bases, the ring depth, the flag offset, the op flags and slot
// counts -- comes first and is preloaded into SGPRs with the wave (-amdgpu-kernarg-preload-count; a by-value struct never
// is), so the first LDS-DMA piece does not wait for a kernarg fetch; the consumers' operands follow in the struct.
template <typename P, int TT>
__global__ __launch_bounds__(256) void draft_chain_kernel(const char* w_o, const char* w_gu, const char* w_down, const char* w_qkv,
                                                          int R, unsigned off_flags, int front_tail, unsigned slots,
                                                          unsigned timeout_ticks, ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cu = blockIdx.x;
  lds_vu* fl = (lds_vu*)((lds_char*)smem + off_flags);
  if (threadIdx.x < F_WORDS) fl[threadIdx.x] = 0;
  __syncthreads();                                   // the only workgroup barrier of the launch
  const int T = a.T;
  int tr_n = 0; (void)tr_n;
  CH_STAMP();                                        // 0: wave start
  const long long deadline = (long long)wall_clock64() + (long long)timeout_ticks;
  lds_vu* dead = fl + F_DEAD;
  // slots per CU of each op (rows per CU x bytes per row / 16 KiB); H = 2048, I = 8192 (umb_chain_ok)
  const int ncu = gridDim.x;
  const int s_o = a.front ? (a.H / ncu) * a.H * 2 / CH_SLOT : 0;
  const int s_gu = a.front ? (2 * a.I / ncu) * a.H * 2 / CH_SLOT : 0;
  const int s_dn = a.front ? (a.H / ncu) * a.I * 2 / CH_SLOT : 0;
  const int s_q = a.tail ? (a.NQKV / ncu) * a.H * 2 / CH_SLOT : 0;

  if (wv == 0) {
    // ------------------------------------------------------------------ LOADER: this CU's weight stream, all ops in order
    // A slot = 16 pieces of 1 KiB (one global_load_lds_dwordx4 per piece); the instruction's immediate offset moves the
    // global AND the LDS address, so four pieces share one M0 / base setting.  Up to four slots in flight (vmcnt is
    // six bits: the 64th outstanding piece stalls the issue, which is the throttle); slot i - 3 is published once all but
    // the newest 48 pieces have landed.
    const unsigned ring0 = (unsigned)(size_t)(lds_char*)smem;
    const unsigned voff = (unsigned)lane * 16u;
    const bool lfront = (front_tail & 1) != 0, ltail = (front_tail & 2) != 0, thin = (front_tail & 4) != 0;
    int i = 0, p = 0;                                  // global slot index, ring position
#ifdef UMB_CHAIN_TRACE
    unsigned long long full_ticks = 0, full_n = 0;     // ring-full episodes of this loader
#endif
    auto flag_landed = [&](int slot_i) {               // slot_i has landed: its ring position carries slot_i + 1
      if (lane == 0) fl[F_LANDED + slot_i % R] = (unsigned)slot_i + 1u;
    };
    auto run = [&](const char* w, int nslots) {
      const char* src = w + (size_t)cu * nslots * CH_SLOT;
      int sk = cu % (nslots ? nslots : 1);             // step s reads this CU's slot (s + cu) % nslots (see the consumers)
      for (int s = 0; s < nslots; ++s) {
        if (i >= R) {
          // ring full: nothing to overlap with, so drain what is in flight and publish it, then wait for the slot
          if (__builtin_amdgcn_readfirstlane(fl[F_FREED + p]) < (unsigned)(i - R + 1)) {
#ifdef UMB_CHAIN_TRACE
            const unsigned long long tf0 = wall_clock64();
            ++full_n;
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (i >= 3) flag_landed(i - 3);
            if (i >= 2) flag_landed(i - 2);
            if (i >= 1) flag_landed(i - 1);
            if (!lds_wait_ge(fl + F_FREED + p, (unsigned)(i - R + 1), deadline, dead)) return;
#ifdef UMB_CHAIN_TRACE
            full_ticks += wall_clock64() - tf0;
#endif
          }
        }
        const unsigned dst = ring0 + (unsigned)p * CH_SLOT;
        const char* sp = src + (size_t)sk * CH_SLOT;
        if (++sk == nslots) sk = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %1, %2 nt\n\t"
                       "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                       "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\t"
                       "global_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
                       "s_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(sp + j * 4096), "s"(dst + j * 4096) : "memory");
        }
        CH_STAMP();                                    // loader: slot i issued
        if (thin && __builtin_amdgcn_readfirstlane(fl[F_SWEEP]) != 0) {
          // consumers of this CU are sweeping an edge: their granule loads queue behind this wave's pieces in the CU's
          // memory path -- one fill outstanding instead of four while that lasts (MI355X_MICROARCH.md "gather-pass")
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (i >= 3) flag_landed(i - 3);
          if (i >= 2) flag_landed(i - 2);
          if (i >= 1) flag_landed(i - 1);
          flag_landed(i);
        } else if (i >= 3) {
          asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
          flag_landed(i - 3);
        }
        ++i;
        if (++p == R) p = 0;
      }
    };
    // slots per CU of each op, packed by the host: o | gate/up << 8 | down << 16 | q/k/v << 24
    if (lfront) { run(w_o, (int)(slots & 255u)); run(w_gu, (int)((slots >> 8) & 255u)); run(w_down, (int)((slots >> 16) & 255u)); }
    if (ltail) run(w_qkv, (int)(slots >> 24));
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    if (i >= 3) flag_landed(i - 3);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (i >= 2) flag_landed(i - 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (i >= 1) flag_landed(i - 1);
    CH_STAMP();                                        // loader: everything landed
#ifdef UMB_CHAIN_TRACE
    if (a.trace && lane == 0) a.trace[((size_t)cu * 4) * 32 + 31] = full_ticks | (full_n << 48);
#endif
    return;
  }

  // ---------------------------------------------------------------------- CONSUMERS
  const int cw = wv - 1;                               // 0..2
  const int tl = lane >> 4;                            // the folded sums leave token t in row t (lanes 16 t ..)
  const bool fin = (lane & 15) == 0 && tl < T;         // lane 16 t finishes token t
  const int OOB = (int)0x80000000;
  const unsigned tag_base = a.epoch ? __hip_atomic_load(a.epoch, RLX_AGENT) << 2 : 0u;
  bool ok = true;
  int gslot = 0;                                       // global slot index of the current op's slot 0
  unsigned gath_no = 0;

  // Slot order: step k of an op with n slots per CU is this CU's slot (k + cu) % n (loader and consumers alike), so at any
  // moment the 256 loaders read 256 DIFFERENT offsets of their contiguous row blocks instead of the same one (CU row blocks
  // are 32 - 256 KiB apart: in lock step they would all sit on the same few HBM channels).
  const int orow0 = (a.H / ncu) * cu;                  // this CU's first o / down row (8 rows per CU at 256 CUs)
  const int qrows = a.NQKV / ncu;
  const int sa_o = cw < s_o ? (cw + cu) % (s_o ? s_o : 1) : 0;                 // this consumer's o slot (step cw)
  const int sa_q = s_q ? (cw + cu) % s_q : 0;                                  // and q/k/v slot

  // ---- prologue A: what the first op needs (plain loads; everything here was written before this launch), issued at once
  const auto rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.front ? a.attn : a.hw), 0, (unsigned)(T * a.H * 2), 0x00020000);
  u32x4 x[TT][4];                                      // K = 2048 operand: lane l holds k = 512 kc + 8 l .. + 7
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      x[t][kc] = __builtin_amdgcn_raw_buffer_load_b128(rx0, (t * a.H + kc * 512 + lane * 8) * 2, 0, 0);
  // o-projection residual: lane 16 t reads its token's four values of this consumer's o slot
  const auto rh = __builtin_amdgcn_make_buffer_rsrc(a.h, 0, a.front ? (unsigned)(T * a.H * 2) : 0u, 0x00020000);
  const u32x2 h0v = __builtin_amdgcn_raw_buffer_load_b64(rh, (fin && cw < s_o) ? (tl * a.H + orow0 + 4 * sa_o) * 2 : OOB, 0, 0);
  const auto rn2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.norm2), 0, a.norm2 ? (unsigned)a.H * 2u : 0u, 0x00020000);
  const auto rnn = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.next_norm), 0, a.next_norm ? (unsigned)a.H * 2u : 0u, 0x00020000);
  u32x4 wn2[4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) wn2[kc] = __builtin_amdgcn_raw_buffer_load_b128(rn2, (kc * 512 + lane * 8) * 2, 0, 0);
  const auto rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.pos), 0, a.tail ? (unsigned)T * 4u : 0u, 0x00020000);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(a.slot), 0, a.tail ? (unsigned)T * 4u : 0u, 0x00020000);
  const auto rcos = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.cosT), 0, a.tail ? 0x7fffffffu : 0u, 0x00020000);
  const auto rsin = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(a.sinT), 0, a.tail ? 0x7fffffffu : 0u, 0x00020000);
  int pos_t = 0, slot_t = 0;
  unsigned short rc[2][4];                             // RoPE operands of this consumer's q/k/v slot: per pair cos lo, hi, sin lo, hi
  auto load_pos = [&]() {
    pos_t = (int)__builtin_amdgcn_raw_buffer_load_b32(rp, fin ? tl * 4 : OOB, 0, 0);
    slot_t = (int)__builtin_amdgcn_raw_buffer_load_b32(rs, fin ? tl * 4 : OOB, 0, 0);
  };
  auto load_rope = [&]() {                             // waits for pos_t: issued where that round trip has long finished
    const int D = a.D, half = D / 2;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int n = qrows * cu + 4 * sa_q + 2 * pr, m = (n % D) >> 1;
      const int cb = fin ? (pos_t * D + m) * 2 : OOB;
      rc[pr][0] = __builtin_amdgcn_raw_buffer_load_b16(rcos, cb, 0, 0);
      rc[pr][1] = __builtin_amdgcn_raw_buffer_load_b16(rcos, fin ? cb + half * 2 : OOB, 0, 0);
      rc[pr][2] = __builtin_amdgcn_raw_buffer_load_b16(rsin, cb, 0, 0);
      rc[pr][3] = __builtin_amdgcn_raw_buffer_load_b16(rsin, fin ? cb + half * 2 : OOB, 0, 0);
    }
  };
  // tail-only launch (the forward's first q/k/v): 1/rms from the producer's plain sums of squares, as gv_kernel reads them
  float sqv[TT][4];
  if (!a.front) {
    const auto rsq = __builtin_amdgcn_make_buffer_rsrc(a.ssq, 0, (unsigned)((T - 1) * a.ssq_stride + a.ssq_groups_in) * 4u, 0x00020000);
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gq = lane + 64 * q;
        sqv[t][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            rsq, (t < T && gq < a.ssq_groups_in) ? (t * a.ssq_stride + gq) * 4 : OOB, 0, 0));
      }
    load_pos();
    load_rope();
  }

  // ---- helpers
  auto wait_slot = [&](int gi) -> const lds_char* {        // ring position of global slot gi, once it has landed
    const int p = gi % R;
    if (ok && !lds_wait_ge(fl + F_LANDED + p, (unsigned)gi + 1u, deadline, dead)) ok = false;
    asm volatile("" ::: "memory");
    return (const lds_char*)smem + (size_t)p * CH_SLOT;
  };
  auto free_slot = [&](int gi) {                       // the slot's bytes are in registers
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) fl[F_FREED + gi % R] = (unsigned)gi + 1u;
  };
  // four K = 2048 rows of one slot: load4 pulls them into registers and frees the ring slot -- also AHEAD of a dependency
  // edge (the operand of the op is not there yet, but registers are idle then and the loader gets its slot back: the ring
  // grows by what the consumers hold); compute4 runs them against the TT operand rows.  The 4 TT accumulators advance
  // TOGETHER, one k pair at a time (a v_dot2c waits for its own previous result: one chain alone issues every ~9 cycles);
  // each chain still adds its products in gv_kernel's order.  acc[r]: row t of the wave = token t's sum for weight row r
  auto load4 = [&](int gi, u32x4 (&wr)[4][4]) {
    const lds_char* sp = wait_slot(gi);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) wr[r][kc] = *(const __attribute__((address_space(3))) u32x4*)(sp + r * 4096 + kc * 1024 + lane * 16);
    free_slot(gi);
  };
  auto compute4 = [&](const u32x4 (&wr)[4][4], float (&acc)[4]) {
    float part[4][TT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < TT; ++t) part[r][t] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < TT; ++t) part[r][t] = ch_dot2<P>(wr[r][kc][e], x[t][kc][e], part[r][t]);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = ch_fold<TT>(part[r]);
  };
  // sweep an edge's granules into LDS staging (flat u32 index == flat granule index), then meet the other consumers.
  // NCH: this consumer's chunks (1024 granules each), ALL polled in one pass -- an edge completes with its slowest
  // producer, whose granules sit in every chunk alike: chunk after chunk would put NCH passes behind that arrival, this one
  // Before the full sweep, ONE 512-byte piece of the edge is polled (a different piece on every CU and consumer, so the
  // pollers spread over the edge): re-reading a whole incomplete edge moves as many bytes per CU as the weight stream
  // itself (three waves x 8 KiB per pass) and slowed the loaders by a third.  The piece is a hint -- its 64 granules come
  // from 4 - 16 producers, the others finish around the same time -- the tags of the full sweep decide.
  auto gather = [&](auto nch, gu64* g, int chunks, unsigned tag, lds_u* stage, auto&& idle) {
    constexpr int NCH = decltype(nch)::value;
    if (lane == 0) __hip_atomic_fetch_add((lds_u*)(fl + F_SWEEP), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (a.hint) {
      // hint 1: every consumer polls a piece of its own first chunk; hint 2: ONE wave of the CU polls (consumer 0), the other
      // two wait on an LDS word.  One loop, one idle() site (a second copy of the preload code cost 441 spilled registers)
      const bool poller = a.hint == 2 ? cw == 0 : cw < chunks;
      const bool waiter = a.hint == 2 && cw != 0;
      gu64* hp = g + (size_t)(a.hint == 2 ? 0 : cw) * 1024 + (size_t)((cu + 5 * cw) & 15) * 64 + lane;
      unsigned spins = 0;
      while (poller || waiter) {
        bool done;
        if (poller) {
          const unsigned long long gv = __hip_atomic_load(hp, RLX_AGENT);
          done = __all((unsigned)(gv >> 32) == tag);
        } else {
          done = __builtin_amdgcn_readfirstlane(fl[F_GO]) >= gath_no + 1;
        }
        if (done || !ok) break;
        idle();
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 127u) == 0) {
          if (__builtin_amdgcn_readfirstlane(*dead)) ok = false;
          else if (__builtin_amdgcn_readfirstlane((int)((long long)wall_clock64() > deadline))) { *dead = 1; ok = false; }
        }
      }
      if (a.hint == 2 && cw == 0 && lane == 0) fl[F_GO] = gath_no + 1;
    }
    unsigned pend = 0;                                 // bit j: chunk cw + 3 j still incomplete
#pragma unroll
    for (int j = 0; j < NCH; ++j) pend |= (cw + 3 * j < chunks) ? (1u << j) : 0u;
    unsigned spins = 0;
    while (pend) {
      unsigned val[NCH][16];
#pragma unroll
      for (int j = 0; j < NCH; ++j)
        if (pend & (1u << j)) {
          gu64* gp = g + (size_t)(cw + 3 * j) * 1024 + lane;
          bool good = true;
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const unsigned long long gv = __hip_atomic_load(gp + 64 * k, RLX_AGENT);
            val[j][k] = (unsigned)gv;
            good &= (unsigned)(gv >> 32) == tag;
          }
          if (__all(good) || !ok) {
#pragma unroll
            for (int k = 0; k < 16; ++k) stage[(cw + 3 * j) * 1024 + lane + 64 * k] = val[j][k];
            pend &= ~(1u << j);
          }
        }
      if (!pend) break;
      ch_sleep();
      if ((++spins & 255u) == 0) {
        if (__builtin_amdgcn_readfirstlane(*dead)) ok = false;
        else if (__builtin_amdgcn_readfirstlane((int)((long long)wall_clock64() > deadline))) { *dead = 1; ok = false; }
      }
    }
    ++gath_no;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((lds_u*)(fl + F_SWEEP), 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) __hip_atomic_fetch_add((lds_u*)(fl + F_GATH), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ok && !lds_wait_ge(fl + F_GATH, 3u * gath_no, deadline, dead)) ok = false;
    asm volatile("" ::: "memory");
  };
  // K = 2048 operand from a gathered residual stream h (staging [T][2048]): x = round(h * w), 1/rms from the same 16-byte
  // pieces -- lane l's piece kc is sums-of-squares group 64 kc + l of the GEMV family (8 consecutive columns = one
  // workgroup's rows there).  FMA4: the producer summed its rows as two 4-row FMA chains (down), else 8 rounded squares (o)
  float inv = 1.f;                                     // 1/rms of this lane's token (row t of the wave)
  auto build_x = [&](const lds_char* st, const u32x4 (&wn)[4], auto fma4) {
    constexpr bool FMA4 = decltype(fma4)::value;
    float ssum[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      float grp[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const u32x4 hv = *(const __attribute__((address_space(3))) u32x4*)(st + t * 4096 + kc * 1024 + lane * 16);
        float e[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { e[2 * q] = lo_f<P>(hv[q]); e[2 * q + 1] = hi_f<P>(hv[q]); }
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          o[q] = pack2<P>(e[2 * q] * lo_f<P>(wn[kc][q]), e[2 * q + 1] * hi_f<P>(wn[kc][q]));
        x[t][kc] = o;                                  // (a packed v_pk_mul_f16 here measured no faster and not bit-equal)
        float sg;
        if constexpr (FMA4) {
          float s0 = e[0] * e[0], s1 = e[4] * e[4];
          asm volatile("" : "+v"(s0), "+v"(s1));
#pragma unroll
          for (int q = 1; q < 4; ++q) { s0 = __builtin_fmaf(e[q], e[q], s0); s1 = __builtin_fmaf(e[4 + q], e[4 + q], s1); }
          sg = s0 + s1;
        } else {
          sg = e[0] * e[0];
          asm volatile("" : "+v"(sg));
#pragma unroll
          for (int q = 1; q < 8; ++q) { float sq = e[q] * e[q]; asm volatile("" : "+v"(sq)); sg += sq; }
        }
        grp[kc] = sg;
      }
      ssum[t] = ((grp[0] + grp[1]) + grp[2]) + grp[3];
    }
    inv = rsqrtf(ch_fold<TT>(ssum) / (float)a.H + a.eps);
  };

  lds_u* stH = (lds_u*)((lds_char*)smem + a.off_sth);
  lds_u* stA = (lds_u*)((lds_char*)smem + a.off_sta);
  u32x4 pre[2][4][4];                                  // slots held in registers across an edge (gate/up; then the q/k/v slot)

  if (a.front) {
    // ================================================================= o-projection + residual -> h1 granules
    for (int k = cw; k < s_o; k += 3) {                // one step per consumer at most (2 slots)
      u32x4 wr[4][4];
      float acc[4];
      load4(gslot + k, wr);
      compute4(wr, acc);
      if (fin) {
        const int n0 = orow0 + 4 * sa_o;
        const float r0 = lo_f<P>(h0v[0]), r1 = hi_f<P>(h0v[0]), r2 = lo_f<P>(h0v[1]), r3 = hi_f<P>(h0v[1]);
        const float v0 = rnd<P>(rnd<P>(acc[0]) + r0), v1 = rnd<P>(rnd<P>(acc[1]) + r1);
        const float v2 = rnd<P>(rnd<P>(acc[2]) + r2), v3 = rnd<P>(rnd<P>(acc[3]) + r3);
        gu64* gp = a.g_h1 + (size_t)tl * (a.H / 2) + n0 / 2;
        const unsigned long long tg = (unsigned long long)(tag_base | 1u) << 32;
        if (cu != a.drop_cu) {
          __hip_atomic_store(gp, tg | pack2<P>(v0, v1), RLX_AGENT);
          __hip_atomic_store(gp + 1, tg | pack2<P>(v2, v3), RLX_AGENT);
        }
      }
    }
    gslot += s_o;
    CH_STAMP();                                        // 1: o slots done (granules stored)
    // ---- prologue B: operands of the later ops, in flight while the first edge is gathered
    u32x4 wnn[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) wnn[kc] = __builtin_amdgcn_raw_buffer_load_b128(rnn, (kc * 512 + lane * 8) * 2, 0, 0);
    int sa_d[3];                                       // this consumer's down rows (steps cw, cw + 3, cw + 6)
    unsigned short nwd[3];                             // and the next norm's weight at them
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      sa_d[q] = (cw + 3 * q + cu) % (s_dn ? s_dn : 1);
      nwd[q] = __builtin_amdgcn_raw_buffer_load_b16(rnn, (orow0 + sa_d[q]) * 2, 0, 0);
    }
    load_pos();
    // this consumer's first two gate/up slots go to registers BEFORE the edge: six ring slots return to the loader
    const int gu0 = gslot + cw, gu1 = gslot + cw + 3;
    if (cw < s_gu) load4(gu0, pre[0]);                 // landed long ago
    bool have1 = !(cw + 3 < s_gu);
    auto try_pre1 = [&]() {                            // the second one as soon as it has landed, without holding up the sweep
      if (!have1 && __builtin_amdgcn_readfirstlane(fl[F_LANDED + gu1 % R]) >= (unsigned)gu1 + 1u) { load4(gu1, pre[1]); have1 = true; }
    };
    gather(std::integral_constant<int, 1>{}, a.g_h1, T * (a.H / 2048), tag_base | 1u, stH, try_pre1);
    if (!have1) { load4(gu1, pre[1]); have1 = true; }
    CH_STAMP();                                        // 2: h1 gathered
    if (a.tail) load_rope();                           // pos arrived microseconds ago; lands during gate/up
    build_x((const lds_char*)stH, wn2, std::false_type{});
    float h1d[3] = {0.f, 0.f, 0.f};                    // h1 at this consumer's down rows (lane 16 t: token t)
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (cw + 3 * q < s_dn && fin) h1d[q] = P::to_f(((const __attribute__((address_space(3))) u16*)stH)[tl * a.H + orow0 + sa_d[q]]);
    // ================================================================= gate/up + SiLU -> act granules
    CH_STAMP();                                        // 3: gate/up operand built
    auto gu_step = [&](int k, const u32x4 (&wr)[4][4]) {
      const int sa = (k + cu) % s_gu;
      float acc[4];
      compute4(wr, acc);
      if (fin) {
        const float iv = inv;
        const float g0 = rnd_prod<P>(acc[0], iv), u0 = rnd_prod<P>(acc[1], iv);
        const float g1 = rnd_prod<P>(acc[2], iv), u1 = rnd_prod<P>(acc[3], iv);
        const float a0 = rnd<P>(g0 / (1.f + __expf(-g0))) * u0, a1 = rnd<P>(g1 / (1.f + __expf(-g1))) * u1;
        const int m0 = (a.I / ncu) * cu + 2 * sa;      // first act column of this slot
        __hip_atomic_store(a.g_act + (size_t)tl * (a.I / 2) + m0 / 2,
                           ((unsigned long long)(tag_base | 2u) << 32) | pack2<P>(a0, a1), RLX_AGENT);
      }
    };
    if (cw < s_gu) gu_step(cw, pre[0]);
    if (cw + 3 < s_gu) gu_step(cw + 3, pre[1]);
    for (int k = cw + 6; k < s_gu; k += 3) {
      u32x4 wr[4][4];
      load4(gslot + k, wr);
      gu_step(k, wr);
    }
    gslot += s_gu;
    CH_STAMP();                                        // 4: gate/up slots done
    // the first two down rows of this consumer go to registers before the edge (as above)
    u32x4 dpre[2][16];
    auto load16 = [&](int gi, u32x4 (&w16)[16]) {
      const lds_char* sp = wait_slot(gi);
#pragma unroll
      for (int kc = 0; kc < 16; ++kc) w16[kc] = *(const __attribute__((address_space(3))) u32x4*)(sp + kc * 1024 + lane * 16);
      free_slot(gi);
    };
    const int dn0 = gslot + cw, dn1 = gslot + cw + 3;
    bool dhave0 = !(cw < s_dn), dhave1 = !(cw + 3 < s_dn);
    auto try_dpre = [&]() {
      if (!dhave0 && __builtin_amdgcn_readfirstlane(fl[F_LANDED + dn0 % R]) >= (unsigned)dn0 + 1u) { load16(dn0, dpre[0]); dhave0 = true; }
      if (dhave0 && !dhave1 && __builtin_amdgcn_readfirstlane(fl[F_LANDED + dn1 % R]) >= (unsigned)dn1 + 1u) { load16(dn1, dpre[1]); dhave1 = true; }
    };
    try_dpre();
    gather(std::integral_constant<int, (TT * 4 + 2) / 3>{}, a.g_act, T * (a.I / 2048), tag_base | 2u, stA, try_dpre);
    if (!dhave0) { load16(dn0, dpre[0]); dhave0 = true; }
    if (!dhave1) { load16(dn1, dpre[1]); dhave1 = true; }
    CH_STAMP();                                        // 5: act gathered
    // ================================================================= down-projection + residual -> h2 (plain + granules)
    // one K = 8192 row per slot; the operand stays in LDS staging (lane l: k = 512 kc + 8 l .. + 7) and is read with the
    // weights, two 2048-element k-slices at a time: 2 TT accumulators advance together, each slice is finished (folded) on
    // its own and the four are added in slice order -- gv_kernel<.., 4, 4, ..>'s sum
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int k = cw + 3 * q;
      if (k >= s_dn) break;
      const int sa = sa_d[q];
      const lds_char* sp = q < 2 ? (const lds_char*)smem : wait_slot(gslot + k);
      float tot = 0.f;                                 // row t of the wave: token t
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        u32x4 wr[8], xa[TT][8];
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          if (q < 2) wr[kc] = dpre[q < 2 ? q : 0][8 * hf + kc];
          else wr[kc] = *(const __attribute__((address_space(3))) u32x4*)(sp + (8 * hf + kc) * 1024 + lane * 16);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
          for (int kc = 0; kc < 8; ++kc) xa[t][kc] = *(const __attribute__((address_space(3))) u32x4*)((const lds_char*)stA + t * 16384 + (8 * hf + kc) * 1024 + lane * 16);
        if (q >= 2 && hf == 1) free_slot(gslot + k);
        float part[2][TT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int t = 0; t < TT; ++t) part[j][t] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int t = 0; t < TT; ++t) part[j][t] = ch_dot2<P>(wr[4 * j + kc][e], xa[t][4 * j + kc][e], part[j][t]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float u = ch_fold<TT>(part[j]);
          tot = (hf == 0 && j == 0) ? u : tot + u;
        }
      }
      const int n = orow0 + sa;
      float h2 = 0.f;
      if (fin) {
        h2 = rnd<P>(rnd<P>(tot) + h1d[q]);
        const long off = (long)tl * a.H + n;
        a.h[off] = P::from_f(h2);
        if (a.next_norm) a.hw[off] = P::from_f(h2 * P::to_f(nwd[q]));
        fl[F_HV + sa * 4 + tl] = (unsigned)P::from_f(h2);
        ((lds_vf*)fl)[F_SQ + sa * 4 + tl] = h2;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned oldp = 0, oldc = 0;
      if (lane == 0) {
        oldp = __hip_atomic_fetch_add((lds_u*)(fl + F_PAIR + (sa >> 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        oldc = __hip_atomic_fetch_add((lds_u*)(fl + F_SQCNT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      oldp = __builtin_amdgcn_readfirstlane(oldp);
      oldc = __builtin_amdgcn_readfirstlane(oldc);
      if (oldp == 1 && a.tail && fin) {                // second finisher of rows (2j, 2j + 1): one granule per token
        const int j = sa >> 1;
        const unsigned lo = fl[F_HV + (2 * j) * 4 + tl], hi = fl[F_HV + (2 * j + 1) * 4 + tl];
        __hip_atomic_store(a.g_h2 + (size_t)tl * (a.H / 2) + (orow0 + 2 * j) / 2,
                           ((unsigned long long)(tag_base | 3u) << 32) | lo | (hi << 16), RLX_AGENT);
      }
      if (oldc == (unsigned)s_dn - 1 && fin && a.ssq) {   // last finisher: the CU's sums-of-squares group, gv_kernel's order
        lds_vf* sq = (lds_vf*)fl + F_SQ;
        float s0 = sq[0 * 4 + tl] * sq[0 * 4 + tl], s1 = sq[4 * 4 + tl] * sq[4 * 4 + tl];
        asm volatile("" : "+v"(s0), "+v"(s1));
#pragma unroll
        for (int r = 1; r < 4; ++r) {
          const float e0 = sq[r * 4 + tl], e1 = sq[(4 + r) * 4 + tl];
          s0 = __builtin_fmaf(e0, e0, s0);
          s1 = __builtin_fmaf(e1, e1, s1);
        }
        a.ssq[(long)tl * a.ssq_stride + cu] = s0 + s1;
      }
    }
    gslot += s_dn;
    CH_STAMP();                                        // 6: down slots done
    if (a.tail) {
      const int q0 = gslot + cw;
      bool qhave = !(cw < s_q);
      auto try_q = [&]() {                             // this consumer's q/k/v slot, ahead of the edge
        if (!qhave && __builtin_amdgcn_readfirstlane(fl[F_LANDED + q0 % R]) >= (unsigned)q0 + 1u) { load4(q0, pre[0]); qhave = true; }
      };
      try_q();
      gather(std::integral_constant<int, 1>{}, a.g_h2, T * (a.H / 2048), tag_base | 3u, stH, try_q);
      if (!qhave) { load4(q0, pre[0]); qhave = true; }
      CH_STAMP();                                      // 7: h2 gathered
      build_x((const lds_char*)stH, wnn, std::true_type{});
    }
    // the next launch reads the epoch after this one has ended; CU 0 has seen every CU's granules by now
    if (cu == 0 && cw == 0 && lane == 0 && a.epoch) __hip_atomic_store(a.epoch, (tag_base >> 2) + 1u, RLX_AGENT);
  } else {
    // tail-only launch: x = hw as loaded, 1/rms from the plain sums of squares
    float ssum[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) ssum[t] = ((sqv[t][0] + sqv[t][1]) + sqv[t][2]) + sqv[t][3];
    inv = rsqrtf(ch_fold<TT>(ssum) / (float)a.H + a.eps);
  }

  if (a.tail) {
    // ================================================================= q/k/v + RoPE + KV append (plain stores)
    for (int k = cw; k < s_q; k += 3) {                // one step per consumer (3 slots)
      float acc[4];
      if (a.front) {
        compute4(pre[0], acc);
      } else {
        u32x4 wr[4][4];
        load4(gslot + k, wr);
        compute4(wr, acc);
      }
      if (!fin) continue;
      const float iv = inv;
      const int D = a.D, half = D / 2;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int n = qrows * cu + 4 * sa_q + 2 * pr;
        const int head = n / D, dp = n % D, m = dp >> 1;
        const float a0 = rnd_prod<P>(acc[2 * pr], iv), b0 = rnd_prod<P>(acc[2 * pr + 1], iv);
        if (head < a.Hq + a.Hkv) {
          const float cl = P::to_f(rc[pr][0]), chh = P::to_f(rc[pr][1]), sl_ = P::to_f(rc[pr][2]), sh = P::to_f(rc[pr][3]);
          const float lo0 = rnd<P>(mul_rnd<P>(a0, cl) + mul_rnd<P>(-b0, sl_));
          const float hi0 = rnd<P>(mul_rnd<P>(b0, chh) + mul_rnd<P>(a0, sh));
          if (head < a.Hq) {
            u16* dst = a.q_out + ((long)tl * a.Hq + head) * D;
            dst[m] = P::from_f(lo0);
            dst[m + half] = P::from_f(hi0);
          } else {
            u16* dst = a.kc + (long)(head - a.Hq) * a.Lmax * D;
            dst[kc_off(slot_t, m, D)] = P::from_f(lo0);
            dst[kc_off(slot_t, m + half, D)] = P::from_f(hi0);
          }
        } else {
          u16* dst = a.vt + (long)(head - a.Hq - a.Hkv) * D * VT_LD(a.Lmax);
          dst[vt_off(dp, slot_t, D)] = P::from_f(a0);
          dst[vt_off(dp + 1, slot_t, D)] = P::from_f(b0);
        }
      }
    }
  }
  CH_STAMP();                                          // last: wave done
  if (!ok && lane == 0 && a.status) __hip_atomic_fetch_or(a.status, 0xDEAD0000u | (unsigned)(cu & 0xffff), RLX_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------------
// lm_head of the <= 4-row forwards on the same engine (round 5): logits[r][v] = round(1/rms_r * sum_k hw[r][k] W[v][k]) for the
// tied embedding table W [V][2048] (row-major as it is: no second copy).  Reference: `self.lm_head(hidden_states)` behind the final
// norm, umbrella/models/llama.py:130-133.  The MFMA kernel walks its 1002 blocks in two rounds per CU and pays the ramp of a
// launch in each (525 MB in 96 us = 5.4 TB/s; the same kernel reads the 70B's 2.1 GB head at 6.4); here ONE workgroup per CU
// streams its 125 / 126 four-row slots through the LDS ring without a break -- no hand-off between CUs, so nothing can stall
// but the loader / consumer ring inside a workgroup.  Operand rows and 1/rms as in the chain's tail-only launch.
template <typename P, int TT>
__global__ __launch_bounds__(256) void draft_head_kernel(const char* w, int R, unsigned off_flags, int nslots_total, unsigned timeout_ticks,
                                                         const u16* __restrict__ hw, const float* __restrict__ ssq, int ssq_stride,
                                                         int ssq_groups, float eps, float* __restrict__ logits, int T, int V,
                                                         int x_fm_tt) {
  // TT <= 4: token t's sums fold into row t of a wave (ch_fold); TT = 5 .. 8: two folds, tokens 0-3 and 4-7, lane 16 t finishes
  // tokens t and 4 + t.  x_fm_tt > 0: the operand rows are in FM order with that many token tiles (the low-latency schedule's hw).
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = 2048;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cu = blockIdx.x, ncu = gridDim.x;
  lds_vu* fl = (lds_vu*)((lds_char*)smem + off_flags);
  if (threadIdx.x < F_WORDS) fl[threadIdx.x] = 0;
  __syncthreads();
  const long long deadline = (long long)wall_clock64() + (long long)timeout_ticks;
  lds_vu* dead = fl + F_DEAD;
  // slots of this CU: the first (total % ncu) CUs take one more
  const int base = nslots_total / ncu, rem = nslots_total - base * ncu;
  const int n = base + (cu < rem ? 1 : 0);
  const int first = cu * base + min(cu, rem);
  if (wv == 0) {
    // ---- loader: the chain's (four pieces per M0 setting, up to four slots in flight, slot i - 3 published after slot i is issued)
    const unsigned ring0 = (unsigned)(size_t)(lds_char*)smem;
    const unsigned voff = (unsigned)lane * 16u;
    const char* src = w + (size_t)first * CH_SLOT;
    int sk = cu % n, p = 0;
    auto flag_landed = [&](int slot_i) { if (lane == 0) fl[F_LANDED + slot_i % R] = (unsigned)slot_i + 1u; };
    for (int i = 0; i < n; ++i) {
      if (i >= R && __builtin_amdgcn_readfirstlane(fl[F_FREED + p]) < (unsigned)(i - R + 1)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (i >= 3) flag_landed(i - 3);
        if (i >= 2) flag_landed(i - 2);
        if (i >= 1) flag_landed(i - 1);
        if (!lds_wait_ge(fl + F_FREED + p, (unsigned)(i - R + 1), deadline, dead)) return;
      }
      const unsigned dst = ring0 + (unsigned)p * CH_SLOT;
      const char* sp = src + (size_t)sk * CH_SLOT;
      if (++sk == n) sk = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2 nt\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sp + j * 4096), "s"(dst + j * 4096) : "memory");
      }
      if (i >= 3) {
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        flag_landed(i - 3);
      }
      if (++p == R) p = 0;
    }
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    if (n >= 3) flag_landed(n - 3);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (n >= 2) flag_landed(n - 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (n >= 1) flag_landed(n - 1);
    return;
  }
  // ---- consumers
  const int cw = wv - 1;
  const int tl = lane >> 4;
  constexpr int TA = TT < 4 ? TT : 4, TB = TT > 4 ? TT - 4 : 1;
  const bool finA = (lane & 15) == 0 && tl < T, finB = TT > 4 && (lane & 15) == 0 && 4 + tl < T;
  const int OOB = (int)0x80000000;
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(hw), 0, (unsigned)((x_fm_tt ? x_fm_tt * 16 : T) * H * 2), 0x00020000);
  u32x4 x[TT][4];                                      // lane l holds k = 512 kc + 8 l .. + 7 of every operand row
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int f = kc * 512 + lane * 8;               // FM: 8 consecutive features of one token are one 16-byte piece (common.h fm_off)
      const int off = x_fm_tt ? ((((f >> 5) * x_fm_tt + (t >> 4)) * 64 + ((f >> 3) & 3) * 16 + (t & 15)) << 3) : t * H + f;
      x[t][kc] = __builtin_amdgcn_raw_buffer_load_b128(rx, t < T ? off * 2 : OOB, 0, 0);
    }
  const auto rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ssq), 0, (unsigned)((T - 1) * ssq_stride + ssq_groups) * 4u, 0x00020000);
  float ssum[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    float sq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gq = lane + 64 * q;
      sq[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsq, (t < T && gq < ssq_groups) ? (t * ssq_stride + gq) * 4 : OOB, 0, 0));
    }
    ssum[t] = ((sq[0] + sq[1]) + sq[2]) + sq[3];
  }
  float sA[TA], sB[TB];
#pragma unroll
  for (int t = 0; t < TA; ++t) sA[t] = ssum[t];
#pragma unroll
  for (int t = 0; t < TB; ++t) sB[t] = TT > 4 ? ssum[4 + t] : 0.f;
  const float invA = rsqrtf(ch_fold<TA>(sA) / (float)H + eps);       // row t of the wave: token t
  const float invB = TT > 4 ? rsqrtf(ch_fold<TB>(sB) / (float)H + eps) : 1.f;   // ... token 4 + t
  bool ok = true;
  for (int k = cw; k < n; k += 3) {
    const int sa = (k + cu) % n;                       // the loader's slot order
    const int pos = k % R;
    if (ok && !lds_wait_ge(fl + F_LANDED + pos, (unsigned)k + 1u, deadline, dead)) ok = false;
    asm volatile("" ::: "memory");
    const lds_char* sp = (const lds_char*)smem + (size_t)pos * CH_SLOT;
    u32x4 wr[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) wr[r][kc] = *(const __attribute__((address_space(3))) u32x4*)(sp + r * 4096 + kc * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) fl[F_FREED + pos] = (unsigned)k + 1u;
    float part[4][TT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < TT; ++t) part[r][t] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < TT; ++t) part[r][t] = ch_dot2<P>(wr[r][kc][e], x[t][kc][e], part[r][t]);
    float accA[4], accB[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pa[TA], pb[TB];
#pragma unroll
      for (int t = 0; t < TA; ++t) pa[t] = part[r][t];
#pragma unroll
      for (int t = 0; t < TB; ++t) pb[t] = TT > 4 ? part[r][4 + t] : 0.f;
      accA[r] = ch_fold<TA>(pa);
      accB[r] = TT > 4 ? ch_fold<TB>(pb) : 0.f;
    }
    if (finA) {
      f32x4 o = {rnd_prod<P>(accA[0], invA), rnd_prod<P>(accA[1], invA), rnd_prod<P>(accA[2], invA), rnd_prod<P>(accA[3], invA)};
      *reinterpret_cast<f32x4*>(logits + (long)tl * V + (long)(first + sa) * 4) = o;
    }
    if (finB) {
      f32x4 o = {rnd_prod<P>(accB[0], invB), rnd_prod<P>(accB[1], invB), rnd_prod<P>(accB[2], invB), rnd_prod<P>(accB[3], invB)};
      *reinterpret_cast<f32x4*>(logits + (long)(4 + tl) * V + (long)(first + sa) * 4) = o;
    }
  }
}

// ---- host side
static int chain_ring(int T) {
  // 160 KiB per CU: staging [T][H] (4 KiB per row) + [T][I] (16 KiB per row) + flags, the rest is the ring
  const int left = 160 * 1024 - T * (4096 + 16384) - 1024;
  int r = left / CH_SLOT;
  return r > CH_MAXRING ? CH_MAXRING : r;
}

// Per-DEVICE facts and one-time settings (a process may drive several GPUs: the dynamic-LDS attribute and the CU count belong to
// the device that is current at the call, not to the process).
static inline int chain_dev() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : 0; }
static inline int chain_ncu() {
  static int ncu[64] = {};
  const int d = chain_dev();
  if (ncu[d] == 0) {
    hipDeviceProp_t pr;
    ncu[d] = hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : -1;
  }
  return ncu[d];
}
extern "C" int umb_chain_ok(int T, int H, int I, int NQKV, int D, int has_bias) {
  static const bool off = getenv("UMB_NO_CHAIN") != nullptr;
  if (off || has_bias) return 0;
  // T = 4 would keep 256 operand registers for the down-projection (spills); 4-row levels stay on the GEMV launches
  if (T < 1 || T > 3 || H != 2048 || I != 8192 || NQKV % 1024 || D % 2) return 0;
  const int ncu = chain_ncu();
  // rows per CU in whole slots: 8 o / down rows (the GEMV family's sums-of-squares groups), whole 4-row q/k/v slots, one
  // q/k/v slot per consumer at most (its RoPE operands are loaded up front)
  if (ncu != 256) return 0;
  if ((NQKV / ncu) % 4 || NQKV / ncu > 12 || NQKV % ncu) return 0;
  return chain_ring(T) >= 4;
}

extern "C" size_t umb_chain_xchg_bytes(int Tmax, int H, int I) {
  return (size_t)Tmax * (H / 2 + I / 2 + H / 2) * 8 + 256;       // granules | epoch word, status word
}

static size_t chain_tail_off(int Tmax, int H, int I) { return (size_t)Tmax * (H / 2 + I / 2 + H / 2) * 8; }

// granules zero (tag 0 never matches: epochs start at 1), epoch word 1, status 0
extern "C" int umb_chain_xchg_init(void* xchg, int Tmax, int H, int I, hipStream_t st) {
  if (!xchg || Tmax < 1) return UMB_EINVAL;
  if (hipMemsetAsync(xchg, 0, umb_chain_xchg_bytes(Tmax, H, I), st) != hipSuccess) return UMB_EHIP;
  const unsigned one = 1;
  if (hipMemcpyAsync((char*)xchg + chain_tail_off(Tmax, H, I), &one, 4, hipMemcpyHostToDevice, st) != hipSuccess) return UMB_EHIP;
  return hipStreamSynchronize(st) == hipSuccess ? UMB_OK : UMB_EHIP;
}

// host read of the sticky give-up word (synchronises the stream); 0 = every hand-off of every launch so far completed
extern "C" int umb_chain_status(const void* xchg, int Tmax, int H, int I, uint32_t* status_out, hipStream_t st) {
  if (!xchg || !status_out) return UMB_EINVAL;
  if (hipMemcpyAsync(status_out, (const char*)xchg + chain_tail_off(Tmax, H, I) + 64, 4, hipMemcpyDeviceToHost, st) != hipSuccess)
    return UMB_EHIP;
  return hipStreamSynchronize(st) == hipSuccess ? UMB_OK : UMB_EHIP;
}

extern "C" int umb_draft_chain(const UmbChain* c, int dtype, hipStream_t st) {
  if (!c || !c->xchg) return UMB_EINVAL;
  const int T = c->T;
  if (!umb_chain_ok(T, c->H, c->I, c->NQKV, c->D, 0)) return UMB_EINVAL;
  if (!c->front && !c->tail) return UMB_EINVAL;
  if (c->front && (!c->w_o || !c->w_gu || !c->w_down || !c->attn || !c->h || !c->hw || !c->norm2)) return UMB_EINVAL;
  if (c->tail && (!c->w_qkv || !c->pos || !c->slot || !c->cosT || !c->sinT || !c->q_out || !c->k_cache || !c->vt_cache)) return UMB_EINVAL;
  if (c->tail && c->front && !c->next_norm) return UMB_EINVAL;
  if (!c->front && (!c->hw || !c->ssq || c->ssq_groups_in < 1 || c->ssq_groups_in > 256)) return UMB_EINVAL;
  if (c->Lmax % 32 || c->D % 32) return UMB_EINVAL;
  ChainArgs a = {};
  a.w_o = (const char*)c->w_o; a.w_gu = (const char*)c->w_gu; a.w_down = (const char*)c->w_down; a.w_qkv = (const char*)c->w_qkv;
  a.attn = (const u16*)c->attn; a.h = (u16*)c->h; a.hw = (u16*)c->hw; a.ssq = c->ssq;
  a.norm2 = (const u16*)c->norm2; a.next_norm = (const u16*)c->next_norm;
  a.pos = c->pos; a.slot = c->slot; a.cosT = (const u16*)c->cosT; a.sinT = (const u16*)c->sinT; a.q_out = (u16*)c->q_out;
  a.kc = (u16*)c->k_cache; a.vt = (u16*)c->vt_cache;
  char* xb = (char*)c->xchg;
  const size_t gh = (size_t)c->Tmax * (c->H / 2) * 8, ga = (size_t)c->Tmax * (c->I / 2) * 8;
  a.g_h1 = (gu64*)xb; a.g_act = (gu64*)(xb + gh); a.g_h2 = (gu64*)(xb + gh + ga);
  a.epoch = (gu32*)(xb + 2 * gh + ga); a.status = (gu32*)(xb + 2 * gh + ga + 64);
  if (T > c->Tmax) return UMB_EINVAL;
  a.T = T; a.R = chain_ring(T); a.front = c->front; a.tail = c->tail; a.H = c->H; a.I = c->I; a.NQKV = c->NQKV;
  a.ssq_stride = c->ssq_stride; a.ssq_groups_in = c->ssq_groups_in; a.Hq = c->Hq; a.Hkv = c->Hkv; a.D = c->D; a.Lmax = c->Lmax;
  a.eps = c->eps;
  a.off_sth = (unsigned)a.R * CH_SLOT; a.off_sta = a.off_sth + (unsigned)T * 4096u;
  const unsigned off_flags = a.off_sta + (unsigned)T * 16384u;
  const char* etm = getenv("UMB_CHAIN_TIMEOUT_MS");    // 100 MHz wall clock: 20 ms by default (read per call: tests change it)
  a.timeout_ticks = (etm ? atoll(etm) : 20ll) * 100000ll;
  const char* edr = getenv("UMB_CHAIN_TEST_DROP_CU");
  a.drop_cu = edr ? atoi(edr) : -1;
  static const int hint_on = getenv("UMB_CHAIN_HINT") ? atoi(getenv("UMB_CHAIN_HINT")) : 2;
  static const int thin_on = getenv("UMB_CHAIN_THIN") ? atoi(getenv("UMB_CHAIN_THIN")) : 1;
  a.hint = hint_on;
#ifdef UMB_CHAIN_TRACE
  const char* etr = getenv("UMB_CHAIN_TRACE_PTR");
  static int trace_seq = 0;                             // one region per launch (64 regions of [256][4][32] stamps)
  a.trace = etr ? (unsigned long long*)strtoull(etr, nullptr, 0) + (size_t)(trace_seq++ % 64) * 32768 : nullptr;
#endif
  const unsigned lds = off_flags + F_WORDS * 4u;
  const unsigned slots = (unsigned)((c->H / 256) * c->H * 2 / CH_SLOT) | (unsigned)((2 * c->I / 256) * c->H * 2 / CH_SLOT) << 8 |
                         (unsigned)((c->H / 256) * c->I * 2 / CH_SLOT) << 16 | (unsigned)((c->NQKV / 256) * c->H * 2 / CH_SLOT) << 24;
  if (c->front && c->ssq && c->ssq_stride < 256) return UMB_EINVAL;
#define CH_GO(TTV)                                                                                                     \
  do {                                                                                                                 \
    static bool attr_done_dev[64] = {};                                                                                \
    bool& attr_done = attr_done_dev[chain_dev()];                                                                      \
    if (!attr_done) {                                                                                                  \
      if (hipFuncSetAttribute((const void*)draft_chain_kernel<P, TTV>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              160 * 1024) != hipSuccess) return UMB_EHIP;                                              \
      attr_done = true;                                                                                                \
    }                                                                                                                  \
    hipLaunchKernelGGL((draft_chain_kernel<P, TTV>), dim3(256), dim3(256), lds, st, a.w_o, a.w_gu, a.w_down, a.w_qkv,  \
                       a.R, off_flags, (c->front ? 1 : 0) | (c->tail ? 2 : 0) | (thin_on ? 4 : 0), slots,               \
                       (unsigned)a.timeout_ticks, a);                                                                 \
  } while (0)
  DISPATCH_DTYPE(dtype, {
    if (T == 1) CH_GO(1);
    else if (T == 2) CH_GO(2);
    else CH_GO(3);
  })
#undef CH_GO
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}

// lm_head of a <= 4-row forward on the streaming engine (see draft_head_kernel).  w_rows: the head's weights as plain row-major
// [V][2048] rows (a tied model's embedding table as it is).  x: [rows][2048] = h * final-norm weight, ssq: the producer's sums of
// squares ([rows][ssq_stride], `groups` valid), logits: fp32 [rows][V].
extern "C" int umb_head_stream_ok(int rows, int V, int H) {
  static const bool off = getenv("UMB_NO_HEAD_STREAM") != nullptr;
  if (off || rows < 1 || rows > 8 || H != 2048 || V % 4 || V < 4 * 256 * 8) return 0;
  return chain_ncu() == 256;
}
extern "C" int umb_head_stream(float* logits, const void* x, const float* ssq, int ssq_stride, int groups, float eps,
                               const void* w_rows, int rows, int V, int H, int x_fm_tt, int dtype, hipStream_t st) {
  if (x_fm_tt < 0 || (x_fm_tt && x_fm_tt * 16 < rows)) return UMB_EINVAL;
  if (!umb_head_stream_ok(rows, V, H) || !logits || !x || !ssq || !w_rows || groups < 1 || groups > 256 || groups > ssq_stride) return UMB_EINVAL;
  const int R = 8;                                      // ring: 8 x 16 KiB; the operand rows live in registers
  const unsigned off_flags = (unsigned)R * CH_SLOT;
  const unsigned lds = off_flags + F_WORDS * 4u;
  // No give-up deadline here: this kernel's waits are between the loader wave and the consumer waves of ONE workgroup (no other
  // workgroup, no residency assumption), so they always end; a deadline could only turn a slow launch into silently wrong logits
  // (there is no status word on this path).  ~43 s of the 100 MHz clock = never.
  const unsigned ticks = 0xFFFFFFFFu;
#define UMB_HEAD_(PT, TTV)                                                                                                       \
  do {                                                                                                                           \
    static bool once_dev[64] = {};                                                                                               \
    bool& once = once_dev[chain_dev()];                                                                                          \
    if (!once) {                                                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&draft_head_kernel<PT, TTV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds) != hipSuccess) return UMB_EHIP;                                                          \
      once = true;                                                                                                               \
    }                                                                                                                            \
    hipLaunchKernelGGL((draft_head_kernel<PT, TTV>), dim3(256), dim3(256), lds, st, (const char*)w_rows, R, off_flags, V / 4, ticks, \
                       (const u16*)x, ssq, ssq_stride, groups, eps, logits, rows, V, x_fm_tt);                                         \
  } while (0)
  DISPATCH_DTYPE(dtype, {
    if (rows == 1) UMB_HEAD_(P, 1); else if (rows == 2) UMB_HEAD_(P, 2); else if (rows == 3) UMB_HEAD_(P, 3); else if (rows == 4) UMB_HEAD_(P, 4);
    else if (rows == 5) UMB_HEAD_(P, 5); else if (rows == 6) UMB_HEAD_(P, 6); else if (rows == 7) UMB_HEAD_(P, 7); else UMB_HEAD_(P, 8);
  })
#undef UMB_HEAD_
  UMB_LAUNCH_CHECK();
  return UMB_OK;
}
