// Fused attention for gfx950:  O = softmax(Q K^T * scale) V, scores never leave registers.
//
// Replaces CrossAttention.forward's einsum / softmax / einsum (`ldm/modules/attention.py:227-249`), which
// materialises a (B*heads, Tq, Tk) fp32 score tensor (1.07 GB at 512x512).  Numerics kept from
// the reference: q.k^T accumulated in fp32 and scaled in fp32 (`:230-233`), softmax in fp32.
//
// Structure (per workgroup: 4 waves x 32 query rows, one (batch, head); key tiles of 64 streamed through LDS):
//  * S^T = K Q^T on v_mfma_f32_32x32x16_f16 with K as the A operand: the accumulator then has the query on
//    the lane and 16 keys in registers, so the row max / row sum of the online softmax are in-lane
//    reductions plus ONE exchange with lane^32.
//  * P^T stays in registers: registers 8s..8s+7 of the score accumulator, packed to fp16, ARE the B operand
//    of the next MFMA (O^T += V^T P^T) with no LDS round trip.  The k-order inside a step is permuted
//    (element j of lane-half h is key 16s + 8(j>>2) + 4h + (j&3)); the V^T fragment is fetched with two
//    8-byte LDS reads at exactly those key offsets.
//  * V arrives row-major ([B*TkS][heads*d], i.e. a column block of the fused q/k/v projection) and is staged
//    row-major in LDS ([key][d]); the V^T fragments of the O^T += V^T P^T MFMA are fetched with the transposing
//    LDS read ds_read_b64_tr_b16 (four keys of one channel per lane), so no transposed copy of V exists anywhere.
//    The key-row stride of the LDS tile is 64 or 192 bytes mod 256: the four rows a 32-lane half reads then
//    cover all 64 banks once.
//  * head dims 40 / 80 / 160 (and 8..64 for the reduced test config): QK^T pads d to a multiple of 16 with
//    zero chunks in LDS, PV pads to a multiple of 32 rows.
//  * K / V tiles are register-staged and double-buffered: the next tile's global loads are issued before
//    the MFMA phase and written to the other LDS buffer after it.
#include <type_traits>

#include "kernels.h"

namespace sdeo {

typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// bytes per key of the row-major V tile in LDS (see the file comment)
constexpr int attn_vrow(int dt) { return dt == 1 ? 64 : (dt <= 3 ? 192 : 320); }

struct AP {
  f16* o;
  const f16* q;
  const f16* k;
  const f16* v;
  int ldo, ldq, ldk, ldv;
  int H, Tq, Tk, TkS, TkSv, d;
  float scale_log2;
  int causal;        // 1: key j is visible to query t only when j <= t (CLIP text transformer)
};

struct AP2 { AP k[1]; };

// lane ^ 32 exchange on the VALU (gfx950 v_permlane32_swap): after the swap the pair (lo, hi) holds the value of lanes 0..31 and
// of lanes 32..63 in every lane, so a reduction over the two halves needs no LDS round trip (ds_bpermute) in the softmax chain.
// Inline asm, not __builtin_amdgcn_permlane32_swap: hipcc 7.2 maps both results of the builtin to ONE register for float users
// (r0 + r1 is emitted as v_add v1, v1, v1).  The s_nop covers the VALU-write -> permlane read hazard the compiler cannot see.
__device__ __forceinline__ void swap32(float v, float& lo, float& hi) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}
// v_max_f32 without the canonicalising v_max v, v, v the compiler puts in front of fmaxf on values it cannot prove quiet
// (results of inline asm, loop-carried state); the operands here are never signalling NaNs
__device__ __forceinline__ float max_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float xor32_max(float v) {
  float lo, hi;
  swap32(v, lo, hi);
  return max_raw(lo, hi);
}
__device__ __forceinline__ float xor32_sum(float v) {
  float lo, hi;
  swap32(v, lo, hi);
  return lo + hi;
}

// KS = 1: four waves, each owning 32 queries and walking all keys.  KS = 2 ("key split"): eight waves, the two waves of a
// pair own the same 32 queries and each takes one 32-key half of every staged tile, so the serial per-tile chain
// (QK^T -> max -> exp -> PV) per wave halves and twice as many waves hide it; the two partial (m, l, O) states are merged
// through LDS at the end.  Reduction order is fixed (half 0 then half 1), so results stay deterministic.
// MPAD (head dims with a free K-dim slot, d % 16 == 8: d = 40): the softmax's per-score fma moves into the QK^T MFMA.  Q is scaled
// by scale * log2 e once (fp16), the K tile's pad channel d holds 1, and Q's pad channel d holds -m, the row's running reference
// (kept as an fp16-exact value), so the MFMA delivers  s * scale * log2 e - m  and the inner loop is max / exp / cvt only.
// QB: 32-query blocks per workgroup (4: 128 queries; 2: 64 queries, twice the workgroups, each staging the K / V tiles for half as
// many queries but with only QB * KS waves behind one barrier).
template <int D16, int KS, bool MPAD, int QB = 4>
__global__ __launch_bounds__(64 * QB * KS, (KS == 2 ? (D16 <= 4 ? 4 : 2) : (D16 <= 2 ? 4 : (D16 <= 5 ? 2 : 1))))
void attention_kernel(const AP2 pp) {
  kernarg_warm<sizeof(AP2)>();
  // the whole parameter block in one batch, pinned in SGPRs (otherwise ~5 dependent s_load round trips before the first Q load);
  // blockIdx.y is always 0 (one problem per launch)
  AP pl = pp.k[0];
  // (integers / floats only: a pointer that has been through the asm loses its address space and its accesses become flat_ ones)
  asm volatile("" : "+s"(pl.ldo), "+s"(pl.ldq), "+s"(pl.ldk), "+s"(pl.ldv), "+s"(pl.H),
               "+s"(pl.Tq), "+s"(pl.Tk), "+s"(pl.TkS), "+s"(pl.TkSv), "+s"(pl.d), "+s"(pl.scale_log2), "+s"(pl.causal));
  const AP& p = pl;
  constexpr int NT = 64 * QB * KS;                           // threads per workgroup
  constexpr int NKB = 2 / KS;                                // 32-key blocks of a tile each wave handles
  constexpr int DT = (D16 + 1) / 2;                          // 32-row tiles of O^T
  constexpr int KROW = D16 * 32 + ((D16 * 2) % 2 == 0 ? 16 : 0);  // K tile row bytes (odd multiple of 16)
  constexpr int VROW = attn_vrow(DT);                        // V tile bytes per key: >= DT*64, == 64 or 192 (mod 256)
  constexpr int KBYTES = 64 * KROW;
  constexpr int VBYTES = 64 * VROW;
  constexpr int STAGE = KBYTES + VBYTES;
  constexpr int KCH = D16 * 2;                               // 16-byte chunk slots per K / V row
  constexpr int KITEMS = 64 * KCH, KPASS = (KITEMS + NT - 1) / NT;
  // Row sum for free (odd D16 only): P V pads d to DT*32 = 16 (D16 + 1) rows of O^T, QK^T only to 16 D16, so column R1 = 16 D16 of
  // the V tile is never a real channel.  Filled with ones, the MFMA leaves sum_k P[k][q] in that row of O^T (register 8 of the
  // last tile, lanes 0..31): no per-score add, no cross-half exchange per tile, and the running sum is rescaled with O.
  constexpr bool ONES = (D16 % 2) == 1;
  constexpr int R1 = 16 * D16;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int qb = KS == 2 ? (wave >> 1) : wave;               // 32-query block of this wave
  const int kh = KS == 2 ? (wave & 1) : 0;                   // key half of this wave (KS == 2)
  // XCD-aware placement (speed only): workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous
  // run of (batch, head) pairs so the K / V^T of a head stay in ONE private L2 instead of being streamed by all eight
  const int qtiles = (p.Tq + 32 * QB - 1) / (32 * QB);
  int wg = blockIdx.x, nwg = gridDim.x;
  {
    const int q_ = nwg >> 3, r_ = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + idx;
  }
  const int bh = wg / qtiles;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = (wg - bh * qtiles) * (32 * QB) + qb * 32;
  const int qrow = q0 + lq;
  const bool qvalid = qrow < p.Tq;
  const int d = p.d;
  // loop-invariant scalars pinned in SGPRs (otherwise re-read from the kernel-argument segment inside the key loop, an
  // s_load + lgkmcnt(0) in the softmax chain of every tile)
  float sl2 = p.scale_log2;
  int Tk = p.Tk;
  int causal = p.causal;
  asm volatile("" : "+s"(sl2), "+s"(Tk), "+s"(causal));

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[qrow][ks*16 + 8*lh + 0..7]
  f16x8 qf[D16];
  {
    const f16* qp = p.q + ((size_t)b * p.Tq + (qvalid ? qrow : 0)) * p.ldq + h * d;
    // unconditional loads (rows past Tq read row 0, chunks past d read chunk 0; zeroed at their first use below): a load under an
    // exec mask cannot be counted by the compiler's wait pass, and every later wait of the prologue would become vmcnt(0)
#pragma unroll
    for (int ks = 0; ks < D16; ++ks) {
      const int c = (ks * 2 + lh) * 8;
      qf[ks] = *reinterpret_cast<const f16x8*>(qp + (c < d ? c : 0));
    }
  }

  const f16* kbase = p.k + (size_t)b * p.TkS * p.ldk + h * d;
  const f16* vbase = p.v + (size_t)b * p.TkSv * p.ldv + h * d;
  const int ntiles = (p.Tk + 63) / 64;
  const f16x8 zero8 = f16x8{0, 0, 0, 0, 0, 0, 0, 0};

  // per-thread staging state, hoisted out of the key loop: source column and LDS destinations of the 16-byte chunks this thread moves.
  // DEEP: two register sets, the loads of a tile are issued TWO tiles ahead (small head dims, where a tile's compute is
  // shorter than a global round trip); otherwise one set, loads issued at the top of the previous tile.
  // Every load is UNCONDITIONAL (rows past Tk read row Tk - 1 again: those keys are masked to -inf, so P is 0 against a finite
  // V; threads without a chunk read chunk 0 and never store): with loads under exec masks or uniform branches the compiler
  // cannot count them and turns every wait into vmcnt(0), which serialises the prefetch with the compute it should hide behind.
  constexpr bool DEEP = D16 <= 5 && QB == 4;      // 64-query workgroups: one set (two passes per thread already fill the registers)
  f16x8 kr[DEEP ? 2 : 1][KPASS], vr[DEEP ? 2 : 1][KPASS];
  int krow[KPASS], kcol[KPASS], kdst[KPASS], vdst[KPASS];
#pragma unroll
  for (int i = 0; i < KPASS; ++i) {
    const int it = tid + i * NT;
    const int row = it / KCH, c = it - row * KCH;
    const bool use = it < KITEMS && c * 8 < d;
    krow[i] = row;
    kcol[i] = use ? c * 16 : 0;                      // bytes
    kdst[i] = use ? row * KROW + c * 16 : -1;        // chunk slots past d keep their zeros (K) / zeros and the ones column (V)
    vdst[i] = use ? row * VROW + c * 16 : -1;
  }
  const int last_key = Tk - 1;
  // byte offsets from the (wave-uniform) head bases stay below 2^32 and their factors below 2^24: one v_mad_u32_u24 per
  // address and the scalar-base form of global_load instead of a 64-bit multiply-add chain per load
  unsigned ldk2 = (unsigned)p.ldk * 2u, ldv2 = (unsigned)p.ldv * 2u;
  asm volatile("" : "+s"(ldk2), "+s"(ldv2));
  auto load_tile = [&](auto SET, int kt) {
    constexpr int rs = SET.value;
    const int key0 = kt * 64;
#pragma unroll
    for (int i = 0; i < KPASS; ++i) {
      const unsigned row = (unsigned)min(key0 + krow[i], last_key);
      kr[rs][i] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(kbase) + (__umul24(row, ldk2) + (unsigned)kcol[i]));
      vr[rs][i] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(vbase) + (__umul24(row, ldv2) + (unsigned)kcol[i]));
    }
  };
  auto store_tile = [&](auto SET, int stage) {
    constexpr int rs = SET.value;
    char* ks_ = smem + stage * STAGE;
    char* vs_ = ks_ + KBYTES;
#pragma unroll
    for (int i = 0; i < KPASS; ++i) {
      if (kdst[i] >= 0) *reinterpret_cast<f16x8*>(ks_ + kdst[i]) = kr[rs][i];
      if (vdst[i] >= 0) *reinterpret_cast<f16x8*>(vs_ + vdst[i]) = vr[rs][i];
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // Prologue order: EVERY global load of the prologue (Q above, tile 0, tile 1) is in flight before the LDS is initialised, so the
  // zero fill and its barrier pass under one memory round trip instead of standing between two of them.
  load_tile(S0{}, 0);
  if constexpr (DEEP) load_tile(S1{}, 1);           // set 1 carries the odd tiles, set 0 the even ones
  // both stages start as zeros (+ the ones column of V): staging only ever writes the chunk slots below d
  for (int off = tid * 16; off < 2 * STAGE; off += NT * 16) *reinterpret_cast<f16x8*>(smem + off) = zero8;
  __syncthreads();
  if (ONES && tid < 128) *reinterpret_cast<f16*>(smem + (tid >> 6) * STAGE + KBYTES + (tid & 63) * VROW + R1 * 2) = (f16)1.f;
  if (MPAD && tid < 128) *reinterpret_cast<f16*>(smem + (tid >> 6) * STAGE + (tid & 63) * KROW + d * 2) = (f16)1.f;   // K[key][d] = 1

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // the first real use of the Q fragments HERE (MPAD: scaled by scale * log2 e), before the later prefetch loads are issued: their
  // wait is then placed in front of the loop.  Left to the first MFMA of the loop the compiler must assume Q still in flight on
  // every iteration and waits vmcnt(0) at the top of each tile, which serialises the prefetched K / V loads with the compute they
  // were meant to hide behind.
#pragma unroll
  for (int ks = 0; ks < D16; ++ks) {
    const bool live = qvalid && (ks * 2 + lh) * 8 < d;
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[ks][j] = live ? (MPAD ? (f16)((float)qf[ks][j] * sl2) : qf[ks][j]) : (f16)0.f;
  }
#pragma unroll
  for (int ks = 0; ks < D16; ++ks) asm volatile("" : "+v"(qf[ks]));
  store_tile(S0{}, 0);
  if constexpr (DEEP) load_tile(S0{}, 2);
  __syncthreads();

  // one 64-key tile; at its end the NEXT tile (loaded two iterations ago) goes from registers to the other LDS buffer and the
  // freed register set takes the loads of the tile after that: a global round trip has two tiles of compute to hide behind
  // the compute of one 64-key tile (QK^T, online softmax, P V) out of LDS buffer `cur`
  auto compute = [&](int kt, int cur) __attribute__((always_inline)) {
    const char* ks_ = smem + cur * STAGE;
    const char* vs_ = ks_ + KBYTES;

    // ---- S^T = K Q^T for the two 32-key blocks of this tile
    f32x16 s[NKB];
#pragma unroll
    for (int ki = 0; ki < NKB; ++ki) {
      const int kb = KS == 2 ? kh : ki;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[ki][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < D16; ++ks) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(ks_ + (kb * 32 + lq) * KROW + (ks * 2 + lh) * 16);
        s[ki] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[ki], 0, 0, 0);
      }
    }
    // ---- V^T fragments of this tile, fetched NOW (transposing LDS reads): their latency passes under the softmax below
    //      instead of in front of every P V MFMA.  Lane (q = (lane & 15) >> 2, pp = lane & 3) of each 16-lane group addresses key
    //      row q, channels 4pp..4pp+3 of the group's 4-key x 16-channel block and receives its own channel for the four keys.
    f16x8 vf[NKB][2][DT];
#pragma unroll
    for (int ki = 0; ki < NKB; ++ki)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int kb = KS == 2 ? kh : ki;
        const char* vblk = vs_ + (kb * 32 + 16 * st + 4 * lh + ((lane & 15) >> 2)) * VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          typedef __attribute__((address_space(3))) h16x4* lds_h4;
          const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(vblk + t * 64));
          const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(vblk + t * 64 + 8 * VROW));
          vf[ki][st][t] = __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    // ---- online softmax (base-2), key index of s[kb][r] = kt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*lh
    //      the running max is kept in scaled units (score * scale * log2 e); the scale itself is folded into one fma
    // masking only on a tile that needs it: ONE uniform branch per tile (inside the register loop it became a branch per score register)
    if ((kt + 1) * 64 > Tk || causal) {
#pragma unroll
      for (int ki = 0; ki < NKB; ++ki)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kb = KS == 2 ? kh : ki;
          const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= Tk || (causal && key > qrow)) s[ki][r] = -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int ki = 0; ki < NKB; ++ki)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[ki][r]);
    float rs = 0.f;
    if constexpr (MPAD) {
      // the scores arrive as s * scale * log2 e - m_run (0 instead of m_run while the row has no reference yet)
      mx = xor32_max(mx);
      const bool unset = m_run == -INFINITY;
      if (__any(mx > 8.0f || (unset && mx > -INFINITY))) {       // wave-uniform, rare (see the lazy rescale below)
        const float m_abs = unset ? mx : m_run + fmaxf(mx, 0.f);  // the row's reference in absolute units: only ever raised
        const float m_q = (float)(f16)m_abs;                       // ... as the fp16 value Q's pad channel can hold
        const bool valid = m_abs > -INFINITY;                      // false: no visible key so far, the row stays unset
        const float delta = valid ? m_q - (unset ? 0.f : m_run) : 0.f;
        const float alpha = unset ? 0.f : __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int ki = 0; ki < NKB; ++ki)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[ki][r] -= delta;          // this tile was formed against the old reference
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        if (valid) {
          m_run = m_q;
          if (lh == 1) qf[D16 - 1][0] = (f16)(-m_q);               // channel d of this row's Q: lanes lh = 1, last k-step, element 0
        }
      }
#pragma unroll
      for (int ki = 0; ki < NKB; ++ki)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[ki][r]);
          s[ki][r] = pv;
          if (!ONES) rs += pv;
        }
    } else {
    mx = xor32_max(mx) * sl2;
    // Lazy rescale: the running reference m_run moves only when some row's new maximum exceeds it by more than 8 (base-2 units),
    // i.e. P = 2^(s - m_run) stays below 2^8 -- exact in the fp32 accumulators and far inside fp16 for the P V operand.  The
    // result is the same softmax (any common reference per row cancels in O / l); with the eager form ~70 % of the tiles of a
    // 32-query block took the 19-instruction rescale of O on random scores, with this one only the first few do.
    if (__any(mx > m_run + 8.0f)) {    // wave-uniform; -inf + 8 = -inf, so the first valid tile always takes it
      const float m_new = max_raw(m_run, mx);
      // a row that has not seen a valid key yet (m_new = -inf; its O and l are still 0) must not form -inf - -inf
      const float alpha = m_new == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    // a key half that has not seen a single valid key yet (KS == 2, Tk <= 32) keeps m = -inf: exponentiate against 0
    const float m_use = (KS == 2 && m_run == -INFINITY) ? 0.f : m_run;
#pragma unroll
    for (int ki = 0; ki < NKB; ++ki)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(s[ki][r], sl2, -m_use));
        s[ki][r] = pv;
        if (!ONES) rs += pv;
      }
    }
    if (!ONES) rs = xor32_sum(rs);
    l_run += rs;

    // ---- O^T += V^T P^T : P^T fragments straight from the score registers
#pragma unroll
    for (int ki = 0; ki < NKB; ++ki)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        f16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (f16)s[ki][8 * st + j];
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[ki][st][t], pf, o[t], 0, 0, 0);
      }
  };
  if constexpr (DEEP) {
    // at the end of a tile the NEXT tile (loaded two iterations ago) goes from registers to the other LDS buffer and the freed
    // register set takes the loads of the tile after that: a global round trip has two tiles of compute to hide behind
    for (int kt = 0; kt < ntiles; kt += 2) {      // an odd tile count runs one fully masked tile (P = 0) at the end
      compute(kt, 0);
      store_tile(S1{}, 1);
      load_tile(S1{}, kt + 3);
      __syncthreads();
      compute(kt + 1, 1);
      store_tile(S0{}, 0);
      load_tile(S0{}, kt + 4);
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < ntiles; ++kt) {
      const int cur = kt & 1;
      load_tile(S0{}, kt + 1);
      compute(kt, cur);
      store_tile(S0{}, cur ^ 1);
      __syncthreads();
    }
  }

  if constexpr (KS == 2) {
    // ---- merge the two key halves of each query block through LDS (field-major, lane-contiguous: conflict-free).
    //      The loop's last __syncthreads() already separates the final tile reads from these writes.
    // A tile of a half that lies entirely past Tk leaves m = -inf, l = 0 there; exp2(-inf - m) = 0 handles it, and half 0
    // always sees key 0, so m below is finite.
    constexpr int NF = 2 + DT * 16;
    float* mg = reinterpret_cast<float*>(smem) + qb * NF * 64 + lane;
    if (kh == 1) {
      mg[0] = m_run;
      mg[64] = l_run;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mg[(2 + t * 16 + r) * 64] = o[t][r];
    }
    __syncthreads();
    if (kh == 1) return;
    const float m1 = mg[0], l1 = mg[64];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    l_run = l_run * a0 + l1 * a1;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = o[t][r] * a0 + mg[(2 + t * 16 + r) * 64] * a1;
  }

  if constexpr (ONES) l_run = __shfl(o[DT - 1][8], lq, 64);     // row R1 of O^T: register 8 of the last tile in lanes 0..31
  // ---- normalise and store: o[t][4g..4g+3] = O[qrow][t*32 + 8g + 4lh + 0..3]
  if (qvalid) {
    const float inv = 1.0f / l_run;
    f16* op = p.o + ((size_t)b * p.Tq + qrow) * p.ldo + h * d;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dc = t * 32 + 8 * g + 4 * lh;
        if (dc < d) {
          f16x4 ov;
#pragma unroll
          for (int j = 0; j < 4; ++j) ov[j] = (f16)(o[t][4 * g + j] * inv);
          *reinterpret_cast<f16x4*>(op + dc) = ov;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Wide heads (d = 256 / 512: the VAE AttnBlock, `ldm/modules/diffusionmodules/model.py:179-203`, one head of 512 channels).
// 32 queries x d fp32 of O do not fit one wave's registers, so the four waves of a workgroup share ONE block of 32 queries and
// split the CHANNELS: wave w owns channels [w*DS, (w+1)*DS).  Per 32-key tile each wave forms the partial S^T = K[:, slice] Q[:, slice]^T
// of its slice, the four partials are summed through LDS in a fixed order (identical full scores in every wave, deterministic),
// every wave runs the same online softmax on them (redundant VALU work, but no second exchange), and multiplies P into ITS slice
// of V: O^T[slice] += V[:, slice]^T P^T.  MFMA layouts, the in-register P^T operand and the transposing V reads are those of
// attention_kernel above.  K / V tiles: 32 keys x d, register-staged, double-buffered in LDS.
// ------------------------------------------------------------------------------------------------
template <int DS>
__global__ __launch_bounds__(256, 1) void attention_wide_kernel(const AP p) {
  constexpr int D = 4 * DS;                                   // head dim
  constexpr int KS16 = DS / 16;                               // QK^T k-steps per wave (even)
  constexpr int DT = DS / 32;                                 // 32-row tiles of O^T per wave
  constexpr int KROW = D * 2 + 16;                            // K tile row bytes: odd multiple of 16
  constexpr int VROW = D * 2 + 64;                            // V tile row bytes: 64 (mod 256), see attn_vrow
  constexpr int KBYTES = 32 * KROW, VBYTES = 32 * VROW, STAGE = KBYTES + VBYTES;
  constexpr int CH = D / 8;                                   // 16-byte chunks per row
  constexpr int PASS = 32 * CH / 256;                         // chunks per thread per tile (K and V each)
  static_assert(KROW % 32 == 16 && VROW % 256 == 64 && (32 * CH) % 256 == 0, "tile geometry");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xch = smem + 2 * STAGE;                               // partial-score exchange: [wave][4][lane] f32x4 (16 KB)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, lh = lane >> 5;
  const int qtiles = (p.Tq + 31) >> 5;
  const int bh = blockIdx.x / qtiles;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = (blockIdx.x - bh * qtiles) * 32;
  const int qrow = q0 + lq;
  const bool qvalid = qrow < p.Tq;
  const int c0 = wave * DS;                                   // first channel of this wave's slice

  f16x8 qf[KS16];
  {
    const f16* qp = p.q + ((size_t)b * p.Tq + (qvalid ? qrow : 0)) * p.ldq + h * D + c0;
#pragma unroll
    for (int ks = 0; ks < KS16; ++ks)
      qf[ks] = *reinterpret_cast<const f16x8*>(qp + (ks * 2 + lh) * 8);       // row 0 for queries past Tq: never stored
  }

  const f16* kbase = p.k + (size_t)b * p.TkS * p.ldk + h * D;
  const f16* vbase = p.v + (size_t)b * p.TkSv * p.ldv + h * D;
  // loop-invariant scalars pinned in SGPRs (see attention_kernel)
  float sl2 = p.scale_log2;
  int Tk = p.Tk;
  unsigned ldk2 = (unsigned)p.ldk * 2u, ldv2 = (unsigned)p.ldv * 2u;
  asm volatile("" : "+s"(sl2), "+s"(Tk), "+s"(ldk2), "+s"(ldv2));
  const int ntiles = (Tk + 31) / 32;
  const int last_key = Tk - 1;
  // K / V tiles are register-staged two tiles ahead (two register sets): with one workgroup per CU and one wave per SIMD nothing
  // else hides a global round trip, and one tile of compute (~1 us) is shorter than it (measured: 2.7 us per tile with the
  // loads issued one tile ahead)
  f16x8 kr[2][PASS], vr[2][PASS];
  auto load_tile = [&](auto SET, int kt) {
    constexpr int rs = SET.value;
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const int it = tid + i * 256;
      const int row = it / CH, c = it - row * CH;
      // unconditional, counted loads (see attention_kernel): rows >= Tk re-read row Tk - 1, their keys are masked (P = 0, V finite)
      const unsigned key = (unsigned)min(kt * 32 + row, last_key);
      kr[rs][i] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(kbase) + (__umul24(key, ldk2) + (unsigned)c * 16u));
      vr[rs][i] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(vbase) + (__umul24(key, ldv2) + (unsigned)c * 16u));
    }
  };
  auto store_tile = [&](auto SET, int stage) {
    constexpr int rs = SET.value;
    char* ks_ = smem + stage * STAGE;
    char* vs_ = ks_ + KBYTES;
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const int it = tid + i * 256;
      const int row = it / CH, c = it - row * CH;
      *reinterpret_cast<f16x8*>(ks_ + row * KROW + c * 16) = kr[rs][i];
      *reinterpret_cast<f16x8*>(vs_ + row * VROW + c * 16) = vr[rs][i];
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  load_tile(S0{}, 0);
  load_tile(S1{}, 1);                                 // set 1 carries the odd tiles, set 0 the even ones
  store_tile(S0{}, 0);
#pragma unroll
  for (int ks = 0; ks < KS16; ++ks) asm volatile("" : "+v"(qf[ks]));     // Q's wait in front of the loop (see attention_kernel)
  load_tile(S0{}, 2);
  __syncthreads();

  auto compute = [&](int kt, int cur) __attribute__((always_inline)) {
    const char* ks_ = smem + cur * STAGE;
    const char* vs_ = ks_ + KBYTES;

    // ---- partial S^T over this wave's channel slice
    f32x16 s, s1;                                     // two accumulation chains: a 32x32x16 MFMA has 16 passes of latency
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS16; ks += 2) {
      const f16x8 kf0 = *reinterpret_cast<const f16x8*>(ks_ + lq * KROW + (c0 / 8 + ks * 2 + lh) * 16);
      const f16x8 kf1 = *reinterpret_cast<const f16x8*>(ks_ + lq * KROW + (c0 / 8 + ks * 2 + 2 + lh) * 16);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qf[ks], s, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qf[ks + 1], s1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] += s1[r];
    {
      f32x4* dst = reinterpret_cast<f32x4*>(xch) + wave * 256 + lane;       // [wave][4 register groups][lane]: lane-contiguous
#pragma unroll
      for (int g = 0; g < 4; ++g) dst[g * 64] = f32x4{s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
    }
    __syncthreads();
    // ---- full scores: the four partials in a fixed order (every wave computes the same bits)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const f32x4* src = reinterpret_cast<const f32x4*>(xch) + w * 256 + lane;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = src[g * 64];
        s[4 * g] += v[0]; s[4 * g + 1] += v[1]; s[4 * g + 2] += v[2]; s[4 * g + 3] += v[3];
      }
    }
    // ---- online softmax (base 2); key of s[r] = kt*32 + (r&3) + 8*(r>>2) + 4*lh
    float mx = -INFINITY;
    if ((kt + 1) * 32 > Tk) {                         // one uniform branch per tile
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= Tk) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = xor32_max(mx) * sl2;
    // lazy rescale of the running reference (threshold 2^8, see attention_kernel): O is DT x 16 registers per wave here, and the
    // eager form multiplied all of them by alpha on every tile
    if (__any(mx > m_run + 8.0f)) {                   // tile 0 always holds key 0: m_run becomes finite there
      const float m_new = max_raw(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      m_run = m_new;
    }
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -m_run));
      s[r] = pv;
      rs += pv;
    }
    l_run += xor32_sum(rs);
    // ---- O^T[slice] += V[:, slice]^T P^T
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      f16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (f16)s[8 * st + j];
      const char* vblk = vs_ + (16 * st + 4 * lh + ((lane & 15) >> 2)) * VROW + (c0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        typedef __attribute__((address_space(3))) h16x4* lds_h4;
        const h16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(vblk + t * 64));
        const h16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4)(vblk + t * 64 + 8 * VROW));
        const f16x8 vf = __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[t], 0, 0, 0);
      }
    }
  };
  // at the end of a tile the NEXT tile (loaded two iterations ago) goes from registers to the other LDS buffer and the freed
  // register set takes the loads of the tile after that; the closing barrier also frees the exchange area
  for (int kt = 0; kt < ntiles; kt += 2) {            // an odd tile count runs one fully masked tile (P = 0) at the end
    compute(kt, 0);
    store_tile(S1{}, 1);
    load_tile(S1{}, kt + 3);
    __syncthreads();
    compute(kt + 1, 1);
    store_tile(S0{}, 0);
    load_tile(S0{}, kt + 4);
    __syncthreads();
  }

  if (qvalid) {
    const float inv = 1.0f / l_run;
    f16* op = p.o + ((size_t)b * p.Tq + qrow) * p.ldo + h * D + c0;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = (f16)(o[t][4 * g + j] * inv);
        *reinterpret_cast<f16x4*>(op + t * 32 + 8 * g + 4 * lh) = ov;
      }
  }
}

template <int DS>
static int launch_attn_wide(const AP& ap, int B, hipStream_t stream) {
  constexpr int D = 4 * DS;
  constexpr int smem = 2 * (32 * (D * 2 + 16) + 32 * (D * 2 + 64)) + 4 * 64 * 16 * 4;
  static_assert(smem <= 160 * 1024, "LDS");
  static DeviceOnce attr_done;
  if (attr_done.need()) {
    SDEO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_wide_kernel<DS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done.mark();
  }
  hipLaunchKernelGGL((attention_wide_kernel<DS>), dim3(cdiv(ap.Tq, 32) * B * ap.H), dim3(256), smem, stream, ap);
  SDEO_HIP(hipGetLastError());
  return 0;
}

template <int D16, int KS, bool MPAD, int QB = 4>
static int launch_attn_ks(const AP2& ap2, int count, int B, hipStream_t stream) {
  const AP& ap = ap2.k[0];
  constexpr int DT = (D16 + 1) / 2;
  constexpr int KROW = D16 * 32 + ((D16 * 2) % 2 == 0 ? 16 : 0);
  constexpr int stage2 = 2 * (64 * KROW + 64 * attn_vrow(DT));
  constexpr int merge = KS == 2 ? QB * (2 + DT * 16) * 64 * 4 : 0;    // LDS of the key-half merge
  constexpr int smem = stage2 > merge ? stage2 : merge;
  static DeviceOnce attr_done;
  if (attr_done.need()) {
    SDEO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<D16, KS, MPAD, QB>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done.mark();
  }
  dim3 grid(cdiv(ap.Tq, 32 * QB) * B * ap.H, count);
  hipLaunchKernelGGL((attention_kernel<D16, KS, MPAD, QB>), grid, dim3(64 * QB * KS), smem, stream, ap2);
  SDEO_HIP(hipGetLastError());
  return 0;
}

// key split pays when there are enough keys for the serial chain to dominate and the head dim keeps the merge small
static int attn_key_split(const AP& ap, int d16) {
  static const int forced = [] { const char* e = getenv("SDEO_ATTN_KS"); return e ? atoi(e) : 0; }();
  if (d16 > 5) return 1;
  if (forced) return forced;
  return ap.Tk >= 128 ? 2 : 1;
}

template <int D16>
static int launch_attn(const AP2& ap, int count, int B, hipStream_t stream) {
  if constexpr (D16 <= 5) {
    // a free K-dim slot right after the head's channels (d = 8, 24, 40, 56, 72): the reference-in-the-pad form
    static const bool mpad_on = [] { const char* e = getenv("SDEO_ATTN_MPAD"); return !e || atoi(e) != 0; }();
    const bool mpad = mpad_on && ap.k[0].d % 16 == 8;
    if (attn_key_split(ap.k[0], D16) == 2) {
      if constexpr (D16 == 3 || D16 == 5) {
        // 64-query workgroups where 128-query ones leave CUs empty (T = 1024 on 16 (batch, head) pairs: 128 workgroups on 256 CUs;
        // measured 19.3 -> 17.2 us at d = 80); with the chip already full they lose (T = 4096: 71 -> 95 us: twice the staging per
        // query, one prefetch set).  SDEO_ATTN_QB = 2 / 4 forces one form (measurement).  Instantiated for d = 40 / 80.
        static const int qb_forced = [] { const char* e = getenv("SDEO_ATTN_QB"); return e ? atoi(e) : 0; }();
        const bool small_grid = cdiv(ap.k[0].Tq, 128) * B * ap.k[0].H < 256;
        if (qb_forced ? qb_forced == 2 : small_grid)
          return mpad ? launch_attn_ks<D16, 2, true, 2>(ap, count, B, stream) : launch_attn_ks<D16, 2, false, 2>(ap, count, B, stream);
      }
      return mpad ? launch_attn_ks<D16, 2, true>(ap, count, B, stream) : launch_attn_ks<D16, 2, false>(ap, count, B, stream);
    }
    return mpad ? launch_attn_ks<D16, 1, true>(ap, count, B, stream) : launch_attn_ks<D16, 1, false>(ap, count, B, stream);
  } else {
    return launch_attn_ks<D16, 1, false>(ap, count, B, stream);
  }
}

static int attn_prepare(AP& ap, const AttnArgs& a) {
  SDEO_CHECK(a.o && a.q && a.k && a.v, "attention: null operand");
  SDEO_CHECK(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0 && a.TkS >= a.Tk && a.TkSv >= a.Tk,
             "attention: bad sizes B=%d H=%d Tq=%d Tk=%d TkS=%d TkSv=%d", a.B, a.H, a.Tq, a.Tk, a.TkS, a.TkSv);
  SDEO_CHECK((a.d % 8 == 0 && a.d >= 8 && a.d <= 160) || a.d == 256 || a.d == 512,
             "attention: head dim %d unsupported (multiple of 8 up to 160, or 256 / 512)", a.d);
  SDEO_CHECK(a.d <= 160 || !a.causal, "attention: causal masking is not built for head dim %d", a.d);
  SDEO_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0,
             "attention: strides must keep 16-byte alignment (ldq=%d ldk=%d ldv=%d ldo=%d)", a.ldq, a.ldk, a.ldv, a.ldo);
  SDEO_CHECK((reinterpret_cast<uintptr_t>(a.q) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.k) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(a.v) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.o) & 7) == 0,
             "attention: operands must be 16-byte aligned");
  SDEO_CHECK(!a.causal || a.Tq == a.Tk, "attention: causal needs Tq == Tk (got %d, %d)", a.Tq, a.Tk);
  ap = AP{a.o, a.q, a.k, a.v, a.ldo, a.ldq, a.ldk, a.ldv, a.H, a.Tq, a.Tk, a.TkS, a.TkSv, a.d, a.scale * 1.4426950408889634f, a.causal ? 1 : 0};
  return 0;
}

static int attn_dispatch(const AP2& ap, int count, int B, hipStream_t stream) {
  const int d = ap.k[0].d;
  switch (cdiv(d, 16)) {
    case 1: return launch_attn<1>(ap, count, B, stream);
    case 2: return launch_attn<2>(ap, count, B, stream);
    case 3: return launch_attn<3>(ap, count, B, stream);
    case 4: return launch_attn<4>(ap, count, B, stream);
    case 5: return launch_attn<5>(ap, count, B, stream);
    case 6: return launch_attn<6>(ap, count, B, stream);
    case 8: return launch_attn<8>(ap, count, B, stream);
    case 10: return launch_attn<10>(ap, count, B, stream);
    default: return fail("attention: head dim %d not instantiated", d);
  }
}

int attention(const AttnArgs& a, hipStream_t stream) {
  AP2 ap{};
  if (int rc = attn_prepare(ap.k[0], a)) return rc;
  if (a.d == 512) return launch_attn_wide<128>(ap.k[0], a.B, stream);
  if (a.d == 256) return launch_attn_wide<64>(ap.k[0], a.B, stream);
  return attn_dispatch(ap, 1, a.B, stream);
}

int attention(f16* o, int ldo, const f16* q, int ldq, const f16* k, int ldk, const f16* v, int ldv, int B, int H,
              int Tq, int Tk, int TkS, int TkSv, int d, float scale, hipStream_t stream, int causal) {
  return attention(AttnArgs{o, q, k, v, ldo, ldq, ldk, ldv, B, H, Tq, Tk, TkS, TkSv, d, scale, causal}, stream);
}

}  // namespace sdeo
