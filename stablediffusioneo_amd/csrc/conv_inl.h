// Device-side pieces shared by the implicit-GEMM kernels (conv_gemm.hip) and the halo-reuse 3x3 kernel (conv_halo.hip):
// kernel parameter block, epilogue, XCD-aware tile order, LDS-DMA / fragment-read helpers.
#pragma once
#include <utility>

#include "kernels.h"

namespace sdeo {

struct KP {
  const f16* x;
  const f16* w;
  f16* y;
  float* y32;
  const float* bias;
  const float* bias2;
  const f16* res;
  float* ws;
  int M, N, K;
  int Hi, Wi, Cin, Ho, Wo, S, stride, pad, ups;
  int HoWo;
  int ldx, ldw, ldy, ldres, ld_bias2;
  int act, bias_per_row;
  float scale;
  int nk, nk_per_split, splitk;
  int tiles_m, tiles_n;
  // measurement builds only (-DSDEO_DEBUG_KERNELS, `python -m stablediffusioneo_amd.build --debug` -> libsdeo_dbg.so; the
  // field is ignored by the production kernels): SDEO_DBG_GEMM bits, results are wrong under them: 1 = activation DMAs read
  // the zero page, 2 = weight DMAs do, 4 = no MFMAs, 8 = no DMAs after the prologue, 16 = no fragment reads, 32 = no epilogue
  int dbg;
  int n_fastest;     // tile order inside an XCD's contiguous run: 1 = all N tiles of an M tile are neighbours
  int coalesce;      // host-checked: the LDS-transposed 16-byte epilogue applies (see epilogue_rows)
  // ---- LayerNorm folded into this GEMM (the consumer of a pre-LN residual stream, `attention.py:381-385`):
  //      LN(x) W^T + b = rstd * (x W'^T - mean * s) + b'   with W' = W * gamma (fp16), s[n] = sum_k W'[n][k],
  //      b'[n] = b[n] + sum_k beta[k] W[n][k] (passed as `bias`).  The K loop runs on the RAW x; the epilogue applies the two
  //      per-row scalars.  (mean, rstd) come from per-row partial (sum, sum of squares) written by the PRODUCER of x.
  const float* ln_stats;   // [rows][ln_ld][2] partial (sum, sumsq) per row of x, ln_strips valid partials per row; null = off
  const float* ln_s;       // [N]
  int ln_strips, ln_ld;
  float ln_invc, ln_eps;
  // ---- fp8 weights (conv_gemm_dma_kernel<..., W8>): `w` points at bytes, ldw counts bytes, the accumulators are multiplied by the
  //      per-output-channel scale before anything else in the epilogue (or in the split-K reduce)
  const float* wscale;     // [N]; null = fp16 weights
  // ---- producer side: per-row partial (sum, sumsq) of the values this launch stores, one partial per TN-wide strip
  float* stats_out;        // [M][stats_ld][2]; null = off
  int stats_ld;
  // ---- producer side, GroupNorm (`util.py:217-219`): per-(image, M tile of the image, group) partial (sum, sumsq) of the values this
  //      launch stores, so that the GroupNorm that consumes y needs no statistics pass.  Every (image, tile, group) entry is written
  //      by exactly one workgroup (host-checked: strips cover whole groups, tiles do not straddle images, unsplit fp16 plan);
  //      the consumer sums the gn_slots entries of a group in a fixed order (deterministic, no atomics).
  float* gn_out;           // [B][gn_slots][gn_groups][2]; null = off
  int gn_cpg, gn_slots, gn_groups;
  // ---- consumer side, GroupNorm (+ SiLU) of the INPUT (conv3x3_halo_kernel): x is the raw tensor, gn_in the partials its producer's
  //      epilogue wrote (gn_out of that launch).  The MFMA waves turn them into per-channel (a, b) = (rstd gamma, beta - mean rstd gamma) of
  //      the workgroup's image and channel range in LDS while the first DMAs fly; the loader waves rewrite every staged patch in place
  //      (y = silu(a x + b), padding left at zero) before the barrier that shows it to the MFMA waves: no GroupNorm launch, no
  //      normalised copy of the tensor in memory.
  const float* gn_in;       // [B][gn_in_slots][32][2]; null = off
  const float* gn_gamma;    // [Cin]
  const float* gn_beta;     // [Cin]
  int gn_in_slots, gn_in_cpg, gn_in_silu;
  float gn_in_eps, gn_in_inv;   // 1 / (cpg * H * W)
  // ---- block-scaled fp8 operands (conv_gemm_dma_kernel<..., MX>, GEMM only): x and w hold OCP e4m3fn codes (one byte per element,
  //      addressed through this block as fp16 arrays of half the length: K, Cin, ldx, ldw count PAIRS of codes), mx_sx / mx_sw one
  //      e8m0 scale byte per 32 codes of a row.  The MFMA is v_mfma_scale_f32_16x16x128_f8f6f4: 128 codes per K-step.
  const unsigned char* mx_sx;   // [M][mx_ldsx]; null = off
  const unsigned char* mx_sw;   // [N][mx_ldsw]
  int mx_ldsx, mx_ldsw;
};

// what a kernel receives (blockIdx.y indexes the problem; always one)
struct KP2 { KP k[1]; };

// Ablation / stamp switches exist only in the measurement build: in the production library dbg_on() is the constant false and
// every branch on it (and the stamp code) is compiled out of the K loops.
#ifdef SDEO_DEBUG_KERNELS
__device__ __forceinline__ bool dbg_on(const KP& p, int bit) { return (p.dbg & bit) != 0; }
#else
__device__ __forceinline__ constexpr bool dbg_on(const KP&, int) { return false; }
#endif

// SDEO_DBG_GEMM bit 6 (64): per-workgroup phase stamps (s_memrealtime, 100 MHz, chip-wide) written by MFMA wave 0, lane 0:
// [0] kernel entry, [1] address set-up done, [2] first K-step visible, [3] K loop done, [4] epilogue stores issued,
// [5] stores complete.  One copy of the buffer per translation unit; read with sdeo_debug_read_stamps.
constexpr int kStampWGs = 4096, kStampSlots = 16;
static __device__ unsigned long long g_stamps[kStampWGs * kStampSlots];
__device__ __forceinline__ void stamp_cycles(const KP& p, int slot) {      // shader-clock counter (for the in-kernel clock rate)
  if (dbg_on(p, 64) && threadIdx.x == 0) {
    const int wg = blockIdx.x + gridDim.x * blockIdx.z;
    if (wg < kStampWGs) g_stamps[wg * kStampSlots + slot] = __builtin_amdgcn_s_memtime();
  }
}
__device__ __forceinline__ void stamp(const KP& p, int slot) {
  if (dbg_on(p, 64) && threadIdx.x == 0) {
    const int wg = blockIdx.x + gridDim.x * blockIdx.z;
    if (wg < kStampWGs) g_stamps[wg * kStampSlots + slot] = __builtin_amdgcn_s_memrealtime();
  }
}

// 256 bytes of zeros: DMA source for padded / out-of-range rows
static __device__ __attribute__((aligned(256))) const unsigned int g_zero_page[64] = {0};

template <int BK>
__device__ __forceinline__ int swz_chunk(int row, int chunk) {
  // conflict-free for the 16-lane groups of ds_read_b128 when lanes read rows r..r+15 at one k-chunk
  if (BK == 64) return chunk ^ ((row >> 1) & 7);
  return chunk ^ (((row >> 3) & 1) * 3);
}

// ---- epilogue shared by all kernels: lane holds n = nb + i*16 + fq*4 + {0..3} (4 consecutive channels) of output row
//      mrow[j] (-1: no such row) for accumulator tile (i, j); nb = first column of this wave's TN-wide strip
// `scratch` (optional): wave-private LDS, >= epilogue_scratch_bytes(TN) bytes, free of pending DMAs and of other waves'
// fragment reads.  With it (and p.coalesce set by the host: fp16 output, no split-K, N % 8 == 0, 16-byte aligned rows) the
// wave transposes each 16-row block of fp32 results through LDS so that every lane then moves 8 consecutive channels:
// 16-byte residual loads and 16-byte stores in runs of TN*2 contiguous bytes per row, instead of 8-byte pieces in 32-byte
// runs (measured: the 8-byte epilogue took ~4 us of a 24 us conv).  Same arithmetic, same single rounding.
constexpr int epilogue_scratch_bytes(int tn) { return 16 * (tn * 4 + 16) + 32 * tn; }   // + [16 rows][tn / 8] float2 row-statistics partials
                                                                                        //   or [16 rows][tn] fp16 copy of the stored block (GroupNorm partials)

// The epilogue reads ~25 scalar parameters.  Left to the compiler they are loaded from the kernel-argument segment where first
// used, one s_load + s_waitcnt lgkmcnt(0) after the other (the loads are invariant, so it prefers re-loading to keeping SGPRs
// live across the K loop): a serial chain of scalar-cache round trips at the start of every epilogue.  Kernels therefore copy the
// parameters right BEFORE the pre-epilogue barrier and pin the epilogue's fields in SGPRs here: one batch of wide loads whose
// latency hides behind the barrier.
__device__ __forceinline__ unsigned long long pin_u64(unsigned long long v) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int pin_i32(int v) {
  int r = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" : "+s"(r));
  return r;
}
__device__ __forceinline__ float pin_f32(float v) { return __builtin_bit_cast(float, pin_i32(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ void pin_epilogue_scalars(KP& q) {
  // pointers travel through the asm as integers and come back as GLOBAL pointers: pinned as generic pointers they would lose
  // their address space and every access would become a flat_ one (which also counts on lgkmcnt and would serialise with the
  // epilogue's LDS traffic)
#define SDEO_PIN_PTR(f) q.f = (decltype(q.f))(__attribute__((address_space(1))) std::remove_pointer_t<decltype(q.f)>*)pin_u64((unsigned long long)q.f);
  SDEO_PIN_PTR(y) SDEO_PIN_PTR(y32) SDEO_PIN_PTR(bias) SDEO_PIN_PTR(bias2) SDEO_PIN_PTR(res) SDEO_PIN_PTR(ws)
  SDEO_PIN_PTR(wscale) SDEO_PIN_PTR(stats_out) SDEO_PIN_PTR(ln_stats) SDEO_PIN_PTR(ln_s) SDEO_PIN_PTR(gn_out)
#undef SDEO_PIN_PTR
  q.M = pin_i32(q.M); q.N = pin_i32(q.N); q.HoWo = pin_i32(q.HoWo); q.ldy = pin_i32(q.ldy); q.ldres = pin_i32(q.ldres);
  q.ld_bias2 = pin_i32(q.ld_bias2); q.act = pin_i32(q.act); q.bias_per_row = pin_i32(q.bias_per_row); q.scale = pin_f32(q.scale);
  q.splitk = pin_i32(q.splitk); q.coalesce = pin_i32(q.coalesce); q.stats_ld = pin_i32(q.stats_ld); q.ln_strips = pin_i32(q.ln_strips);
  q.ln_ld = pin_i32(q.ln_ld); q.ln_invc = pin_f32(q.ln_invc); q.ln_eps = pin_f32(q.ln_eps);
  q.gn_cpg = pin_i32(q.gn_cpg); q.gn_slots = pin_i32(q.gn_slots); q.gn_groups = pin_i32(q.gn_groups);
}

// blocks staged per round: the largest divisor of MI whose scratch fits `budget` bytes per wave and whose residual registers
// (PASSES 16-byte vectors per block) stay within 8 vectors
constexpr int epilogue_blocks(int tn, int mi, int budget) {
  const int passes = (16 * (tn / 8) + 63) / 64;
  int best = 1;
  for (int nb = 1; nb <= mi; ++nb)
    if (mi % nb == 0 && nb * epilogue_scratch_bytes(tn) <= budget && nb * passes <= 8) best = nb;
  return best;
}

// one 16-row block of the coalesced epilogue, flags resolved at compile time (a scalar branch per flag per tile cost more
// than the arithmetic: ~60 taken branches per wave)
template <int NI, int TN, bool B2, int ACT>
__device__ __forceinline__ void stage_block(const KP& p, const f32x4 (&accj)[NI], const f32x4 (&bias)[NI], int m, int nb, int fq,
                                            int frow, char* scratch) {
  constexpr int ROWB = TN * 4 + 16;
  const float* b2p = nullptr;
  if (B2) b2p = p.bias2 + (size_t)(m >= 0 ? m / p.HoWo : 0) * p.ld_bias2 + nb + fq * 4;
  f32x4 b2[NI];
  if (B2) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      b2[i] = (nb + i * 16 + fq * 4 < p.N) ? *reinterpret_cast<const f32x4*>(b2p + i * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    f32x4 v = accj[i] + bias[i];
    if (B2) v += b2[i];
    if (ACT == 1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = silu_f(v[t]);
    } else if (ACT == 2) {
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = quick_gelu_f(v[t]);
    }
    v *= p.scale;
    *reinterpret_cast<f32x4*>(scratch + frow * ROWB + (i * 16 + fq * 4) * 4) = v;
  }
}

// (mean, rstd) of one row of the LayerNorm'ed operand from the producer's per-strip partials (fixed order: deterministic)
__device__ __forceinline__ void ln_row_scalars(const KP& p, int row, float& r, float& rm) {
  const float2* src = reinterpret_cast<const float2*>(p.ln_stats) + (size_t)row * p.ln_ld;
  float s = 0.f, q = 0.f;
  for (int t0 = 0; t0 < p.ln_strips; t0 += 8) {       // 8 independent loads in flight, summed in a fixed order
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = t0 + u < p.ln_strips ? src[t0 + u] : make_float2(0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += v[u].x; q += v[u].y; }
  }
  const float mean = s * p.ln_invc;
  float var = q * p.ln_invc - mean * mean;
  var = var < 0.f ? 0.f : var;
  r = rsqrtf(var + p.ln_eps);
  rm = r * mean;
}

// acc <- rstd * acc - rstd * mean * s   (see KP::ln_stats).  `lnrow` (optional, LDS): the (rstd, rstd * mean) of this wave's
// rows, indexed j * 16 + frow, computed at kernel start while the first K-step was in flight (conv_gemm_dma_kernel); without it
// each lane sums its rows' partials here.
template <int NI, int MI>
__device__ __forceinline__ void ln_correct(const KP& p, f32x4 (&acc)[NI][MI], const int (&mrow)[MI], int nb, int fq,
                                           const float2* lnrow) {
  const int frow = threadIdx.x & 15;
  float r[MI], rm[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    r[j] = 0.f; rm[j] = 0.f;
    if (lnrow) { const float2 v = lnrow[j * 16 + frow]; r[j] = v.x; rm[j] = v.y; }
    else if (mrow[j] >= 0) ln_row_scalars(p, mrow[j], r[j], rm[j]);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = nb + i * 16 + fq * 4;
    const f32x4 sv = n < p.N ? *reinterpret_cast<const f32x4*>(p.ln_s + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = acc[i][j] * r[j] - sv * rm[j];
  }
}

// NB = 16-row blocks staged per LDS round trip (scratch >= NB * epilogue_scratch_bytes(TN) per wave).  With NB = 1 a wave tile of
// MI blocks costs MI dependent write -> read -> store chains; staging several blocks first issues their residual loads, LDS
// reads and global stores back to back (measured with tools/gemm_ablate.py: the epilogue is ~40 % of a many-tile short-K GEMM
// and of the order of a third of every ~8 us conv / GEMM launch).
template <int NI, int MI, int TN, int NB = 1>
__device__ __forceinline__ void epilogue_rows(const KP& p, f32x4 (&acc)[NI][MI], const int (&mrow)[MI], int nb, int fq, int z,
                                              const f32x4 (&bpre)[NI], bool use_bpre, char* scratch = nullptr,
                                              const float2* lnrow = nullptr) {
  static_assert(NB >= 1 && MI % NB == 0, "blocks per staging round");
  if (p.wscale && p.splitk == 1) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nb + i * 16 + fq * 4;
      const f32x4 sc = n < p.N ? *reinterpret_cast<const f32x4*>(p.wscale + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] *= sc;
    }
  }
  if (p.ln_stats && p.splitk == 1) ln_correct<NI, MI>(p, acc, mrow, nb, fq, lnrow);
  if constexpr (NI % 2 == 0 && TN % 32 == 0) {
    if (scratch && p.coalesce && p.act == 3) {
      // GEGLU pair epilogue (see below) through the same LDS transposition: value * gelu(gate) is formed in the accumulator
      // layout, staged as fp32 [16 rows][TN / 2] and stored in 16-byte pieces, TN bytes contiguous per row
      constexpr int TNO = TN / 2, ROWB = TNO * 4 + 16, G = TNO / 8, PASSES = (16 * G + 63) / 64;
      constexpr int BLOCKB = epilogue_scratch_bytes(TN);
      const int lane = threadIdx.x & 63, frow = lane & 15;
      f32x4 bias[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = nb + i * 16 + fq * 4;
        bias[i] = use_bpre ? bpre[i] : ((p.bias && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f});
      }
      const int nbo = nb >> 1, No = p.N >> 1;
#pragma unroll
      for (int j0 = 0; j0 < MI; j0 += NB) {
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          char* sc = scratch + jb * BLOCKB;
#pragma unroll
          for (int i = 0; i < NI; i += 2) {
            const f32x4 v = acc[i][j0 + jb] + bias[i], g = acc[i + 1][j0 + jb] + bias[i + 1];
            f32x4 o;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = v[t] * gelu_erf_f(g[t]);
            *reinterpret_cast<f32x4*>(sc + frow * ROWB + ((i >> 1) * 16 + fq * 4) * 4) = o;
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          const char* sc = scratch + jb * BLOCKB;
          const int m = mrow[j0 + jb];
#pragma unroll
          for (int t = 0; t < PASSES; ++t) {
            const int id = t * 64 + lane;
            const int r = id / G, n = nbo + (id - r * G) * 8;
            const int msrc = __shfl(m, r & 15, 64);
            if (id < 16 * G && n < No && msrc >= 0) {
              const char* src = sc + r * ROWB + (n - nbo) * 4;
              const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 16);
              f16x8 o;
#pragma unroll
              for (int u = 0; u < 4; ++u) { o[u] = (f16)lo[u]; o[4 + u] = (f16)hi[u]; }
              *reinterpret_cast<f16x8*>(p.y + (size_t)msrc * p.ldy + n) = o;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
  }
  if (scratch && p.coalesce && p.act != 3) {
    constexpr int ROWB = TN * 4 + 16;            // odd multiple of 16 bytes: the 16 rows of a block start in different banks
    constexpr int G = TN / 8;                    // 8-channel groups per row
    constexpr int PASSES = (16 * G + 63) / 64;
    constexpr int BLOCKB = epilogue_scratch_bytes(TN);
    const int lane = threadIdx.x & 63, frow = lane & 15;
    // per-channel bias of this lane's tiles (zeros when absent), second-phase coordinates: the same for every block
    f32x4 bias[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nb + i * 16 + fq * 4;
      bias[i] = use_bpre ? bpre[i] : ((p.bias && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
    int rr[PASSES], nn[PASSES];
    bool live[PASSES];
#pragma unroll
    for (int t = 0; t < PASSES; ++t) {
      const int id = t * 64 + lane;
      rr[t] = id / G;
      nn[t] = nb + (id - rr[t] * G) * 8;
      live[t] = id < 16 * G && nn[t] < p.N;
    }
    const int mode = (p.bias2 ? 3 : 0) + p.act;   // act is 0..2 here (3 = GEGLU never takes this path)
    // GroupNorm partials (KP::gn_out): lane c (and c + 64 ...) sums column c of every stored block, read back from an fp16 copy of
    // the block in LDS -- two registers per column instead of sixteen per (row, vector) slot
    constexpr int GCH = (TN + 63) / 64;
    float gsum[GCH], gsq[GCH];
#pragma unroll
    for (int c = 0; c < GCH; ++c) { gsum[c] = 0.f; gsq[c] = 0.f; }
    stamp(p, 8);
#pragma unroll
    for (int j0 = 0; j0 < MI; j0 += NB) {
      // residual loads of these blocks first: they are independent of the staging
      int mm[NB][PASSES];
      f16x8 resv[NB][PASSES];
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const int m = mrow[j0 + jb];
#pragma unroll
        for (int t = 0; t < PASSES; ++t) {
          const int msrc = __shfl(m, rr[t] & 15, 64);          // lane r (fq == 0) holds the row index of block row r
          mm[jb][t] = live[t] ? msrc : -1;
          resv[jb][t] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
      if (p.res) {
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
          for (int t = 0; t < PASSES; ++t)
            if (mm[jb][t] >= 0) resv[jb][t] = *reinterpret_cast<const f16x8*>(p.res + (size_t)mm[jb][t] * p.ldres + nn[t]);
      }
      if (j0 == 0) stamp(p, 9);
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const int m = mrow[j0 + jb];
        char* sc = scratch + jb * BLOCKB;
        f32x4 accj[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) accj[i] = acc[i][j0 + jb];
        if (dbg_on(p, 2048)) continue;          // ablation: no staging writes
        switch (mode) {
          case 0: stage_block<NI, TN, false, 0>(p, accj, bias, m, nb, fq, frow, sc); break;
          case 1: stage_block<NI, TN, false, 1>(p, accj, bias, m, nb, fq, frow, sc); break;
          case 2: stage_block<NI, TN, false, 2>(p, accj, bias, m, nb, fq, frow, sc); break;
          case 3: stage_block<NI, TN, true, 0>(p, accj, bias, m, nb, fq, frow, sc); break;
          case 4: stage_block<NI, TN, true, 1>(p, accj, bias, m, nb, fq, frow, sc); break;
          default: stage_block<NI, TN, true, 2>(p, accj, bias, m, nb, fq, frow, sc); break;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (j0 == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(p, 10); }
      if (dbg_on(p, 1024)) continue;            // ablation: staging only
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const char* sc = scratch + jb * BLOCKB;
        float2* spart = reinterpret_cast<float2*>(scratch + jb * BLOCKB + 16 * ROWB);     // [16 rows][G] partial (sum, sumsq) of the stored values
#pragma unroll
        for (int t = 0; t < PASSES; ++t) {
          float ssum = 0.f, ssq = 0.f;
          if (mm[jb][t] >= 0) {
            const char* src = sc + rr[t] * ROWB + (nn[t] - nb) * 4;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 16);
            f16x8 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              o[u] = (f16)(lo[u] + (float)resv[jb][t][u]);
              o[4 + u] = (f16)(hi[u] + (float)resv[jb][t][4 + u]);
            }
            if (!dbg_on(p, 512)) *reinterpret_cast<f16x8*>(p.y + (size_t)mm[jb][t] * p.ldy + nn[t]) = o;
            if (p.stats_out) {
#pragma unroll
              for (int u = 0; u < 8; ++u) { const float f = (float)o[u]; ssum += f; ssq += f * f; }
            }
            if (p.gn_out) *reinterpret_cast<f16x8*>(scratch + jb * BLOCKB + 16 * ROWB + (rr[t] * TN + (nn[t] - nb)) * 2) = o;
          } else if (p.gn_out && t * 64 + lane < 16 * G) {     // rows / columns past the problem count as zeros
            *reinterpret_cast<f16x8*>(scratch + jb * BLOCKB + 16 * ROWB + (rr[t] * TN + (nn[t] - nb)) * 2) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
          }
          if (p.stats_out && t * 64 + lane < 16 * G) spart[t * 64 + lane] = make_float2(ssum, ssq);     // index = row * G + group
        }
      }
      if (p.gn_out) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          const f16* blk = reinterpret_cast<const f16*>(scratch + jb * BLOCKB + 16 * ROWB);
#pragma unroll
          for (int c = 0; c < GCH; ++c) {
            const int col = c * 64 + lane;
            if (col < TN) {
#pragma unroll
              for (int r = 0; r < 16; ++r) { const float f = (float)blk[r * TN + col]; gsum[c] += f; gsq[c] += f * f; }
            }
          }
        }
      }
      if (p.stats_out) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
          const int m = mrow[j0 + jb];
          const float2* spart = reinterpret_cast<const float2*>(scratch + jb * BLOCKB + 16 * ROWB);
          if (lane < 16 && m >= 0 && nb < p.N) {   // lanes 0..15 have fq == 0 and frow == lane: `m` is the row index of block row `lane`;
                                                   // a strip wholly past N (tile padding) has no slot in the statistics row
            float ts = 0.f, tq = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) { const float2 v = spart[lane * G + g]; ts += v.x; tq += v.y; }
            reinterpret_cast<float2*>(p.stats_out)[(size_t)m * p.stats_ld + nb / TN] = make_float2(ts, tq);
          }
        }
      }
      if (j0 == 0) stamp(p, 11);
      __builtin_amdgcn_wave_barrier();        // the next round overwrites the scratch rows
    }
    if (p.gn_out) {      // this wave's per-column (sum, sumsq) at the start of its scratch: gn_partials_finish combines the waves
#pragma unroll
      for (int c = 0; c < GCH; ++c)
        if (c * 64 + lane < TN) reinterpret_cast<float2*>(scratch)[c * 64 + lane] = make_float2(gsum[c], gsq[c]);
    }
    stamp(p, 12);
    return;
  }
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = mrow[j];
    if (m < 0) continue;
    if constexpr (NI % 2 == 0 && TN % 32 == 0) {
      if (p.act == 3) {
        // GEGLU (`attention.py:49-56`) fused into ff.net.0.proj: the weight rows were interleaved at load time in blocks of
        // 16 (block 2q = value channels 16q..16q+15, block 2q+1 = their gates), so accumulator tiles i and i+1 hold, in the
        // same lane and register, a value and its gate.  Output column = value channel; ldy counts N/2 columns.
#pragma unroll
        for (int i = 0; i < NI; i += 2) {
          const int n = nb + i * 16 + fq * 4;
          if (n >= p.N) continue;
          f32x4 v = acc[i][j], g = acc[i + 1][j];
          if (p.bias) {
            if (use_bpre) { v += bpre[i]; g += bpre[i + 1]; }
            else { v += *reinterpret_cast<const f32x4*>(p.bias + n); g += *reinterpret_cast<const f32x4*>(p.bias + n + 16); }
          }
          f16x4 o;
#pragma unroll
          for (int t = 0; t < 4; ++t) o[t] = (f16)(v[t] * gelu_erf_f(g[t]));
          *reinterpret_cast<f16x4*>(p.y + (size_t)m * p.ldy + ((nb + i * 16) >> 1) + fq * 4) = o;
        }
        continue;
      }
    }
    const int b = p.bias2 ? m / p.HoWo : 0;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = nb + i * 16 + fq * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.splitk > 1) {
        *reinterpret_cast<f32x4*>(p.ws + ((size_t)z * p.M + m) * p.N + n) = v;
        continue;
      }
      if (p.bias) {
        if (p.bias_per_row) v += p.bias[m];
        else if (use_bpre) v += bpre[i];
        else v += *reinterpret_cast<const f32x4*>(p.bias + n);
      }
      if (p.bias2) v += *reinterpret_cast<const f32x4*>(p.bias2 + (size_t)b * p.ld_bias2 + n);
      if (p.act == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = silu_f(v[t]);
      } else if (p.act == 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = quick_gelu_f(v[t]);
      }
      v *= p.scale;
      if (p.res) {
        const f16x4 r = *reinterpret_cast<const f16x4*>(p.res + (size_t)m * p.ldres + n);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] += (float)r[t];
      }
      if (p.y32) {
        *reinterpret_cast<f32x4*>(p.y32 + (size_t)m * p.ldy + n) = v;
      } else {
        f16x4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = (f16)v[t];
        *reinterpret_cast<f16x4*>(p.y + (size_t)m * p.ldy + n) = o;
        if (p.stats_out) {
#pragma unroll
          for (int t = 0; t < 4; ++t) { const float f = (float)o[t]; ssum += f; ssq += f * f; }
        }
      }
    }
    if (p.stats_out && p.splitk == 1) {
      // the four lanes fq = 0..3 with this frow hold the strip's columns of row m between them
      ssum += __shfl_xor(ssum, 16, 64); ssq += __shfl_xor(ssq, 16, 64);
      ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
      if (fq == 0 && nb < p.N) reinterpret_cast<float2*>(p.stats_out)[(size_t)m * p.stats_ld + nb / TN] = make_float2(ssum, ssq);
    }
  }
}

// implicit-GEMM tiles: accumulator tile (i, j) of wave (wm, wn) is output row m0 + wm*TM + j*16 + frow
template <int NI, int MI, int TM, int TN, int NB = 1>
__device__ __forceinline__ void epilogue(const KP& p, f32x4 (&acc)[NI][MI], int m0, int n0, int wm, int wn, int frow, int fq, int z,
                                         const f32x4 (&bpre)[NI], bool use_bpre, char* scratch = nullptr,
                                         const float2* lnrow = nullptr) {
  int mrow[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + wm * TM + j * 16 + frow;
    mrow[j] = m < p.M ? m : -1;
  }
  epilogue_rows<NI, MI, TN, NB>(p, acc, mrow, n0 + wn * TN, fq, z, bpre, use_bpre, scratch, lnrow);
}

// GroupNorm partials, second half (KP::gn_out).  After epilogue_rows every wave's scratch starts with the per-column (sum, sumsq)
// of the rows it stored.  Called by ONE wave per TN-wide strip after a workgroup barrier: sums the NW waves that share the strip
// (scratch bases `base + w * stride`), then the gn_cpg columns of each group, and writes entry (image b, tile slot, group).
template <int TN, int NW>
__device__ __forceinline__ void gn_partials_finish(const KP& p, char* base, int stride, int lane, int b, int slot, int nb) {
  constexpr int GCH = (TN + 63) / 64;
  float2* tot = reinterpret_cast<float2*>(base) + TN;          // behind wave 0's own column sums
#pragma unroll
  for (int c = 0; c < GCH; ++c) {
    const int col = c * 64 + lane;
    if (col < TN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { const float2 v = reinterpret_cast<const float2*>(base + w * stride)[col]; s += v.x; q += v.y; }
      tot[col] = make_float2(s, q);
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int cpg = p.gn_cpg;
  const int g = (nb / cpg) + lane;                             // strips start on a group boundary (host-checked)
  if (lane * cpg < TN && nb + lane * cpg < p.N) {
    float s = 0.f, q = 0.f;
    for (int k = 0; k < cpg; ++k) { const float2 v = tot[lane * cpg + k]; s += v.x; q += v.y; }
    reinterpret_cast<float2*>(p.gn_out)[((size_t)b * p.gn_slots + slot) * p.gn_groups + g] = make_float2(s, q);
  }
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (observed, speed only), so give each
// XCD a contiguous run of tiles (bijective for any tile count): neighbouring tiles share a weight panel in L2.
__device__ __forceinline__ int xcd_remap(int wg, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// compile-time loop (the LDS offsets of the inline-asm fragment reads must be immediates)
template <int N, typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}
template <int OFF>
__device__ __forceinline__ void lds_read128(f16x8& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// a weight fragment as it sits in LDS: 16 bytes of fp16 or 8 bytes of fp8
template <int OFF>
__device__ __forceinline__ void lds_read_w(f16x8& dst, unsigned addr) { lds_read128<OFF>(dst, addr); }
template <int OFF>
__device__ __forceinline__ void lds_read_w(uint2& dst, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// halo-reuse 3x3 kernel family (conv_halo.hip): variant table + launcher used by conv_gemm's planner
struct HaloCfg { int ph, pw, bn; const char* name; };
extern const HaloCfg kHaloCfgs[];
extern const int kNumHaloCfgs;
int launch_halo(int variant, const KP2& kp, int count, int tiles_m, int tiles_n, hipStream_t stream);
int halo_ring_bytes(int variant);

}  // namespace sdeo
