// Implicit-GEMM convolution / GEMM for gfx950 (CDNA4) on v_mfma_f32_16x16x32_f16.
//
// Replaces, for the CNSD hot path, every F.conv2d / F.linear the reference issues through ATen
// (ResBlock convs `openaimodel.py:200-240`, Down/Upsample `:90-159`, SpatialTransformer proj_in/out
// and the CrossAttention / GEGLU Linear layers `attention.py:49-76,154-179`, VAE convs `model.py`).
//
// Design (MI355X-first, not a cuDNN/cuBLAS shape):
//  * both operands are K-contiguous: activations NHWC (channel fastest), weights KRSC; every MFMA
//    fragment is one 16-byte LDS read, every global access a 16-byte vector.
//  * the im2col matrix is never materialised: the A tile is gathered straight from the NHWC tensor,
//    one filter tap x BK channels per K-step; zero padding, stride 2 and the nearest-x2 Upsample are
//    folded into the gather address.
//  * weights are the MFMA "A" operand (rows = output channel), activations the "B" operand (cols =
//    output pixel): each lane then ends up with 4 consecutive output channels of one pixel, i.e. an
//    8-byte NHWC store, and bias / time-embedding / residual / SiLU / scale are applied in registers.
//  * main kernel (`conv_gemm_dma_kernel`): global_load_lds (LDS-DMA, 16 B per lane) into a ring of LDS
//    stages, counted s_waitcnt vmcnt(N) + raw s_barrier so 1-3 K-steps of loads stay in flight across
//    the barrier; padding rows read a zero page so every lane issues the same number of DMAs; the XOR
//    swizzle that makes the ds_read_b128 fragment reads bank-conflict free is applied to the per-lane
//    SOURCE address (the DMA destination is lane-linear) and to the fragment reads.
//  * fallback kernel (`conv_gemm_kernel`): register-staged double buffering, used when Cin is not a
//    multiple of 64 (first convs, hint block, reduced test configs).
//  * split-K over blockIdx.z for the weight-bandwidth-bound deep levels (M = 128..512, K up to 23k) and to
//    fill 256 CUs when M x N is small.
#include <stdlib.h>

#include <array>
#include <map>

#include <type_traits>
#include <utility>

#include "conv_inl.h"

namespace sdeo {

// ------------------------------------------------------------------------------------------------
// main kernel: LDS-DMA ring, Cin % 64 == 0
// ------------------------------------------------------------------------------------------------
// WS ("wave-specialised"): 8 waves, waves 0-3 only read fragments and issue MFMAs, waves 4-7 only issue the LDS-DMAs.
// An LDS-DMA instruction stalls its wave for ~60-180 clocks at issue (measured); with one
// workgroup per CU that stall sat in front of every K-step's MFMAs.  Same ring protocol, one barrier per K-step for
// both roles: the loaders retire step `it` (counted vmcnt) BEFORE barrier `it`, the MFMA waves read it AFTER; the
// loaders refill slot (it-1) % STAGES after barrier `it`, which every MFMA wave only reaches once its reads of step
// it-1 have fed its MFMAs.
// W8: the weights are fp8 (OCP e4m3fn, one byte per element, per-output-channel power-of-two scale applied in the epilogue):
// a weight-tile row is 64 bytes per K-step instead of 128, one DMA pass of the 256 loading threads covers 64 rows, and the
// MFMA waves widen each 8-byte fragment to fp16 in registers (v_cvt_scalef32_pk_f16_fp8, exact) right before its MFMAs.
// The activations, the accumulation and the epilogue are those of the fp16 kernel; what changes is the bytes streamed.
// KPB (K-steps per barrier): the ring still holds STAGES 64-deep steps and every step is issued and retired as before, but loaders
// and MFMA waves meet only once per GROUP of KPB steps: the loaders make a whole group visible with one counted wait + barrier and
// then refill the KPB slots of the previous group.  A small tile has 10 - 20 MFMAs per wave per step (160 - 320 clocks) against a
// barrier / wait skeleton of ~240 clocks per step; the halo kernel gained 14 - 35 % from the same change
// (conv_halo.hip TPB).  Needs STAGES >= 2 KPB; STAGES - 2 KPB steps of loads stay in flight across a barrier.
// MX: both operands are block-scaled fp8 (KP::mx_sx): the loaders move the same 128-byte rows (128 codes instead of 64 halves: the
// address arithmetic is that of an fp16 problem of half the length), the MFMA waves read each lane's 32 consecutive codes as two
// 16-byte chunks and issue v_mfma_scale_f32_16x16x128_f8f6f4 with the two e8m0 block scales of the lane's row and 32-code block:
// 4x the K per MFMA at twice its cycles, half the operand bytes per FLOP.  GEMM (1x1) only, wave-specialised only.
template <int BM, int BN, int STAGES, bool UPS, bool WS, bool W8 = false, int KPB = 1, bool MX = false>
__global__ __launch_bounds__(WS ? 512 : 256) void conv_gemm_dma_kernel(const KP2 pp) {
  kernarg_warm<sizeof(KP2)>();
  // The scalars both roles need before their first DMA / fragment read, fetched in ONE batch and pinned in SGPRs: left to the
  // compiler each is an s_load + s_waitcnt lgkmcnt(0) next to its first use, ten dependent scalar round trips in front of the
  // first DMA of a launch whose whole K loop lasts a few microseconds.  (blockIdx.y is always 0: one problem per launch.)
  KP pl = pp.k[0];
  // (integers only: a pointer that has been through the asm loses its address space and every access through it becomes a flat_ one)
  asm volatile("" : "+s"(pl.M), "+s"(pl.N), "+s"(pl.K), "+s"(pl.Hi), "+s"(pl.Wi), "+s"(pl.Cin), "+s"(pl.Wo),
               "+s"(pl.S), "+s"(pl.stride), "+s"(pl.pad), "+s"(pl.HoWo), "+s"(pl.ldx), "+s"(pl.ldw), "+s"(pl.nk),
               "+s"(pl.nk_per_split), "+s"(pl.tiles_m), "+s"(pl.tiles_n), "+s"(pl.n_fastest));
  const KP& p = pl;
  static_assert(!MX || (WS && !W8 && !UPS && KPB == 1), "the block-scaled fp8 path exists for the plain wave-specialised GEMM");
  constexpr int BK = 64, RPP = 32;
  constexpr int WRPP = W8 ? 64 : 32;                           // weight rows per DMA pass
  constexpr int XP = BM / RPP, WP = (BN + WRPP - 1) / WRPP, L = XP + WP;     // DMA instructions per loading thread per stage
  constexpr int TM = BM / 2, TN = BN / 2, MI = TM / 16, NI = TN / 16;
  constexpr int XBYTES = BM * BK * 2, WBYTES = WP * 4096, STAGE = XBYTES + WBYTES;
  constexpr int PF = STAGES - KPB;                              // K-steps of loads issued ahead of the compute (initial fill)
  constexpr int FLY = STAGES - 2 * KPB;                         // steps of loads that stay in flight across a barrier
  static_assert(KPB >= 1 && FLY >= 0 && PF >= 1 && STAGES <= 10, "ring depth");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_load = !WS || wave_all >= 4;                   // wave-uniform roles
  const bool do_mma = !WS || wave_all < 4;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // indices inside the role
  const int wm = wave & 1, wn = wave >> 1;
  int m0, n0;
  auto set_tile = [&](int t) {             // t: position in the (XCD-aware) tile order
    const int tile = xcd_remap(t, p.tiles_m * p.tiles_n);
    const int tm = p.n_fastest ? tile / p.tiles_n : tile % p.tiles_m;
    const int tn = p.n_fastest ? tile % p.tiles_n : tile / p.tiles_m;
    m0 = tm * BM; n0 = tn * BN;
  };
  set_tile(blockIdx.x);
  const int z = blockIdx.z;
  const int kbeg = z * p.nk_per_split;
  const int kend = min(p.nk, kbeg + p.nk_per_split);
  const int nk = kend - kbeg;

  stamp(p, 0);
  const int chunk = tid & 7, lrow = tid >> 3;
  const int cl = chunk ^ ((lrow >> 1) & 7);        // logical k-chunk this lane fetches into its (linear) LDS slot
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // provably uniform: the DMA's LDS base stays in SGPRs

  // Gather state, all hoisted out of the K loop (the loop body must stay MFMA-bound, not address-bound):
  //   xrow[i]  address of this row's tap (0,0) pixel at channel chunk cl (may lie outside the tensor for border
  //            rows: it is only dereferenced when the tap's bit in vmask[i] is set)
  //   vmask[i] bit r*S+s = tap (r,s) of row i reads inside the image (zero padding otherwise)
  const char* xrow[XP];
  unsigned vmask[XP];
  int pixbase[UPS ? XP : 1], hb[UPS ? XP : 1], wb[UPS ? XP : 1];
  const char* wptr[WP];
  int winc[WP];
  const int tapsteps = p.Cin >> 6;
  int st_c, st_r, st_s;                                                                         // next K-step to issue
  auto loader_setup = [&]() {              // gather state of tile (m0, n0)
    st_c = kbeg % tapsteps; st_r = (kbeg / tapsteps) / p.S; st_s = (kbeg / tapsteps) % p.S;
    const int Hv = UPS ? 2 * p.Hi : p.Hi, Wv = UPS ? 2 * p.Wi : p.Wi;
    const int R = p.K / (p.S * p.Cin);
    const bool linear = !UPS && p.K == p.Cin && p.stride == 1 && p.pad == 0;   // Linear / conv1x1: row m IS pixel m
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int m = m0 + lrow + i * RPP;
      const bool mv = m < p.M;
      if (linear) {
        vmask[i] = (mv && !dbg_on(p, 1)) ? 1u : 0u;
        xrow[i] = reinterpret_cast<const char*>(p.x) + ((long)m * p.ldx + cl * 8) * 2;
        continue;
      }
      const int mm = mv ? m : 0;
      const int b = mm / p.HoWo;
      const int rem = mm - b * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride - p.pad;
      // taps inside the image: rows rlo..rhi-1, columns slo..shi-1
      const int rlo = max(0, -h0), rhi = min(R, Hv - h0), slo = max(0, -w0), shi = min(p.S, Wv - w0);
      const unsigned sm = (mv && shi > slo && !dbg_on(p, 1)) ? (((1u << shi) - 1u) & ~((1u << slo) - 1u)) : 0u;
      unsigned vm = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r)          // R <= 4 (checked on the host): straight-line selects instead of a divergent loop
        if (r >= rlo && r < rhi) vm |= sm << (r * p.S);
      vmask[i] = vm;
      if (UPS) {
        pixbase[i] = b * p.Hi * p.Wi; hb[i] = h0; wb[i] = w0;
        xrow[i] = nullptr;
      } else {
        xrow[i] = reinterpret_cast<const char*>(p.x) + ((long)(b * p.Hi * p.Wi + h0 * p.Wi + w0) * p.ldx + cl * 8) * 2;
      }
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      if constexpr (!W8) {
        const int n = n0 + lrow + i * RPP;
        const bool nv = n < p.N && !dbg_on(p, 2);
        wptr[i] = nv ? reinterpret_cast<const char*>(p.w + (size_t)n * p.ldw + (size_t)kbeg * BK + cl * 8) : zero;
        winc[i] = nv ? BK * 2 : 0;
      } else {
        // 64-byte rows: thread (row = tid >> 2, 16-byte chunk c = tid & 3) of each pass; the LDS image rotates the four chunks
        // of row r by (r >> 2) & 3 (source-side, the DMA destination is lane-linear) so that the 8-byte fragment reads of a
        // 32-lane half (16 rows x 2 k-groups) cover all 64 banks once
        const int r8 = (tid >> 2) + i * 64, c = tid & 3;
        const int n = n0 + r8;
        const bool nv = r8 < BN && n < p.N && !dbg_on(p, 2);
        const int csrc = (c - ((r8 >> 2) & 3)) & 3;
        wptr[i] = nv ? reinterpret_cast<const char*>(p.w) + (size_t)n * p.ldw + (size_t)kbeg * BK + csrc * 16 : zero;
        winc[i] = nv ? BK : 0;
      }
    }
  };
  if (do_load) loader_setup();

  // issue the DMAs of the NEXT K-step into ring slot `slot` (every loading lane issues exactly L of them)
  auto issue = [&](int slot) {
    const int tapbit = st_r * p.S + st_s;
    char* xs = smem + slot * STAGE + wave_u * 1024;
    char* wsm = xs + XBYTES;
    if (!UPS) {
      const long toff = ((long)(st_r * p.Wi + st_s) * p.ldx + (st_c << 6)) * 2;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const char* src = ((vmask[i] >> tapbit) & 1) ? xrow[i] + toff : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xs + i * 4096), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        const int hi = (hb[i] + st_r) >> 1, wi = (wb[i] + st_s) >> 1;
        const char* src = ((vmask[i] >> tapbit) & 1)
                              ? reinterpret_cast<const char*>(p.x) + ((long)(pixbase[i] + hi * p.Wi + wi) * p.ldx + (st_c << 6) + cl * 8) * 2
                              : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xs + i * 4096), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wptr[i],
                                       (__attribute__((address_space(3))) void*)(wsm + i * 4096), 16, 0, 0);
      wptr[i] += winc[i];
    }
    if (++st_c == tapsteps) {
      st_c = 0;
      if (++st_s == p.S) { st_s = 0; ++st_r; }
    }
  };
  // retire the DMAs of group g (K-steps g KPB .. g KPB + KPB - 1): all but the (newer) steps still allowed in flight.  Steps are
  // issued in order: PF in the prologue, then KPB more after every barrier, so behind group g at most FLY steps are younger.
  auto retire = [&](int g) {
    const int ahead = max(0, min(FLY, nk - (g + 1) * KPB));
    static_for<FLY + 1>([&](auto A) {
      if (ahead == A.value) wait_vmcnt<A.value * L>();
    });
  };
  // after barrier g every MFMA wave holds group g - 1 in registers: refill its slots (group 0: the KPB slots not used yet)
  auto refill = [&](int g) {
#pragma unroll
    for (int j = 0; j < KPB; ++j) {
      const int step = PF + g * KPB + j;
      if (step < nk && !dbg_on(p, 8)) issue(step % STAGES);
    }
  };
  const int ngroups = (nk + KPB - 1) / KPB;

  if constexpr (WS) {
    if (!do_mma) {                         // ---- loader waves
      if (nk > 0) {
#pragma unroll
        for (int s = 0; s < PF; ++s)
          if (s < nk) issue(s);
        for (int g = 0; g < ngroups; ++g) {
          retire(g);
          __builtin_amdgcn_s_barrier();
          refill(g);
        }
      }
      __builtin_amdgcn_s_barrier();          // matches the MFMA waves' pre-epilogue barrier (LDS becomes epilogue scratch)
      return;
    }
  }

  const int frow = lane & 15, fq = lane >> 4;
  constexpr int g0 = 0;
  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LayerNorm folded into this GEMM: (rstd, rstd * mean) of the tile's rows from the producer's per-strip partials, summed by
  // the MFMA waves of column 0 while the first K-step is in flight (they would wait on the first barrier otherwise) and parked
  // in LDS behind the ring; the epilogue reads two floats per row instead of re-summing up to N/32 partials per lane.
  float2* lnsm = reinterpret_cast<float2*>(smem + STAGES * STAGE);
  const bool ln_lds = p.ln_stats != nullptr && p.splitk == 1;
  if (ln_lds && wn == 0) {
    for (int rr = lane; rr < TM; rr += 64) {
      const int m = m0 + wm * TM + rr;
      float r = 0.f, rm = 0.f;
      if (m < p.M) ln_row_scalars(p, m, r, rm);
      lnsm[wm * TM + rr] = make_float2(r, rm);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // written before this wave's next barrier, read after the pre-epilogue one
  }

  // the bias is needed only by the epilogue: fetch it now (issued before, hence older than, every DMA) so the
  // epilogue does not start with a dependent global round trip
  f32x4 bpre[NI];
  const bool use_bpre = p.bias && !p.bias_per_row && p.splitk == 1;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = n0 + wn * TN + i * 16 + fq * 4;
    bpre[i] = (use_bpre && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // One K-step of MFMAs.  hipcc waits lgkmcnt(0) before the first MFMA after ANY group of LDS reads in this kernel (it
  // never emits a partial count here), and left alone it keeps a minimal fragment set: four full LDS round trips per
  // K-step in front of ~20 MFMAs (measured 700 clocks per step for 320 clocks of MFMA).  So on tiles with room for two
  // fragment sets the reads are inline asm (invisible to the compiler's wait insertion) with hand-counted waits: all reads
  // of the step are issued, lgkmcnt(NI+MI) releases the 32-deep half 0 (LDS returns in order), its MFMAs run while half 1
  // lands.  sched_barrier(0) after each wait keeps hipcc from hoisting MFMAs above it.
  constexpr bool DB = (NI + MI) * 8 + NI * MI * 4 <= 200;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int swl = (frow >> 1) & 7;                      // == (row >> 1) & 7: every fragment row is frow + multiple of 16
  const unsigned xa0 = lds0 + (wm * TM + frow) * 128 + ((fq ^ swl) << 4), xa1 = lds0 + (wm * TM + frow) * 128 + (((4 + fq) ^ swl) << 4);
  // weight fragment addresses: fp16 rows of 128 bytes (chunk swizzle as for the activations) or fp8 rows of 64 bytes, 8-byte
  // k-group g = kk * 4 + fq stored at group (g + 2 ((row >> 2) & 3)) & 7
  const int g8 = (fq + 2 * ((frow >> 2) & 3)) & 7;
  const unsigned wa0 = W8 ? lds0 + XBYTES + (wn * TN + frow) * 64 + g8 * 8 : xa0 + XBYTES + (wn * TN - wm * TM) * 128;
  const unsigned wa1 = W8 ? lds0 + XBYTES + (wn * TN + frow) * 64 + (g8 ^ 4) * 8 : xa1 + XBYTES + (wn * TN - wm * TM) * 128;
  constexpr int WSTEP = W8 ? 1024 : 2048;               // bytes between fragment rows i and i + 1 (16 tile rows)
  typedef std::conditional_t<W8, uint2, f16x8> wfrag;   // a weight fragment as it sits in LDS
  auto widen = [](const wfrag& w) -> f16x8 {
    if constexpr (W8) {
      const f16x2 a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.x, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.x, 1.0f, true);
      const f16x2 c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.y, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w.y, 1.0f, true);
      return f16x8{a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
    } else {
      return w;
    }
  };
  auto read_half = [&](const char* xs, const char* wsm, int kk, wfrag (&wf)[NI], f16x8 (&xf)[MI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = wn * TN + i * 16 + frow;
      if constexpr (W8) wf[i] = *reinterpret_cast<const uint2*>(wsm + row * 64 + ((kk * 4 + fq + 2 * ((row >> 2) & 3)) & 7) * 8);
      else wf[i] = *reinterpret_cast<const f16x8*>(wsm + row * 128 + swz_chunk<64>(row, kk * 4 + fq) * 16);
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int row = wm * TM + j * 16 + frow;
      xf[j] = *reinterpret_cast<const f16x8*>(xs + row * 128 + swz_chunk<64>(row, kk * 4 + fq) * 16);
    }
  };
  auto mma_half = [&](const wfrag (&wf)[NI], const f16x8 (&xf)[MI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const f16x8 w = widen(wf[i]);
#pragma unroll
      for (int j = 0; j < MI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xf[j], acc[i][j], 0, 0, 0);
    }
  };
  auto compute = [&](int it) {
    if constexpr (DB) {
      const unsigned sb = ((g0 + it) % STAGES) * STAGE;
      wfrag wf0[NI], wf1[NI];
      f16x8 xf0[MI], xf1[MI];
      static_for<NI>([&](auto I) { lds_read_w<I.value * WSTEP>(wf0[I.value], wa0 + sb); });
      static_for<MI>([&](auto J) { lds_read128<J.value * 2048>(xf0[J.value], xa0 + sb); });
      static_for<NI>([&](auto I) { lds_read_w<I.value * WSTEP>(wf1[I.value], wa1 + sb); });
      static_for<MI>([&](auto J) { lds_read128<J.value * 2048>(xf1[J.value], xa1 + sb); });
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NI + MI) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      mma_half(wf0, xf0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mma_half(wf1, xf1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      const char* xs = smem + ((g0 + it) % STAGES) * STAGE;
      const char* wsm = xs + XBYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        wfrag wf[NI];
        f16x8 xf[MI];
        read_half(xs, wsm, kk, wf, xf);
        __builtin_amdgcn_sched_barrier(0);
        mma_half(wf, xf);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if (nk > 0) {
    if constexpr (WS) {                    // ---- MFMA waves
      // retire every scalar (kernel-argument) load here: while one may be pending the compiler's wait-count pass has to
      // assume out-of-order LGKM returns and turns each partial lgkmcnt(N) in the loop into lgkmcnt(0)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if constexpr (MX) {
        // Two fragment register sets by step parity.  Operand map of the instruction (found with exact integer data, tools/mx_probe*.py:
        // CK's "32 consecutive k per lane" does not describe it): the 128-deep step is two 64-deep halves, lane (r = l & 15, g = l >> 4)
        // holds k = 16 g .. 16 g + 15 of EACH half (bytes 0-15: first half, bytes 16-31: second half) -- chunks g and 4 + g of the 128-byte
        // LDS row, the very chunks of the fp16 fragments -- while the scale it supplies belongs to block g = k in [32 g, 32 g + 32).
        // The reads of step it + 1 are issued before the MFMAs of step it.
        typedef int v8i __attribute__((ext_vector_type(8)));
        typedef int v4i __attribute__((ext_vector_type(4)));
        const unsigned xm0 = lds0 + (wm * TM + frow) * 128 + ((fq ^ swl) << 4), xm1 = lds0 + (wm * TM + frow) * 128 + (((4 + fq) ^ swl) << 4);
        const unsigned wm0 = xm0 + XBYTES + (wn * TN - wm * TM) * 128, wm1 = xm1 + XBYTES + (wn * TN - wm * TM) * 128;
        f16x8 wlo[NI], whi[NI], xlo[MI], xhi[MI];
        int swv[NI], sxv[MI];
        // scale bytes of this lane's rows: block index = 4 * (K-step) + fq; rows past the problem read row 0 (their results are dropped)
        const unsigned char* swp[NI];
        const unsigned char* sxp[MI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int n = n0 + wn * TN + i * 16 + frow;
          swp[i] = p.mx_sw + (size_t)(n < p.N ? n : 0) * p.mx_ldsw + kbeg * 4 + fq;
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const int m = m0 + wm * TM + j * 16 + frow;
          sxp[j] = p.mx_sx + (size_t)(m < p.M ? m : 0) * p.mx_ldsx + kbeg * 4 + fq;
        }
        auto rd = [&](unsigned sb) {
          static_for<NI>([&](auto I) { lds_read128<I.value * 2048>(wlo[I.value], wm0 + sb); lds_read128<I.value * 2048>(whi[I.value], wm1 + sb); });
          static_for<MI>([&](auto J) { lds_read128<J.value * 2048>(xlo[J.value], xm0 + sb); lds_read128<J.value * 2048>(xhi[J.value], xm1 + sb); });
        };
        auto cat = [](const f16x8& lo, const f16x8& hi) -> v8i {
          const v4i a = __builtin_bit_cast(v4i, lo), b = __builtin_bit_cast(v4i, hi);
          return v8i{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        };
        // ONE fragment set (two spilled at the 256-register cap of an 8-wave workgroup): an MFMA reads its operands at issue, so the
        // reads of step it + 1 are issued right behind the last MFMA of step it and land while the matrix pipe works through them
        __builtin_amdgcn_s_barrier();        // step 0 visible
        rd(0u);
        for (int it = 0; it < nk; ++it) {
          int swn[NI], sxn[MI];
#pragma unroll
          for (int i = 0; i < NI; ++i) swv[i] = swp[i][it * 4];
#pragma unroll
          for (int j = 0; j < MI; ++j) sxv[j] = sxp[j][it * 4];
          (void)swn; (void)sxn;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const v8i w = cat(wlo[i], whi[i]);
#pragma unroll
            for (int j = 0; j < MI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, cat(xlo[j], xhi[j]), acc[i][j], 0, 0, 0, swv[i], 0, sxv[j]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (it + 1 < nk) {
            __builtin_amdgcn_s_barrier();    // step it+1 visible; every MFMA wave has issued (= read the operands of) step `it`
            rd(((it + 1) % STAGES) * STAGE);
          }
        }
      } else if constexpr (DB) {
        // Rotated loop: barrier `it+1` and the half-0 reads of step it+1 are issued BEFORE the half-1 MFMAs of step it,
        // the half-1 reads of step it+1 right after them, so every LDS round trip and the barrier hide under 32-deep
        // halves of MFMAs.  A wave reaches barrier it+1 only after lgkmcnt(0), i.e. with all its reads of step `it` in
        // registers, which is what allows the loaders to refill that slot.  Register sets are rewritten only after the
        // MFMAs reading them have issued (an LDS return is >60 clocks away, an MFMA reads its operands at issue).
        wfrag wf0[NI], wf1[NI];
        f16x8 xf0[MI], xf1[MI];
        auto reads0 = [&](unsigned sb) {
          if (dbg_on(p, 16)) return;
          static_for<NI>([&](auto I) { lds_read_w<I.value * WSTEP>(wf0[I.value], wa0 + sb); });
          static_for<MI>([&](auto J) { lds_read128<J.value * 2048>(xf0[J.value], xa0 + sb); });
        };
        auto reads1 = [&](unsigned sb) {
          if (dbg_on(p, 16)) return;
          static_for<NI>([&](auto I) { lds_read_w<I.value * WSTEP>(wf1[I.value], wa1 + sb); });
          static_for<MI>([&](auto J) { lds_read128<J.value * 2048>(xf1[J.value], xa1 + sb); });
        };
        stamp(p, 1);
        __builtin_amdgcn_s_barrier();        // step 0 visible
        stamp(p, 2);
        reads0((g0 % STAGES) * STAGE);
        reads1((g0 % STAGES) * STAGE);
        int ing = 1;                         // position of step it + 1 inside its group (KPB = 1: always 0 = opens a group)
        for (int it = 0; it < nk; ++it) {
          const bool more = it + 1 < nk;
          if (ing == KPB) ing = 0;
          const bool newgroup = KPB == 1 || ing == 0;      // step it + 1 opens a group: meet the loaders before reading it
          const unsigned sbn = ((g0 + it + 1) % STAGES) * STAGE;
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NI + MI) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          if (!dbg_on(p, 4)) mma_half(wf0, xf0);
          __builtin_amdgcn_sched_barrier(0);
          if (newgroup || !more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
              __builtin_amdgcn_s_barrier();  // group of step it+1 visible; every MFMA wave holds step `it` in registers
              reads0(sbn);
            }
          } else {
            reads0(sbn);                     // same group, already visible: the half-1 fragments of step `it` only have to be in
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NI + MI) : "memory");     // registers (older than the NI + MI reads just issued)
          }
          __builtin_amdgcn_sched_barrier(0);
          if (!dbg_on(p, 4)) mma_half(wf1, xf1);
          __builtin_amdgcn_sched_barrier(0);
          if (more) reads1(sbn);
          ++ing;
        }
      } else {
        for (int it = 0; it < nk; ++it) {
          if (it % KPB == 0) __builtin_amdgcn_s_barrier();      // group of step `it` visible (its loaders retired it before arriving here)
          if (!dbg_on(p, 4)) compute(it);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (s < nk) issue(s);
      for (int it = 0; it < nk; ++it) {
        if (it % KPB == 0) {
          retire(it / KPB);
          __builtin_amdgcn_s_barrier();    // group of step `it` visible to every wave; everyone is done reading the previous group's slots
          refill(it / KPB);
        }
        compute(it);
      }
    }
  }
  // every wave is done reading fragments and no DMA is pending (the last K-step was retired with vmcnt(0)): the ring becomes
  // the waves' private epilogue scratch
  KP pe = p;                               // the epilogue's scalars in one batch of loads, in flight across the barrier
  pin_epilogue_scalars(pe);
  __builtin_amdgcn_s_barrier();
  stamp(p, 3);
  if (dbg_on(p, 32)) return;
  static_assert(4 * epilogue_scratch_bytes(TN) <= STAGES * STAGE, "epilogue scratch");
  constexpr int SCR = (STAGES * STAGE / 4) & ~15;        // LDS each wave may use as epilogue scratch
  constexpr int NBLK = epilogue_blocks(TN, MI, SCR);
  char* scratch = smem + wave * SCR;
  epilogue<NI, MI, TM, TN, NBLK>(pe, acc, m0, n0, wm, wn, frow, fq, z, bpre, use_bpre, scratch, ln_lds ? lnsm + wm * TM : nullptr);
  if (pe.gn_out) {                         // GroupNorm partials of this tile (loader waves, if any, are gone: the barrier counts the rest)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 0) {                         // waves 2 wn and 2 wn + 1 share strip wn
      const int bimg = m0 / pe.HoWo;
      gn_partials_finish<TN, 2>(pe, smem + 2 * wn * SCR, SCR, lane, bimg, (m0 - bimg * pe.HoWo) / BM, n0 + wn * TN);
    }
  }
  if (dbg_on(p, 64)) {
    stamp(p, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(p, 5);
  }
}

// ------------------------------------------------------------------------------------------------
// fallback kernel: register-staged double buffer (any Cin % 8 == 0)
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, bool GENERIC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const KP2 pp) {
  kernarg_warm<sizeof(KP2)>();
  const KP& p = pp.k[blockIdx.y];
  constexpr int CPR = BK / 8;         // 16-byte chunks per tile row
  constexpr int RPP = 256 / CPR;      // tile rows covered per pass of the 256 threads
  constexpr int XP = BM / RPP;        // passes for the activation tile
  constexpr int WP = BN / RPP;        // passes for the weight tile
  constexpr int TM = BM / 2, TN = BN / 2;   // wave tile (2x2 waves)
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2, STAGE = XBYTES + WBYTES;
  static_assert(XP >= 1 && WP >= 1, "tile too small for 256 threads");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  const int tile = blockIdx.x;
  const int tm = tile % p.tiles_m, tn = tile / p.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int kbeg = z * p.nk_per_split;
  const int kend = min(p.nk, kbeg + p.nk_per_split);

  const int chunk = tid % CPR;
  const int lrow = tid / CPR;

  int pixbase[XP], hb[XP], wb[XP];
  bool mvalid[XP];
  const int Hv = p.ups ? 2 * p.Hi : p.Hi, Wv = p.ups ? 2 * p.Wi : p.Wi;
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int m = m0 + lrow + i * RPP;
    mvalid[i] = m < p.M;
    const int mm = mvalid[i] ? m : 0;
    const int b = mm / p.HoWo;
    const int rem = mm - b * p.HoWo;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    pixbase[i] = b * p.Hi * p.Wi;
    hb[i] = ho * p.stride - p.pad;
    wb[i] = wo * p.stride - p.pad;
  }
  const f16* wrow[WP];
  bool nvalid[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int n = n0 + lrow + i * RPP;
    nvalid[i] = n < p.N;
    wrow[i] = p.w + (size_t)(nvalid[i] ? n : 0) * p.ldw + chunk * 8;
  }

  uint4 xr[XP], wr[WP];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  auto load_tile = [&](int kt) {
    if (!GENERIC) {
      const int tapsteps = p.Cin / BK;
      const int tap = kt / tapsteps;
      const int c0 = (kt - tap * tapsteps) * BK + chunk * 8;
      const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        int hi = hb[i] + r, wi = wb[i] + s;
        const bool ok = mvalid[i] && hi >= 0 && hi < Hv && wi >= 0 && wi < Wv;
        if (p.ups) { hi >>= 1; wi >>= 1; }
        const f16* src = p.x + (size_t)(pixbase[i] + hi * p.Wi + wi) * p.ldx + c0;
        xr[i] = ok ? *reinterpret_cast<const uint4*>(src) : zero4;
      }
#pragma unroll
      for (int i = 0; i < WP; ++i)
        wr[i] = nvalid[i] ? *reinterpret_cast<const uint4*>(wrow[i] + (size_t)kt * BK) : zero4;
    } else {
      const int kg = kt * BK + chunk * 8;
      const bool kok = kg < p.K;
      const int tap = kg / p.Cin;
      const int c0 = kg - tap * p.Cin;
      const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
      for (int i = 0; i < XP; ++i) {
        int hi = hb[i] + r, wi = wb[i] + s;
        const bool ok = kok && mvalid[i] && hi >= 0 && hi < Hv && wi >= 0 && wi < Wv;
        if (p.ups) { hi >>= 1; wi >>= 1; }
        const f16* src = p.x + (size_t)(pixbase[i] + hi * p.Wi + wi) * p.ldx + c0;
        xr[i] = ok ? *reinterpret_cast<const uint4*>(src) : zero4;
      }
#pragma unroll
      for (int i = 0; i < WP; ++i)
        wr[i] = (kok && nvalid[i]) ? *reinterpret_cast<const uint4*>(wrow[i] + (size_t)kt * BK) : zero4;
    }
  };
  auto store_tile = [&](int stage) {
    char* xs = smem + stage * STAGE;
    char* wsm = xs + XBYTES;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int row = lrow + i * RPP;
      *reinterpret_cast<uint4*>(xs + row * (BK * 2) + swz_chunk<BK>(row, chunk) * 16) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int row = lrow + i * RPP;
      *reinterpret_cast<uint4*>(wsm + row * (BK * 2) + swz_chunk<BK>(row, chunk) * 16) = wr[i];
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;

  if (kbeg < kend) {
    load_tile(kbeg);
    store_tile(0);
    __syncthreads();
    for (int kt = kbeg; kt < kend; ++kt) {
      const int cur = (kt - kbeg) & 1;
      const bool more = kt + 1 < kend;
      if (more) load_tile(kt + 1);
      const char* xs = smem + cur * STAGE;
      const char* wsm = xs + XBYTES;
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        f16x8 wf[NI], xf[MI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int row = wn * TN + i * 16 + frow;
          wf[i] = *reinterpret_cast<const f16x8*>(wsm + row * (BK * 2) + swz_chunk<BK>(row, kk * 4 + fq) * 16);
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const int row = wm * TM + j * 16 + frow;
          xf[j] = *reinterpret_cast<const f16x8*>(xs + row * (BK * 2) + swz_chunk<BK>(row, kk * 4 + fq) * 16);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
      if (more) store_tile(cur ^ 1);
      __syncthreads();
    }
  }
  f32x4 nob[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) nob[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  static_assert(4 * epilogue_scratch_bytes(TN) <= 2 * STAGE, "epilogue scratch");
  epilogue<NI, MI, TM, TN>(p, acc, m0, n0, wm, wn, frow, fq, z, nob, false, smem + wave * epilogue_scratch_bytes(TN));   // the K loop ends on a barrier
}

// split-K: sum the fp32 partial slabs and apply the epilogue. One thread per 4 output channels.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const KP2 pp) {
  kernarg_warm<sizeof(KP2)>();
  // every scalar the loads below depend on, in one batch (see conv_gemm_dma_kernel); blockIdx.y is always 0
  KP pl = pp.k[0];
  asm volatile("" : "+s"(pl.M), "+s"(pl.N), "+s"(pl.HoWo), "+s"(pl.ldy), "+s"(pl.ldres), "+s"(pl.ld_bias2), "+s"(pl.act),
               "+s"(pl.bias_per_row), "+s"(pl.splitk));     // integers only: pinned pointers would turn their accesses into flat_ ones
  const KP& p = pl;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N / 4;
  if (idx >= (int64_t)p.M * n4) return;
  const int m = (int)(idx / n4);
  const int n = (int)(idx - (int64_t)m * n4) * 4;
  // every load below is independent: issue them all before the first add (a dependent round trip costs ~0.5 us)
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* src = p.ws + (size_t)m * p.N + n;
  const size_t zs = (size_t)p.M * p.N;
  f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, b2 = bv;
  f16x4 rv = f16x4{0, 0, 0, 0};
  if (p.bias) {
    if (p.bias_per_row) bv += p.bias[m];
    else bv = *reinterpret_cast<const f32x4*>(p.bias + n);
  }
  if (p.bias2) b2 = *reinterpret_cast<const f32x4*>(p.bias2 + (size_t)(m / p.HoWo) * p.ld_bias2 + n);
  if (p.res) rv = *reinterpret_cast<const f16x4*>(p.res + (size_t)m * p.ldres + n);
  for (int z0 = 0; z0 < p.splitk; z0 += 8) {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      t[u] = (z0 + u < p.splitk) ? *reinterpret_cast<const f32x4*>(src + (size_t)(z0 + u) * zs) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  if (p.wscale) v *= *reinterpret_cast<const f32x4*>(p.wscale + n);       // fp8 weights: per-output-channel scale
  if (p.ln_stats) {                 // LayerNorm folded into the GEMM: see KP::ln_stats
    float r, rm;
    ln_row_scalars(p, m, r, rm);
    v = v * r - *reinterpret_cast<const f32x4*>(p.ln_s + n) * rm;
  }
  v += bv;
  v += b2;
  if (p.act == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = silu_f(v[t]);
  } else if (p.act == 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = quick_gelu_f(v[t]);
  }
  v *= p.scale;
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] += (float)rv[t];
  if (p.y32) {
    *reinterpret_cast<f32x4*>(p.y32 + (size_t)m * p.ldy + n) = v;
  } else {
    f16x4 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = (f16)v[t];
    *reinterpret_cast<f16x4*>(p.y + (size_t)m * p.ldy + n) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// host side: tile / split-K selection and launch
// ------------------------------------------------------------------------------------------------
enum TileKind { TK_DMA, TK_GENERIC, TK_HALO };
// TK_HALO (conv_halo.hip): bm = patch pixels, `stages` = index into kHaloCfgs, bk = 64
struct TileCfg { int bm, bn, bk, stages; TileKind kind; float weight; int wg_per_cu; const char* name; };
static const TileCfg kTiles[] = {
    {128, 128, 64, 3, TK_DMA, 1.00f, 1, "conv_gemm_dma_kernel<128,128,3>"},
    {128, 64, 64, 3, TK_DMA, 0.85f, 2, "conv_gemm_dma_kernel<128,64,3>"},
    {64, 64, 64, 4, TK_DMA, 0.65f, 2, "conv_gemm_dma_kernel<64,64,4>"},
    {128, 64, 32, 2, TK_GENERIC, 0.85f, 3, "conv_gemm_kernel<128,64,32,true>"},
    {64, 64, 32, 2, TK_GENERIC, 0.65f, 4, "conv_gemm_kernel<64,64,32,true>"},
    {256, 128, 64, 3, TK_DMA, 1.25f, 1, "conv_gemm_dma_kernel<256,128,3>"},
    // 160-wide weight panels: the channel counts of this model are multiples of 320, and what bounds these kernels is the
    // bytes a CU has to take in per K-step, so the tile menu is built to hit 256 equal workgroups
    {64, 160, 64, 3, TK_DMA, 0.90f, 1, "conv_gemm_dma_kernel<64,160,3>"},
    {128, 160, 64, 3, TK_DMA, 1.10f, 1, "conv_gemm_dma_kernel<128,160,3>"},
    {256, 160, 64, 3, TK_DMA, 1.30f, 1, "conv_gemm_dma_kernel<256,160,3>"},
    {32, 160, 64, 4, TK_DMA, 0.60f, 2, "conv_gemm_dma_kernel<32,160,4>"},
    {64, 160, 64, 5, TK_DMA, 0.90f, 1, "conv_gemm_dma_kernel<64,160,5>"},
    {128, 160, 64, 4, TK_DMA, 1.10f, 1, "conv_gemm_dma_kernel<128,160,4>"},
    {128, 64, 64, 5, TK_DMA, 0.85f, 1, "conv_gemm_dma_kernel<128,64,5>"},
    // halo-reuse 3x3 kernels (stride 1, pad 1, Cin % 64 == 0, image a multiple of the patch)
    {128, 80, 64, 0, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,80,4>"},
    {128, 160, 64, 1, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,160,4>"},
    {64, 80, 64, 2, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,8,80,4>"},
    {64, 160, 64, 3, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,8,160,4>"},
    {128, 64, 64, 4, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,64,4>"},
    {128, 128, 64, 5, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,128,4>"},
    // deep rings: a K-step is bound by bytes in flight per CU (a DMA lands ~1 us after issue under load), so small tiles get
    // as many stages as LDS holds
    {64, 64, 64, 8, TK_DMA, 0.65f, 1, "conv_gemm_dma_kernel<64,64,8>"},
    {32, 160, 64, 6, TK_DMA, 0.60f, 1, "conv_gemm_dma_kernel<32,160,6>"},
    {128, 128, 64, 4, TK_DMA, 1.00f, 1, "conv_gemm_dma_kernel<128,128,4>"},
    // halo kernels with two MFMA waves per SIMD
    {128, 80, 64, 6, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,80,8>"},
    {128, 160, 64, 7, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,160,8>"},
    {128, 64, 64, 8, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,64,8>"},
    {128, 128, 64, 9, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,128,8>"},
    // many-tile, short-K problems (ff.net.0.proj, q|k|v: thousands of tiles of 5 - 10 K-steps): a tile's prologue, K loop and
    // epilogue are each ~2 us and serial inside a workgroup (tools/stamps.py), so what overlaps them is OTHER workgroups on the
    // same CU.  Four-wave workgroups (no loader waves: half the register footprint per workgroup) on a two-slot ring fit three
    // or four to a CU.
    {128, 64, 64, 2, TK_DMA, 0.85f, 3, "conv_gemm_dma_kernel<128,64,2>"},
    {64, 64, 64, 2, TK_DMA, 0.65f, 4, "conv_gemm_dma_kernel<64,64,2>"},
    {128, 128, 64, 2, TK_DMA, 1.00f, 2, "conv_gemm_dma_kernel<128,128,2>"},
    {128, 160, 64, 2, TK_DMA, 1.10f, 2, "conv_gemm_dma_kernel<128,160,2>"},
    {64, 160, 64, 2, TK_DMA, 0.90f, 2, "conv_gemm_dma_kernel<64,160,2>"},
    {32, 160, 64, 2, TK_DMA, 0.60f, 3, "conv_gemm_dma_kernel<32,160,2>"},
    {64, 64, 64, 3, TK_DMA, 0.65f, 3, "conv_gemm_dma_kernel<64,64,3>"},
    {128, 64, 64, 3, TK_DMA, 0.85f, 2, "conv_gemm_dma_kernel<128,64,3>/4w"},
    // halo kernels with one barrier per filter row (three taps): indices 34..40 = kHaloCfgs 10..16
    {128, 80, 64, 10, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,80,4,3>"},
    {128, 80, 64, 11, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,80,8,3>"},
    {128, 64, 64, 12, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,64,4,3>"},
    {64, 80, 64, 13, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,8,80,4,3>"},
    {64, 160, 64, 14, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,8,160,4,3>"},
    {128, 128, 64, 15, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,128,8,3>"},
    {128, 64, 64, 16, TK_HALO, 1.0f, 1, "conv3x3_halo_kernel<8,16,64,8,3>"},
    // implicit-GEMM tiles with one barrier per GROUP of K-steps (conv_gemm_dma_kernel KPB): name<BM,BN,STAGES,KPB>
    {32, 160, 64, 6, TK_DMA, 0.60f, 1, "conv_gemm_dma_kernel<32,160,6,k2>"},
    {64, 64, 64, 8, TK_DMA, 0.65f, 1, "conv_gemm_dma_kernel<64,64,8,k2>"},
    {64, 64, 64, 9, TK_DMA, 0.65f, 1, "conv_gemm_dma_kernel<64,64,9,k3>"},
    {64, 160, 64, 5, TK_DMA, 0.90f, 1, "conv_gemm_dma_kernel<64,160,5,k2>"},
    {128, 64, 64, 6, TK_DMA, 0.85f, 1, "conv_gemm_dma_kernel<128,64,6,k2>"},
    {64, 64, 64, 6, TK_DMA, 0.65f, 1, "conv_gemm_dma_kernel<64,64,6,k2>"},
    {128, 128, 64, 4, TK_DMA, 1.00f, 1, "conv_gemm_dma_kernel<128,128,4,k2>"},
    {32, 160, 64, 5, TK_DMA, 0.60f, 1, "conv_gemm_dma_kernel<32,160,5,k2>"},
};
static const int kNumTiles = 49;
static bool tile_is_grouped(int t) { return t >= 41 && t <= 48; }
static bool tile_is_light(int t) { return t >= 26 && t <= 33; }     // four-wave (non-specialised) instantiations
static const int kNumCU = 256;

struct Plan { int tile; int splitk; int nk; int tiles_m, tiles_n; };

static bool is_fast(const ConvGemm& p) { return p.Cin % 64 == 0; }
static bool halo_ok(const ConvGemm& p, const TileCfg& c) {
  if (c.kind != TK_HALO) return false;
  const HaloCfg& h = kHaloCfgs[c.stages];
  return p.R == 3 && p.S == 3 && p.stride == 1 && p.pad == 1 && !p.ups && p.Cin % 64 == 0 && p.act != 3 && !p.bias_per_row &&
         p.Hi % h.ph == 0 && p.Wi % h.pw == 0 && p.Ho == p.Hi && p.Wo == p.Wi;
}
// tiles and K-steps of a plan: a halo tile is a patch of one image and steps through Cin in 64-channel slices (9 taps each)
static int plan_tiles_m(const ConvGemm& p, const TileCfg& c) {
  if (c.kind == TK_HALO) { const HaloCfg& h = kHaloCfgs[c.stages]; return p.B * (p.Hi / h.ph) * (p.Wi / h.pw); }
  return cdiv(p.M, c.bm);
}
static int plan_nk(const ConvGemm& p, const TileCfg& c) { return c.kind == TK_HALO ? p.Cin / 64 : cdiv(p.K, c.bk); }

// tuning hook (tools/tune_gemm.py): force the tile configuration / split-K factor of every following launch
static int g_force_tile = -1, g_force_splitk = 0, g_force_order = -1;
void conv_gemm_debug_force_order(int order) { g_force_order = order; }
void conv_gemm_debug_force(int tile, int splitk) { g_force_tile = tile; g_force_splitk = splitk; }

// plans measured on this device by conv_gemm_autotune (shape -> tile, split-K); consulted before the heuristic
typedef std::array<int, 10> ShapeKey;
static std::map<ShapeKey, std::pair<int, int>> g_tuned;
// (the `ups` slot doubles as the epilogue class: 2 = GEGLU pair epilogue, whose tile menu is restricted)
// (... and + 4 marks fp8 weights: a different kernel family with its own measurements)
static ShapeKey key_of(const ConvGemm& p) {
  return {p.M, p.N, p.K, p.Cin, p.R, p.stride, (p.act == 3 ? 2 : p.ups) + (p.wscale ? 4 : 0) + (p.mx_sx ? 8 : 0), p.Hi, p.Wi, p.B};
}
// tiles instantiated for block-scaled fp8 operands (the MFMA-bound GEMMs: many rows)
static bool tile_has_mx(int t) { return t == 0 || t == 1 || t == 6 || t == 21; }      // (<128,160,3> spills at the 256-register cap)
// tiles instantiated with fp8 weights (the weight-bound shapes: few rows, long K)
static bool tile_has_w8(int t) { return t == 1 || t == 2 || t == 6 || t == 9 || t == 19 || t == 20; }

static Plan make_plan(const ConvGemm& p) {
  Plan best{};
  const bool fast = is_fast(p);
  const int force_tile = p.force_tile >= 0 ? p.force_tile : g_force_tile;
  const int force_sk = p.force_splitk > 0 ? p.force_splitk : g_force_splitk;
  const bool pair = p.act == 3;     // GEGLU epilogue: needs an even number of 16-wide accumulator tiles per wave, no split-K
  if (force_tile < 0 && force_sk <= 0 && fast) {
    auto it = g_tuned.find(key_of(p));
    if (it != g_tuned.end() && !(pair && ((kTiles[it->second.first].bn / 2) % 32 != 0 || it->second.second != 1))) {
      const TileCfg& c = kTiles[it->second.first];
      if ((c.kind != TK_HALO || halo_ok(p, c)) && (!p.wscale || tile_has_w8(it->second.first)) && (!p.mx_sx || tile_has_mx(it->second.first)))
        return Plan{it->second.first, it->second.second, plan_nk(p, c), plan_tiles_m(p, c), cdiv(p.N, c.bn)};
    }
  }
  float best_t = 1e30f;
  for (int t = 0; t < kNumTiles; ++t) {
    const TileCfg& c = kTiles[t];
    auto usable = [&](int ti) {
      const TileCfg& cc = kTiles[ti];
      if (p.wscale && !tile_has_w8(ti)) return false;
      if (p.mx_sx && !tile_has_mx(ti)) return false;
      if (cc.kind == TK_HALO) return halo_ok(p, cc);
      return (cc.kind == TK_DMA) == fast && !(pair && (cc.bn / 2) % 32 != 0);
    };
    if (!usable(t)) continue;
    if (c.kind == TK_HALO && force_tile != t) continue;      // halo tiles enter through the measured plan table or a forced plan only
    if (tile_is_light(t) && force_tile != t) continue;       // so do the four-wave tiles
    if (tile_is_grouped(t) && force_tile != t) continue;     // and the grouped-barrier tiles
    if (force_tile >= 0 && usable(force_tile) && force_tile != t) continue;
    const int tmn = plan_tiles_m(p, c), tnn = cdiv(p.N, c.bn);
    const int tiles = tmn * tnn;
    const int nk = plan_nk(p, c);
    const bool halo = c.kind == TK_HALO;
    // candidate split-K factors: 1 and whatever gives every CU one or two workgroups
    int cands[3] = {1, 0, 0};
    int ncand = 1;
    if (tiles < kNumCU && (nk >= 8 || halo) && !pair) {
      for (int fill = 1; fill <= 2; ++fill) {
        int sk = fill * kNumCU / tiles;
        if (sk > nk / (halo ? 1 : 4)) sk = nk / (halo ? 1 : 4);
        if (sk > 16) sk = 16;
        if (sk >= 2 && sk != cands[ncand - 1]) cands[ncand++] = sk;
      }
    }
    if (force_sk > 0 && !pair) { cands[0] = force_sk > nk ? nk : force_sk; ncand = 1; }
    for (int ci = 0; ci < ncand; ++ci) {
      int sk = cands[ci];
      const int per = cdiv(nk, sk);
      sk = cdiv(nk, per);
      // time model in clocks: a CU takes in ~28 B/clk through the vector-memory path whatever
      // the source, so a K-step of a workgroup costs max((bm+bn) * 128 B / 28, MFMA time of the tile); the workgroups
      // are spread over 256 CUs; ~9k clocks of launch / prologue / epilogue per kernel
      const float rounds = (float)cdiv(tiles * sk, kNumCU);
      const float step_clk = halo ? fmaxf(((float)c.bm * 0.16f + c.bn) * 4.57f, (float)(c.bm * c.bn) / 31.8f)
                                  : fmaxf((float)(c.bm + c.bn) * 4.57f, (float)(c.bm * c.bn) / 31.8f);
      float tcost = 9000.f + rounds * ((float)per * (halo ? 9.f : 1.f) * step_clk + 1500.f);
      if (sk > 1) tcost += 8000.f + ((float)sk + 1.f) * p.M * p.N * 4.0f / 1700.f;   // reduce launch + fp32 slabs at ~4 TB/s
      if (tcost < best_t) { best_t = tcost; best = Plan{t, sk, nk, tmn, tnn}; }
    }
  }
  return best;
}

// block-scaled fp8 operands are addressed as an fp16 problem of half the length (KP::mx_sx): every plan query sees that view
static ConvGemm mx_view(const ConvGemm& p0) {
  ConvGemm p = p0;
  if (p.mx_sx) { p.K /= 2; p.Cin /= 2; p.ldx /= 2; p.ldw /= 2; }
  return p;
}

// columns per epilogue strip of a plan: the wave tile width of the implicit-GEMM kernels, the whole tile of the halo kernel
static int plan_tn(const Plan& pl) { return kTiles[pl.tile].kind == TK_HALO ? kTiles[pl.tile].bn : kTiles[pl.tile].bn / 2; }

int conv_gemm_stats_strips(const ConvGemm& p0) {
  const ConvGemm p = mx_view(p0);
  const Plan pl = make_plan(p);
  return pl.splitk == 1 ? cdiv(p.N, plan_tn(pl)) : 0;
}

bool conv_gemm_plan_is_halo(const ConvGemm& p) { return kTiles[make_plan(mx_view(p)).tile].kind == TK_HALO; }

// GroupNorm partials from the epilogue (KP::gn_out): entries per image the plan of p writes for groups of `cpg` channels, or 0 when it
// cannot (split-K, a strip that cuts a group, a tile that straddles two images, the register-staged fallback kernel, an epilogue
// that is not the LDS-transposed one)
bool conv_gemm_gn_in_ok(const ConvGemm& p) {
  if (p.mx_sx || p.Cin % 32 || p.R != 3) return false;
  const Plan pl = make_plan(p);
  const TileCfg& c = kTiles[pl.tile];
  if (c.kind != TK_HALO || !halo_ok(p, c)) return false;
  const int per = cdiv(pl.nk, pl.splitk);
  return per <= 32 && halo_ring_bytes(c.stages) + per * 512 <= 160 * 1024;      // (the MFMA waves build the table 2048 channels at a time)
}

static int gn_slots_of(const ConvGemm& p, int cpg);
int conv_gemm_gn_slots(const ConvGemm& p0, int cpg) { return gn_slots_of(mx_view(p0), cpg); }
static int gn_slots_of(const ConvGemm& p, int cpg) {       // p: the view the kernels see (mx_view)
  if (cpg <= 0 || p.N % cpg || p.act == 3 || p.y32 || !p.y || p.bias_per_row || p.stats_out) return 0;
  const Plan pl = make_plan(p);
  const TileCfg& c = kTiles[pl.tile];
  if (pl.splitk != 1 || c.kind == TK_GENERIC || plan_tn(pl) % cpg) return 0;
  if (p.N % 8 || p.ldy % 8 || (reinterpret_cast<uintptr_t>(p.y) & 15) || (p.res && (p.ldres % 8 || (reinterpret_cast<uintptr_t>(p.res) & 15)))) return 0;
  if (c.kind == TK_HALO) { const HaloCfg& h = kHaloCfgs[c.stages]; return (p.Hi / h.ph) * (p.Wi / h.pw); }
  const int hw = p.Ho * p.Wo;
  return hw % c.bm == 0 ? hw / c.bm : 0;
}

size_t conv_gemm_workspace_bytes(const ConvGemm& p0) {
  const ConvGemm p = mx_view(p0);
  const Plan pl = make_plan(p);
  return pl.splitk > 1 ? (size_t)pl.splitk * p.M * p.N * sizeof(float) : 0;
}

const char* conv_gemm_kernel_name(const ConvGemm& p) { return kTiles[make_plan(mx_view(p)).tile].name; }
int conv_gemm_plan_splitk(const ConvGemm& p) { return make_plan(mx_view(p)).splitk; }

template <typename K>
static int launch_k(K kernel, int smem, DeviceOnce* attr_done, const KP2& kp, int count, int tiles, hipStream_t stream, int threads = 256) {
  if (attr_done->need()) {
    SDEO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done->mark();
  }
  dim3 grid(tiles, count, kp.k[0].splitk);
  hipLaunchKernelGGL(kernel, grid, dim3(threads), smem, stream, kp);
  SDEO_HIP(hipGetLastError());
  return 0;
}

static int g_ws = -1;     // wave-specialised variant: -1 = SDEO_GEMM_WS env (default on)
static bool use_ws() {
  if (g_ws < 0) { const char* e = getenv("SDEO_GEMM_WS"); g_ws = e ? atoi(e) : 1; }
  return g_ws != 0;
}

// fp8-weight instantiations: wave-specialised only, no folded Upsample (the weight-bound convs have neither)
template <int BM, int BN, int ST>
static int launch_dma_w8(const KP2& kp, int count, int tiles, hipStream_t stream) {
  static DeviceOnce done;
  constexpr int smem = ST * (BM * 128 + (BN + 63) / 64 * 4096) + BM * 8;
  return launch_k(&conv_gemm_dma_kernel<BM, BN, ST, false, true, true>, smem, &done, kp, count, tiles, stream, 512);
}

template <int BM, int BN, int ST, int KPB = 1>
static int launch_dma(int ups, const KP2& kp, int count, int tiles, hipStream_t stream) {
  static DeviceOnce done[4];
  constexpr int smem = ST * (BM + BN) * 128 + BM * 8;      // ring + the LayerNorm row scalars (conv_gemm_dma_kernel: lnsm)
  static_assert(smem <= 160 * 1024, "LDS");
  if (use_ws() || KPB > 1) {
    return ups ? launch_k(&conv_gemm_dma_kernel<BM, BN, ST, true, true, false, KPB>, smem, &done[3], kp, count, tiles, stream, 512)
               : launch_k(&conv_gemm_dma_kernel<BM, BN, ST, false, true, false, KPB>, smem, &done[2], kp, count, tiles, stream, 512);
  }
  return ups ? launch_k(&conv_gemm_dma_kernel<BM, BN, ST, true, false>, smem, &done[1], kp, count, tiles, stream)
             : launch_k(&conv_gemm_dma_kernel<BM, BN, ST, false, false>, smem, &done[0], kp, count, tiles, stream);
}

// four-wave workgroups on a two-slot ring (tiles 26..28)
template <int BM, int BN, int ST>
static int launch_dma_light(int ups, const KP2& kp, int count, int tiles, hipStream_t stream) {
  static DeviceOnce done[2];
  constexpr int smem = ST * (BM + BN) * 128 + BM * 8;
  return ups ? launch_k(&conv_gemm_dma_kernel<BM, BN, ST, true, false>, smem, &done[1], kp, count, tiles, stream)
             : launch_k(&conv_gemm_dma_kernel<BM, BN, ST, false, false>, smem, &done[0], kp, count, tiles, stream);
}

// argument checks + plan + kernel parameters of one problem
static int prepare(const ConvGemm& p, Plan& pl, KP& kp) {
  SDEO_CHECK(p.x && p.w && (p.y || p.y32), "conv_gemm: null operand");
  SDEO_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "conv_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  SDEO_CHECK(p.N % 4 == 0, "conv_gemm: N=%d must be a multiple of 4", p.N);
  SDEO_CHECK(p.Cin % 8 == 0 && p.ldx % 8 == 0 && p.ldw % 8 == 0, "conv_gemm: Cin=%d ldx=%d ldw=%d must be multiples of 8",
             p.Cin, p.ldx, p.ldw);
  SDEO_CHECK(p.K == p.R * p.S * p.Cin, "conv_gemm: K=%d != R*S*Cin=%d", p.K, p.R * p.S * p.Cin);
  SDEO_CHECK(p.R >= 1 && p.R <= 4 && p.S >= 1 && p.S <= 4, "conv_gemm: filter %dx%d unsupported (1..4)", p.R, p.S);
  SDEO_CHECK(p.ldw >= p.K, "conv_gemm: ldw=%d < K=%d", p.ldw, p.K);
  SDEO_CHECK(p.M == p.B * p.Ho * p.Wo, "conv_gemm: M=%d != B*Ho*Wo=%d", p.M, p.B * p.Ho * p.Wo);
  SDEO_CHECK(p.ldy % 4 == 0 && p.ldy >= (p.act == 3 ? p.N / 2 : p.N), "conv_gemm: ldy=%d", p.ldy);
  SDEO_CHECK(!p.res || p.ldres % 4 == 0, "conv_gemm: ldres=%d", p.ldres);
  SDEO_CHECK((reinterpret_cast<uintptr_t>(p.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.w) & 15) == 0,
             "conv_gemm: operands must be 16-byte aligned");
  {
    const int Hv = p.ups ? 2 * p.Hi : p.Hi, Wv = p.ups ? 2 * p.Wi : p.Wi;
    SDEO_CHECK((Hv + 2 * p.pad - p.R) / p.stride + 1 == p.Ho && (Wv + 2 * p.pad - p.S) / p.stride + 1 == p.Wo,
               "conv_gemm: geometry mismatch Hi=%d Wi=%d -> Ho=%d Wo=%d (R=%d S=%d stride=%d pad=%d ups=%d)", p.Hi, p.Wi,
               p.Ho, p.Wo, p.R, p.S, p.stride, p.pad, p.ups);
  }
  if (p.act == 3)
    SDEO_CHECK(p.N % 32 == 0 && p.y && !p.y32 && !p.res && !p.bias2 && !p.bias_per_row && p.ldy >= p.N / 2 && p.scale == 1.0f,
               "conv_gemm: GEGLU epilogue needs N %% 32 == 0 (N=%d), fp16 output with ldy >= N/2 and no residual / bias2", p.N);
  if (p.wscale)
    SDEO_CHECK(p.Cin % 64 == 0 && !p.ups && p.ldw % 16 == 0 && !p.bias_per_row,
               "conv_gemm: fp8 weights need Cin %% 64 == 0 (Cin=%d), no folded upsample and 16-byte aligned rows (ldw=%d bytes)", p.Cin, p.ldw);
  pl = make_plan(p);
  SDEO_CHECK(!p.wscale || (tile_has_w8(pl.tile) && kTiles[pl.tile].kind == TK_DMA), "conv_gemm: no fp8-weight plan for this shape");
  SDEO_CHECK(!p.mx_sx || (tile_has_mx(pl.tile) && kTiles[pl.tile].kind == TK_DMA), "conv_gemm: no block-scaled fp8 plan for this shape");
  SDEO_CHECK(p.act != 3 || (pl.splitk == 1 && (kTiles[pl.tile].bn / 2) % 32 == 0), "conv_gemm: no GEGLU-capable plan");
  kp = KP{};
  kp.x = p.x; kp.w = p.w; kp.y = p.y; kp.y32 = p.y32; kp.bias = p.bias; kp.bias2 = p.bias2; kp.res = p.res;
  kp.ws = p.workspace;
  kp.M = p.M; kp.N = p.N; kp.K = p.K;
  kp.Hi = p.Hi; kp.Wi = p.Wi; kp.Cin = p.Cin; kp.Ho = p.Ho; kp.Wo = p.Wo; kp.S = p.S; kp.stride = p.stride; kp.pad = p.pad;
  kp.ups = p.ups; kp.HoWo = p.Ho * p.Wo;
  kp.ldx = p.ldx; kp.ldw = p.ldw; kp.ldy = p.ldy; kp.ldres = p.ldres; kp.ld_bias2 = p.ld_bias2;
  kp.act = p.act; kp.bias_per_row = p.bias_per_row; kp.scale = p.scale;
  kp.wscale = p.wscale;
  if (p.mx_sx) {       // (p arrives with K, Cin, ldx, ldw already halved: conv_gemm())
    SDEO_CHECK(p.mx_sw && p.R == 1 && p.S == 1 && p.stride == 1 && !p.ups && !p.wscale && p.Cin % 64 == 0 && !p.bias_per_row,
               "conv_gemm: block-scaled fp8 operands go with a plain GEMM whose K is a multiple of 128");
    SDEO_CHECK(p.mx_ldsx >= p.K / 16 && p.mx_ldsw >= p.K / 16, "conv_gemm: scale rows too short (%d, %d < %d)", p.mx_ldsx, p.mx_ldsw, p.K / 16);
    kp.mx_sx = p.mx_sx; kp.mx_sw = p.mx_sw; kp.mx_ldsx = p.mx_ldsx; kp.mx_ldsw = p.mx_ldsw;
  }
  if (p.ln_stats) {
    SDEO_CHECK(p.ln_s && p.ln_strips >= 1 && p.ln_ld >= p.ln_strips && p.ln_c > 0, "conv_gemm: incomplete LayerNorm-fold arguments");
    SDEO_CHECK(!p.bias_per_row && !p.y32, "conv_gemm: the LayerNorm fold applies to row-major fp16 products only");
    kp.ln_stats = p.ln_stats; kp.ln_s = p.ln_s; kp.ln_strips = p.ln_strips; kp.ln_ld = p.ln_ld;
    kp.ln_invc = 1.0f / (float)p.ln_c; kp.ln_eps = p.ln_eps;
  }
  if (p.stats_out) {
    const int strips = pl.splitk == 1 ? cdiv(p.N, plan_tn(pl)) : 0;
    SDEO_CHECK(strips >= 1 && p.stats_ld >= strips && p.y && !p.y32 && p.act != 3,
               "conv_gemm: row statistics need an unsplit fp16 plan (strips %d, stats_ld %d)", strips, p.stats_ld);
    kp.stats_out = p.stats_out; kp.stats_ld = p.stats_ld;
  }
  if (p.gn_in) {
    SDEO_CHECK(p.gn_gamma && p.gn_beta && p.gn_in_slots >= 1 && conv_gemm_gn_in_ok(p),
               "conv_gemm: GroupNorm of the input needs a halo-reuse 3x3 plan with LDS left for its table");
    kp.gn_in = p.gn_in; kp.gn_gamma = p.gn_gamma; kp.gn_beta = p.gn_beta; kp.gn_in_slots = p.gn_in_slots;
    kp.gn_in_cpg = p.Cin / 32; kp.gn_in_silu = p.gn_in_silu; kp.gn_in_eps = p.gn_in_eps;
    kp.gn_in_inv = 1.0f / ((float)(p.Cin / 32) * (float)p.Hi * (float)p.Wi);
  }
  if (p.gn_out) {
    const int slots = gn_slots_of(p, p.gn_cpg);
    SDEO_CHECK(slots > 0 && slots == p.gn_slots && p.gn_groups * p.gn_cpg == p.N,
               "conv_gemm: GroupNorm partials need an unsplit fp16 plan whose strips cover whole groups (slots %d, expected %d)", slots, p.gn_slots);
    kp.gn_out = p.gn_out; kp.gn_cpg = p.gn_cpg; kp.gn_slots = p.gn_slots; kp.gn_groups = p.gn_groups;
  }
  kp.nk = pl.nk; kp.splitk = pl.splitk; kp.nk_per_split = cdiv(pl.nk, pl.splitk);
  kp.tiles_m = pl.tiles_m; kp.tiles_n = pl.tiles_n;
  {
    // Each of the 8 XCDs has a private L2 and gets a contiguous run of tiles.  With N fastest every activation row is
    // fetched by one XCD but every XCD streams all the weights; with M fastest it is the other way round (each XCD
    // sees about tiles_n/8 weight panels and every activation row min(8, tiles_n) times).  Pick the cheaper fetch.
    const double xb = (double)p.M * p.Cin * 2.0 * (p.R * p.S > 1 ? 1.3 : 1.0), wb = (double)p.N * p.K * 2.0;
    const double cost_n_fast = xb + 8.0 * wb;
    const double cost_m_fast = (pl.tiles_n < 8 ? pl.tiles_n : 8) * xb + wb * (pl.tiles_n < 8 ? 1.6 : 1.0);
#ifdef SDEO_DEBUG_KERNELS
    { static const int dbg = [] { const char* e = getenv("SDEO_DBG_GEMM"); return e ? atoi(e) : 0; }(); kp.dbg = dbg; }
#endif
    kp.n_fastest = g_force_order >= 0 ? g_force_order : (cost_n_fast < cost_m_fast ? 1 : 0);
    static const int epi = [] { const char* e = getenv("SDEO_EPI_COALESCE"); return e ? atoi(e) : 1; }();
    kp.coalesce = epi && p.y && !p.y32 && pl.splitk == 1 && !p.bias_per_row && p.N % (p.act == 3 ? 16 : 8) == 0 && p.ldy % 8 == 0 &&
                  (reinterpret_cast<uintptr_t>(p.y) & 15) == 0 &&
                  (!p.res || (p.ldres % 8 == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0));
    SDEO_CHECK(!p.gn_out || kp.coalesce, "conv_gemm: GroupNorm partials need the LDS-transposed epilogue");
  }
  if (pl.splitk > 1) {
    const size_t need = (size_t)pl.splitk * p.M * p.N * sizeof(float);
    SDEO_CHECK(p.workspace && p.workspace_bytes >= need, "conv_gemm: split-K workspace too small (%zu < %zu)",
               p.workspace_bytes, need);
  }
  if (kTiles[pl.tile].kind == TK_HALO) SDEO_CHECK(halo_ok(p, kTiles[pl.tile]), "conv_gemm: halo plan on an ineligible problem");
  return 0;
}

// launch the problem(s) in kp at plan `pl` (count is always 1)
// block-scaled fp8 instantiations
template <int BM, int BN, int ST>
static int launch_dma_mx(const KP2& kp, int count, int tiles, hipStream_t stream) {
  static DeviceOnce done;
  constexpr int smem = ST * (BM + BN) * 128 + BM * 8;
  return launch_k(&conv_gemm_dma_kernel<BM, BN, ST, false, true, false, 1, true>, smem, &done, kp, count, tiles, stream, 512);
}

static int dispatch(const Plan& pl, int ups, bool w8, const KP2& kp, int count, hipStream_t stream) {
  const int tiles = pl.tiles_m * pl.tiles_n;
  int rc = 0;
  if (kp.k[0].mx_sx) {
    switch (pl.tile) {
      case 0: rc = launch_dma_mx<128, 128, 3>(kp, count, tiles, stream); break;
      case 1: rc = launch_dma_mx<128, 64, 3>(kp, count, tiles, stream); break;
      case 6: rc = launch_dma_mx<64, 160, 3>(kp, count, tiles, stream); break;
      case 21: rc = launch_dma_mx<128, 128, 4>(kp, count, tiles, stream); break;
      default: return fail("conv_gemm: tile %d has no block-scaled fp8 instantiation", pl.tile);
    }
  } else if (w8) {
    switch (pl.tile) {
      case 1: rc = launch_dma_w8<128, 64, 3>(kp, count, tiles, stream); break;
      case 2: rc = launch_dma_w8<64, 64, 4>(kp, count, tiles, stream); break;
      case 6: rc = launch_dma_w8<64, 160, 3>(kp, count, tiles, stream); break;
      case 9: rc = launch_dma_w8<32, 160, 4>(kp, count, tiles, stream); break;
      case 19: rc = launch_dma_w8<64, 64, 8>(kp, count, tiles, stream); break;
      case 20: rc = launch_dma_w8<32, 160, 6>(kp, count, tiles, stream); break;
      default: return fail("conv_gemm: tile %d has no fp8-weight instantiation", pl.tile);
    }
  } else
  switch (pl.tile) {
    case 0: rc = launch_dma<128, 128, 3>(ups, kp, count, tiles, stream); break;
    case 1: rc = launch_dma<128, 64, 3>(ups, kp, count, tiles, stream); break;
    case 2: rc = launch_dma<64, 64, 4>(ups, kp, count, tiles, stream); break;
    case 3: { static DeviceOnce d; rc = launch_k(&conv_gemm_kernel<128, 64, 32, true>, 2 * (128 + 64) * 64, &d, kp, count, tiles, stream); break; }
    case 4: { static DeviceOnce d; rc = launch_k(&conv_gemm_kernel<64, 64, 32, true>, 2 * (64 + 64) * 64, &d, kp, count, tiles, stream); break; }
    case 5: rc = launch_dma<256, 128, 3>(ups, kp, count, tiles, stream); break;
    case 6: rc = launch_dma<64, 160, 3>(ups, kp, count, tiles, stream); break;
    case 7: rc = launch_dma<128, 160, 3>(ups, kp, count, tiles, stream); break;
    case 8: rc = launch_dma<256, 160, 3>(ups, kp, count, tiles, stream); break;
    case 9: rc = launch_dma<32, 160, 4>(ups, kp, count, tiles, stream); break;
    case 10: rc = launch_dma<64, 160, 5>(ups, kp, count, tiles, stream); break;
    case 11: rc = launch_dma<128, 160, 4>(ups, kp, count, tiles, stream); break;
    case 12: rc = launch_dma<128, 64, 5>(ups, kp, count, tiles, stream); break;
    case 19: rc = launch_dma<64, 64, 8>(ups, kp, count, tiles, stream); break;
    case 20: rc = launch_dma<32, 160, 6>(ups, kp, count, tiles, stream); break;
    case 21: rc = launch_dma<128, 128, 4>(ups, kp, count, tiles, stream); break;
    case 26: rc = launch_dma_light<128, 64, 2>(ups, kp, count, tiles, stream); break;
    case 27: rc = launch_dma_light<64, 64, 2>(ups, kp, count, tiles, stream); break;
    case 28: rc = launch_dma_light<128, 128, 2>(ups, kp, count, tiles, stream); break;
    case 29: rc = launch_dma_light<128, 160, 2>(ups, kp, count, tiles, stream); break;
    case 30: rc = launch_dma_light<64, 160, 2>(ups, kp, count, tiles, stream); break;
    case 31: rc = launch_dma_light<32, 160, 2>(ups, kp, count, tiles, stream); break;
    case 32: rc = launch_dma_light<64, 64, 3>(ups, kp, count, tiles, stream); break;
    case 33: rc = launch_dma_light<128, 64, 3>(ups, kp, count, tiles, stream); break;
    case 41: rc = launch_dma<32, 160, 6, 2>(ups, kp, count, tiles, stream); break;
    case 42: rc = launch_dma<64, 64, 8, 2>(ups, kp, count, tiles, stream); break;
    case 43: rc = launch_dma<64, 64, 9, 3>(ups, kp, count, tiles, stream); break;
    case 44: rc = launch_dma<64, 160, 5, 2>(ups, kp, count, tiles, stream); break;
    case 45: rc = launch_dma<128, 64, 6, 2>(ups, kp, count, tiles, stream); break;
    case 46: rc = launch_dma<64, 64, 6, 2>(ups, kp, count, tiles, stream); break;
    case 47: rc = launch_dma<128, 128, 4, 2>(ups, kp, count, tiles, stream); break;
    case 48: rc = launch_dma<32, 160, 5, 2>(ups, kp, count, tiles, stream); break;
    case 13: case 14: case 15: case 16: case 17: case 18: case 22: case 23: case 24: case 25:
    case 34: case 35: case 36: case 37: case 38: case 39: case 40:
      rc = launch_halo(kTiles[pl.tile].stages, kp, count, pl.tiles_m, pl.tiles_n, stream);
      break;
    default: return fail("conv_gemm: bad tile %d", pl.tile);
  }
  if (rc) return rc;
  if (pl.splitk > 1) {
    const int64_t n = (int64_t)kp.k[0].M * (kp.k[0].N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(n, 256), count), dim3(256), 0, stream, kp);
    SDEO_HIP(hipGetLastError());
  }
  return 0;
}

int conv_gemm(const ConvGemm& p0, hipStream_t stream) {
  if (p0.mx_sx)        // fp8 codes addressed as an fp16 problem of half the length (KP::mx_sx)
    SDEO_CHECK(p0.K % 128 == 0 && p0.ldx % 16 == 0 && p0.ldw % 16 == 0 && p0.K == p0.Cin, "conv_gemm: block-scaled fp8 GEMM needs K %% 128 == 0 (K=%d, ldx=%d, ldw=%d)", p0.K, p0.ldx, p0.ldw);
  const ConvGemm p = mx_view(p0);
  Plan pl;
  KP2 kk{};
  if (int rc = prepare(p, pl, kk.k[0])) return rc;
  return dispatch(pl, p.ups, p.wscale != nullptr, kk, 1, stream);
}

int conv_gemm_read_stamps(unsigned long long* out, int n) {
  SDEO_HIP(hipDeviceSynchronize());
  SDEO_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)(n < kStampWGs * kStampSlots ? n : kStampWGs * kStampSlots)));
  return 0;
}

// tuned-plan table I/O (stablediffusioneo_amd/tuned_plans_gfx950.json is produced by tools/tune_plans.py on an MI355X and
// loaded at start-up, so production runs neither re-measure nor vary their plans from run to run)
void conv_gemm_set_tuned(const int key[10], int tile, int splitk) {
  ShapeKey k;
  for (int i = 0; i < 10; ++i) k[i] = key[i];
  if (tile >= 0 && tile < kNumTiles && kTiles[tile].kind != TK_GENERIC && splitk >= 1) g_tuned[k] = {tile, splitk};
}

std::string conv_gemm_tuned_json() {
  std::string out = "[";
  char buf[256];
  bool first = true;
  for (auto& kv : g_tuned) {
    snprintf(buf, sizeof(buf), "%s[%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d]", first ? "" : ",", kv.first[0], kv.first[1], kv.first[2],
             kv.first[3], kv.first[4], kv.first[5], kv.first[6], kv.first[7], kv.first[8], kv.first[9], kv.second.first,
             kv.second.second);
    out += buf;
    first = false;
  }
  return out + "]";
}

// ------------------------------------------------------------------------------------------------
// autotune: measure every (tile, split-K) candidate for one problem on the device and remember the fastest.
// Called once per distinct shape from sdeo_configure (never on the hot path, never under graph capture).
// ------------------------------------------------------------------------------------------------
static const size_t kTuneWorkspaceCap = (size_t)256 << 20;

size_t conv_gemm_autotune_workspace_bytes(const ConvGemm& p) {
  if (!is_fast(p)) return conv_gemm_workspace_bytes(p);
  const size_t per = (size_t)p.M * p.N * sizeof(float);
  size_t want = per * 16;
  if (want > kTuneWorkspaceCap) want = kTuneWorkspaceCap / per * per;
  const size_t heur = conv_gemm_workspace_bytes(p);
  return want > heur ? want : heur;
}

int conv_gemm_autotune(const ConvGemm& p, hipStream_t stream) {
  if (p.mx_sx) return 0;                                   // block-scaled fp8 GEMMs run on the heuristic plan among their few tiles
  if (!is_fast(p) || g_force_tile >= 0 || g_force_splitk > 0) return 0;
  const ShapeKey key = key_of(p);
  if (g_tuned.count(key)) return 0;
  static const int tiles[] = {0, 1, 2, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48};
  static const int sks[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20};
  hipEvent_t a, b;
  SDEO_HIP(hipEventCreate(&a));
  SDEO_HIP(hipEventCreate(&b));
  float best = 1e30f;
  std::pair<int, int> pick(-1, 1);
  for (int t : tiles) {
    const bool halo = kTiles[t].kind == TK_HALO;
    if (halo && !halo_ok(p, kTiles[t])) continue;
    if (p.wscale && !tile_has_w8(t)) continue;
    const int nk = plan_nk(p, kTiles[t]);
    const int wgs1 = plan_tiles_m(p, kTiles[t]) * cdiv(p.N, kTiles[t].bn);
    if (p.act == 3 && (kTiles[t].bn / 2) % 32 != 0) continue;
    for (int sk : sks) {
      if (sk > 1 && p.act == 3) continue;
      if (sk > 1 && (nk / sk < (halo ? 1 : 4) || (size_t)sk * p.M * p.N * sizeof(float) > p.workspace_bytes || !p.workspace)) continue;
      if (sk > 1 && make_plan([&] { ConvGemm q = p; q.force_tile = t; q.force_splitk = sk; return q; }()).splitk != sk) continue;   // duplicate of a smaller factor
      if (sk > 1 && wgs1 >= 4 * kNumCU) continue;      // already several rounds of workgroups: splitting K only adds traffic
      ConvGemm q = p;
      q.force_tile = t;
      q.force_splitk = sk;
      q.gn_out = nullptr;                              // candidates are timed without the GroupNorm partials (not every plan can emit them)
      q.gn_in = nullptr;
      if (int rc = conv_gemm(q, stream)) return rc;    // warm-up (also sets the function attributes)
      // best of two rounds of 8 back-to-back launches: single short rounds flipped plans from run to run
      const int reps = 8;
      float ms = 1e30f;
      for (int round = 0; round < 2; ++round) {
        SDEO_HIP(hipEventRecord(a, stream));
        for (int r = 0; r < reps; ++r)
          if (int rc = conv_gemm(q, stream)) return rc;
        SDEO_HIP(hipEventRecord(b, stream));
        SDEO_HIP(hipEventSynchronize(b));
        float t = 0.f;
        SDEO_HIP(hipEventElapsedTime(&t, a, b));
        ms = fminf(ms, t / reps);
      }
      // a 3x3 conv of a ResBlock feeds a GroupNorm: a plan whose epilogue cannot emit the GroupNorm partials (split-K, strips that
      // cut a group) costs that GroupNorm a statistics launch (~4 us, tools/plan_ab.py) where it runs as two launches (HW >= 1024)
      if (p.R == 3 && p.N % 32 == 0 && p.Ho * p.Wo >= 1024 && p.y && conv_gemm_gn_slots(q, p.N / 32) == 0) ms += 0.004f;
      if (ms < best) { best = ms; pick = {t, make_plan(q).splitk}; }
    }
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (pick.first >= 0) g_tuned[key] = pick;
  return 0;
}

}  // namespace sdeo
