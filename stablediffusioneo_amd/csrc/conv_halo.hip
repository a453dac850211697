// Halo-reuse 3x3 convolution (stride 1, pad 1) for gfx950 on v_mfma_f32_16x16x32_f16.
//
// Same arithmetic as the implicit-GEMM kernel of conv_gemm.hip (ResBlock / ResnetBlock 3x3 convs, `openaimodel.py:200-240`,
// `model.py:129-149`), different data movement.  At batch 1 the 3x3 convs of this model are bound by what a CU can take in
// through its vector-memory path (~28 B/clk/CU from L2), not by the matrix pipe: an im2col tile brings
// every activation in nine times (once per filter tap).  Here a workgroup owns a PH x PW patch of output pixels of ONE
// image and BN output channels, and for each 64-channel slice of Cin stages the (PH+2) x (PW+2) halo patch ONCE; the nine
// taps are nine K-steps that read their B fragments from the same LDS patch at shifted rows.  Per K-step a CU takes in
// BN rows of weights + 1/9 of the patch instead of BN + PH*PW rows.
//
//  * 8 waves: 0-3 read fragments and issue MFMAs (each owns PH*PW/4 pixels x all BN channels), 4-7 issue LDS-DMAs
//    (global_load_lds, 16 B per lane, counted s_waitcnt vmcnt + raw s_barrier, weights in a 4-slot ring, patch double-buffered).
//  * LDS images are the implicit-GEMM kernel's: 128-byte rows (64 channels), 16-byte chunk c of row r stored at chunk
//    c ^ ((r >> 1) & 7); the permutation is applied to the per-lane SOURCE address and to the fragment reads.
//  * halo pixels outside the image read a zero page (padding = 1), so every loader lane issues the same number of DMAs.
//  * split-K over blockIdx.z in units of 64-channel slices; epilogue shared with the implicit-GEMM kernel.
#include "conv_inl.h"

namespace sdeo {

constexpr int halo_xbytes(int ph, int pw) { return ((ph + 2) * (pw + 2) + 31) / 32 * 32 * 128; }
constexpr int halo_wbytes(int bn) { return (bn + 31) / 32 * 32 * 128; }
// ring slots; a slot holds the weights of TPB consecutive taps (TPB = 1: one barrier per tap; TPB = 3: one per filter row)
constexpr int halo_stages(int ph, int pw, int bn, int tpb = 1) {
  const int s = (160 * 1024 - 2 * halo_xbytes(ph, pw)) / (tpb * halo_wbytes(bn));
  const int cap = tpb == 1 ? 8 : 4;
  return s > cap ? cap : s;
}

// NMW = 4: one MFMA wave per SIMD beside one loader wave.  NMW = 8: two MFMA waves per SIMD (waves w and w + 4 share one), each
// owning half as many pixels: an in-order wave alone on its SIMD exposes every fragment-read issue, wait and barrier between
// its MFMA batches (measured 740 clocks per K-step for 320 clocks of MFMA); with a partner the SIMD interleaves the two.
// TPB = 3: the loaders and the MFMA waves meet at ONE barrier per filter row (three taps, 192-deep) instead of one per tap: the
// per-step skeleton (counted wait + barrier + loop bookkeeping, ~120 ns per tap by the ablations of DESIGN.md section 11) is paid a
// third as often and the MFMA waves run three taps of fragment reads and MFMAs without meeting the loaders.
template <int PH, int PW, int BN, int NMW, int TPB = 1>
__global__ __launch_bounds__(NMW * 64 + 256) void conv3x3_halo_kernel(const KP2 pp) {
  kernarg_warm<sizeof(KP2)>();
  // prologue scalars in one batch, pinned in SGPRs (see conv_gemm_dma_kernel); blockIdx.y is always 0
  KP pl = pp.k[0];
  // (integers only: a pointer that has been through the asm loses its address space and every access through it becomes a flat_ one)
  asm volatile("" : "+s"(pl.M), "+s"(pl.N), "+s"(pl.Hi), "+s"(pl.Wi), "+s"(pl.Cin), "+s"(pl.ldx), "+s"(pl.ldw),
               "+s"(pl.nk_per_split), "+s"(pl.tiles_m), "+s"(pl.tiles_n));
  const KP& p = pl;
  constexpr int BM = PH * PW;
  constexpr int MI = BM / (16 * NMW), NI = BN / 16;
  constexpr int HWD = PW + 2, XREAL = (PH + 2) * HWD;
  constexpr int LXW = (XREAL + 31) / 32;             // patch DMAs per loader wave per Cin slice (8 rows each, 4 loader waves)
  constexpr int XBYTES = LXW * 32 * 128;
  constexpr int LW = (BN + 31) / 32;                 // weight DMAs per loader wave per K-step
  constexpr int WBYTES = LW * 32 * 128;
  // weight ring as deep as LDS allows (<= 8 slots).  What bounds a K-step here is bytes in flight per CU, not bandwidth: a DMA
  // takes ~1 us to land under load, so a CU takes in (bytes in flight) / 1 us; 3 steps ahead gave ~40 GB/s, 7 give ~90.
  constexpr int WST = halo_stages(PH, PW, BN, TPB), PF = WST - 1;      // slots of TPB taps each; PF slots of loads ahead of the compute
  constexpr int WBASE = 2 * XBYTES;
  static_assert(TPB == 1 || TPB == 3, "taps per barrier");
  static_assert(PF * TPB >= 2 && PF <= 8, "ring depth");
  static_assert(BM % (16 * NMW) == 0 && BN % 16 == 0, "tile");
  static_assert((NI + MI) * 8 + NI * MI * 4 <= 200, "fragment double-buffering needs the registers");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int H = p.Hi, W = p.Wi;
  const int tpx = W / PW, tpi = (H / PH) * tpx;       // patches per row / per image
  const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tm = tile % p.tiles_m, tn = tile / p.tiles_m;   // patches fastest: an XCD's run of tiles shares one weight panel in L2
  const int b = tm / tpi, trem = tm - b * tpi;
  const int y0 = (trem / tpx) * PH, x0 = (trem % tpx) * PW;
  const int n0 = tn * BN;
  const int z = blockIdx.z;
  const int nch = p.Cin >> 6;
  const int c0 = z * p.nk_per_split;
  const int c1 = min(nch, c0 + p.nk_per_split);
  const int nsteps = (c1 - c0) * 9;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  stamp(p, 0);
  stamp_cycles(p, 14);
  if (dbg_on(p, 256) && lane == 0) {          // which SIMD each wave of the workgroup landed on (HW_REG_HW_ID)
    const int wg = blockIdx.x + gridDim.x * blockIdx.z;
    if (wg < kStampWGs) g_stamps[wg * kStampSlots + wave_all] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }

  if (wave_all >= NMW) {
    // ------------------------------------------------------------------ loader waves
    if (nsteps <= 0) { __builtin_amdgcn_s_barrier(); return; }
    const int lw = wave_all - NMW;
    const int lrow = lane >> 3, pc = lane & 7;
    const char* xsrc[LXW];
    int xinc[LXW];
#pragma unroll
    for (int q = 0; q < LXW; ++q) {
      const int hr = (q * 4 + lw) * 8 + lrow;
      const int cl = pc ^ ((hr >> 1) & 7);
      const int hy = hr / HWD, hx = hr - hy * HWD;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = hr < XREAL && y >= 0 && y < H && x >= 0 && x < W;
      xsrc[q] = ok ? reinterpret_cast<const char*>(p.x) + (((long)(b * H + y) * W + x) * p.ldx + c0 * 64 + cl * 8) * 2 : zero;
      xinc[q] = ok ? 128 : 0;
    }
    const char* wsrc[LW];
    bool wok[LW];
#pragma unroll
    for (int q = 0; q < LW; ++q) {
      const int wr = (q * 4 + lw) * 8 + lrow;
      const int cl = pc ^ ((wr >> 1) & 7);
      const int n = n0 + wr;
      wok[q] = wr < BN && n < p.N;
      wsrc[q] = reinterpret_cast<const char*>(p.w) + ((size_t)(wok[q] ? n : 0) * p.ldw + cl * 8) * 2;
    }
    auto issue_x = [&](int buf) {
      char* dst = smem + buf * XBYTES + lw * 1024;
#pragma unroll
      for (int q = 0; q < LXW; ++q) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xsrc[q],
                                         (__attribute__((address_space(3))) void*)(dst + q * 4096), 16, 0, 0);
        xsrc[q] += xinc[q];
      }
    };
    int st_tap = 0, st_ch = c0;                        // K-step the next issue_w stages
    auto issue_w = [&](int slot) {
#pragma unroll
      for (int tt = 0; tt < TPB; ++tt) {
        char* dst = smem + WBASE + (slot * TPB + tt) * WBYTES + lw * 1024;
        const long koff = ((long)st_tap * p.Cin + (st_ch << 6)) * 2;
#pragma unroll
        for (int q = 0; q < LW; ++q) {
          const char* src = wok[q] ? wsrc[q] + koff : zero;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst + q * 4096), 16, 0, 0);
        }
        if (++st_tap == 9) { st_tap = 0; ++st_ch; }
      }
    };
    // a "step" of the loaders is one ring slot = TPB taps; SPS steps per Cin slice
    constexpr int SPS = 9 / TPB, LWS = LW * TPB;
    const int lsteps = nsteps / TPB;
    // ---- GroupNorm (+ SiLU) of the input (KP::gn_in): every piece (16 bytes = 8 channels of one halo pixel) this lane staged is
    //      rewritten in place once its DMA has landed.  The lane's logical channel chunk is the same for all its pieces
    //      ((hr >> 1) & 7 does not depend on the piece index), so one (a, b) set per slice serves them all.  All LDS traffic of the
    //      loaders is inline asm: the compiler must not see LDS accesses next to LDS-DMAs (it would wait vmcnt(0) for them).
    const bool gnin = p.gn_in != nullptr;
    const int clq = pc ^ ((lw * 4 + (lrow >> 1)) & 7);
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned tab0 = ldsb + 2 * XBYTES + WST * TPB * WBYTES + clq * 64;      // (a, b) pairs of this lane's chunk, slice 0
    auto transform = [&](int buf, int slice_rel, int qa, int qb) {
      f32x4 t0, t1, t2, t3;
      const unsigned ta = tab0 + slice_rel * 512;
      asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(ta) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(t1) : "v"(ta) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(t2) : "v"(ta) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(t3) : "v"(ta) : "memory");
      const unsigned xb = ldsb + buf * XBYTES + lw * 1024 + lane * 16;
      f16x8 v[LXW];
#pragma unroll
      for (int q = 0; q < LXW; ++q)
        if (q >= qa && q < qb && xinc[q]) asm volatile("ds_read_b128 %0, %1" : "=v"(v[q]) : "v"(xb + q * 4096) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : : "memory");
      __builtin_amdgcn_sched_barrier(0);
      const float av[8] = {t0[0], t0[2], t1[0], t1[2], t2[0], t2[2], t3[0], t3[2]};
      const float bv[8] = {t0[1], t0[3], t1[1], t1[3], t2[1], t2[3], t3[1], t3[3]};
#pragma unroll
      for (int q = 0; q < LXW; ++q) {
        if (q >= qa && q < qb && xinc[q]) {
          asm volatile("" : "+v"(v[q]));
          f16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float f = fmaf((float)v[q][j], av[j], bv[j]);
            if (p.gn_in_silu) f = silu_f(f);
            o[j] = (f16)f;
          }
          asm volatile("ds_write_b128 %0, %1" : : "v"(xb + q * 4096), "v"(o) : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // the (a, b) loads of a transform and the pieces it rewrites: TPB = 3: half of them at step 1, half at step 2 of the previous
    // slice; TPB = 1: one piece per step from step 2 on
    constexpr int TR0 = TPB == 3 ? 1 : 2;               // first step of a slice that transforms the NEXT slice's patch
    int wsince = 0;                                     // W stages issued since the last patch (they are younger than it)
    issue_x(0);
#pragma unroll
    for (int s = 0; s < PF; ++s) issue_w(s);           // lsteps >= SPS > PF
    if (gnin) {
      __builtin_amdgcn_s_barrier();                    // T: the MFMA waves have built the (a, b) table
      wait_vmcnt<PF * LWS>();                          // patch 0 has landed (the PF weight stages behind it may still fly)
      transform(0, 0, 0, LXW);
    }
    int tap = 0, cr = 0;                               // step inside the slice / relative Cin slice of step `it`
    const int ncr = c1 - c0;
    for (int it = 0; it < lsteps; ++it) {
      // retire W(it) (and, being older, the patch of its slice).  Younger DMAs may stay in flight: the W stages of the next
      // min(PF-1, remaining) steps, plus the next slice's patch when it was issued in one of the last PF-1 iterations.
      // (the patch is issued at step 0 of a slice BEFORE that iteration's W stage, so it is younger than W(it) for steps 1 .. PF-1)
      const int a = min(PF - 1, lsteps - 1 - it);
      const bool xin = tap >= 1 && tap <= PF - 1 && cr + 1 < ncr;
      if (a == PF - 1) {
        if (xin) wait_vmcnt<(PF - 1) * LWS + LXW>();
        else wait_vmcnt<(PF - 1) * LWS>();
      } else {                                           // the last PF-1 steps: nothing is issued any more (xin is false there
        static_for<PF - 1>([&](auto A) {                 // unless the range has a single slice, where it is false anyway)
          if (a == A.value) wait_vmcnt<A.value * LWS>();
        });
      }
      __builtin_amdgcn_s_barrier();
      // every MFMA wave now holds step it-1 in registers: its W slot and (at step 0) the previous slice's patch are free
      if (dbg_on(p, 8)) { if (++tap == SPS) { tap = 0; ++cr; } continue; }      // SDEO_DBG_GEMM ablation: no DMAs after the prologue
      if (tap == 0 && cr + 1 < ncr) { issue_x((cr + 1) & 1); wsince = 0; }
      if (it + PF < lsteps) { issue_w((it + PF) % WST); ++wsince; }
      if (gnin && cr + 1 < ncr && tap >= TR0) {
        // the next slice's patch, issued at step 0 of this slice: it must have landed; only the weight stages issued since may fly
        if (tap == TR0) {
          static_for<TR0 + 2>([&](auto A) {
            if (wsince == A.value) wait_vmcnt<A.value * LWS>();
          });
        }
        if (TPB == 3) {
          if (tap == 1) transform((cr + 1) & 1, cr + 1, 0, LXW / 2);
          else transform((cr + 1) & 1, cr + 1, LXW / 2, LXW);
        } else {
          const int q = tap - TR0;
          if (q < LXW) transform((cr + 1) & 1, cr + 1, q, q + 1);
        }
      }
      if (++tap == SPS) { tap = 0; ++cr; }
    }
    __builtin_amdgcn_s_barrier();            // matches the MFMA waves' pre-epilogue barrier
    return;
  }

  // -------------------------------------------------------------------- MFMA waves
  const int wave = wave_all;
  const int frow = lane & 15, fq = lane >> 4;
  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias prefetch (keeps the epilogue from starting with a dependent global round trip) only where the registers allow
  constexpr bool BPRE = NI <= 5;
  f32x4 bpre[NI];
  const bool use_bpre = BPRE && p.bias && !p.bias_per_row && p.splitk == 1;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = n0 + i * 16 + fq * 4;
    bpre[i] = (use_bpre && n < p.N) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if (p.gn_in && nsteps > 0) {
    // (a, b) of every channel of this workgroup's range for its image: each wave sums the producer's partials of all 32 groups
    // (lane pair = group, fixed order: deterministic), then takes its share of the channels; the group of a channel is fetched
    // from the lane that holds it.  Runs while the loaders' first DMAs are in flight.
    // every load of this block is issued before the first use (one memory round trip instead of three dependent ones): gamma / beta
    // of this lane's channels first, then the partials of its group
    constexpr int CPL = 8;                                // channels per lane: covers (c1 - c0) * 64 <= NMW * 64 * CPL
    const int nchan = (c1 - c0) * 64;
    float gam[CPL], bet[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int cc = wave * 64 + lane + k * NMW * 64;
      const int ch = c0 * 64 + (cc < nchan ? cc : 0);
      gam[k] = p.gn_gamma[ch];
      bet[k] = p.gn_beta[ch];
    }
    const int g = lane >> 1, half = lane & 1;
    const float2* src = reinterpret_cast<const float2*>(p.gn_in) + ((size_t)b * p.gn_in_slots) * 32 + g;
    float ts = 0.f, tq = 0.f;
    for (int s0 = half; s0 < p.gn_in_slots; s0 += 32) {
      float2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = s0 + 2 * u < p.gn_in_slots ? src[(size_t)(s0 + 2 * u) * 32] : make_float2(0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 16; ++u) { ts += v[u].x; tq += v[u].y; }
    }
    const float os = __shfl_xor(ts, 1, 64), oq = __shfl_xor(tq, 1, 64);
    const float s_all = half ? os + ts : ts + os, q_all = half ? oq + tq : tq + oq;      // half 0 + half 1 in both lanes
    const float mean = s_all * p.gn_in_inv;
    float var = q_all * p.gn_in_inv - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + p.gn_in_eps);
    float* tab = reinterpret_cast<float*>(smem + 2 * XBYTES + WST * TPB * WBYTES);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int cc = wave * 64 + lane + k * NMW * 64;       // whole waves are inside or outside the range (nchan % 64 == 0)
      if (cc < nchan) {
        const int gg = (c0 * 64 + cc) / p.gn_in_cpg;
        const float m_ = __shfl(mean, 2 * gg, 64), r_ = __shfl(rstd, 2 * gg, 64);
        const float a_ = r_ * gam[k];
        *reinterpret_cast<float2*>(tab + 2 * cc) = make_float2(a_, bet[k] - m_ * a_);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // T: table visible to the loader waves
  }

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // patch fragment offsets (k-half 0) of this lane's pixel in each of its 16-pixel groups, for the nine taps;
  // k-half 1 is logical chunk 4 + fq = the same offset with byte bit 6 flipped
  unsigned xrel[9][MI];
  int mrow[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int pi = (wave * MI + j) * 16 + frow;
    const int py = pi / PW, px = pi - py * PW;
    mrow[j] = (b * H + y0 + py) * W + x0 + px;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hr = (py + t / 3) * HWD + px + t % 3;
      xrel[t][j] = hr * 128 + ((fq ^ ((hr >> 1) & 7)) << 4);
    }
  }
  const int swl = (frow >> 1) & 7;
  const unsigned wa0 = lds0 + WBASE + frow * 128 + ((fq ^ swl) << 4), wa1 = lds0 + WBASE + frow * 128 + (((4 + fq) ^ swl) << 4);

  if (nsteps > 0) {
    __builtin_amdgcn_s_waitcnt(0xc07f);      // retire the kernel-argument loads: see conv_gemm.hip (partial lgkmcnt waits below)
    f16x8 wf0[NI], xf0[MI], wf1[NI], xf1[MI];
    auto reads0 = [&](auto T, unsigned wsl, unsigned xoff) {
      if (dbg_on(p, 16)) return;
      static_for<NI>([&](auto I) { lds_read128<I.value * 2048>(wf0[I.value], wa0 + wsl); });
      static_for<MI>([&](auto J) { lds_read128<0>(xf0[J.value], lds0 + xoff + xrel[T.value][J.value]); });
    };
    auto reads1 = [&](auto T, unsigned wsl, unsigned xoff) {
      if (dbg_on(p, 16)) return;
      static_for<NI>([&](auto I) { lds_read128<I.value * 2048>(wf1[I.value], wa1 + wsl); });
      static_for<MI>([&](auto J) { lds_read128<0>(xf1[J.value], lds0 + xoff + (xrel[T.value][J.value] ^ 64u)); });
    };
    auto mma_half = [&](const f16x8 (&wf)[NI], const f16x8 (&xf)[MI]) {
      if (dbg_on(p, 4)) return;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    };
    stamp(p, 1);
    __builtin_amdgcn_s_barrier();            // K-step 0 (and the first patch) visible
    stamp(p, 2);
    reads0(std::integral_constant<int, 0>{}, 0u, 0u);
    reads1(std::integral_constant<int, 0>{}, 0u, 0u);
    int it = 0;
    unsigned slot = 0;                       // W ring position (in taps: slot * TPB + tap inside the slot) of K-step `it`
    for (int cr = 0; cr < c1 - c0; ++cr) {
      const unsigned xcur = (cr & 1) * XBYTES;
      static_for<9>([&](auto T) {
        constexpr int NT = (T.value + 1) % 9;                     // tap of the next K-step
        constexpr bool NEWSLOT = NT % TPB == 0;                   // the next K-step opens a ring slot: meet the loaders first
        const bool more = it + 1 < nsteps;
        const unsigned nslot = slot + 1 == WST * TPB ? 0u : slot + 1;
        const unsigned xnext = T.value == 8 ? (unsigned)XBYTES - xcur : xcur;
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NI + MI) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_half(wf0, xf0);
        __builtin_amdgcn_sched_barrier(0);
        if (NEWSLOT) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
          if (NEWSLOT) __builtin_amdgcn_s_barrier();      // the next slot is visible; every MFMA wave holds this slot's last tap in registers
          reads0(std::integral_constant<int, NT>{}, nslot * WBYTES, xnext);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!NEWSLOT) {                      // no barrier here: the half-1 fragments only have to be in registers
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NI + MI) : "memory");     // (issued before the NI + MI reads of reads0 above)
          __builtin_amdgcn_sched_barrier(0);
        }
        mma_half(wf1, xf1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) reads1(std::integral_constant<int, NT>{}, nslot * WBYTES, xnext);
        slot = nslot;
        ++it;
      });
    }
  }
  // all fragment reads done, no DMA pending: the patch / ring LDS becomes the waves' private epilogue scratch
  KP pe = p;                               // the epilogue's scalars in one batch of loads, in flight across the barrier (conv_inl.h)
  pin_epilogue_scalars(pe);
  __builtin_amdgcn_s_barrier();
  stamp(p, 3);
  stamp_cycles(p, 15);
  if (dbg_on(p, 32)) return;
  static_assert(NMW * epilogue_scratch_bytes(BN) <= 2 * XBYTES + WST * TPB * WBYTES, "epilogue scratch");
  constexpr int SCR = ((2 * XBYTES + WST * TPB * WBYTES) / NMW) & ~15;     // LDS each MFMA wave may use as epilogue scratch
  constexpr int NBLK = epilogue_blocks(BN, MI, SCR);
  epilogue_rows<NI, MI, BN, NBLK>(pe, acc, mrow, n0, fq, z, bpre, use_bpre, smem + wave * SCR);
  if (pe.gn_out) {                         // GroupNorm partials of this patch: the loader waves are gone, the barrier counts the rest
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == 0) gn_partials_finish<BN, NMW>(pe, smem, SCR, lane, b, trem, n0);
  }
  if (dbg_on(p, 64)) {
    stamp(p, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(p, 5);
    if (dbg_on(p, 128)) {        // the same epilogue once more, now with its code resident: cold instruction fetch vs work
      epilogue_rows<NI, MI, BN>(p, acc, mrow, n0, fq, z, bpre, use_bpre, smem + wave * epilogue_scratch_bytes(BN));
      stamp(p, 6);
    }
  }
}

int conv_halo_read_stamps(unsigned long long* out, int n) {
  SDEO_HIP(hipDeviceSynchronize());
  SDEO_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)(n < kStampWGs * kStampSlots ? n : kStampWGs * kStampSlots)));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// variants + launcher
// ------------------------------------------------------------------------------------------------
const HaloCfg kHaloCfgs[] = {
    {8, 16, 80, "conv3x3_halo_kernel<8,16,80,4>"},
    {8, 16, 160, "conv3x3_halo_kernel<8,16,160,4>"},
    {8, 8, 80, "conv3x3_halo_kernel<8,8,80,4>"},
    {8, 8, 160, "conv3x3_halo_kernel<8,8,160,4>"},
    {8, 16, 64, "conv3x3_halo_kernel<8,16,64,4>"},
    {8, 16, 128, "conv3x3_halo_kernel<8,16,128,4>"},
    {8, 16, 80, "conv3x3_halo_kernel<8,16,80,8>"},
    {8, 16, 160, "conv3x3_halo_kernel<8,16,160,8>"},
    {8, 16, 64, "conv3x3_halo_kernel<8,16,64,8>"},
    {8, 16, 128, "conv3x3_halo_kernel<8,16,128,8>"},
    // one barrier per filter row (TPB = 3)
    {8, 16, 80, "conv3x3_halo_kernel<8,16,80,4,3>"},
    {8, 16, 80, "conv3x3_halo_kernel<8,16,80,8,3>"},
    {8, 16, 64, "conv3x3_halo_kernel<8,16,64,4,3>"},
    {8, 8, 80, "conv3x3_halo_kernel<8,8,80,4,3>"},
    {8, 8, 160, "conv3x3_halo_kernel<8,8,160,4,3>"},
    {8, 16, 128, "conv3x3_halo_kernel<8,16,128,8,3>"},
    {8, 16, 64, "conv3x3_halo_kernel<8,16,64,8,3>"},
};
const int kNumHaloCfgs = 17;

template <int PH, int PW, int BN, int NMW, int TPB = 1>
static int launch_halo_t(const KP2& kp, int count, int tiles_m, int tiles_n, hipStream_t stream) {
  constexpr int smem = 2 * halo_xbytes(PH, PW) + halo_stages(PH, PW, BN, TPB) * TPB * halo_wbytes(BN);
  static_assert(smem <= 160 * 1024, "LDS");
  static DeviceOnce done;
  if (done.need()) {
    SDEO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<PH, PW, BN, NMW, TPB>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.mark();
  }
  // GroupNorm of the input (KP::gn_in): the (a, b) table of the workgroup's channel range sits behind the ring
  const int extra = kp.k[0].gn_in ? kp.k[0].nk_per_split * 512 : 0;
  SDEO_CHECK(smem + extra <= 160 * 1024, "conv3x3_halo_kernel: no LDS left for the GroupNorm table (%d + %d bytes)", smem, extra);
  hipLaunchKernelGGL((conv3x3_halo_kernel<PH, PW, BN, NMW, TPB>), dim3(tiles_m * tiles_n, count, kp.k[0].splitk), dim3(NMW * 64 + 256), smem + extra, stream,
                     kp);
  SDEO_HIP(hipGetLastError());
  return 0;
}

// LDS the ring of a variant takes (what is left of 160 KiB can hold the GroupNorm table of KP::gn_in)
int halo_ring_bytes(int variant) {
  static const int tpb[] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 3, 3, 3, 3, 3, 3, 3};
  if (variant < 0 || variant >= kNumHaloCfgs) return 1 << 30;
  const HaloCfg& h = kHaloCfgs[variant];
  return 2 * halo_xbytes(h.ph, h.pw) + halo_stages(h.ph, h.pw, h.bn, tpb[variant]) * tpb[variant] * halo_wbytes(h.bn);
}

int launch_halo(int variant, const KP2& kp, int count, int tiles_m, int tiles_n, hipStream_t stream) {
  switch (variant) {
    case 0: return launch_halo_t<8, 16, 80, 4>(kp, count, tiles_m, tiles_n, stream);
    case 1: return launch_halo_t<8, 16, 160, 4>(kp, count, tiles_m, tiles_n, stream);
    case 2: return launch_halo_t<8, 8, 80, 4>(kp, count, tiles_m, tiles_n, stream);
    case 3: return launch_halo_t<8, 8, 160, 4>(kp, count, tiles_m, tiles_n, stream);
    case 4: return launch_halo_t<8, 16, 64, 4>(kp, count, tiles_m, tiles_n, stream);
    case 5: return launch_halo_t<8, 16, 128, 4>(kp, count, tiles_m, tiles_n, stream);
    case 6: return launch_halo_t<8, 16, 80, 8>(kp, count, tiles_m, tiles_n, stream);
    case 7: return launch_halo_t<8, 16, 160, 8>(kp, count, tiles_m, tiles_n, stream);
    case 8: return launch_halo_t<8, 16, 64, 8>(kp, count, tiles_m, tiles_n, stream);
    case 9: return launch_halo_t<8, 16, 128, 8>(kp, count, tiles_m, tiles_n, stream);
    case 10: return launch_halo_t<8, 16, 80, 4, 3>(kp, count, tiles_m, tiles_n, stream);
    case 11: return launch_halo_t<8, 16, 80, 8, 3>(kp, count, tiles_m, tiles_n, stream);
    case 12: return launch_halo_t<8, 16, 64, 4, 3>(kp, count, tiles_m, tiles_n, stream);
    case 13: return launch_halo_t<8, 8, 80, 4, 3>(kp, count, tiles_m, tiles_n, stream);
    case 14: return launch_halo_t<8, 8, 160, 4, 3>(kp, count, tiles_m, tiles_n, stream);
    case 15: return launch_halo_t<8, 16, 128, 8, 3>(kp, count, tiles_m, tiles_n, stream);
    case 16: return launch_halo_t<8, 16, 64, 8, 3>(kp, count, tiles_m, tiles_n, stream);
    default: return fail("launch_halo: bad variant %d", variant);
  }
}

}  // namespace sdeo
