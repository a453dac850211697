// GroupNorm(+SiLU), LayerNorm and row-softmax for gfx950.
//
// GroupNorm replaces the reference's TensorRT plugin (`plugin/groupNormPlugin/groupNormKernel.cu:49-238`,
// `groupNormPlugin.cpp:179-228`): same operand contract (NHWC fp16 in/out, fp32 gamma/beta, optional Swish),
// but (a) epsilon is honoured (the plugin drops it), (b) any channel count with C % 8 == 0
// and C % groups == 0 is accepted (the plugin hard-codes 320/480/256/128-channel blocks), (c) the per-group
// statistics are a deterministic two-level reduction (per-thread fp32 partials -> LDS -> per-chunk slab ->
// fixed-order sum) instead of atomicAdd into a memset workspace, so results are bitwise reproducible.
// Numerics follow PyTorch's F.group_norm (`util.py:217-219`, `attention.py:88-89`): fp32 mean / biased
// variance over (C/G)*H*W elements, y = (x-mean)*rsqrt(var+eps)*gamma+beta.
#include <stdlib.h>

#include "kernels.h"

namespace sdeo {

// statistics chunks per image: enough blocks to fill 256 CUs at the 64x64 level, capped so the
// second-level sum stays short
// upper bound of the statistics chunks per image (sizes the partial-sum workspace)
// operands of one GroupNorm problem
struct GnOne {
  f16* y; const f16* x; const float* gamma; const float* beta; float* partials; int ldy, ldx;
};
struct GnPair { GnOne k[1]; };

int gn_chunks(int HW) { const int c = cdiv(HW, 8); return c > 128 ? 128 : (c < 1 ? 1 : c); }

// Channel vectors (8 x fp16 = 16 B) handled by one block: the largest divisor of C/8 that is <= 256 and
// covers whole groups.  Returns 0 when no such split exists.
static int gn_vec_per_block(int C, int groups) {
  const int nvec = C / 8, cpg = C / groups;
  for (int parts = 1; parts <= nvec; ++parts) {
    if (nvec % parts) continue;
    const int nvb = nvec / parts;
    if (nvb > 256) continue;
    if ((nvb * 8) % cpg) continue;
    return nvb;
  }
  return 0;
}

// grid: (chunks, channel-parts, B).  Thread (prow, cv) walks pixels prow, prow+P, ... of its chunk with a
// fixed 8-channel vector cv, so a wave reads whole contiguous NHWC rows.
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnPair gp, int Bper, int HW, int C, int cpg, int nvb, int nchunks,
                                                       int groups, int ppc) {
  kernarg_warm<sizeof(GnPair) + 24>();
  const GnOne& g1 = gp.k[0];
  const f16* __restrict__ x = g1.x;
  const int ldx = g1.ldx;
  float* __restrict__ partials = g1.partials;
  __shared__ float s_sum[256 * 8];
  __shared__ float s_sq[256 * 8];
  const int tid = threadIdx.x;
  const int P = 256 / nvb;
  const int cv = tid % nvb, prow = tid / nvb;
  const int chunk = blockIdx.x, part = blockIdx.y, b = blockIdx.z;
  const int c0 = (part * nvb + cv) * 8;
  const int pbeg = chunk * ppc;
  const int pend = min(HW, pbeg + ppc);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  if (prow < P) {
    const f16* base = x + ((size_t)b * HW) * ldx + c0;
    // 4 independent 16-byte loads in flight per thread (the loop is latency-bound otherwise)
    int pix = pbeg + prow;
    for (; pix + 7 * P < pend; pix += 8 * P) {
      f16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f16x8*>(base + (size_t)(pix + u * P) * ldx);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)v[u][j];
          s[j] += f;
          q[j] += f * f;
        }
    }
    for (; pix + 3 * P < pend; pix += 4 * P) {
      f16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f16x8*>(base + (size_t)(pix + u * P) * ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)v[u][j];
          s[j] += f;
          q[j] += f * f;
        }
    }
    for (; pix < pend; pix += P) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(base + (size_t)pix * ldx);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        s[j] += f;
        q[j] += f * f;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { s_sum[tid * 8 + j] = s[j]; s_sq[tid * 8 + j] = q[j]; }
  __syncthreads();
  // tpg threads per group: each sums a fixed slice of the (pixel row, channel) entries, then a fixed-order butterfly
  const int gpb = nvb * 8 / cpg;    // groups per block
  int tpg = 256 / gpb;
  tpg = tpg >= 32 ? 32 : (tpg >= 16 ? 16 : (tpg >= 8 ? 8 : (tpg >= 4 ? 4 : (tpg >= 2 ? 2 : 1))));   // power of two <= 32
  const int gl = tid / tpg, l = tid - gl * tpg;
  float ts = 0.f, tq = 0.f;
  if (gl < gpb) {
    const int n = P * cpg;            // entries of this group: (pr, c) with c in [gl*cpg, gl*cpg + cpg)
    for (int e = l; e < n; e += tpg) {
      const int pr = e / cpg, c = gl * cpg + (e - pr * cpg);
      const int idx = (pr * nvb + (c >> 3)) * 8 + (c & 7);
      ts += s_sum[idx];
      tq += s_sq[idx];
    }
  }
  for (int off = tpg >> 1; off >= 1; off >>= 1) {
    ts += __shfl_xor(ts, off, 64);
    tq += __shfl_xor(tq, off, 64);
  }
  if (gl < gpb && l == 0) {
    const int g = part * gpb + gl;
    float* dst = partials + (((size_t)b * nchunks + chunk) * groups + g) * 2;
    dst[0] = ts;
    dst[1] = tq;
  }
}

// grid: (chunks, channel-parts, B): fold mean/rstd/gamma/beta into per-channel a,b in LDS, then y = x*a+b.
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnPair gp, int Bper, int HW, int C, int cpg, int nvb,
                                                       int nchunks, int groups, float eps, int with_silu, int ppc, int nsc) {
  kernarg_warm<sizeof(GnPair) + 24>();
  const GnOne& g1 = gp.k[0];
  f16* __restrict__ y = g1.y;
  const f16* __restrict__ x = g1.x;
  const int ldy = g1.ldy, ldx = g1.ldx;
  const float* __restrict__ gamma = g1.gamma;
  const float* __restrict__ beta = g1.beta;
  const float* __restrict__ partials = g1.partials;
  __shared__ float s_a[256 * 8];
  __shared__ float s_b[256 * 8];
  __shared__ float s_mean[64];
  __shared__ float s_rstd[64];
  const int tid = threadIdx.x;
  const int P = 256 / nvb;
  const int cv = tid % nvb, prow = tid / nvb;
  const int chunk = blockIdx.x, part = blockIdx.y, b = blockIdx.z;
  const int gpb = nvb * 8 / cpg;
  const bool active = prow < P;
  const int c0 = (part * nvb + cv) * 8;
  const int pbeg = chunk * ppc;
  const int pend = min(HW, pbeg + ppc);
  const f16* xb = x + ((size_t)b * HW) * ldx + c0;
  f16* yb = y + ((size_t)b * HW) * ldy + c0;
  // The kernel is a chain of dependent round trips (partials -> mean/rstd -> gamma/beta -> x -> y) on tensors of a
  // few MB, i.e. latency-bound: issue every load that does not depend on the statistics FIRST.
  f16x8 xv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int pix = pbeg + prow + u * P;
    if (active && pix < pend) xv[u] = *reinterpret_cast<const f16x8*>(xb + (size_t)pix * ldx);
  }
  float gpre[8], bpre[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = tid + k * 256;
    if (c < nvb * 8) { gpre[k] = gamma[part * nvb * 8 + c]; bpre[k] = beta[part * nvb * 8 + c]; }
  }
  {
    // second-level reduction over the statistics chunks: tpg threads per group, fixed order => deterministic
    const int tpg = 256 / gpb;
    const int gl = tid / tpg, l = tid - gl * tpg;
    float ts = 0.f, tq = 0.f;
    if (gl < gpb) {
      const float2* src = reinterpret_cast<const float2*>(partials) + ((size_t)b * nsc * groups + part * gpb + gl);
      for (int c0 = l; c0 < nsc; c0 += 8 * tpg) {       // 8 independent loads in flight, summed in a fixed order
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = c0 + u * tpg;
          v[u] = c < nsc ? src[(size_t)c * groups] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { ts += v[u].x; tq += v[u].y; }
      }
    }
    s_a[tid] = ts;
    s_b[tid] = tq;
  }
  __syncthreads();
  if (tid < gpb) {
    const int tpg = 256 / gpb;
    float ts = 0.f, tq = 0.f;
    for (int l = 0; l < tpg; ++l) { ts += s_a[tid * tpg + l]; tq += s_b[tid * tpg + l]; }
    const float inv = 1.0f / ((float)cpg * (float)HW);
    const float mean = ts * inv;
    float var = tq * inv - mean * mean;
    var = var < 0.f ? 0.f : var;
    s_mean[tid] = mean;
    s_rstd[tid] = rsqrtf(var + eps);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = tid + k * 256;
    if (c < nvb * 8) {
      const int g = c / cpg;
      const float a = s_rstd[g] * gpre[k];
      s_a[c] = a;
      s_b[c] = bpre[k] - s_mean[g] * a;
    }
  }
  __syncthreads();
  if (!active) return;
  float a[8], bb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = s_a[cv * 8 + j]; bb[j] = s_b[cv * 8 + j]; }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int pix = pbeg + prow + u * P;
    if (pix < pend) {
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (float)xv[u][j] * a[j] + bb[j];
        if (with_silu) f = silu_f(f);
        o[j] = (f16)f;
      }
      *reinterpret_cast<f16x8*>(yb + (size_t)pix * ldy) = o;
    }
  }
  for (int pix = pbeg + prow + 4 * P; pix < pend; pix += P) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(xb + (size_t)pix * ldx);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (float)v[j] * a[j] + bb[j];
      if (with_silu) f = silu_f(f);
      o[j] = (f16)f;
    }
    *reinterpret_cast<f16x8*>(yb + (size_t)pix * ldy) = o;
  }
}

// producer-side partials with more entries per group than the apply kernel should re-sum in every workgroup: one block per
// (image, group) folds them to a single entry, fixed order (per-thread strided sums, then a fixed tree)
__global__ __launch_bounds__(256) void gn_fold_partials_kernel(float2* __restrict__ out, const float2* __restrict__ in, int nsc, int groups) {
  __shared__ float2 red[256];
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups, tid = threadIdx.x;
  float s = 0.f, q = 0.f;
  for (int c = tid; c < nsc; c += 256) { const float2 v = in[((size_t)b * nsc + c) * groups + g]; s += v.x; q += v.y; }
  red[tid] = make_float2(s, q);
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) { red[tid].x += red[tid + off].x; red[tid].y += red[tid + off].y; }
    __syncthreads();
  }
  if (tid == 0) out[(size_t)b * groups + g] = red[0];
}

int groupnorm_fold_partials(float* out, const float* in, int B, int nsc, int groups, hipStream_t stream) {
  SDEO_CHECK(out && in && B > 0 && nsc > 0 && groups > 0, "groupnorm_fold_partials: bad argument");
  hipLaunchKernelGGL(gn_fold_partials_kernel, dim3(B * groups), dim3(256), 0, stream, reinterpret_cast<float2*>(out),
                     reinterpret_cast<const float2*>(in), nsc, groups);
  SDEO_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Single-launch GroupNorm for tensors whose per-(image, group set) slice fits in one workgroup's LDS.
// Every kernel on this chip costs ~4 us of launch + dependent round trips whatever its size (DESIGN.md
// section 10), so below the 64x64 level the two-launch path above is mostly fixed cost.
// grid (channel parts, B); a part is the smallest run of channel vectors that covers whole groups
// (lcm(cpg, 8) channels).  The slice is pulled into LDS by LDS-DMA (every piece in flight at once, no
// registers), thread t owning vector t of each NT-vector sweep; statistics are reduced deterministically
// (per-thread partials -> LDS -> one wave per (vector, slot) column in a fixed order); the second pass
// normalises from LDS.  A vector of 8 channels spans at most two groups ("slots"), checked on the host.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void gn_fused_kernel(const GnPair gp, int Bper, int HW, int cpg, int nvw, int nsweep, float eps,
                                                      int with_silu) {
  kernarg_warm<sizeof(GnPair) + 24>();
  const GnOne& g1 = gp.k[0];
  f16* __restrict__ y = g1.y;
  const f16* __restrict__ x = g1.x;
  const int ldy = g1.ldy, ldx = g1.ldx;
  const float* __restrict__ gamma = g1.gamma;
  const float* __restrict__ beta = g1.beta;
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int P = NT / nvw;                       // pixel rows per sweep
  const int v = tid % nvw, prow = tid / nvw;
  const bool active = prow < P;
  const int part = blockIdx.x, b = blockIdx.y;
  const int c0 = (part * nvw + v) * 8;          // parts start on a group boundary
  char* slice = gsm;                                                                  // [nsweep][NT] 16-byte vectors
  float2* part_sq = reinterpret_cast<float2*>(gsm + (size_t)nsweep * NT * 16);        // [2 * nvw][P]
  float2* tot = part_sq + (size_t)2 * nvw * P;                                        // [2 * nvw]
  float* s_mean = reinterpret_cast<float*>(tot + 2 * nvw);                            // [groups per part]
  float* s_rstd = s_mean + 32;

  const f16* xb = x + ((size_t)b * HW) * ldx + c0;
  f16* yb = y + ((size_t)b * HW) * ldy + c0;
  {
    for (int k = 0; k < nsweep; ++k) {
      const int pix = prow + k * P;
      // lanes without a pixel fetch a valid dummy (the first pixel of their vector); they are masked below
      const f16* src = (active && pix < HW) ? xb + (size_t)pix * ldx : x + c0 % 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slice + (size_t)k * NT * 16 + wave_u * 1024), 16, 0, 0);
    }
  }
  f32x4 gm[2], bt[2];
  gm[0] = *reinterpret_cast<const f32x4*>(gamma + c0);
  gm[1] = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
  bt[0] = *reinterpret_cast<const f32x4*>(beta + c0);
  bt[1] = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // own DMAs landed; each thread only reads back its own vectors

  const int g_lo = (v * 8) / cpg;                 // group (inside the part) of this vector's first channel
  const int nb = (g_lo + 1) * cpg - v * 8;        // channels of the vector that belong to g_lo (>= 1; may exceed 8)
  float sj[8], qj[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sj[j] = 0.f; qj[j] = 0.f; }
  const char* mine = slice + (size_t)tid * 16;
  for (int k = 0; k < nsweep; ++k) {
    if (active && prow + k * P < HW) {
      const f16x8 xv = *reinterpret_cast<const f16x8*>(mine + (size_t)k * NT * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)xv[j];
        sj[j] += f;
        qj[j] += f * f;
      }
    }
  }
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < nb) { s0 += sj[j]; q0 += qj[j]; }
    else { s1 += sj[j]; q1 += qj[j]; }
  }
  if (active) {
    part_sq[(size_t)(v * 2) * P + prow] = make_float2(s0, q0);
    part_sq[(size_t)(v * 2 + 1) * P + prow] = make_float2(s1, q1);
  }
  __syncthreads();
  for (int pair = wave; pair < 2 * nvw; pair += NT / 64) {
    float ts = 0.f, tq = 0.f;
    const float2* col = part_sq + (size_t)pair * P;
    for (int e = lane; e < P; e += 64) { ts += col[e].x; tq += col[e].y; }
    ts = wave_sum(ts);                            // DPP / permlane reduction: no LDS round trip per step (common.h)
    tq = wave_sum(tq);
    if (lane == 0) tot[pair] = make_float2(ts, tq);
  }
  __syncthreads();
  const int gpw = nvw * 8 / cpg;                  // groups in this part
  if (tid < gpw) {
    float ts = 0.f, tq = 0.f;
    for (int vv = 0; vv < nvw; ++vv) {
      const int gl = (vv * 8) / cpg;
      if (gl == tid) { ts += tot[vv * 2].x; tq += tot[vv * 2].y; }
      if (gl + 1 == tid) { ts += tot[vv * 2 + 1].x; tq += tot[vv * 2 + 1].y; }
    }
    const float inv = 1.0f / ((float)cpg * (float)HW);
    const float mean = ts * inv;
    float var = tq * inv - mean * mean;
    var = var < 0.f ? 0.f : var;
    s_mean[tid] = mean;
    s_rstd[tid] = rsqrtf(var + eps);
  }
  __syncthreads();
  if (!active) return;
  float a[8], bb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int g = j < nb ? g_lo : g_lo + 1;
    const float sc = s_rstd[g] * gm[j >> 2][j & 3];
    a[j] = sc;
    bb[j] = bt[j >> 2][j & 3] - s_mean[g] * sc;
  }
  for (int k = 0; k < nsweep; ++k) {
    const int pix = prow + k * P;
    if (pix < HW) {
      const f16x8 xv = *reinterpret_cast<const f16x8*>(mine + (size_t)k * NT * 16);
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = (float)xv[j] * a[j] + bb[j];
        if (with_silu) f = silu_f(f);
        o[j] = (f16)f;
      }
      *reinterpret_cast<f16x8*>(yb + (size_t)pix * ldy) = o;
    }
  }
}

// vectors per part of the single-launch path (0 = not applicable): lcm(cpg, 8) / 8, every vector within two groups
static int gn_fused_vecs(int C, int cpg) {
  if (cpg < 4) return 0;
  int l = cpg;
  while (l % 8) l += cpg;
  const int nvw = l / 8;
  if (nvw > 16 || (C / 8) % nvw) return 0;
  for (int v = 0; v < nvw; ++v)
    if ((v * 8 + 7) / cpg - (v * 8) / cpg > 1) return 0;
  return nvw;
}

static const size_t kGnFusedLdsCap = 144 * 1024;

template <int NT>
static size_t gn_fused_smem(int HW, int nvw) {
  const int P = NT / nvw, nsweep = cdiv(HW, P);
  return (size_t)nsweep * NT * 16 + ((size_t)2 * nvw * P + 2 * nvw) * sizeof(float2) + 64 * sizeof(float);
}

template <int NT>
static int launch_gn_fused(const GnPair& gp, int count, int B, int HW, int C, int cpg, int nvw, float eps, int with_silu, hipStream_t stream) {
  static DeviceOnce attr_done;
  if (attr_done.need()) {
    SDEO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_fused_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)kGnFusedLdsCap));
    attr_done.mark();
  }
  const int P = NT / nvw, nsweep = cdiv(HW, P);
  hipLaunchKernelGGL((gn_fused_kernel<NT>), dim3((C / 8) / nvw, B * count), dim3(NT), gn_fused_smem<NT>(HW, nvw), stream, gp, B, HW,
                     cpg, nvw, nsweep, eps, with_silu);
  SDEO_HIP(hipGetLastError());
  return 0;
}

static int gn_check(const GnArgs& a) {
  SDEO_CHECK(a.y && a.x && a.gamma && a.beta && (a.partials || (a.ext_partials && a.ext_nsc <= 128)), "groupnorm: null operand");
  SDEO_CHECK(a.B > 0 && a.HW > 0 && a.C > 0, "groupnorm: empty tensor");
  SDEO_CHECK(a.groups > 0 && a.groups <= 64 && a.C % a.groups == 0, "groupnorm: C=%d not divisible into %d groups", a.C, a.groups);
  SDEO_CHECK(a.C % 8 == 0 && a.ldx % 8 == 0 && a.ldy % 8 == 0, "groupnorm: C=%d ldx=%d ldy=%d must be multiples of 8", a.C, a.ldx, a.ldy);
  return 0;
}

// 0: two-launch path; 256 / 1024: threads of the single-launch kernel
static int gn_fused_threads(const GnArgs& a) {
  const int HW = a.HW, C = a.C, cpg = C / a.groups;
  static const int two_pass = [] { const char* e = getenv("SDEO_GN_TWO_PASS"); return e ? atoi(e) : 0; }();
  const int nvw = two_pass ? 0 : gn_fused_vecs(C, cpg);
  if (nvw <= 0) return 0;
  // small slices: 256 threads (more workgroups per CU); otherwise 1024 threads, if the slice fits in LDS
  // Measured (tools/small_bench.py): one launch wins up to 16x16 pixels (6.1 vs 9.0 us at 8x8, 8.4 vs 9.5 at 16x16);
  // from 32x32 on, the many-workgroup two-launch path is faster than one workgroup per part (9.9 vs 12.3 us).
  // SDEO_GN_FUSED_MAX_HW overrides the crossover for measurements.
  static const int max_hw = [] { const char* e = getenv("SDEO_GN_FUSED_MAX_HW"); return e ? atoi(e) : 256; }();
  if (HW > max_hw) return 0;
  if (cdiv(HW, 256 / nvw) <= 8 && gn_fused_smem<256>(HW, nvw) <= kGnFusedLdsCap) return 256;
  if (cdiv(HW, 1024 / nvw) <= 60 && gn_fused_smem<1024>(HW, nvw) <= kGnFusedLdsCap) return 1024;
  return 0;
}

bool groupnorm_is_single_launch(const GnArgs& a) {
  return a.groups > 0 && a.C % a.groups == 0 && a.C % 8 == 0 && gn_fused_threads(a) != 0;
}

static int gn_dispatch(const GnArgs& a, const GnPair& gp, int count, hipStream_t stream) {
  const int B = a.B, HW = a.HW, C = a.C, groups = a.groups, with_silu = a.with_silu;
  const float eps = a.eps;
  const int cpg = C / groups;
  if (a.ext_partials) {
    // statistics came out of the producer's epilogue: normalise only
    SDEO_CHECK(count == 1 && a.ext_nsc >= 1, "groupnorm: bad producer partials");
    const int nvb = gn_vec_per_block(C, groups);
    SDEO_CHECK(nvb > 0, "groupnorm: unsupported channel split C=%d groups=%d", C, groups);
    const int parts = (C / 8) / nvb, P = 256 / nvb;
    int chunks = cdiv(HW, 4 * P);
    if (chunks > 2048) chunks = 2048;
    const int ppc = cdiv(HW, chunks);
    GnPair g2 = gp;
    int nsc = a.ext_nsc;
    if (nsc > 128) {
      hipLaunchKernelGGL(gn_fold_partials_kernel, dim3(B * groups), dim3(256), 0, stream, reinterpret_cast<float2*>(a.partials),
                         reinterpret_cast<const float2*>(a.ext_partials), nsc, groups);
      nsc = 1;
    } else {
      g2.k[0].partials = const_cast<float*>(a.ext_partials);
    }
    hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, parts, B), dim3(256), 0, stream, g2, B, HW, C, cpg, nvb, chunks, groups, eps, with_silu, ppc, nsc);
    SDEO_HIP(hipGetLastError());
    return 0;
  }
  {
    const int nt = gn_fused_threads(a);
    const int nvw = gn_fused_vecs(C, cpg);
    if (nt == 256) return launch_gn_fused<256>(gp, count, B, HW, C, cpg, nvw, eps, with_silu, stream);
    if (nt == 1024) return launch_gn_fused<1024>(gp, count, B, HW, C, cpg, nvw, eps, with_silu, stream);
  }
  const int nvb = gn_vec_per_block(C, groups);
  SDEO_CHECK(nvb > 0, "groupnorm: unsupported channel split C=%d groups=%d", C, groups);
  const int parts = (C / 8) / nvb;
  // Both kernels are chains of dependent memory round trips (~0.5 us each on this chip), so the chunking is chosen to
  // give every thread ONE batch of independent loads: 4 pixels per thread in apply (prefetched before the statistics
  // are known), 8 in the statistics pass; P = pixel rows a block covers per sweep.
  const int P = 256 / nvb;
  int chunks = cdiv(HW, 4 * P);
  if (chunks > 2048) chunks = 2048;
  const int ppc = cdiv(HW, chunks);
  dim3 grid(chunks, parts, B * count);
  int sc = cdiv(HW, 8 * P);
  { const int mxs = gn_chunks(HW) < 128 ? gn_chunks(HW) : 128; sc = sc < 1 ? 1 : (sc > mxs ? mxs : sc); }
  const int sppc = cdiv(HW, sc);
  dim3 sgrid(sc, parts, B * count);
  hipLaunchKernelGGL(gn_stats_kernel, sgrid, dim3(256), 0, stream, gp, B, HW, C, cpg, nvb, sc, groups, sppc);
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), 0, stream, gp, B, HW, C, cpg, nvb, chunks, groups, eps, with_silu, ppc, sc);
  SDEO_HIP(hipGetLastError());
  return 0;
}

static GnOne gn_one(const GnArgs& a) { return GnOne{a.y, a.x, a.gamma, a.beta, a.partials, a.ldy, a.ldx}; }

int groupnorm_nhwc(const GnArgs& a, hipStream_t stream) {
  if (int rc = gn_check(a)) return rc;
  GnPair gp{};
  gp.k[0] = gn_one(a);
  return gn_dispatch(a, gp, 1, stream);
}

int groupnorm_nhwc(f16* y, int ldy, const f16* x, int ldx, const float* gamma, const float* beta, int B, int HW, int C,
                   int groups, float eps, int with_silu, float* partials, hipStream_t stream) {
  return groupnorm_nhwc(GnArgs{y, x, gamma, beta, partials, ldy, ldx, B, HW, C, groups, eps, with_silu}, stream);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (`attention.py:372-374`, nn.LayerNorm eps 1e-5): one wave per row, the row lives in
// registers (<= 4 vectors of 8 per lane => C <= 2048), exact two-pass mean / variance in fp32.
// ------------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_kernel(f16* __restrict__ y, int ldy, const f16* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = C / 8;
  f16x8 v[VPL];
  f32x4 gv[VPL][2], bv[VPL][2];      // gamma / beta fetched up front: independent of the row statistics
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      v[i] = *reinterpret_cast<const f16x8*>(x + (size_t)row * ldx + vi * 8);
      gv[i][0] = *reinterpret_cast<const f32x4*>(gamma + vi * 8); gv[i][1] = *reinterpret_cast<const f32x4*>(gamma + vi * 8 + 4);
      bv[i][0] = *reinterpret_cast<const f32x4*>(beta + vi * 8); bv[i][1] = *reinterpret_cast<const f32x4*>(beta + vi * 8 + 4);
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (float)v[i][j] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        o[j] = (f16)(((float)v[i][j] - mean) * rstd * gv[i][j >> 2][j & 3] + bv[i][j >> 2][j & 3]);
      *reinterpret_cast<f16x8*>(y + (size_t)row * ldy + vi * 8) = o;
    }
  }
}

int layernorm(f16* y, int ldy, const f16* x, int ldx, const float* gamma, const float* beta, int rows, int C, float eps,
              hipStream_t stream) {
  SDEO_CHECK(y && x && gamma && beta, "layernorm: null operand");
  SDEO_CHECK(rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "layernorm: rows=%d C=%d ldx=%d ldy=%d", rows, C,
             ldx, ldy);
  SDEO_CHECK(C <= 2048, "layernorm: C=%d > 2048 unsupported", C);
  const int vpl = cdiv(C / 8, 64);
  dim3 grid(cdiv(rows, 4));
  if (vpl <= 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, stream, y, ldy, x, ldx, gamma, beta, rows, C, eps);
  else if (vpl == 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, stream, y, ldy, x, ldx, gamma, beta, rows, C, eps);
  else hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, stream, y, ldy, x, ldx, gamma, beta, rows, C, eps);
  SDEO_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// row softmax of fp32 scores (VAE AttnBlock, `model.py:186-193`): one 256-thread block per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(f16* __restrict__ p, int ldp, const float* __restrict__ s, int lds,
                                                           int cols, float scale) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const float* src = s + (size_t)row * lds;
  float mx = -INFINITY;
  for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, src[c] * scale);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = tid; c < cols; c += 256) sum += __expf(src[c] * scale - mx);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.0f / sum;
  for (int c = tid; c < cols; c += 256) p[(size_t)row * ldp + c] = (f16)(__expf(src[c] * scale - mx) * inv);
}

int softmax_rows(f16* p, int ldp, const float* s, int lds, int rows, int cols, float scale, hipStream_t stream) {
  SDEO_CHECK(p && s && rows > 0 && cols > 0, "softmax_rows: bad operand");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, p, ldp, s, lds, cols, scale);
  SDEO_HIP(hipGetLastError());
  return 0;
}

}  // namespace sdeo
