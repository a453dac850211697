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
//  * V is consumed transposed: the caller supplies V^T ([heads*d][B*TkS], produced directly by the
//    projection GEMM with swapped operands), so both MFMA operands are contraction-contiguous in memory and
//    the kernel needs no in-LDS transpose.
//  * head dims 40 / 80 / 160 (and 8..64 for the reduced test config): QK^T pads d to a multiple of 16 with
//    zero chunks in LDS, PV pads to a multiple of 32 rows.
//  * K / V^T tiles are register-staged and double-buffered: the next tile's global loads are issued before
//    the MFMA phase and written to the other LDS buffer after it.
#include "kernels.h"

namespace sdeo {

struct AP {
  f16* o;
  const f16* q;
  const f16* k;
  const f16* vt;
  int ldo, ldq, ldk, ldvt;
  int H, Tq, Tk, TkS, TkSv, d;
  float scale_log2;
};

template <int D16>
__global__ __launch_bounds__(256) void attention_kernel(const AP p) {
  constexpr int DT = (D16 + 1) / 2;                          // 32-row tiles of O^T
  constexpr int KROW = D16 * 32 + ((D16 * 2) % 2 == 0 ? 16 : 0);  // K tile row bytes (odd multiple of 16)
  constexpr int VROW = 64 * 2 + 8;                           // V^T tile row bytes (odd multiple of 8)
  constexpr int KBYTES = 64 * KROW;
  constexpr int VBYTES = DT * 32 * VROW;
  constexpr int STAGE = KBYTES + VBYTES;
  constexpr int KCH = D16 * 2;                               // 16-byte chunk slots per K row
  constexpr int KITEMS = 64 * KCH, KPASS = (KITEMS + 255) / 256;
  constexpr int VITEMS = DT * 32 * 8, VPASS = (VITEMS + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = q0 + lq;
  const bool qvalid = qrow < p.Tq;
  const int d = p.d;

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[qrow][ks*16 + 8*lh + 0..7]
  f16x8 qf[D16];
  {
    const f16* qp = p.q + ((size_t)b * p.Tq + (qvalid ? qrow : 0)) * p.ldq + h * d;
#pragma unroll
    for (int ks = 0; ks < D16; ++ks) {
      const int c = (ks * 2 + lh) * 8;
      if (qvalid && c < d) qf[ks] = *reinterpret_cast<const f16x8*>(qp + c);
      else qf[ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  const f16* kbase = p.k + (size_t)b * p.TkS * p.ldk + h * d;
  const f16* vbase = p.vt + (size_t)h * d * p.ldvt + (size_t)b * p.TkSv;
  const int ntiles = (p.Tk + 63) / 64;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  uint4 kr[KPASS], vr[VPASS];
  auto load_tile = [&](int kt) {
    const int key0 = kt * 64;
#pragma unroll
    for (int i = 0; i < KPASS; ++i) {
      const int it = tid + i * 256;
      const int row = it / KCH, c = it - row * KCH;
      const int key = key0 + row;
      kr[i] = (it < KITEMS && key < p.Tk && c * 8 < d)
                  ? *reinterpret_cast<const uint4*>(kbase + (size_t)key * p.ldk + c * 8) : zero4;
    }
#pragma unroll
    for (int i = 0; i < VPASS; ++i) {
      const int it = tid + i * 256;
      const int row = it >> 3, c = it & 7;
      const int key = key0 + c * 8;
      uint4 v = zero4;
      if (it < VITEMS && row < d && key < p.Tk) {
        v = *reinterpret_cast<const uint4*>(vbase + (size_t)row * p.ldvt + key);
        if (key + 8 > p.Tk) {  // partially valid chunk: zero the keys >= Tk (P is 0 there, V must be finite)
          f16x8 t = *reinterpret_cast<f16x8*>(&v);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (key + j >= p.Tk) t[j] = (f16)0.f;
          v = *reinterpret_cast<uint4*>(&t);
        }
      }
      vr[i] = v;
    }
  };
  auto store_tile = [&](int stage) {
    char* ks_ = smem + stage * STAGE;
    char* vs_ = ks_ + KBYTES;
#pragma unroll
    for (int i = 0; i < KPASS; ++i) {
      const int it = tid + i * 256;
      const int row = it / KCH, c = it - row * KCH;
      if (it < KITEMS) *reinterpret_cast<uint4*>(ks_ + row * KROW + c * 16) = kr[i];
    }
#pragma unroll
    for (int i = 0; i < VPASS; ++i) {
      const int it = tid + i * 256;
      const int row = it >> 3, c = it & 7;
      if (it < VITEMS) {
        uint2* dst = reinterpret_cast<uint2*>(vs_ + row * VROW + c * 16);   // rows are only 8-byte aligned
        dst[0] = make_uint2(vr[i].x, vr[i].y);
        dst[1] = make_uint2(vr[i].z, vr[i].w);
      }
    }
  };

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < ntiles;
    if (more) load_tile(kt + 1);
    const char* ks_ = smem + cur * STAGE;
    const char* vs_ = ks_ + KBYTES;

    // ---- S^T = K Q^T for the two 32-key blocks of this tile
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < D16; ++ks) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(ks_ + (kb * 32 + lq) * KROW + (ks * 2 + lh) * 16);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
    // ---- online softmax (base-2), key index of s[kb][r] = kt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*lh
    const bool tail = (kt + 1) * 64 > p.Tk;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = s[kb][r] * p.scale_log2;
        if (tail) {
          const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= p.Tk) v = -INFINITY;
        }
        s[kb][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
        s[kb][r] = pv;
        rs += pv;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

    // ---- O^T += V^T P^T : P^T fragments straight from the score registers
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        f16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (f16)s[kb][8 * st + j];
        const int kofs = (kb * 32 + 16 * st + 4 * lh) * 2;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const char* vrow = vs_ + (t * 32 + lq) * VROW + kofs;
          const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
          const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16);
          uint4 vv = make_uint4(lo.x, lo.y, hi.x, hi.y);
          const f16x8 vf = *reinterpret_cast<f16x8*>(&vv);
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[t], 0, 0, 0);
        }
      }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

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

template <int D16>
static int launch_attn(const AP& ap, int B, hipStream_t stream) {
  constexpr int DT = (D16 + 1) / 2;
  constexpr int KROW = D16 * 32 + ((D16 * 2) % 2 == 0 ? 16 : 0);
  constexpr int smem = 2 * (64 * KROW + DT * 32 * (64 * 2 + 8));
  static bool attr_done = false;
  if (!attr_done) {
    SDEO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<D16>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done = true;
  }
  dim3 grid(cdiv(ap.Tq, 128), B * ap.H);
  hipLaunchKernelGGL((attention_kernel<D16>), grid, dim3(256), smem, stream, ap);
  SDEO_HIP(hipGetLastError());
  return 0;
}

int attention(f16* o, int ldo, const f16* q, int ldq, const f16* k, int ldk, const f16* vt, int ldvt, int B, int H,
              int Tq, int Tk, int TkS, int TkSv, int d, float scale, hipStream_t stream) {
  SDEO_CHECK(o && q && k && vt, "attention: null operand");
  SDEO_CHECK(B > 0 && H > 0 && Tq > 0 && Tk > 0 && TkS >= Tk && TkSv >= Tk, "attention: bad sizes B=%d H=%d Tq=%d Tk=%d TkS=%d TkSv=%d", B,
             H, Tq, Tk, TkS, TkSv);
  SDEO_CHECK(d % 8 == 0 && d >= 8 && d <= 160, "attention: head dim %d unsupported (multiple of 8, <= 160)", d);
  SDEO_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && TkSv % 8 == 0,
             "attention: strides must keep 16-byte alignment (ldq=%d ldk=%d ldvt=%d ldo=%d TkSv=%d)", ldq, ldk, ldvt, ldo, TkSv);
  AP ap{o, q, k, vt, ldo, ldq, ldk, ldvt, H, Tq, Tk, TkS, TkSv, d, scale * 1.4426950408889634f};
  const int d16 = cdiv(d, 16);
  switch (d16) {
    case 1: return launch_attn<1>(ap, B, stream);
    case 2: return launch_attn<2>(ap, B, stream);
    case 3: return launch_attn<3>(ap, B, stream);
    case 4: return launch_attn<4>(ap, B, stream);
    case 5: return launch_attn<5>(ap, B, stream);
    case 6: return launch_attn<6>(ap, B, stream);
    case 8: return launch_attn<8>(ap, B, stream);
    case 10: return launch_attn<10>(ap, B, stream);
    default: return fail("attention: head dim %d not instantiated", d);
  }
}

}  // namespace sdeo
