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
//  * register-staged double buffering (global -> VGPR early, VGPR -> LDS after the MFMA phase), LDS
//    tiles XOR-swizzled so the ds_read_b128 fragment reads are bank-conflict free.
//  * split-K over blockIdx.z for the weight-bandwidth-bound deep levels (M = 128..512, K up to 23k).
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
  int tiles_m;
};

template <int BK>
__device__ __forceinline__ int swz_chunk(int row, int chunk) {
  // conflict-free for the 16-lane groups of ds_read_b128 when lanes read rows r..r+15 at one k-chunk
  if (BK == 64) return chunk ^ ((row >> 1) & 7);
  return chunk ^ (((row >> 3) & 1) * 3);
}

template <int BM, int BN, int BK, bool GENERIC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const KP p) {
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

  // ---- per-thread gather state for the activation rows this thread stages
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

  // ---- epilogue: lane holds n = nb + fq*4 + {0..3} (4 consecutive output channels) of pixel m
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + wm * TM + j * 16 + frow;
    if (m >= p.M) continue;
    const int b = p.bias2 ? m / p.HoWo : 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = n0 + wn * TN + i * 16 + fq * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.splitk > 1) {
        *reinterpret_cast<f32x4*>(p.ws + ((size_t)z * p.M + m) * p.N + n) = v;
        continue;
      }
      if (p.bias) {
        if (p.bias_per_row) {
          v += p.bias[m];
        } else {
          v += *reinterpret_cast<const f32x4*>(p.bias + n);
        }
      }
      if (p.bias2) v += *reinterpret_cast<const f32x4*>(p.bias2 + (size_t)b * p.ld_bias2 + n);
      if (p.act == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = silu_f(v[t]);
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
      }
    }
  }
}

// split-K: sum the fp32 partial slabs and apply the epilogue. One thread per 4 output channels.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const KP p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = p.N / 4;
  if (idx >= (int64_t)p.M * n4) return;
  const int m = (int)(idx / n4);
  const int n = (int)(idx - (int64_t)m * n4) * 4;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.splitk; ++z) v += *reinterpret_cast<const f32x4*>(p.ws + ((size_t)z * p.M + m) * p.N + n);
  if (p.bias) {
    if (p.bias_per_row) v += p.bias[m];
    else v += *reinterpret_cast<const f32x4*>(p.bias + n);
  }
  if (p.bias2) v += *reinterpret_cast<const f32x4*>(p.bias2 + (size_t)(m / p.HoWo) * p.ld_bias2 + n);
  if (p.act == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = silu_f(v[t]);
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
  }
}

// ------------------------------------------------------------------------------------------------
// host side: tile / split-K selection and launch
// ------------------------------------------------------------------------------------------------
struct TileCfg { int bm, bn, bk; bool generic; float weight; };
static const TileCfg kTiles[] = {
    {128, 128, 64, false, 1.00f},
    {128, 64, 64, false, 0.90f},
    {64, 64, 64, false, 0.72f},
    {128, 64, 32, true, 0.90f},
    {64, 64, 32, true, 0.72f},
};
static const int kNumTiles = 5;
static const int kNumCU = 256;

struct Plan { int tile; int splitk; int nk; int tiles_m, tiles_n; };

static bool is_fast(const ConvGemm& p) { return p.Cin % 64 == 0; }

static Plan make_plan(const ConvGemm& p) {
  Plan pl{};
  const bool fast = is_fast(p);
  float best = -1.f;
  for (int t = 0; t < kNumTiles; ++t) {
    const TileCfg& c = kTiles[t];
    if (c.generic == fast) continue;
    if (p.force_tile >= 0 && p.force_tile != t) continue;
    const int tmn = cdiv(p.M, c.bm), tnn = cdiv(p.N, c.bn);
    const float eff = ((float)p.M * p.N) / ((float)tmn * c.bm * tnn * c.bn);
    const float tiles = (float)tmn * tnn;
    // fraction of the chip busy (up to 2 resident blocks per CU), discounted by tile efficiency
    float fill = tiles / (2.0f * kNumCU);
    if (fill > 1.f) fill = 1.f;
    const float score = eff * c.weight * (0.25f + 0.75f * fill);
    if (score > best) { best = score; pl.tile = t; pl.tiles_m = tmn; pl.tiles_n = tnn; }
  }
  const TileCfg& c = kTiles[pl.tile];
  pl.nk = cdiv(p.K, c.bk);
  int sk = 1;
  const int tiles = pl.tiles_m * pl.tiles_n;
  if (p.force_splitk > 0) {
    sk = p.force_splitk;
  } else if (tiles < kNumCU && pl.nk >= 16) {
    sk = cdiv(kNumCU + kNumCU / 2, tiles);
    if (sk > pl.nk / 8) sk = pl.nk / 8;
    if (sk > 16) sk = 16;
    if (sk < 1) sk = 1;
  }
  if (sk > pl.nk) sk = pl.nk;
  // drop empty splits
  const int per = cdiv(pl.nk, sk);
  sk = cdiv(pl.nk, per);
  pl.splitk = sk;
  return pl;
}

size_t conv_gemm_workspace_bytes(const ConvGemm& p) {
  const Plan pl = make_plan(p);
  return pl.splitk > 1 ? (size_t)pl.splitk * p.M * p.N * sizeof(float) : 0;
}

const char* conv_gemm_kernel_name(const ConvGemm& p) {
  static const char* names[] = {"conv_gemm_kernel<128,128,64,false>", "conv_gemm_kernel<128,64,64,false>",
                                "conv_gemm_kernel<64,64,64,false>", "conv_gemm_kernel<128,64,32,true>",
                                "conv_gemm_kernel<64,64,32,true>"};
  return names[make_plan(p).tile];
}

template <int BM, int BN, int BK, bool G>
static int launch(const KP& kp, int tiles, hipStream_t stream) {
  constexpr int smem = 2 * (BM + BN) * BK * 2;
  static bool attr_done = false;
  if (!attr_done) {
    SDEO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, BK, G>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done = true;
  }
  dim3 grid(tiles, 1, kp.splitk);
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, G>), grid, dim3(256), smem, stream, kp);
  SDEO_HIP(hipGetLastError());
  return 0;
}

int conv_gemm(const ConvGemm& p, hipStream_t stream) {
  SDEO_CHECK(p.x && p.w && (p.y || p.y32), "conv_gemm: null operand");
  SDEO_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "conv_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  SDEO_CHECK(p.N % 4 == 0, "conv_gemm: N=%d must be a multiple of 4", p.N);
  SDEO_CHECK(p.Cin % 8 == 0 && p.ldx % 8 == 0 && p.ldw % 8 == 0, "conv_gemm: Cin=%d ldx=%d ldw=%d must be multiples of 8",
             p.Cin, p.ldx, p.ldw);
  SDEO_CHECK(p.K == p.R * p.S * p.Cin, "conv_gemm: K=%d != R*S*Cin=%d", p.K, p.R * p.S * p.Cin);
  SDEO_CHECK(p.ldw >= p.K, "conv_gemm: ldw=%d < K=%d", p.ldw, p.K);
  SDEO_CHECK(p.M == p.B * p.Ho * p.Wo, "conv_gemm: M=%d != B*Ho*Wo=%d", p.M, p.B * p.Ho * p.Wo);
  SDEO_CHECK(p.ldy % 4 == 0 && p.ldy >= p.N, "conv_gemm: ldy=%d", p.ldy);
  SDEO_CHECK(!p.res || p.ldres % 4 == 0, "conv_gemm: ldres=%d", p.ldres);
  {
    const int Hv = p.ups ? 2 * p.Hi : p.Hi, Wv = p.ups ? 2 * p.Wi : p.Wi;
    SDEO_CHECK((Hv + 2 * p.pad - p.R) / p.stride + 1 == p.Ho && (Wv + 2 * p.pad - p.S) / p.stride + 1 == p.Wo,
               "conv_gemm: geometry mismatch Hi=%d Wi=%d -> Ho=%d Wo=%d (R=%d S=%d stride=%d pad=%d ups=%d)", p.Hi, p.Wi,
               p.Ho, p.Wo, p.R, p.S, p.stride, p.pad, p.ups);
  }
  const Plan pl = make_plan(p);
  const TileCfg& c = kTiles[pl.tile];
  KP kp{};
  kp.x = p.x; kp.w = p.w; kp.y = p.y; kp.y32 = p.y32; kp.bias = p.bias; kp.bias2 = p.bias2; kp.res = p.res;
  kp.ws = p.workspace;
  kp.M = p.M; kp.N = p.N; kp.K = p.K;
  kp.Hi = p.Hi; kp.Wi = p.Wi; kp.Cin = p.Cin; kp.Ho = p.Ho; kp.Wo = p.Wo; kp.S = p.S; kp.stride = p.stride; kp.pad = p.pad;
  kp.ups = p.ups; kp.HoWo = p.Ho * p.Wo;
  kp.ldx = p.ldx; kp.ldw = p.ldw; kp.ldy = p.ldy; kp.ldres = p.ldres; kp.ld_bias2 = p.ld_bias2;
  kp.act = p.act; kp.bias_per_row = p.bias_per_row; kp.scale = p.scale;
  kp.nk = pl.nk; kp.splitk = pl.splitk; kp.nk_per_split = cdiv(pl.nk, pl.splitk);
  kp.tiles_m = pl.tiles_m;
  if (pl.splitk > 1) {
    const size_t need = (size_t)pl.splitk * p.M * p.N * sizeof(float);
    SDEO_CHECK(p.workspace && p.workspace_bytes >= need, "conv_gemm: split-K workspace too small (%zu < %zu)",
               p.workspace_bytes, need);
  }
  const int tiles = pl.tiles_m * pl.tiles_n;
  int rc = 0;
  switch (pl.tile) {
    case 0: rc = launch<128, 128, 64, false>(kp, tiles, stream); break;
    case 1: rc = launch<128, 64, 64, false>(kp, tiles, stream); break;
    case 2: rc = launch<64, 64, 64, false>(kp, tiles, stream); break;
    case 3: rc = launch<128, 64, 32, true>(kp, tiles, stream); break;
    case 4: rc = launch<64, 64, 32, true>(kp, tiles, stream); break;
    default: return fail("conv_gemm: bad tile %d", pl.tile);
  }
  if (rc) return rc;
  if (pl.splitk > 1) {
    const int64_t n = (int64_t)p.M * (p.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, stream, kp);
    SDEO_HIP(hipGetLastError());
  }
  (void)c;
  return 0;
}

}  // namespace sdeo
