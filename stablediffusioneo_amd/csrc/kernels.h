// Host-side launchers of the hand-written gfx950 kernels.  All are asynchronous on `stream`,
// never allocate and never synchronise (hipGraph-capturable).
#pragma once
#include "common.h"

namespace sdeo {

// ------------------------------------------------------------------------------------------
// Implicit-GEMM convolution / GEMM on MFMA:   Y[M][N] = epi( sum_k A(m,k) * W[n][k] )
//   A(m,k) is gathered on the fly from an NHWC fp16 tensor (im2col never materialised):
//   m -> (b, ho, wo), k -> (r, s, c);  R=S=1 gives a plain row-major GEMM (Linear / conv1x1).
//   W is K-contiguous: [N][R][S][Cin] (KRSC) == nn.Linear's native [out][in].
//   epilogue:  v = acc + bias[n] + bias2[b][n];  v = act(v);  v = v*scale + res[m][n]
// ------------------------------------------------------------------------------------------
struct ConvGemm {
  const f16* x = nullptr;      // activations, pixel stride ldx elements
  const f16* w = nullptr;      // weights [N][K], row stride ldw
  f16* y = nullptr;            // fp16 output (row stride ldy) ...
  float* y32 = nullptr;        // ... or fp32 output when set
  const float* bias = nullptr;   // [N]   (or [M] when bias_per_row)
  const float* bias2 = nullptr;  // [B][ld_bias2], indexed by the batch of row m (time-embedding add)
  const f16* res = nullptr;      // residual [M][ldres]
  float* workspace = nullptr;    // split-K partials, >= splitk*M*N floats
  size_t workspace_bytes = 0;
  int M = 0, N = 0, K = 0;
  int B = 1, Hi = 1, Wi = 1, Cin = 0;  // source tensor geometry (Cin = channels consumed per tap)
  int Ho = 1, Wo = 1, R = 1, S = 1, stride = 1, pad = 0;
  int ups = 0;                 // 1: the conv sees the source nearest-upsampled x2 (Upsample folded in)
  int ldx = 0, ldw = 0, ldy = 0, ldres = 0, ld_bias2 = 0;
  int act = 0;                 // 0 none, 1 SiLU, 2 quick-GELU  v * sigmoid(1.702 v)  (CLIP MLP),
                               // 3 GEGLU pair: W rows interleaved value/gate in blocks of 16, y gets N/2 columns v * gelu(g)
  int bias_per_row = 0;
  float scale = 1.0f;
  int force_tile = -1;         // testing hook: tile config index
  int force_splitk = 0;
  // LayerNorm folded into the GEMM (see KP::ln_stats in conv_inl.h): x is used raw, w = W * gamma, bias = b + W beta,
  // ln_s[n] = sum_k w[n][k]; ln_stats = the per-row partial (sum, sumsq) the producer of x emitted (ln_strips valid of ln_ld)
  const float* ln_stats = nullptr;
  const float* ln_s = nullptr;
  int ln_strips = 0, ln_ld = 0;
  int ln_c = 0;                // channels normalised over (= K for rows mode)
  float ln_eps = 1e-5f;
  // fp8 weights (OCP e4m3fn): `w` points at [N][K] BYTES (ldw in bytes), wscale[N] = per-output-channel scale (a power of two:
  // the product is then bit-identical to the fp16 kernel run on the dequantised weights).  Implicit-GEMM DMA kernel only.
  const float* wscale = nullptr;
  // emit per-row partial (sum, sumsq) of the stored fp16 values: stats_out[m][stats_ld][2], conv_gemm_stats_strips(p) valid
  float* stats_out = nullptr;
  int stats_ld = 0;
  // emit GroupNorm partials of the stored fp16 values: gn_out[b][gn_slots][gn_groups][2] = (sum, sumsq) of the gn_cpg channels of a
  // group over the rows of one M tile of image b; gn_slots must equal conv_gemm_gn_slots(p, gn_cpg) (> 0)
  float* gn_out = nullptr;
  int gn_cpg = 0, gn_slots = 0, gn_groups = 0;
  // GroupNorm(32 groups) + optional SiLU of the INPUT applied inside the conv (halo-reuse 3x3 kernels only, conv_gemm_gn_in_ok):
  // x is the raw tensor, gn_in the partials the producer of x emitted ([B][gn_in_slots][32][2], ConvGemm::gn_out of that launch)
  const float* gn_in = nullptr;
  const float* gn_gamma = nullptr;
  const float* gn_beta = nullptr;
  int gn_in_slots = 0, gn_in_silu = 0;
  float gn_in_eps = 1e-5f;
  // block-scaled fp8 GEMM (both operands OCP e4m3fn codes + one e8m0 scale byte per 32 codes of a row, quantize_mx): x and w point at
  // BYTES, K / ldx / ldw count codes, K % 128 == 0, R = S = 1.  Runs on v_mfma_scale_f32_16x16x128_f8f6f4.
  const uint8_t* mx_sx = nullptr;  // [M][mx_ldsx]
  const uint8_t* mx_sw = nullptr;  // [N][mx_ldsw]
  int mx_ldsx = 0, mx_ldsw = 0;
};
int conv_gemm(const ConvGemm& p, hipStream_t stream);
// M tiles per image of the plan chosen for p when its epilogue can emit GroupNorm partials for groups of cpg channels, else 0
int conv_gemm_gn_slots(const ConvGemm& p, int cpg);
// whether the plan chosen for p can apply a GroupNorm(32) of its input itself (ConvGemm::gn_in): a halo-reuse 3x3 kernel with LDS left
// for the (a, b) table of a workgroup's channel range
bool conv_gemm_gn_in_ok(const ConvGemm& p);
// whether the plan chosen for p is a halo-reuse 3x3 kernel (which has no fp8-weight variant)
bool conv_gemm_plan_is_halo(const ConvGemm& p);
// strips (partials per row) a launch of p writes to stats_out; 0 when the chosen plan cannot emit them (split-K)
int conv_gemm_stats_strips(const ConvGemm& p);
// measurement only: phase stamps of the last launch of an implicit-GEMM / halo kernel (SDEO_DBG_GEMM bit 6), see conv_inl.h
int conv_gemm_read_stamps(unsigned long long* out, int n);
int conv_halo_read_stamps(unsigned long long* out, int n);
size_t conv_gemm_workspace_bytes(const ConvGemm& p);
int conv_gemm_plan_splitk(const ConvGemm& p);       // split-K factor of the plan chosen for p
// name of the kernel instantiation the launcher will pick (for profiles; matches the rocprof kernel name's template args)
const char* conv_gemm_kernel_name(const ConvGemm& p);
void conv_gemm_debug_force(int tile, int splitk);
void conv_gemm_debug_force_order(int order);       // -1 heuristic, 0 M-fastest, 1 N-fastest tile order
// one-time on-device plan search for p's shape (p needs valid scratch pointers); workspace to reserve for it
int conv_gemm_autotune(const ConvGemm& p, hipStream_t stream);
size_t conv_gemm_autotune_workspace_bytes(const ConvGemm& p);
void conv_gemm_set_tuned(const int key[10], int tile, int splitk);
std::string conv_gemm_tuned_json();   // [[M,N,K,Cin,R,stride,ups,Hi,Wi,B,tile,splitk],...]   // tuning hook: -1 / 0 restore the heuristic

// ------------------------------------------------------------------------------------------
// GroupNorm (NHWC fp16, fp32 statistics, eps honoured) + optional SiLU; two launches:
// per-chunk partial sums, then normalise.  `partials` >= B*gn_chunks(HW)*groups*2 floats.
// ------------------------------------------------------------------------------------------
int gn_chunks(int HW);
int groupnorm_nhwc(f16* y, int ldy, const f16* x, int ldx, const float* gamma, const float* beta, int B, int HW, int C,
                   int groups, float eps, int with_silu, float* partials, hipStream_t stream);

struct GnArgs {
  f16* y; const f16* x; const float* gamma; const float* beta; float* partials;
  int ldy, ldx, B, HW, C, groups;
  float eps;
  int with_silu;
  // statistics already exist: ext_partials[b][ext_nsc][groups][2] = (sum, sumsq) partials written by the epilogue of the conv / GEMM
  // that produced x (ConvGemm::gn_out).  Then no statistics pass runs: ONE launch normalises (plus a small reduction launch when
  // ext_nsc > 128, the VAE's large images: `partials` is its workspace, >= B * groups * 2 floats)
  const float* ext_partials = nullptr;
  int ext_nsc = 0;
};
// out[b][g] = sum over the nsc entries of in[b][.][g] (fixed order): producer-side partials of a large image folded to one entry per group
int groupnorm_fold_partials(float* out, const float* in, int B, int nsc, int groups, hipStream_t stream);
// whether groupnorm_nhwc(a) runs as ONE launch with its slice in LDS
bool groupnorm_is_single_launch(const GnArgs& a);
int groupnorm_nhwc(const GnArgs& a, hipStream_t stream);

// LayerNorm over the last dim of [rows][C] fp16 (two-pass in registers, fp32 math).
int layernorm(f16* y, int ldy, const f16* x, int ldx, const float* gamma, const float* beta, int rows, int C, float eps,
              hipStream_t stream);

// ------------------------------------------------------------------------------------------
// Fused attention  O = softmax(Q K^T * scale) V   (flash-style, scores never materialised)
//   Q[(b*Tq+t)*ldq + h*d + i], K[(b*TkS+j)*ldk + h*d + i], V[(b*TkSv+j)*ldv + h*d + i]   (all row-major)
//   O[(b*Tq+t)*ldo + h*d + i];  keys j >= Tk are masked;  TkS / TkSv = per-batch row strides of K / V.
//   causal: key j additionally masked for query t when j > t (needs Tq == Tk).
// ------------------------------------------------------------------------------------------
int attention(f16* o, int ldo, const f16* q, int ldq, const f16* k, int ldk, const f16* v, int ldv, int B, int H,
              int Tq, int Tk, int TkS, int TkSv, int d, float scale, hipStream_t stream, int causal = 0);
struct AttnArgs {
  f16* o; const f16* q; const f16* k; const f16* v;
  int ldo, ldq, ldk, ldv, B, H, Tq, Tk, TkS, TkSv, d;
  float scale;
  int causal;
};
int attention(const AttnArgs& a, hipStream_t stream);
// vt[c][b*TkSv + t] = v[(b*T + t)*ldv + c]   (VAE AttnBlock's materialised-score path)
int transpose_pad(f16* vt, int ldvt, const f16* v, int ldv, int B, int T, int TkSv, int C, hipStream_t stream);

// row softmax: fp32 scores [rows][ld] -> fp16 probabilities (VAE single-head attention)
int softmax_rows(f16* p, int ldp, const float* s, int lds, int rows, int cols, float scale, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// elementwise / layout kernels
// ------------------------------------------------------------------------------------------
// y[m][0:C] = a[m][0:C] * gelu(a[m][C:2C])      (GEGLU, exact erf GELU)
int geglu(f16* y, int ldy, const f16* a, int lda, int rows, int C, hipStream_t stream);
// y = a + b*scale (b may be null), fp16, strided rows of C channels
int add_scaled(f16* y, int ldy, const f16* a, int lda, const f16* b, int ldb, float scale, int rows, int C,
               hipStream_t stream);
// sinusoidal timestep embedding (util.py:154-174): out[b][dim] fp16
int timestep_embedding(f16* out, const int64_t* t, int B, int dim, hipStream_t stream);
int silu(f16* y, const f16* x, int64_t n, hipStream_t stream);
// layout / dtype conversion at the NCHW boundary
int nchw_f32_to_nhwc_f16(f16* y, int ldy, const float* x, int B, int C, int HW, float scale, hipStream_t stream);
// context [B][T][C] fp32 -> fp16 [B][Tpad][C], rows >= T zero
int pad_rows_f32_to_f16(f16* y, const float* x, int B, int T, int Tpad, int C, hipStream_t stream);
int nhwc_f16_to_nchw_f32(float* y, const f16* x, int ldx, int B, int C, int HW, float scale, hipStream_t stream);
int nhwc_f16_to_nhwc_u8(uint8_t* y, const f16* x, int ldx, int64_t pixels, int C, hipStream_t stream);
// weights: fp32 [O][I][R][S] -> fp16 [O][R][S][Ipad]
int oihw_f32_to_ohwi_f16(f16* y, const float* w, int O, int I, int R, int S, int Ipad, hipStream_t stream);
int zero_f16(f16* y, int64_t n, hipStream_t stream);
int f32_to_f16(f16* y, const float* x, int64_t n, hipStream_t stream);
int f16_to_f32(float* y, const f16* x, int64_t n, hipStream_t stream);
// GEGLU projection [2H][cols] fp32 -> fp16 with value / gate rows interleaved in blocks of 16 (see conv_gemm act = 3); bias variant
int geglu_interleave_f32_to_f16(f16* y, const float* x, int H, int cols, hipStream_t stream);
int geglu_interleave_f32(float* y, const float* x, int H, hipStream_t stream);
// LayerNorm folded into a Linear (KP::ln_stats): w_out = fp16(w * gamma), s = row sums of w_out, b_out = bias + w beta
int fold_layernorm(f16* w_out, float* s_out, float* b_out, const f16* w, const float* gamma, const float* beta,
                   const float* bias, int rows, int C, hipStream_t stream);
// stats[r][ld][2] <- one (sum, sumsq) partial per row of x [rows][C]
int row_stats(float* stats, int ld, const f16* x, int ldx, int rows, int C, hipStream_t stream);
// fp8 weight pack: q[r][0:cols] = e4m3fn codes of w[r] / scale[r] (scale = power of two, absmax / scale <= 448) and w <- code * scale
int quantize_fp8_rows(uint8_t* q, float* scale, f16* w, int rows, int cols, int ldw, int ldq, hipStream_t stream);
int row_sums_f16(float* s_out, const f16* w, int rows, int C, hipStream_t stream);
// [ (Wp W2) | Wp ] and Wp b2 + bp: ff.net.2 and proj_out of a SpatialTransformer as one Linear over [g | t] (elementwise.hip)
int compose_proj(f16* w_out, float* b_out, const f16* wp, const float* bp, const f16* w2, const float* b2, int C, int K2,
                 hipStream_t stream);
// block-scaled fp8 pack of a [rows][cols] fp16 matrix (cols % 32 == 0): per 32 consecutive elements of a row one e8m0 scale byte
// (2^(s - 127): the smallest power of two with amax / scale <= 448; 127 for an all-zero block) and 32 e4m3fn codes of x / scale
int quantize_mx(uint8_t* q, uint8_t* scales, const f16* x, int rows, int cols, int ldx, int ldq, int lds, hipStream_t stream);
// CLIP text embeddings: out[(b*T + t)][0:W] = tok_emb[ids[b*T + t]][0:W] + pos_emb[t][0:W]   (ids are clamped to [0, vocab))
int embed_tokens(f16* out, const int32_t* ids, const f16* tok_emb, const f16* pos_emb, int B, int T, int W, int vocab,
                 hipStream_t stream);
// cv2.Canny(img, low, high) (aperture 3, L1 magnitude) on an HWC uint8 image of 1..4 channels -> edges [H][W] uint8 0 / 255 and / or
// control [3][H][W] fp32 = edges / 255.  Asynchronous, capturable (hysteresis = union-find labelling).  csrc/canny.hip
size_t canny_workspace_bytes(int H, int W);
int canny_u8(const uint8_t* img, int H, int W, int C, float low_threshold, float high_threshold, uint8_t* edges, float* control,
             void* workspace, size_t workspace_bytes, hipStream_t stream);
// cv2.resize on 8-bit HWC images (csrc/resize.hip); the coefficient tables are built on the host (annotator/util.py)
int resize_lanczos4_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dh, int dw, const int* x0, const short* ax,
                       const int* y0, const short* by, hipStream_t stream);
int resize_area_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dh, int dw, const int* xstart, const int* xidx,
                   const float* xw, const int* ystart, const int* yidx, const float* yw, hipStream_t stream);
int resize_area_fast_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dh, int dw, hipStream_t stream);
// classifier-free guidance + DDIM update on NCHW fp32 latents (ddim_hacked.py:192,208-231)
int cfg_ddim_pair(float* x, float* pred_x0, const f16* eps, int lde, f16* x0, int ld0, int b, int C, int HW, float cfg_scale, float a_t,
                  float a_prev, float sqrt_one_minus_at, hipStream_t stream);
int latent_pair_to_nhwc(f16* x0, int ld0, const float* x, int b, int C, int HW, hipStream_t stream);
int cfg_ddim_step(float* x_prev, float* pred_x0, const float* x, const float* eps_c, const float* eps_u, const float* noise,
                  float cfg_scale, float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at, int64_t n,
                  hipStream_t stream);

}  // namespace sdeo
