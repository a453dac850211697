// C ABI: error convention and the op-level entry points declared in include/sdeo.h.
#include <stdarg.h>

#include "../../include/sdeo.h"
#include "sdeo_internal.h"
#include "kernels.h"

namespace sdeo {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return 1;
}

}  // namespace sdeo

using namespace sdeo;

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

const char* sdeo_last_error(void) { return g_last_error.c_str(); }
int sdeo_version(void) { return SDEO_ABI_VERSION; }
void sdeo_debug_force_gemm_plan(int tile, int splitk) { conv_gemm_debug_force(tile, splitk); }
void sdeo_debug_force_gemm_order(int order) { conv_gemm_debug_force_order(order); }
void sdeo_set_tuned_gemm_plan(const int* key10, int tile, int splitk) { conv_gemm_set_tuned(key10, tile, splitk); }
const char* sdeo_tuned_gemm_plans_json(void) {
  static std::string s;
  s = conv_gemm_tuned_json();
  return s.c_str();
}
int sdeo_debug_silu(void* y, const void* x, int64_t n, void* stream) { return silu((f16*)y, (const f16*)x, n, S(stream)); }

size_t sdeo_groupnorm_workspace_bytes(int n, int hw, int groups) {
  return (size_t)n * gn_chunks(hw) * groups * 2 * sizeof(float);
}

int sdeo_groupnorm_nhwc_f16(void* y, const void* x, const float* gamma, const float* beta, int n, int h, int w, int c,
                            int groups, float eps, int with_swish, void* workspace, void* stream) {
  return groupnorm_nhwc((f16*)y, c, (const f16*)x, c, gamma, beta, n, h * w, c, groups, eps, with_swish, (float*)workspace,
                        S(stream));
}

static thread_local const void* g_next_q8 = nullptr;
static thread_local const float* g_next_q8_scale = nullptr;
// sdeo_debug_next_weights_fp8 is one-shot: every conv / GEMM entry point disarms it FIRST (before any validation can fail), so a
// rejected call can never leave it armed for an unrelated later one
struct Fp8Arm { const void* q; const float* s; };
static Fp8Arm disarm_fp8() {
  Fp8Arm a{g_next_q8, g_next_q8_scale};
  g_next_q8 = nullptr; g_next_q8_scale = nullptr;
  return a;
}
static void take_fp8(ConvGemm& p, const Fp8Arm& a) {
  if (!a.q) return;
  p.w = (const f16*)a.q; p.wscale = a.s; p.ldw = p.K;
}

static int fill_conv(ConvGemm& p, int n, int h, int w, int cin, int cout, int ksize, int stride, int upsample2x) {
  SDEO_CHECK(ksize == 1 || ksize == 3, "conv2d: ksize %d unsupported", ksize);
  SDEO_CHECK(stride == 1 || stride == 2, "conv2d: stride %d unsupported", stride);
  const int pad = ksize / 2;
  const int hv = upsample2x ? 2 * h : h, wv = upsample2x ? 2 * w : w;
  p.B = n; p.Hi = h; p.Wi = w; p.Cin = cin;
  p.R = p.S = ksize; p.stride = stride; p.pad = pad; p.ups = upsample2x;
  p.Ho = (hv + 2 * pad - ksize) / stride + 1;
  p.Wo = (wv + 2 * pad - ksize) / stride + 1;
  p.M = n * p.Ho * p.Wo; p.N = cout; p.K = ksize * ksize * cin;
  p.ldx = cin; p.ldw = p.K; p.ldy = cout; p.ldres = cout; p.ld_bias2 = cout;
  return 0;
}

size_t sdeo_conv2d_workspace_bytes(int n, int h, int w, int cin, int cout, int ksize, int stride, int upsample2x) {
  ConvGemm p;
  if (fill_conv(p, n, h, w, cin, cout, ksize, stride, upsample2x)) return 0;
  return conv_gemm_workspace_bytes(p);
}

size_t sdeo_canny_workspace_bytes(int h, int w) { return canny_workspace_bytes(h, w); }

int sdeo_canny_u8(const uint8_t* img_hwc, int h, int w, int c, float low_threshold, float high_threshold, uint8_t* edges,
                  float* control_chw, void* workspace, size_t workspace_bytes, void* stream) {
  return canny_u8(img_hwc, h, w, c, low_threshold, high_threshold, edges, control_chw, workspace, workspace_bytes, S(stream));
}

int sdeo_resize_lanczos4_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dst_h, int dst_w, const int32_t* x_first_tap,
                            const int16_t* x_coeffs, const int32_t* y_first_tap, const int16_t* y_coeffs, void* stream) {
  return resize_lanczos4_u8(dst, src, h, w, c, dst_h, dst_w, x_first_tap, x_coeffs, y_first_tap, y_coeffs, S(stream));
}

int sdeo_resize_area_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dst_h, int dst_w, const int32_t* x_start,
                        const int32_t* x_index, const float* x_weight, const int32_t* y_start, const int32_t* y_index,
                        const float* y_weight, void* stream) {
  return resize_area_u8(dst, src, h, w, c, dst_h, dst_w, x_start, x_index, x_weight, y_start, y_index, y_weight, S(stream));
}

int sdeo_resize_area_fast_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dst_h, int dst_w, void* stream) {
  return resize_area_fast_u8(dst, src, h, w, c, dst_h, dst_w, S(stream));
}

int sdeo_debug_read_stamps(int which, unsigned long long* out, int n) {
  return which ? conv_halo_read_stamps(out, n) : conv_gemm_read_stamps(out, n);
}

const char* sdeo_debug_conv2d_kernel_name(int n, int h, int w, int cin, int cout, int ksize, int stride, int upsample2x) {
  ConvGemm p;
  if (fill_conv(p, n, h, w, cin, cout, ksize, stride, upsample2x)) return "";
  return conv_gemm_kernel_name(p);
}

int sdeo_conv2d_nhwc_f16(void* y, const void* x, const void* w_krsc, const float* bias, const float* bias2, const void* res,
                         int n, int h, int w, int cin, int cout, int ksize, int stride, int upsample2x, int act, float scale,
                         void* workspace, size_t workspace_bytes, void* stream) {
  const Fp8Arm arm = disarm_fp8();
  ConvGemm p;
  if (int rc = fill_conv(p, n, h, w, cin, cout, ksize, stride, upsample2x)) return rc;
  p.x = (const f16*)x; p.w = (const f16*)w_krsc; p.y = (f16*)y; p.bias = bias; p.bias2 = bias2; p.res = (const f16*)res;
  p.act = act; p.scale = scale; p.workspace = (float*)workspace; p.workspace_bytes = workspace_bytes;
  take_fp8(p, arm);
  return conv_gemm(p, S(stream));
}

// conv3x3 that applies GroupNorm(groups = 32) (+ SiLU) of its INPUT itself (ConvGemm::gn_in) from per-(image, slot, group) partial
// (sum, sumsq) of x.  *ok = 0 when the plan of this shape cannot (nothing is launched then).
int sdeo_debug_conv2d_gnin_f16(void* y, const void* x, const void* w_krsc, const float* bias, int n, int h, int w, int cin, int cout,
                               const float* gamma, const float* beta, const float* partials, int slots, float eps, int with_silu,
                               void* workspace, size_t workspace_bytes, int* ok, void* stream) {
  (void)disarm_fp8();
  SDEO_CHECK(ok && partials && gamma && beta, "conv2d_gnin: null argument");
  ConvGemm p;
  if (int rc = fill_conv(p, n, h, w, cin, cout, 3, 1, 0)) return rc;
  p.x = (const f16*)x; p.w = (const f16*)w_krsc; p.y = (f16*)y; p.bias = bias;
  p.workspace = (float*)workspace; p.workspace_bytes = workspace_bytes;
  *ok = conv_gemm_gn_in_ok(p) ? 1 : 0;
  if (!*ok) return 0;
  p.gn_in = partials; p.gn_in_slots = slots; p.gn_gamma = gamma; p.gn_beta = beta; p.gn_in_eps = eps; p.gn_in_silu = with_silu;
  return conv_gemm(p, S(stream));
}

// conv2d whose epilogue emits the GroupNorm partials of its output, followed by the normalise-only GroupNorm that consumes them
// (the [conv -> GroupNorm] pairs of the networks, csrc/net.hip).  *slots = entries per image the plan writes; 0 = this shape's plan
// cannot emit them (nothing is launched then).  partials >= n * slots * groups * 2 floats (+ n * groups * 2 when slots > 128).
int sdeo_debug_conv2d_gn_f16(void* ynorm, void* y, const void* x, const void* w_krsc, const float* bias, const void* res, int n, int h,
                             int w, int cin, int cout, int ksize, int stride, int upsample2x, const float* gamma, const float* beta,
                             int groups, float eps, int with_silu, float* partials, size_t partial_floats, int* slots, void* stream) {
  (void)disarm_fp8();
  SDEO_CHECK(slots && groups > 0 && cout % groups == 0, "conv2d_gn: bad arguments");
  ConvGemm p;
  if (int rc = fill_conv(p, n, h, w, cin, cout, ksize, stride, upsample2x)) return rc;
  p.x = (const f16*)x; p.w = (const f16*)w_krsc; p.y = (f16*)y; p.bias = bias; p.res = (const f16*)res;
  const int cpg = cout / groups;
  *slots = conv_gemm_gn_slots(p, cpg);
  if (*slots <= 0) { *slots = 0; return 0; }
  const size_t need = (size_t)n * *slots * groups * 2, fold = *slots > 128 ? (size_t)n * groups * 2 : 0;
  SDEO_CHECK(partials && partial_floats >= need + fold, "conv2d_gn: partials buffer too small");
  p.gn_out = partials; p.gn_cpg = cpg; p.gn_slots = *slots; p.gn_groups = groups;
  if (int rc = conv_gemm(p, S(stream))) return rc;
  GnArgs g{(f16*)ynorm, (const f16*)y, gamma, beta, partials + need, cout, cout, n, p.Ho * p.Wo, cout, groups, eps, with_silu};
  g.ext_partials = partials; g.ext_nsc = *slots;
  return groupnorm_nhwc(g, S(stream));
}

static void fill_gemm(ConvGemm& p, int m, int n, int k) {
  p.B = m; p.Hi = p.Wi = p.Ho = p.Wo = 1; p.Cin = k; p.R = p.S = 1; p.stride = 1; p.pad = 0;
  p.M = m; p.N = n; p.K = k;
}

size_t sdeo_gemm_workspace_bytes(int m, int n, int k) {
  ConvGemm p;
  fill_gemm(p, m, n, k);
  return conv_gemm_workspace_bytes(p);
}

int sdeo_gemm_f16(void* y, int ldy, const void* x, int ldx, const void* w, int ldw, const float* bias, const void* res,
                  int ldres, int m, int n, int k, int act, float scale, int out_f32, int bias_per_row, void* workspace,
                  size_t workspace_bytes, void* stream) {
  const Fp8Arm arm = disarm_fp8();
  ConvGemm p;
  fill_gemm(p, m, n, k);
  p.x = (const f16*)x; p.w = (const f16*)w; p.bias = bias; p.res = (const f16*)res;
  if (out_f32) p.y32 = (float*)y; else p.y = (f16*)y;
  p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.ldres = ldres;
  p.act = act; p.scale = scale; p.bias_per_row = bias_per_row;
  p.workspace = (float*)workspace; p.workspace_bytes = workspace_bytes;
  take_fp8(p, arm);
  return conv_gemm(p, S(stream));
}

// block-scaled fp8 ("MX") at op level: pack an fp16 matrix, and the GEMM on two packed operands (v_mfma_scale_f32_16x16x128_f8f6f4)
int sdeo_debug_quantize_mx(void* q_out, void* scales_out, const void* x, int rows, int cols, void* stream) {
  return quantize_mx((uint8_t*)q_out, (uint8_t*)scales_out, (const f16*)x, rows, cols, cols, cols, cols / 32, S(stream));
}
int sdeo_debug_gemm_mx_f16(void* y, int ldy, const void* xq, const void* xs, const void* wq, const void* ws, const float* bias, const void* res,
                           int ldres, int m, int n, int k, int act, void* workspace, size_t workspace_bytes, void* stream) {
  (void)disarm_fp8();
  ConvGemm p;
  fill_gemm(p, m, n, k);
  p.x = (const f16*)xq; p.w = (const f16*)wq; p.y = (f16*)y; p.bias = bias; p.res = (const f16*)res;
  p.ldx = k; p.ldw = k; p.ldy = ldy; p.ldres = ldres; p.act = act;
  p.mx_sx = (const uint8_t*)xs; p.mx_sw = (const uint8_t*)ws; p.mx_ldsx = k / 32; p.mx_ldsw = k / 32;
  p.workspace = (float*)workspace; p.workspace_bytes = workspace_bytes;
  return conv_gemm(p, S(stream));
}

int sdeo_debug_quantize_fp8_rows(void* w_f16_inout, void* q_out, float* scale_out, int rows, int cols, void* stream) {
  return quantize_fp8_rows((uint8_t*)q_out, scale_out, (f16*)w_f16_inout, rows, cols, cols, cols, S(stream));
}
void sdeo_debug_next_weights_fp8(const void* q, const float* scale) { g_next_q8 = q; g_next_q8_scale = scale; }

int sdeo_debug_fold_layernorm(void* w_out, float* s_out, float* b_out, const void* w, const float* gamma, const float* beta,
                              const float* bias, int rows, int c, void* stream) {
  return fold_layernorm((f16*)w_out, s_out, b_out, (const f16*)w, gamma, beta, bias, rows, c, S(stream));
}

int sdeo_debug_compose_proj(void* w_out, float* b_out, const void* wp, const float* bp, const void* w2, const float* b2, int c, int k2,
                            void* stream) {
  return compose_proj((f16*)w_out, b_out, (const f16*)wp, bp, (const f16*)w2, b2, c, k2, S(stream));
}

int sdeo_debug_gemm_stats_f16(void* y, int ldy, const void* x, int ldx, const void* w, int ldw, const float* bias, const void* res,
                              int ldres, int m, int n, int k, float* stats, int stats_ld, int* strips_out, void* stream) {
  SDEO_CHECK(stats && strips_out, "gemm_stats: null argument");
  ConvGemm p;
  fill_gemm(p, m, n, k);
  p.x = (const f16*)x; p.w = (const f16*)w; p.bias = bias; p.res = (const f16*)res; p.y = (f16*)y;
  p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.ldres = ldres;
  *strips_out = conv_gemm_stats_strips(p);
  if (*strips_out == 0) return 0;
  SDEO_CHECK(*strips_out <= stats_ld, "gemm_stats: %d strips do not fit stats_ld %d", *strips_out, stats_ld);
  p.stats_out = stats; p.stats_ld = stats_ld;
  return conv_gemm(p, S(stream));
}

int sdeo_debug_gemm_ln_f16(void* y, int ldy, const void* x, int ldx, const void* w_folded, int ldw, const float* ln_s,
                           const float* bias_folded, const float* stats, int stats_ld, int strips, int ln_c, int m, int n, int k, int act, float eps, void* workspace, size_t workspace_bytes, void* stream) {
  ConvGemm p;
  fill_gemm(p, m, n, k);
  p.x = (const f16*)x; p.w = (const f16*)w_folded; p.bias = bias_folded; p.y = (f16*)y;
  p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.act = act;
  p.ln_stats = stats; p.ln_s = ln_s; p.ln_strips = strips; p.ln_ld = stats_ld; p.ln_c = ln_c; p.ln_eps = eps;
  p.workspace = (float*)workspace; p.workspace_bytes = workspace_bytes;
  return conv_gemm(p, S(stream));
}

int sdeo_debug_row_stats_f16(float* stats, int stats_ld, const void* x, int ldx, int rows, int c, void* stream) {
  return row_stats(stats, stats_ld, (const f16*)x, ldx, rows, c, S(stream));
}

int sdeo_layernorm_f16(void* y, const void* x, const float* gamma, const float* beta, int rows, int c, float eps,
                       void* stream) {
  return layernorm((f16*)y, c, (const f16*)x, c, gamma, beta, rows, c, eps, S(stream));
}

int sdeo_attention_f16(void* o, int ldo, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int b,
                       int heads, int tq, int tk, int tk_stride, int vt_batch_stride, int d, float scale, void* stream) {
  return attention((f16*)o, ldo, (const f16*)q, ldq, (const f16*)k, ldk, (const f16*)v, ldv, b, heads, tq, tk, tk_stride,
                   vt_batch_stride, d, scale, S(stream));
}

int sdeo_attention_causal_f16(void* o, int ldo, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int b,
                              int heads, int tq, int tk, int tk_stride, int vt_batch_stride, int d, float scale, void* stream) {
  return attention((f16*)o, ldo, (const f16*)q, ldq, (const f16*)k, ldk, (const f16*)v, ldv, b, heads, tq, tk, tk_stride,
                   vt_batch_stride, d, scale, S(stream), 1);
}

int sdeo_geglu_f16(void* y, const void* a, int rows, int c, void* stream) {
  return geglu((f16*)y, c, (const f16*)a, 2 * c, rows, c, S(stream));
}

int sdeo_timestep_embedding_f16(void* out, const int64_t* t, int b, int dim, void* stream) {
  return timestep_embedding((f16*)out, t, b, dim, S(stream));
}

int sdeo_cfg_ddim_step(float* x_prev, float* pred_x0, const float* x, const float* eps_c, const float* eps_u,
                       const float* noise, float cfg_scale, float a_t, float a_prev, float sigma_t, float sqrt_one_minus_at,
                       int64_t n, void* stream) {
  return cfg_ddim_step(x_prev, pred_x0, x, eps_c, eps_u, noise, cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, n,
                       S(stream));
}

int sdeo_nchw_f32_to_nhwc_f16(void* y, int ldy, const float* x, int n, int c, int hw, void* stream) {
  return nchw_f32_to_nhwc_f16((f16*)y, ldy, x, n, c, hw, 1.0f, S(stream));
}

int sdeo_nhwc_f16_to_nchw_f32(float* y, const void* x, int ldx, int n, int c, int hw, float scale, void* stream) {
  return nhwc_f16_to_nchw_f32(y, (const f16*)x, ldx, n, c, hw, scale, S(stream));
}

int sdeo_oihw_f32_to_krsc_f16(void* y, const float* w, int o, int i, int r, int s, int i_pad, void* stream) {
  return oihw_f32_to_ohwi_f16((f16*)y, w, o, i, r, s, i_pad, S(stream));
}

}  // extern "C"
