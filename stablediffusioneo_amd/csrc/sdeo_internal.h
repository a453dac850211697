/* Private entry points of libsdeo.so: tuning and measurement hooks used by tools/ and a few op tests.  NOT part of the
 * drop-in boundary (include/sdeo.h): nothing a reference-side binding needs is declared here. */
#ifndef SDEO_INTERNAL_H
#define SDEO_INTERNAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* force tile config / split-K of the following conv/GEMM launches (-1, 0 = plan table / heuristic) */
void sdeo_debug_force_gemm_plan(int tile, int splitk);
void sdeo_debug_force_gemm_order(int order); /* -1 heuristic, 0 M-fastest, 1 N-fastest tile order within an XCD */
/* name of the kernel instantiation sdeo_conv2d_nhwc_f16 would launch for this problem (plan table / forced plan / heuristic) */
const char* sdeo_debug_conv2d_kernel_name(int n, int h, int w, int cin, int cout, int ksize, int stride, int upsample2x);
/* measurement builds only (python -m stablediffusioneo_amd.build --debug, SDEO_DBG_GEMM bit 6): per-workgroup phase stamps of
 * the last GEMM (which = 0) / halo conv (1) launch, 8 x uint64 per workgroup in 10 ns units */
int sdeo_debug_read_stamps(int which, unsigned long long* out, int n);
/* y = silu(x) on n fp16 elements: launch-floor probe for tools/launch_floor.py */
int sdeo_debug_silu(void* y, const void* x, int64_t n, void* stream);

/* op-level hooks for the LayerNorm fold (tests): the networks use these paths internally (csrc/net.hip build_attn).
 * fold: w_out = fp16(w * gamma) [rows][c], s_out[rows] = row sums of w_out, b_out[rows] = bias + w beta (bias may be NULL)
 * gemm_stats: sdeo_gemm_f16 (fp16 out) that also writes per-row (sum, sumsq) partials of y: stats fp32 [m][stats_ld][2];
 *   *strips_out = valid partials per row (0: the plan is split-K and cannot emit them; nothing was launched)
 * gemm_ln: y = act(LN(x) w^T + b) given the FOLDED w / s / b and the statistics of x; eps as nn.LayerNorm
 * row_stats: one (sum, sumsq) partial per row of x */
int sdeo_debug_fold_layernorm(void* w_out, float* s_out, float* b_out, const void* w, const float* gamma, const float* beta,
                              const float* bias, int rows, int c, void* stream);
/* [ (Wp W2) | Wp ] (fp16 [c][k2 + c]) and Wp b2 + bp (fp32 [c]): ff.net.2 and proj_out of a SpatialTransformer as one Linear */
int sdeo_debug_compose_proj(void* w_out, float* b_out, const void* wp, const float* bp, const void* w2, const float* b2, int c, int k2,
                            void* stream);
int sdeo_debug_gemm_stats_f16(void* y, int ldy, const void* x, int ldx, const void* w, int ldw, const float* bias, const void* res,
                              int ldres, int m, int n, int k, float* stats, int stats_ld, int* strips_out, void* stream);
int sdeo_debug_gemm_ln_f16(void* y, int ldy, const void* x, int ldx, const void* w_folded, int ldw, const float* ln_s,
                           const float* bias_folded, const float* stats, int stats_ld, int strips, int ln_c, int m, int n, int k, int act, float eps, void* workspace, size_t workspace_bytes, void* stream);
int sdeo_debug_row_stats_f16(float* stats, int stats_ld, const void* x, int ldx, int rows, int c, void* stream);
/* conv3x3 with GroupNorm(32) (+ SiLU) of its input applied inside the kernel from partial (sum, sumsq) [n][slots][32][2] of x;
 * *ok = 0: the plan of this shape cannot (nothing ran) */
int sdeo_debug_conv2d_gnin_f16(void* y, const void* x, const void* w_krsc, const float* bias, int n, int h, int w, int cin, int cout,
                               const float* gamma, const float* beta, const float* partials, int slots, float eps, int with_silu,
                               void* workspace, size_t workspace_bytes, int* ok, void* stream);
/* [conv whose epilogue emits the GroupNorm partials of its output] -> [normalise-only GroupNorm]: the pair csrc/net.hip builds for
 * every conv that feeds a GroupNorm.  *slots = partial entries per image (0: this shape's plan cannot emit them, nothing ran);
 * partials >= n * slots * groups * 2 floats (+ n * groups * 2 when slots > 128) */
int sdeo_debug_conv2d_gn_f16(void* ynorm, void* y, const void* x, const void* w_krsc, const float* bias, const void* res, int n, int h,
                             int w, int cin, int cout, int ksize, int stride, int upsample2x, const float* gamma, const float* beta,
                             int groups, float eps, int with_silu, float* partials, size_t partial_floats, int* slots, void* stream);

/* fp8 weight pack at op level (tests): quantise [rows][cols] fp16 in place to its dequantised values, codes -> q, scales -> scale;
 * sdeo_debug_next_weights_fp8 makes the NEXT sdeo_gemm_f16 / sdeo_conv2d_nhwc_f16 call of this thread stream these codes
 * (K-contiguous bytes, same [N][K] / KRSC layout) instead of its fp16 weight argument */
/* block-scaled fp8 (e4m3fn codes + one e8m0 scale per 32 elements of a row) at op level: the pack of an fp16 [rows][cols] matrix
 * (q_out [rows][cols] bytes, scales_out [rows][cols / 32] bytes) and y = x w^T on two packed operands with v_mfma_scale_f32_16x16x128_f8f6f4 */
int sdeo_debug_quantize_mx(void* q_out, void* scales_out, const void* x, int rows, int cols, void* stream);
int sdeo_debug_gemm_mx_f16(void* y, int ldy, const void* xq, const void* xs, const void* wq, const void* ws, const float* bias, const void* res,
                           int ldres, int m, int n, int k, int act, void* workspace, size_t workspace_bytes, void* stream);
int sdeo_debug_quantize_fp8_rows(void* w_f16_inout, void* q_out, float* scale_out, int rows, int cols, void* stream);
void sdeo_debug_next_weights_fp8(const void* q, const float* scale);

/* GEMM launches of the configured programs (ControlNet, UNet with and without control) that run on the block-scaled fp8 MFMA
 * (sdeo_set_activation_precision(h, 8, ...)) */
int sdeo_debug_mx_launches(sdeo_handle h);

#ifdef __cplusplus
}
#endif
#endif /* SDEO_INTERNAL_H */
