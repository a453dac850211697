/* libsdeo -- C ABI of the MI355X-native ControlNet / Stable-Diffusion hot path.
 *
 * This is the drop-in boundary for the reference's TensorRT seam.  Paths below are relative to the
 * reference tree (MarToonLi/StableDiffusionEO).
 *
 *   reference seam                                               replaced by
 *   ------------------------------------------------------------ ------------------------------------
 *   Engine.load / activate / allocate_buffers (Engine.py:99-121)  sdeo_create, sdeo_load_weight,
 *                                                                 sdeo_finalize_weights, sdeo_configure
 *   context.set_tensor_address + execute_async_v3                 sdeo_controlnet_forward, sdeo_unet_forward,
 *     (Engine.py:136-137,145,155-157) for ControlNet.plan /       sdeo_vae_decode, sdeo_apply_model
 *     ControlledUnet.plan / Decoder.plan
 *   cudaStreamBeginCapture / cudaGraphLaunch (Engine.py:139-152)  every call is capturable by the caller's hipGraph
 *   p_sample_ddim tail in torch (cldm/ddim_hacked.py:192,208-231) sdeo_cfg_ddim_step
 *   GroupNormPlugin::enqueue (plugin/groupNormPlugin/              sdeo_groupnorm_nhwc_f16
 *     groupNormPlugin.cpp:179-228), libplugin.so via ctypes.CDLL
 *     (onnx2trt_static_plugin.py:7-10)
 *   CUASSERT / "ERROR: inference failed." (Engine.py:38-44,146)   int return codes + sdeo_last_error()
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; sdeo_last_error() returns the message of
 *     the calling thread's last failure (the Python Engine maps non-zero to the reference's
 *     ValueError("ERROR: inference failed.")).
 *   - all pointers are raw DEVICE pointers unless a parameter name starts with `host_`; `stream` is a
 *     hipStream_t passed as void* (NULL = default stream).  Nothing here synchronises, allocates or frees
 *     caller memory, so every call is hipGraph-capturable.
 *   - fp16 activations are NHWC ("pixel rows" of `ld*` elements); the NCHW fp32 tensors of the reference
 *     surface are converted at the net-level entry points.
 *   - a handle is not thread-safe (same as a TensorRT execution context); one handle per GPU / process.
 */
#ifndef SDEO_H
#define SDEO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdeo_handle_s* sdeo_handle;

const char* sdeo_last_error(void);
/* ABI version: bumped whenever the meaning of an argument of an existing entry point changes (not only its C type), so that a binding
 * built against an older header fails loudly instead of computing with re-interpreted operands.
 *   100  first release
 *   101  sdeo_attention_f16 / sdeo_attention_causal_f16: arguments 7, 8, 14 are V ROW-MAJOR with its row stride and per-batch row count
 *        (they were V transposed, ldvt and the batch stride of V^T) */
#define SDEO_ABI_VERSION 101
int sdeo_version(void);
/* GEMM plan table: (tile, split-K) per problem shape key {M,N,K,Cin,R,stride,ups,Hi,Wi,B}.  Known shapes come from the
 * table, which stablediffusioneo_amd/tuned_plans_gfx950.json pre-loads so that runs are reproducible and start fast;
 * unknown shapes use the deterministic heuristic (SDEO_AUTOTUNE=1 measures them on the device at sdeo_configure instead:
 * a tuning aid, plans and hence fp32 summation orders may then differ from run to run). */
void sdeo_set_tuned_gemm_plan(const int* key10, int tile, int splitk);
const char* sdeo_tuned_gemm_plans_json(void);

/* ------------------------------------------------------------------ op-level entry points (used by tests)

 * GroupNorm(+Swish) on NHWC fp16 -- operand contract of the reference plugin (x/y fp16 NHWC, gamma/beta fp32,
 * attrs epsilon + bSwish, groupNormPlugin.cpp:136-160,291-292) with epsilon actually applied.
 * workspace: >= sdeo_groupnorm_workspace_bytes(n, h*w, groups) bytes. */
size_t sdeo_groupnorm_workspace_bytes(int n, int hw, int groups);
int sdeo_groupnorm_nhwc_f16(void* y, const void* x, const float* gamma, const float* beta, int n, int h, int w, int c,
                            int groups, float eps, int with_swish, void* workspace, void* stream);

/* conv2d on NHWC fp16 activations with KRSC fp16 weights (implicit GEMM on MFMA).
 *   y[n][ho][wo][cout] = act( conv(x, w) + bias[cout] + bias2[n][cout] ) * scale + res[n][ho][wo][cout]
 * cin/cout are the STORED channel counts (cin % 8 == 0, cout % 4 == 0); upsample2x=1 folds a nearest x2
 * upsample of x in front of the conv.  bias/bias2 fp32 or NULL, res fp16 NHWC or NULL. act: 0 none, 1 SiLU.
 * workspace (split-K partials): >= sdeo_conv2d_workspace_bytes(...) bytes, may be NULL when that is 0. */
size_t sdeo_conv2d_workspace_bytes(int n, int h, int w, int cin, int cout, int ksize, int stride, int upsample2x);
int sdeo_conv2d_nhwc_f16(void* y, const void* x, const void* w_krsc, const float* bias, const float* bias2, const void* res,
                         int n, int h, int w, int cin, int cout, int ksize, int stride, int upsample2x, int act, float scale,
                         void* workspace, size_t workspace_bytes, void* stream);

/* GEMM  y[m][n] = act(x[m][k] . w[n][k]^T + bias) * scale + res   (both operands K-contiguous; nn.Linear layout).
 * out_f32=1 writes fp32.  bias_per_row=1 indexes bias by m.  ld* in elements. */
size_t sdeo_gemm_workspace_bytes(int m, int n, int k);
int sdeo_gemm_f16(void* y, int ldy, const void* x, int ldx, const void* w, int ldw, const float* bias, const void* res,
                  int ldres, int m, int n, int k, int act, float scale, int out_f32, int bias_per_row, void* workspace,
                  size_t workspace_bytes, void* stream);

/* LayerNorm over the last dim of [rows][c] fp16, fp32 gamma/beta. */
int sdeo_layernorm_f16(void* y, const void* x, const float* gamma, const float* beta, int rows, int c, float eps,
                       void* stream);

/* Fused attention O = softmax(Q K^T scale) V.  Q [b][tq][ldq], K [b][tk_stride][ldk], V [b][v_batch_stride][ldv] (all
 * row-major, head h at column h*d: column blocks of one fused q/k/v projection work as they are);  O [b][tq][ldo].
 * Keys >= tk are masked.  Operands 16-byte aligned, ld* multiples of 8.  Head dim d: a multiple of 8 up to 160 (the UNet /
 * ControlNet / CLIP heads), or 256 / 512 (the VAE AttnBlock's single head, `ldm/modules/diffusionmodules/model.py:179-203`:
 * the four waves of a workgroup split the channels); scores are never materialised for any of them. */
int sdeo_attention_f16(void* o, int ldo, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int b,
                       int heads, int tq, int tk, int tk_stride, int v_batch_stride, int d, float scale, void* stream);

/* as sdeo_attention_f16 with the causal mask of the CLIP text transformer: key j is masked for query t when j > t
 * (needs tq == tk) */
int sdeo_attention_causal_f16(void* o, int ldo, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int b,
                              int heads, int tq, int tk, int tk_stride, int v_batch_stride, int d, float scale,
                              void* stream);

/* GEGLU: y[r][0:c] = a[r][0:c] * gelu_erf(a[r][c:2c]) */
int sdeo_geglu_f16(void* y, const void* a, int rows, int c, void* stream);

/* sinusoidal timestep embedding (util.py:154-174): t int64[b] -> out fp16 [b][dim] */
int sdeo_timestep_embedding_f16(void* out, const int64_t* t, int b, int dim, void* stream);

/* classifier-free guidance + DDIM update on fp32 latents (ddim_hacked.py:192,208-231).
 * eps_u may be NULL (no guidance), noise may be NULL (eta = 0), pred_x0 may be NULL. */
int sdeo_cfg_ddim_step(float* x_prev, float* pred_x0, const float* x, const float* eps_c, const float* eps_u,
                       const float* noise, float cfg_scale, float a_t, float a_prev, float sigma_t,
                       float sqrt_one_minus_at, int64_t n, void* stream);

/* Canny edge map: replaces `cv2.Canny(img, low_threshold, high_threshold)` of annotator/canny/__init__.py:4-6
 * (aperture 3, L1 gradient magnitude; called at canny2image_torch.py:33 on the HWC3 uint8 image) and the control preparation of
 * canny2image_torch.py:34-38.  img_hwc: device uint8 [h][w][c], c in 1..4.  edges (optional): device uint8 [h][w], 0 / 255.
 * control_chw (optional): device fp32 [3][h][w] = edges / 255 on three identical channels (HWC3 + /255 + HWC->CHW).
 * workspace: >= sdeo_canny_workspace_bytes(h, w) bytes of device memory, 4-byte aligned.  Asynchronous on `stream` and
 * hipGraph-capturable like every other entry point (hysteresis is a union-find labelling, not a host-driven iteration). */
size_t sdeo_canny_workspace_bytes(int h, int w);
int sdeo_canny_u8(const uint8_t* img_hwc, int h, int w, int c, float low_threshold, float high_threshold, uint8_t* edges,
                  float* control_chw, void* workspace, size_t workspace_bytes, void* stream);

/* cv2.resize(img, (dst_w, dst_h), interpolation) of annotator/util.py:37 for 8-bit HWC images (c in 1..4), both on the device.
 * The per-axis coefficient tables depend only on the sizes and are built by the host (stablediffusioneo_amd/annotator/util.py):
 *   INTER_LANCZOS4: per destination index the first of its 8 taps (may be negative: indices are clamped to the image) and the
 *     8 coefficients as OpenCV stores them (short, x 2048);
 *   INTER_AREA: per destination index a run [start[d], start[d+1]) of (source index, float weight) pairs. */
int sdeo_resize_lanczos4_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dst_h, int dst_w, const int32_t* x_first_tap,
                            const int16_t* x_coeffs, const int32_t* y_first_tap, const int16_t* y_coeffs, void* stream);
int sdeo_resize_area_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dst_h, int dst_w, const int32_t* x_start,
                        const int32_t* x_index, const float* x_weight, const int32_t* y_start, const int32_t* y_index,
                        const float* y_weight, void* stream);
/* INTER_AREA when both axes shrink by INTEGER factors (h % dst_h == 0, w % dst_w == 0; error otherwise): OpenCV's resizeAreaFast_
 * arithmetic -- 2 x 2 cells (a + b + c + d + 2) >> 2, other cells the integer cell sum times the float 1 / area, ties to even.  No
 * tables.  (cv2.resize picks this path by itself; stablediffusioneo_amd/annotator/util.py: resize_u8 does the same.) */
int sdeo_resize_area_fast_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dst_h, int dst_w, void* stream);

/* layout helpers at the NCHW boundary */
int sdeo_nchw_f32_to_nhwc_f16(void* y, int ldy, const float* x, int n, int c, int hw, void* stream);
int sdeo_nhwc_f16_to_nchw_f32(float* y, const void* x, int ldx, int n, int c, int hw, float scale, void* stream);
int sdeo_oihw_f32_to_krsc_f16(void* y, const float* w, int o, int i, int r, int s, int i_pad, void* stream);

/* ------------------------------------------------------------------ net-level entry points */

typedef struct sdeo_config {
  /* UNet / ControlNet (cldm_v15.yaml values) */
  int in_channels, out_channels, hint_channels, model_channels, num_res_blocks;
  int channel_mult[8];
  int num_levels;
  int attention_resolutions[8];
  int num_attention_resolutions;
  int num_heads, context_dim, context_len;
  /* VAE decoder */
  int vae_ch, vae_out_ch, vae_ch_mult[8], vae_num_levels, vae_num_res_blocks, vae_z_channels;
  float vae_scale_factor;
} sdeo_config;

/* Create / destroy an engine instance on the current HIP device. */
int sdeo_create(const sdeo_config* cfg, sdeo_handle* out);
int sdeo_destroy(sdeo_handle h);

/* Weights: one call per checkpoint tensor, names exactly as in the reference state dict
 * ("model.diffusion_model.*", "control_model.*", "first_stage_model.*"; cldm/model.py:12-21).
 * host_data points to fp32 data in PyTorch layout (OIHW conv, [out][in] linear); it may be a host OR a device
 * pointer (copied with hipMemcpyDefault).
 * Unknown names return 0 and are ignored when `strict` is 0.  After the last tensor call
 * sdeo_finalize_weights (fails listing what is missing). */
int sdeo_load_weight(sdeo_handle h, const char* name, const float* host_data, const int64_t* dims, int ndim, int strict);
int sdeo_finalize_weights(sdeo_handle h);
/* Weight precision of the UNet / ControlNet matrices (the reference's switch is the TensorRT builder flag,
 * onnx2trt_static_plugin.py:40-42): 16 = fp16 (default), 8 = OCP e4m3fn with one power-of-two scale per output channel, packed by
 * sdeo_finalize_weights; activations, accumulation, biases, norms and the VAE stay as they are.  Call before
 * sdeo_finalize_weights.  The weight-bound shapes (<= 512 rows) stream the one-byte codes; every other kernel reads an fp16 copy
 * holding the same dequantised values. */
int sdeo_set_weight_precision(sdeo_handle h, int bits);
/* Activation precision of the large GEMMs (Linear / conv1x1 with >= min_rows rows and K a multiple of 128), meaningful with 8-bit
 * weights only: 16 = fp16 activations (default); 8 = block-scaled fp8 on both sides -- the activations are packed to e4m3fn codes with
 * one e8m0 scale per 32 channels right in front of the GEMM, the matrix is packed the same way by sdeo_finalize_weights, and the
 * product runs on the block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, twice the fp16 rate).  Everything else (3x3 convs,
 * attention, small GEMMs, norms, VAE) stays as with sdeo_set_weight_precision(h, 8).  Call before sdeo_finalize_weights.
 * min_rows <= 0 selects the default threshold (2048). */
int sdeo_set_activation_precision(sdeo_handle h, int bits, int min_rows);
/* Number of expected tensors and the i-th expected name/shape (ndim<=4), for loaders and tests. */
int sdeo_num_weights(sdeo_handle h);
int sdeo_weight_info(sdeo_handle h, int i, const char** name, int64_t dims[4], int* ndim);

/* Fix the problem size (allocate_buffers equivalent): n = images per call of the *_forward functions
 * (the CFG pair counts as 2), latent h x w.  Allocates the activation arena once. */
int sdeo_configure(sdeo_handle h, int n, int latent_h, int latent_w);

/* flags of the *_forward / apply_model entry points.  The hint block depends only on the hint and the
 * cross-attention K / V projections only on the text context, so a sampler that keeps both fixed over the
 * DDIM loop passes the *_CACHED flags after the first step (the corresponding pointers may then be NULL). */
#define SDEO_HINT_CACHED 1
#define SDEO_CONTEXT_CACHED 2
#define SDEO_NO_CONTROL 4 /* apply_model only: the c_concat=None branch (UNet without ControlNet) */
/* The time embedding (timestep_embedding -> time_embed MLP -> every ResBlock's emb_layers, openaimodel.py:777-781, 255-275)
 * depends only on t.  A sampler that knows its schedule hands it over once (sdeo_set_timestep_table) and then passes
 * SDEO_TIMESTEP_ROW(i) instead of a timesteps pointer: every image of the batch runs at timestep i of that table. */
#define SDEO_TIMESTEP_ROW(i) (8 | ((i) << 8))
#define SDEO_MAX_TABLE_STEPS 128

/* ControlNet.forward (cldm/cldm.py:284-305).  NCHW fp32 at the boundary:
 *   x_noisy [n][4][h][w], hint [n][3][8h][8w] in [0,1], timesteps int64 [n], context [n][77][768],
 *   controls[13]: NCHW fp32 outputs in the reference's binding order (export_onnx_all.py:242-256);
 *   entries may be NULL to skip the copy-out. */
int sdeo_controlnet_forward(sdeo_handle h, const float* x_noisy, const float* hint, const int64_t* timesteps,
                            const float* context, float* const* controls, int flags, void* stream);

/* ControlledUnetModel.forward (cldm/cldm.py:22-45).  controls may be NULL (control=None branch);
 * control_scales fp32[13] HOST pointer or NULL (= 1.0), applied as in apply_model (cldm/cldm.py:338);
 * only_mid_control as in the reference. */
int sdeo_unet_forward(sdeo_handle h, const float* x_noisy, const int64_t* timesteps, const float* context,
                      const float* const* controls, const float* host_control_scales, int only_mid_control, float* eps,
                      int flags, void* stream);

/* ControlLDM.apply_model (cldm/cldm.py:328-341) without the NCHW fp32 round trip of the 13 control tensors:
 * ControlNet -> scaled controls -> UNet, all fp16 NHWC inside the arena. */
int sdeo_apply_model(sdeo_handle h, const float* x_noisy, const float* hint, const int64_t* timesteps,
                     const float* context, const float* host_control_scales, int only_mid_control, int flags,
                     float* eps, void* stream);

/* Time-embedding table of a sampling schedule: host_timesteps int64[count] (HOST pointer, count <= SDEO_MAX_TABLE_STEPS), e.g.
 * np.flip(ddim_timesteps) of cldm/ddim_hacked.py:137.  Runs the embedding MLPs of both networks once for all rows; the table stays
 * valid until the next sdeo_configure / sdeo_set_timestep_table.  Not capturable (synchronises `stream`). */
int sdeo_set_timestep_table(sdeo_handle h, const int64_t* host_timesteps, int count, void* stream);

/* One DDIM step of the classifier-free-guidance pair, eta = 0 (p_sample_ddim, cldm/ddim_hacked.py:183-231, with the two
 * apply_model calls batched as [x; x] -- configured n = 2 x latents, conditional half first, as sdeo_apply_model sees it from the
 * sampler): eps = apply_model([x; x]) with the hint block and context K / V cached by an earlier sdeo_apply_model call and the
 * time embedding of row `table_row`; e = eps_u + cfg_scale (eps_c - eps_u); pred_x0 = (x - sqrt(1 - a_t) e) / sqrt(a_t);
 * x <- sqrt(a_prev) pred_x0 + sqrt(1 - a_prev) e.  x [n/2][4][h][w] fp32 is updated IN PLACE, pred_x0 (may be NULL) receives the
 * prediction.  Arithmetic and rounding are those of sdeo_apply_model + sdeo_cfg_ddim_step (bit-identical results); what is saved
 * is the NCHW fp32 round trip of eps and the latent between the two.  flags: SDEO_STEP_LATENT_STAGED = x is exactly what the
 * previous sdeo_ddim_step on this handle left (its fp16 copy is already staged; skips one conversion launch).  Capturable. */
#define SDEO_STEP_LATENT_STAGED 16
int sdeo_ddim_step(sdeo_handle h, float* x, float* pred_x0, int table_row, float cfg_scale, float a_t, float a_prev,
                   float sqrt_one_minus_at, const float* host_control_scales, int only_mid_control, int flags, void* stream);

/* decode_first_stage: z/scale_factor -> post_quant_conv -> Decoder (model.py:619-652).
 * z [n][4][h][w] fp32 NCHW -> images [n][3][8h][8w] fp32 NCHW in [-1,1]; images_u8 (optional, may be NULL)
 * additionally receives the NHWC uint8 post-process of canny2image_torch.py:68. n must be <= configured n. */
int sdeo_vae_decode(sdeo_handle h, const float* z, int n, float* images, uint8_t* images_u8, void* stream);

/* bytes of device memory the handle owns (weights + arena) */
size_t sdeo_device_bytes(sdeo_handle h);

/* ------------------------------------------------------------------ text encoder
 * FrozenCLIPEmbedder.forward (ldm/modules/encoders/modules.py:123-141): token ids -> CLIPTextModel ->
 * last_hidden_state [batch][positions][width].  The tokenizer stays on the host (it needs the vocabulary
 * files the caller supplies by path).  Tensor names are those of the HuggingFace state dict below
 * "text_model." ("cond_stage_model.transformer.text_model.*" in an SD checkpoint: anything up to and including
 * "text_model." is stripped).  openai/clip-vit-large-patch14: vocab 49408, positions 77, width 768, 12 layers,
 * 12 heads, ffn 3072. */
typedef struct sdeo_clip_handle_s* sdeo_clip_handle;
typedef struct sdeo_clip_config {
  int vocab, positions, width, layers, heads, ffn;
} sdeo_clip_config;
int sdeo_clip_create(const sdeo_clip_config* cfg, sdeo_clip_handle* out);
int sdeo_clip_destroy(sdeo_clip_handle h);
int sdeo_clip_load_weight(sdeo_clip_handle h, const char* name, const float* host_data, const int64_t* dims, int ndim, int strict);
int sdeo_clip_finalize_weights(sdeo_clip_handle h);
int sdeo_clip_num_weights(sdeo_clip_handle h);
int sdeo_clip_weight_info(sdeo_clip_handle h, int i, const char** name, int64_t dims[2], int* ndim);
/* fix the number of prompts per call and allocate the activation buffers */
int sdeo_clip_configure(sdeo_clip_handle h, int batch);
/* tokens int32 [batch][positions] (device) -> out fp32 [batch][positions][width] (device); ids outside the
 * vocabulary are clamped */
int sdeo_clip_encode(sdeo_clip_handle h, const int32_t* tokens, int batch, float* out, void* stream);
size_t sdeo_clip_device_bytes(sdeo_clip_handle h);

/* Per-kernel timing for bench.py's roofline: between begin and end every launch of the net-level calls is
 * bracketed by HIP events on the stream it runs on; end synchronises the device and returns a JSON array
 * [{"kernel", "launches", "total_ms", "flops", "bytes"}] (algorithmic flops / bytes summed over the launches).
 * Do not time whole steps while profiling is on (the events serialise nothing but add host overhead). */
int sdeo_profile_begin(sdeo_handle h);
const char* sdeo_profile_end(sdeo_handle h);

#ifdef __cplusplus
}
#endif
#endif /* SDEO_H */
