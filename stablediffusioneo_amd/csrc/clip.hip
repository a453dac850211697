// CLIP text transformer (the step before the DDIM loop): `FrozenCLIPEmbedder.forward`
// (`ldm/modules/encoders/modules.py:123-141`) calls HuggingFace `CLIPTextModel` (transformers, an un-vendored
// dependency of the reference) and returns `last_hidden_state`.  The published algorithm restated here:
//   x = token_embedding[ids] + position_embedding[0..T)
//   per layer:  x += out_proj(softmax_causal((q_proj(LN1 x) * d^-1/2) k_proj(LN1 x)^T) v_proj(LN1 x))
//               x += fc2(quick_gelu(fc1(LN2 x)))          quick_gelu(v) = v * sigmoid(1.702 v)
//   out = final_layer_norm(x)                              (LayerNorm eps 1e-5 throughout)
// Built from the same hand-written kernels as the UNet (layernorm, LDS-DMA GEMM with fused bias / activation /
// residual epilogues, flash attention with a causal mask).  q_proj and k_proj are stacked into one GEMM.
#include <map>
#include <string>
#include <vector>
#include <functional>

#include "../../include/sdeo.h"
#include "kernels.h"

using namespace sdeo;

namespace {

struct CWeight {
  std::string name;
  int64_t dims[2];
  int ndim;
  bool matrix;       // fp16 [rows][cols] (else fp32 vector)
  size_t off;
  bool loaded;
};

}  // namespace

struct sdeo_clip_handle_s {
  sdeo_clip_config cfg{};
  std::vector<CWeight> weights;
  std::map<std::string, int> index;
  char* slab = nullptr;
  size_t slab_bytes = 0;
  float* stage = nullptr;
  size_t stage_bytes = 0;
  bool finalized = false;
  // configured state
  int batch = 0;
  char* act = nullptr;
  size_t act_bytes = 0;
  float* splitk_ws = nullptr;
  size_t splitk_bytes = 0;
  int32_t* tokens = nullptr;
  std::vector<std::function<int(hipStream_t)>> prog;
  f16* out16 = nullptr;
};

namespace {

typedef sdeo_clip_handle_s Clip;

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

static void add_w(Clip* e, const std::string& name, bool matrix, int64_t d0, int64_t d1, size_t off) {
  CWeight w{name, {d0, d1}, matrix ? 2 : 1, matrix, off, false};
  e->index[name] = (int)e->weights.size();
  e->weights.push_back(w);
}

// names follow the HuggingFace state dict below "text_model." (the SD checkpoint stores them as
// "cond_stage_model.transformer.text_model.*"; sdeo_clip_load_weight strips everything up to "text_model.")
static void build_registry(Clip* e) {
  const sdeo_clip_config& c = e->cfg;
  size_t size = 0;
  auto take = [&](size_t bytes) { const size_t off = align256(size); size = off + bytes; return off; };
  const int W = c.width, F = c.ffn;
  add_w(e, "embeddings.token_embedding.weight", true, c.vocab, W, take((size_t)c.vocab * W * 2));
  add_w(e, "embeddings.position_embedding.weight", true, c.positions, W, take((size_t)c.positions * W * 2));
  for (int l = 0; l < c.layers; ++l) {
    const std::string p = "encoder.layers." + std::to_string(l) + ".";
    // q_proj, k_proj and v_proj stacked: one [3W][W] matrix and one [3W] bias (one GEMM; the attention kernel reads V row-major)
    const size_t qkv = take((size_t)3 * W * W * 2), qkvb = take((size_t)3 * W * 4);
    add_w(e, p + "self_attn.q_proj.weight", true, W, W, qkv);
    add_w(e, p + "self_attn.k_proj.weight", true, W, W, qkv + (size_t)W * W * 2);
    add_w(e, p + "self_attn.v_proj.weight", true, W, W, qkv + (size_t)2 * W * W * 2);
    add_w(e, p + "self_attn.q_proj.bias", false, W, 0, qkvb);
    add_w(e, p + "self_attn.k_proj.bias", false, W, 0, qkvb + (size_t)W * 4);
    add_w(e, p + "self_attn.v_proj.bias", false, W, 0, qkvb + (size_t)2 * W * 4);
    add_w(e, p + "self_attn.out_proj.weight", true, W, W, take((size_t)W * W * 2));
    add_w(e, p + "self_attn.out_proj.bias", false, W, 0, take((size_t)W * 4));
    add_w(e, p + "layer_norm1.weight", false, W, 0, take((size_t)W * 4));
    add_w(e, p + "layer_norm1.bias", false, W, 0, take((size_t)W * 4));
    add_w(e, p + "mlp.fc1.weight", true, F, W, take((size_t)F * W * 2));
    add_w(e, p + "mlp.fc1.bias", false, F, 0, take((size_t)F * 4));
    add_w(e, p + "mlp.fc2.weight", true, W, F, take((size_t)W * F * 2));
    add_w(e, p + "mlp.fc2.bias", false, W, 0, take((size_t)W * 4));
    add_w(e, p + "layer_norm2.weight", false, W, 0, take((size_t)W * 4));
    add_w(e, p + "layer_norm2.bias", false, W, 0, take((size_t)W * 4));
  }
  add_w(e, "final_layer_norm.weight", false, W, 0, take((size_t)W * 4));
  add_w(e, "final_layer_norm.bias", false, W, 0, take((size_t)W * 4));
  e->slab_bytes = align256(size);
  size_t mx = 0;
  for (auto& w : e->weights) {
    const size_t n = (size_t)w.dims[0] * (w.ndim == 2 ? (size_t)w.dims[1] : 1);
    mx = n > mx ? n : mx;
  }
  e->stage_bytes = mx * sizeof(float);
}

static const f16* wp(Clip* e, const std::string& n) { return reinterpret_cast<const f16*>(e->slab + e->weights[e->index.at(n)].off); }
static const float* vp(Clip* e, const std::string& n) { return reinterpret_cast<const float*>(e->slab + e->weights[e->index.at(n)].off); }

static void free_configured(Clip* e) {
  if (e->act) (void)hipFree(e->act);
  if (e->splitk_ws) (void)hipFree(e->splitk_ws);
  if (e->tokens) (void)hipFree(e->tokens);
  e->act = nullptr; e->splitk_ws = nullptr; e->tokens = nullptr;
  e->prog.clear();
  e->batch = 0;
}

}  // namespace

extern "C" {

int sdeo_clip_create(const sdeo_clip_config* cfg, sdeo_clip_handle* out) {
  SDEO_CHECK(cfg && out, "sdeo_clip_create: null argument");
  SDEO_CHECK(cfg->vocab > 0 && cfg->positions > 0 && cfg->layers > 0 && cfg->heads > 0 && cfg->ffn > 0, "sdeo_clip_create: empty config");
  SDEO_CHECK(cfg->width % 8 == 0 && cfg->ffn % 8 == 0 && cfg->width % cfg->heads == 0, "sdeo_clip_create: width %d / ffn %d must be multiples of 8, width divisible by heads %d",
             cfg->width, cfg->ffn, cfg->heads);
  const int d = cfg->width / cfg->heads;
  SDEO_CHECK(d % 8 == 0 && d <= 160, "sdeo_clip_create: head dim %d unsupported (multiple of 8, <= 160)", d);
  Clip* e = new Clip();
  e->cfg = *cfg;
  build_registry(e);
  if (hipMalloc((void**)&e->slab, e->slab_bytes) != hipSuccess) {
    const size_t want = e->slab_bytes;
    delete e;
    return fail("sdeo_clip_create: cannot allocate %zu bytes of weights", want);
  }
  *out = e;
  return 0;
}

int sdeo_clip_destroy(sdeo_clip_handle h) {
  if (!h) return 0;
  free_configured(h);
  if (h->slab) (void)hipFree(h->slab);
  if (h->stage) (void)hipFree(h->stage);
  delete h;
  return 0;
}

int sdeo_clip_num_weights(sdeo_clip_handle h) { return h ? (int)h->weights.size() : 0; }

int sdeo_clip_weight_info(sdeo_clip_handle h, int i, const char** name, int64_t dims[2], int* ndim) {
  SDEO_CHECK(h && i >= 0 && i < (int)h->weights.size() && name && dims && ndim, "sdeo_clip_weight_info: bad argument");
  const CWeight& w = h->weights[i];
  *name = w.name.c_str();
  dims[0] = w.dims[0]; dims[1] = w.ndim == 2 ? w.dims[1] : 0;
  *ndim = w.ndim;
  return 0;
}

int sdeo_clip_load_weight(sdeo_clip_handle h, const char* name, const float* host_data, const int64_t* dims, int ndim, int strict) {
  SDEO_CHECK(h && name && host_data && dims, "sdeo_clip_load_weight: null argument");
  std::string key(name);
  const size_t pos = key.find("text_model.");
  if (pos != std::string::npos) key = key.substr(pos + 11);
  auto it = h->index.find(key);
  if (it == h->index.end()) {
    if (strict) return fail("sdeo_clip_load_weight: unexpected tensor '%s'", name);
    return 0;
  }
  CWeight& w = h->weights[it->second];
  SDEO_CHECK(ndim == w.ndim, "sdeo_clip_load_weight: %s has %d dims, expected %d", name, ndim, w.ndim);
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    SDEO_CHECK(dims[i] == w.dims[i], "sdeo_clip_load_weight: %s dim %d is %lld, expected %lld", name, i, (long long)dims[i],
               (long long)w.dims[i]);
    n *= (size_t)dims[i];
  }
  if (!h->stage) SDEO_HIP(hipMalloc((void**)&h->stage, h->stage_bytes));
  SDEO_HIP(hipMemcpy(h->stage, host_data, n * sizeof(float), hipMemcpyDefault));
  void* dst = h->slab + w.off;
  if (w.matrix) {
    if (int rc = f32_to_f16((f16*)dst, h->stage, (int64_t)n, 0)) return rc;
  } else {
    SDEO_HIP(hipMemcpy(dst, h->stage, n * sizeof(float), hipMemcpyDeviceToDevice));
  }
  SDEO_HIP(hipDeviceSynchronize());
  w.loaded = true;
  return 0;
}

int sdeo_clip_finalize_weights(sdeo_clip_handle h) {
  SDEO_CHECK(h, "sdeo_clip_finalize_weights: null handle");
  std::string missing;
  int nmiss = 0;
  for (auto& w : h->weights)
    if (!w.loaded) {
      if (nmiss < 5) missing += (nmiss ? ", " : "") + w.name;
      ++nmiss;
    }
  SDEO_CHECK(nmiss == 0, "sdeo_clip_finalize_weights: %d tensors missing (%s%s)", nmiss, missing.c_str(), nmiss > 5 ? ", ..." : "");
  if (h->stage) { (void)hipFree(h->stage); h->stage = nullptr; }
  h->finalized = true;
  return 0;
}

int sdeo_clip_configure(sdeo_clip_handle h, int batch) {
  SDEO_CHECK(h && h->finalized, "sdeo_clip_configure: weights not finalized");
  SDEO_CHECK(batch >= 1 && batch <= 64, "sdeo_clip_configure: batch=%d out of range", batch);
  free_configured(h);
  Clip* e = h;
  const sdeo_clip_config& c = e->cfg;
  const int B = batch, T = c.positions, W = c.width, F = c.ffn, H = c.heads, d = W / H;
  const int rows = B * T;
  // activation buffers (fp16): xa, xb, a [rows][W]; qkv [rows][3W]; o [rows][W]; hid [rows][F]
  size_t off = 0;
  auto take = [&](size_t elems) { const size_t o = align256(off); off = o + elems * 2; return o; };
  const size_t o_xa = take((size_t)rows * W), o_xb = take((size_t)rows * W), o_a = take((size_t)rows * W),
               o_qkv = take((size_t)rows * 3 * W), o_o = take((size_t)rows * W), o_h = take((size_t)rows * F);
  e->act_bytes = align256(off);
  SDEO_HIP(hipMalloc((void**)&e->act, e->act_bytes));
  SDEO_HIP(hipMemset(e->act, 0, e->act_bytes));
  SDEO_HIP(hipMalloc((void**)&e->tokens, (size_t)rows * sizeof(int32_t)));
  auto P = [&](size_t o) { return reinterpret_cast<f16*>(e->act + o); };
  f16 *xa = P(o_xa), *xb = P(o_xb), *a = P(o_a), *qkv = P(o_qkv), *o = P(o_o), *hid = P(o_h);

  size_t ws = 0;
  auto gemm = [&](const f16* x, int K, const f16* w, int N, const float* bias, int act, const f16* res, f16* y) {
    ConvGemm p;
    p.x = x; p.w = w; p.y = y; p.bias = bias; p.res = res; p.ldres = N;
    p.B = rows; p.Cin = K; p.M = rows; p.N = N; p.K = K; p.ldx = K; p.ldw = K; p.ldy = N; p.act = act;
    const size_t need = conv_gemm_workspace_bytes(p);
    ws = need > ws ? need : ws;
    e->prog.push_back([p, e](hipStream_t s) mutable {
      p.workspace = e->splitk_ws;
      p.workspace_bytes = e->splitk_bytes;
      return conv_gemm(p, s);
    });
  };
  const f16* tok = wp(e, "embeddings.token_embedding.weight");
  const f16* pos = wp(e, "embeddings.position_embedding.weight");
  const int32_t* ids = e->tokens;
  const int vocab = c.vocab;
  e->prog.push_back([=](hipStream_t s) { return embed_tokens(xa, ids, tok, pos, B, T, W, vocab, s); });
  f16* x = xa;
  f16* xn = xb;
  const float scale = 1.0f / sqrtf((float)d);
  for (int l = 0; l < c.layers; ++l) {
    const std::string p = "encoder.layers." + std::to_string(l) + ".";
    const float *g1 = vp(e, p + "layer_norm1.weight"), *b1 = vp(e, p + "layer_norm1.bias");
    const float *g2 = vp(e, p + "layer_norm2.weight"), *b2 = vp(e, p + "layer_norm2.bias");
    { const f16* xi = x; e->prog.push_back([=](hipStream_t s) { return layernorm(a, W, xi, W, g1, b1, rows, W, 1e-5f, s); }); }
    gemm(a, W, wp(e, p + "self_attn.q_proj.weight"), 3 * W, vp(e, p + "self_attn.q_proj.bias"), 0, nullptr, qkv);
    e->prog.push_back([=](hipStream_t s) {
      return attention(o, W, qkv, 3 * W, qkv + W, 3 * W, qkv + 2 * W, 3 * W, B, H, T, T, T, T, d, scale, s, /*causal=*/1);
    });
    gemm(o, W, wp(e, p + "self_attn.out_proj.weight"), W, vp(e, p + "self_attn.out_proj.bias"), 0, x, xn);
    std::swap(x, xn);
    { const f16* xi = x; e->prog.push_back([=](hipStream_t s) { return layernorm(a, W, xi, W, g2, b2, rows, W, 1e-5f, s); }); }
    gemm(a, W, wp(e, p + "mlp.fc1.weight"), F, vp(e, p + "mlp.fc1.bias"), 2, nullptr, hid);
    gemm(hid, F, wp(e, p + "mlp.fc2.weight"), W, vp(e, p + "mlp.fc2.bias"), 0, x, xn);
    std::swap(x, xn);
  }
  {
    const float *g = vp(e, "final_layer_norm.weight"), *b = vp(e, "final_layer_norm.bias");
    const f16* xi = x;
    e->prog.push_back([=](hipStream_t s) { return layernorm(a, W, xi, W, g, b, rows, W, 1e-5f, s); });
    e->out16 = a;
  }
  e->splitk_bytes = ws;
  if (ws) SDEO_HIP(hipMalloc((void**)&e->splitk_ws, ws));
  e->batch = B;
  return 0;
}

int sdeo_clip_encode(sdeo_clip_handle h, const int32_t* tokens, int batch, float* out, void* stream) {
  SDEO_CHECK(h && tokens && out, "sdeo_clip_encode: null argument");
  SDEO_CHECK(h->batch > 0 && batch == h->batch, "sdeo_clip_encode: batch %d != configured %d", batch, h->batch);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t rows = (size_t)batch * h->cfg.positions;
  SDEO_HIP(hipMemcpyAsync(h->tokens, tokens, rows * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  for (auto& op : h->prog)
    if (int rc = op(s)) return rc;
  return f16_to_f32(out, h->out16, (int64_t)rows * h->cfg.width, s);
}

size_t sdeo_clip_device_bytes(sdeo_clip_handle h) { return h ? h->slab_bytes + h->act_bytes + h->splitk_bytes : 0; }

}  // extern "C"
