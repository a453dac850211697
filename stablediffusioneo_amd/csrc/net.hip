// Network executor of libsdeo: weight registry, static launch schedules ("programs") for ControlNet,
// ControlledUnetModel and the VAE decoder, and the net-level C entry points of include/sdeo.h.
//
// This is what stands where the reference's opaque TensorRT engines stood (ControlNet.plan /
// ControlledUnet.plan / Decoder.plan driven by Engine.infer, Engine.py:131-161): the module structure of
// `cldm/cldm.py:22-45,284-305`, `openaimodel.py:255-275` (ResBlock), `attention.py:381-385,431-450`
// (BasicTransformerBlock / SpatialTransformer) and `model.py:619-652` (Decoder) is unrolled ONCE at
// sdeo_configure time into a flat list of kernel launches over a planned activation arena (fp16 NHWC);
// a forward pass then only walks that list: no allocation, no synchronisation, no shape logic, so the
// whole step is hipGraph-capturable.
//
// Fusions decided here (each is arithmetic the reference performs as separate ATen calls):
//   * conv/linear bias, the ResBlock time-embedding add, the residual add, SiLU of the hint block and the
//     control scale are epilogues of the producing conv/GEMM;
//   * nearest-x2 Upsample is folded into the following conv's gather; channel concat is a write into place;
//   * every ResBlock's emb_layers Linear is one stacked GEMM per forward (same input SiLU(emb));
//   * q, k and v projections of self-attention are ONE GEMM (the attention kernel reads V row-major through transposing
//     LDS reads), cross-attention K and V likewise;
//   * every LayerNorm of a BasicTransformerBlock is folded into the GEMM that consumes it (gamma into the weights at load
//     time, mean / rstd as two per-row scalars in the epilogue); the per-row statistics are written by the epilogue of the
//     GEMM that produced the residual stream, so no LayerNorm kernel and no normalised copy of the tokens exists;
//   * cross-attention K / V^T depend only on the text context and the hint block only on the hint: both are
//     computed once per image and cached across the DDIM steps.
#include <functional>
#include <memory>
#include <unordered_map>
#include <vector>

#include "../../include/sdeo.h"
#include "kernels.h"

using namespace sdeo;

namespace {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int round8(int c) { return (c + 7) / 8 * 8; }

const char* NS_UNET = "model.diffusion_model.";
const char* NS_CN = "control_model.";
const char* NS_VAE = "first_stage_model.";

// ------------------------------------------------------------------------------------------------
// architecture plans (mirror of stablediffusioneo_amd/spec.py, itself pinned to the reference
// constructors by tests/golden/manifest_sd15.json)
// ------------------------------------------------------------------------------------------------
enum BlkKind { B_CONV_IN, B_RES, B_ATTN, B_DOWN, B_UP };
struct Blk { BlkKind kind; std::string name; int cin, cout; };
struct UPlan {
  std::vector<std::vector<Blk>> in, out;
  std::vector<Blk> mid;
  std::vector<int> in_ch, in_ds;
};

static bool in_list(const int* v, int n, int x) {
  for (int i = 0; i < n; ++i) if (v[i] == x) return true;
  return false;
}

static UPlan make_uplan(const sdeo_config& c, bool with_decoder) {
  UPlan p;
  const int mc = c.model_channels;
  auto nm = [](const char* pre, int i, int j) { return std::string(pre) + "." + std::to_string(i) + "." + std::to_string(j); };
  p.in.push_back({{B_CONV_IN, "input_blocks.0.0", c.in_channels, mc}});
  p.in_ch.push_back(mc);
  p.in_ds.push_back(1);
  int ch = mc, ds = 1, idx = 1;
  for (int level = 0; level < c.num_levels; ++level) {
    const int mult = c.channel_mult[level];
    for (int r = 0; r < c.num_res_blocks; ++r) {
      std::vector<Blk> layers;
      layers.push_back({B_RES, nm("input_blocks", idx, 0), ch, mult * mc});
      ch = mult * mc;
      if (in_list(c.attention_resolutions, c.num_attention_resolutions, ds))
        layers.push_back({B_ATTN, nm("input_blocks", idx, 1), ch, ch});
      p.in.push_back(layers);
      p.in_ch.push_back(ch);
      p.in_ds.push_back(ds);
      ++idx;
    }
    if (level != c.num_levels - 1) {
      p.in.push_back({{B_DOWN, nm("input_blocks", idx, 0), ch, ch}});
      p.in_ch.push_back(ch);
      ds *= 2;
      p.in_ds.push_back(ds);
      ++idx;
    }
  }
  p.mid = {{B_RES, "middle_block.0", ch, ch}, {B_ATTN, "middle_block.1", ch, ch}, {B_RES, "middle_block.2", ch, ch}};
  if (!with_decoder) return p;
  std::vector<int> stack = p.in_ch;
  int oidx = 0;
  for (int level = c.num_levels - 1; level >= 0; --level) {
    const int mult = c.channel_mult[level];
    for (int i = 0; i <= c.num_res_blocks; ++i) {
      const int ich = stack.back();
      stack.pop_back();
      std::vector<Blk> layers;
      layers.push_back({B_RES, nm("output_blocks", oidx, 0), ch + ich, mc * mult});
      ch = mc * mult;
      if (in_list(c.attention_resolutions, c.num_attention_resolutions, ds))
        layers.push_back({B_ATTN, nm("output_blocks", oidx, 1), ch, ch});
      if (level && i == c.num_res_blocks) {
        layers.push_back({B_UP, nm("output_blocks", oidx, (int)layers.size()), ch, ch});
        ds /= 2;
      }
      p.out.push_back(layers);
      ++oidx;
    }
  }
  return p;
}

struct HintConv { std::string name; int cin, cout, stride; };
static std::vector<HintConv> hint_convs(const sdeo_config& c) {
  const int chans[8][3] = {{-1, 16, 1}, {16, 16, 1}, {16, 32, 2}, {32, 32, 1}, {32, 96, 2}, {96, 96, 1}, {96, 256, 2}, {256, -2, 1}};
  std::vector<HintConv> v;
  for (int i = 0; i < 8; ++i) {
    const int ci = chans[i][0] == -1 ? c.hint_channels : chans[i][0];
    const int co = chans[i][1] == -2 ? c.model_channels : chans[i][1];
    v.push_back({"input_hint_block." + std::to_string(2 * i), ci, co, chans[i][2]});
  }
  return v;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
enum WKind { W_CONV, W_LINEAR, W_VEC, W_GEGLU_W, W_GEGLU_B };   // W_GEGLU_*: ff.net.0.proj, value / gate rows interleaved
struct WEntry {
  std::string name;
  int64_t dims[4];
  int ndim;
  WKind kind;
  size_t off;      // byte offset into the weight slab
  int ipad;        // conv: stored (padded) input channels
  bool loaded;
};

struct Arena {   // first-fit planner over one device allocation; offsets only
  struct Blk_ { size_t off, size; bool free; };
  std::vector<Blk_> blocks;
  size_t end = 0, peak = 0;
  size_t alloc(size_t bytes) {
    bytes = align_up(bytes, 256);
    for (size_t i = 0; i < blocks.size(); ++i) {
      if (blocks[i].free && blocks[i].size >= bytes) {
        if (blocks[i].size > bytes) {
          Blk_ rest{blocks[i].off + bytes, blocks[i].size - bytes, true};
          blocks[i].size = bytes;
          blocks.insert(blocks.begin() + i + 1, rest);
        }
        blocks[i].free = false;
        return blocks[i].off;
      }
    }
    if (!blocks.empty() && blocks.back().free) {   // grow the trailing free block
      blocks.back().size = bytes;
      blocks.back().free = false;
      end = blocks.back().off + bytes;
      peak = std::max(peak, end);
      return blocks.back().off;
    }
    blocks.push_back({end, bytes, false});
    end += bytes;
    peak = std::max(peak, end);
    return blocks.back().off;
  }
  void release(size_t off) {
    for (size_t i = 0; i < blocks.size(); ++i) {
      if (blocks[i].off == off && !blocks[i].free) {
        blocks[i].free = true;
        if (i + 1 < blocks.size() && blocks[i + 1].free) { blocks[i].size += blocks[i + 1].size; blocks.erase(blocks.begin() + i + 1); }
        if (i > 0 && blocks[i - 1].free) { blocks[i - 1].size += blocks[i].size; blocks.erase(blocks.begin() + i); }
        return;
      }
    }
  }
};

struct T {           // fp16 activation view: rows x c, row stride ld; (n,h,w) when it is an image
  f16* p = nullptr;
  size_t off = (size_t)-1;   // arena offset when owned
  int n = 0, h = 0, w = 0, c = 0, ld = 0;
  // GroupNorm partials of this tensor written by the epilogue of the conv / GEMM that produced it (ConvGemm::gn_out), 32 groups
  float* gnp = nullptr;
  size_t gnp_off = (size_t)-1;
  int gn_slots = 0;
  int rows() const { return n * h * w; }
};

struct Op {          // one launch of a program + what it is for the profiler
  std::function<int(hipStream_t)> fn;
  const char* key = "other";
  std::string tag;           // problem shape, shown by the profiler when SDEO_PROFILE_DETAIL=1
  double flops = 0, bytes = 0;
  bool zero_conv = false;    // ControlNet program: a zero conv (skipped when the UNet decoder applies the zero convs itself)
  template <class F>
  Op(F f) : fn(std::move(f)) {}
  int operator()(hipStream_t s) const { return fn(s); }
};
typedef std::vector<Op> Program;

struct ProfRec { std::string key; double flops, bytes; hipEvent_t a, b; };

// LayerNorm folded into a Linear at weight-finalisation time (fold_layernorm): offsets into the weight slab
struct FoldJob { size_t w_out, s_out, b_out, w_in; std::string gamma, beta, bias; int rows, C; };
// ff.net.2 and proj_out of one SpatialTransformer composed into one [C][5C] Linear at weight-finalisation time (compose_proj)
struct ComposeJob { size_t w_out, b_out; std::string wp, bp, w2, b2; int C; };
// a [rows][cols] fp16 matrix of the UNet / ControlNet that a conv / GEMM streams as its weight operand: with fp8 weights
// (sdeo_set_weight_precision) it gets an e4m3fn copy + per-row scales and its fp16 copy is replaced by the dequantised values
struct QRegion { size_t off; int rows, cols; size_t q_off, s_off; };

}  // namespace

struct sdeo_handle_s {
  sdeo_config cfg;
  UPlan uplan, cplan;
  std::vector<HintConv> hconvs;
  // weights
  std::vector<WEntry> weights;
  std::unordered_map<std::string, int> windex;
  std::unordered_map<std::string, size_t> named_off;   // extra named regions (stacked parents, LayerNorm-folded copies)
  std::vector<FoldJob> folds;
  std::vector<ComposeJob> composes;
  std::vector<QRegion> qregions;
  std::unordered_map<size_t, int> qindex;              // fp16 slab offset -> qregions index
  int weight_bits = 16;                                // 8: fp8 e4m3fn weights for the UNet / ControlNet matrices
  int act_bits = 16, mx_min_rows = 2048;               // 8: block-scaled fp8 activations x weights for GEMMs of >= mx_min_rows rows
  char* mxslab = nullptr;                              // block-scaled packs of the matrices (codes + e8m0 scales)
  size_t mx_bytes = 0;
  struct MxRegion { size_t q_off, s_off; int rows, cols; };
  std::unordered_map<size_t, MxRegion> mxindex;        // fp16 slab offset of a matrix -> its block-scaled pack
  int mx_launches = 0;                                 // GEMMs of the current programs that run on the block-scaled fp8 MFMA
  char* q8slab = nullptr;                              // fp8 codes + scales (allocated at the first fp8 finalize)
  size_t q8_bytes = 0;
  char* wslab = nullptr;
  size_t wslab_bytes = 0;
  float* stage = nullptr;
  size_t stage_bytes = 0;
  bool finalized = false;
  // per-net stacked time-embedding projection
  int emb_total[2] = {0, 0};
  std::unordered_map<std::string, int> emb_row;        // "<ns><resblock>" -> row offset
  // problem size + arena
  int N = 0, lh = 0, lw = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  float* splitk_ws = nullptr;
  size_t splitk_ws_bytes = 0;
  float* gn_ws = nullptr;
  size_t gn_ws_bytes = 0;
  // second arena + workspaces + stream: ControlNet runs concurrently with the UNet encoder (both depend only on
  // x, t, context), joined before the decoder consumes the controls
  char* arena2 = nullptr;
  size_t arena2_bytes = 0;
  float* splitk_ws2 = nullptr;
  float* gn_ws2 = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool overlap = true;       // SDEO_OVERLAP=0 runs ControlNet and UNet back to back on one stream
  // boundary buffers (device, owned)
  float* in_x = nullptr; float* in_hint = nullptr; float* in_ctx = nullptr; int64_t* in_t = nullptr;
  float* in_ctrl[13] = {nullptr};
  float* out_eps = nullptr;
  float* out_ctrl[13] = {nullptr};
  float* vae_in = nullptr; float* vae_out = nullptr; uint8_t* vae_u8 = nullptr;
  std::vector<void*> extra_allocs;
  // persistent activations
  T ctrl[13];            // fp16 NHWC controls (ControlNet output / UNet input), unscaled
  T cn_h[13];            // the ControlNet block outputs the zero convs read (kept until the UNet decoder has run)
  float scales[13];
  float eff_scales[13];  // scales with only_mid_control folded in (0 for the twelve skip controls), read at launch by p_unet_dec_fused
  int only_mid = 0;
  bool use_control = true;
  // time embedding (`openaimodel.py:777-781` + every ResBlock's emb_layers): fp32 [N][emb_total] per net, written by p_temb[net] from
  // in_t -- or, for a sampler that announced its schedule (sdeo_set_timestep_table), one row of a table computed once per schedule.
  // The ResBlock convs read their pointer / row stride at LAUNCH time (emb_cur / emb_ld_cur), like the control scales.
  static constexpr int kTabRows = 128;
  float* emb_all[2] = {nullptr, nullptr};
  float* temb_tab[2] = {nullptr, nullptr};     // [kTabRows][emb_total[net]]
  int64_t* tab_t = nullptr;                    // device int64 [kTabRows]
  int tab_count = 0;
  const float* emb_cur[2] = {nullptr, nullptr};
  int emb_ld_cur[2] = {0, 0};
  T x0;                  // fp16 NHWC copy of the latent input, shared by ControlNet and UNet (p_x0 writes it from in_x)
  T eps16;               // the UNet's eps before the NCHW fp32 export (p_eps_export), read directly by sdeo_ddim_step
  // programs
  Program p_hint, p_ctx_cn, p_ctx_unet, p_cn, p_cn_export, p_ctrl_import, p_unet_enc, p_unet_dec, p_unet_noctrl, p_vae;
  Program p_temb[2], p_temb_tab, p_x0, p_eps_export, p_unet_dec_fused;
  std::vector<size_t> ctrl_elems;
  size_t device_bytes = 0;
  // profiling (sdeo_profile_*): HIP events around every launch of the next programs
  bool autotune = false;    // SDEO_AUTOTUNE=1: measure GEMM plans of untabled shapes at configure time (tuning aid; tools/tune_plans.py)
  bool profiling = false;
  std::vector<ProfRec> prof;
  std::string prof_report;
};

namespace {

typedef sdeo_handle_s Engine;

// ------------------------------------------------------------------------------------------------
// registry construction
// ------------------------------------------------------------------------------------------------
struct Registry {
  Engine* e;
  size_t size = 0;
  bool quant = true;          // matrices registered now belong to the UNet / ControlNet (fp8-eligible), not the VAE
  void region(size_t off, int rows, int cols) {
    if (!quant) return;
    e->qindex[off] = (int)e->qregions.size();
    e->qregions.push_back(QRegion{off, rows, cols, 0, 0});
  }
  size_t take(size_t bytes) {
    const size_t off = align_up(size, 256);
    size = off + bytes;
    return off;
  }
  void add(const std::string& name, WKind kind, std::initializer_list<int64_t> dims, size_t off, int ipad = 0) {
    WEntry w{};
    w.name = name; w.kind = kind; w.ndim = (int)dims.size(); w.off = off; w.ipad = ipad; w.loaded = false;
    int i = 0;
    for (auto d : dims) w.dims[i++] = d;
    e->windex[name] = (int)e->weights.size();
    e->weights.push_back(w);
  }
  // conv: weight [cout][cin][k][k] -> fp16 [opad][k][k][ipad]; bias -> fp32 [opad]
  void conv(const std::string& name, int cin, int cout, int k, int opad = -1) {
    const int ip = round8(cin);
    const int op = opad < 0 ? cout : opad;
    const size_t off = take((size_t)op * k * k * ip * 2);
    add(name + ".weight", W_CONV, {cout, cin, k, k}, off, ip);
    region(off, op, k * k * ip);
    add(name + ".bias", W_VEC, {cout}, take((size_t)op * 4));
  }
  void lin(const std::string& name, int cin, int cout, bool bias) {
    const size_t off = take((size_t)cout * cin * 2);
    add(name + ".weight", W_LINEAR, {cout, cin}, off);
    region(off, cout, cin);
    if (bias) add(name + ".bias", W_VEC, {cout}, take((size_t)cout * 4));
  }
  void vec(const std::string& name, int c) { add(name, W_VEC, {c}, take((size_t)c * 4)); }
  void norm(const std::string& name, int c) { vec(name + ".weight", c); vec(name + ".bias", c); }
};

static void reg_res(Registry& r, const std::string& ns, const Blk& b, int emb_dim, size_t emb_w_off, size_t emb_b_off, int& emb_row) {
  const std::string p = ns + b.name;
  r.norm(p + ".in_layers.0", b.cin);
  r.conv(p + ".in_layers.2", b.cin, b.cout, 3);
  // emb_layers.1 lives inside the stacked [sum(cout)][emb_dim] matrix of its network
  r.add(p + ".emb_layers.1.weight", W_LINEAR, {b.cout, emb_dim}, emb_w_off + (size_t)emb_row * emb_dim * 2);
  r.add(p + ".emb_layers.1.bias", W_VEC, {b.cout}, emb_b_off + (size_t)emb_row * 4);
  r.e->emb_row[p] = emb_row;
  emb_row += b.cout;
  r.norm(p + ".out_layers.0", b.cout);
  r.conv(p + ".out_layers.3", b.cout, b.cout, 3);
  if (b.cin != b.cout) r.conv(p + ".skip_connection", b.cin, b.cout, 1);
}

static void reg_attn(Registry& r, const std::string& ns, const Blk& b, int ctx) {
  const std::string p = ns + b.name;
  const int c = b.cin;
  r.norm(p + ".norm", c);
  r.conv(p + ".proj_in", c, c, 1);
  const std::string t = p + ".transformer_blocks.0";
  // attn1: to_q, to_k, to_v stacked as one [3c][c] matrix (raw); the copy the network runs on has norm1 folded in
  const size_t qkv = r.take((size_t)3 * c * c * 2);
  r.add(t + ".attn1.to_q.weight", W_LINEAR, {c, c}, qkv);
  r.add(t + ".attn1.to_k.weight", W_LINEAR, {c, c}, qkv + (size_t)c * c * 2);
  r.add(t + ".attn1.to_v.weight", W_LINEAR, {c, c}, qkv + (size_t)2 * c * c * 2);
  r.lin(t + ".attn1.to_out.0", c, c, true);
  r.add(t + ".attn2.to_q.weight", W_LINEAR, {c, c}, r.take((size_t)c * c * 2));      // raw: only the input of its LayerNorm fold
  // attn2: to_k and to_v stacked as one [2c][ctx] matrix (one GEMM per context)
  const size_t kv = r.take((size_t)2 * c * ctx * 2);
  r.e->named_off[t + ".attn2.to_kv"] = kv;
  r.region(kv, 2 * c, ctx);
  r.add(t + ".attn2.to_k.weight", W_LINEAR, {c, ctx}, kv);
  r.add(t + ".attn2.to_v.weight", W_LINEAR, {c, ctx}, kv + (size_t)c * ctx * 2);
  r.lin(t + ".attn2.to_out.0", c, c, true);
  const size_t ff1 = r.take((size_t)8 * c * c * 2);
  r.add(t + ".ff.net.0.proj.weight", W_GEGLU_W, {8 * c, c}, ff1);
  r.add(t + ".ff.net.0.proj.bias", W_GEGLU_B, {8 * c}, r.take((size_t)8 * c * 4));
  r.lin(t + ".ff.net.2", 4 * c, c, true);
  // LayerNorm-folded copies (built by sdeo_finalize_weights): weights, row sums s, bias'
  auto fold = [&](const std::string& name, size_t w_in, int rows, const std::string& norm, const std::string& bias) {
    FoldJob f;
    f.w_out = r.take((size_t)rows * c * 2);
    f.s_out = r.take((size_t)rows * 4);
    f.b_out = r.take((size_t)rows * 4);
    f.w_in = w_in; f.gamma = norm + ".weight"; f.beta = norm + ".bias"; f.bias = bias; f.rows = rows; f.C = c;
    r.e->named_off[name + ".w"] = f.w_out;
    r.e->named_off[name + ".s"] = f.s_out;
    r.e->named_off[name + ".b"] = f.b_out;
    r.e->folds.push_back(f);
    r.region(f.w_out, rows, c);        // the folded copy is what the network streams (the raw one is only the fold's input)
  };
  fold(t + ".attn1.qkv_ln", qkv, 3 * c, t + ".norm1", "");
  fold(t + ".attn2.q_ln", r.e->weights[r.e->windex.at(t + ".attn2.to_q.weight")].off, c, t + ".norm2", "");
  fold(t + ".ff1_ln", ff1, 8 * c, t + ".norm3", t + ".ff.net.0.proj.bias");
  r.norm(t + ".norm1", c);
  r.norm(t + ".norm2", c);
  r.norm(t + ".norm3", c);
  r.conv(p + ".proj_out", c, c, 1);
  // ff.net.2 + proj_out as one Linear over [GEGLU output | tok2] (build_attn); no fp8 / block-scaled copy: the reference modules the
  // fp8 goldens come from round ff.net.2 and proj_out separately, and this product is formed from exactly those rounded values
  ComposeJob cj;
  cj.w_out = r.take((size_t)c * 5 * c * 2);
  cj.b_out = r.take((size_t)c * 4);
  cj.wp = p + ".proj_out.weight"; cj.bp = p + ".proj_out.bias"; cj.w2 = t + ".ff.net.2.weight"; cj.b2 = t + ".ff.net.2.bias"; cj.C = c;
  r.e->named_off[t + ".ffproj.w"] = cj.w_out;
  r.e->named_off[t + ".ffproj.b"] = cj.b_out;
  r.e->composes.push_back(cj);
}

static int plan_emb_total(const UPlan& p) {
  int t = 0;
  auto acc = [&](const std::vector<Blk>& v) { for (auto& b : v) if (b.kind == B_RES) t += b.cout; };
  for (auto& v : p.in) acc(v);
  acc(p.mid);
  for (auto& v : p.out) acc(v);
  return t;
}

static void reg_unet_like(Registry& r, const std::string& ns, const UPlan& p, const sdeo_config& c, int net) {
  const int emb = 4 * c.model_channels;
  r.lin(ns + "time_embed.0", c.model_channels, emb, true);
  r.lin(ns + "time_embed.2", emb, emb, true);
  const int total = plan_emb_total(p);
  r.e->emb_total[net] = total;
  const size_t ew = r.take((size_t)total * emb * 2);
  const size_t eb = r.take((size_t)total * 4);
  r.e->named_off[ns + "emb_all.weight"] = ew;
  r.region(ew, total, emb);
  r.e->named_off[ns + "emb_all.bias"] = eb;
  int row = 0;
  auto blocks = [&](const std::vector<Blk>& v) {
    for (auto& b : v) {
      switch (b.kind) {
        case B_CONV_IN: r.conv(ns + b.name, b.cin, b.cout, 3); break;
        case B_RES: reg_res(r, ns, b, emb, ew, eb, row); break;
        case B_ATTN: reg_attn(r, ns, b, c.context_dim); break;
        case B_DOWN: r.conv(ns + b.name + ".op", b.cin, b.cout, 3); break;
        case B_UP: r.conv(ns + b.name + ".conv", b.cin, b.cout, 3); break;
      }
    }
  };
  for (auto& v : p.in) blocks(v);
  blocks(p.mid);
  for (auto& v : p.out) blocks(v);
}

static void reg_vae_res(Registry& r, const std::string& p, int cin, int cout) {
  r.norm(p + ".norm1", cin);
  r.conv(p + ".conv1", cin, cout, 3);
  r.norm(p + ".norm2", cout);
  r.conv(p + ".conv2", cout, cout, 3);
  if (cin != cout) r.conv(p + ".nin_shortcut", cin, cout, 1);
}

struct VLevel { int level; std::vector<std::pair<int, int>> blocks; bool up; };
static std::vector<VLevel> vae_levels(const sdeo_config& c, int* block_in_out) {
  const int nl = c.vae_num_levels;
  int bi = c.vae_ch * c.vae_ch_mult[nl - 1];
  *block_in_out = bi;
  std::vector<VLevel> v;
  for (int l = nl - 1; l >= 0; --l) {
    const int bo = c.vae_ch * c.vae_ch_mult[l];
    VLevel L{l, {}, l != 0};
    for (int j = 0; j <= c.vae_num_res_blocks; ++j) { L.blocks.push_back({bi, bo}); bi = bo; }
    v.push_back(L);
  }
  return v;
}

static void build_registry(Engine* e) {
  Registry r{e};
  const sdeo_config& c = e->cfg;
  // UNet
  reg_unet_like(r, NS_UNET, e->uplan, c, 0);
  r.norm(std::string(NS_UNET) + "out.0", c.model_channels);
  r.conv(std::string(NS_UNET) + "out.2", c.model_channels, c.out_channels, 3, (c.out_channels + 3) / 4 * 4);
  // ControlNet
  reg_unet_like(r, NS_CN, e->cplan, c, 1);
  for (size_t i = 0; i < e->cplan.in_ch.size(); ++i)
    r.conv(std::string(NS_CN) + "zero_convs." + std::to_string(i) + ".0", e->cplan.in_ch[i], e->cplan.in_ch[i], 1);
  for (auto& hc : e->hconvs) r.conv(std::string(NS_CN) + hc.name, hc.cin, hc.cout, 3, round8(hc.cout));
  r.conv(std::string(NS_CN) + "middle_block_out.0", e->cplan.in_ch.back(), e->cplan.in_ch.back(), 1);
  // VAE decode path
  r.quant = false;                   // the VAE stays fp16 (BASELINE configs[4])
  const std::string d = std::string(NS_VAE) + "decoder";
  r.conv(std::string(NS_VAE) + "post_quant_conv", c.vae_z_channels, c.vae_z_channels, 1, round8(c.vae_z_channels));
  int bin = 0;
  auto levels = vae_levels(c, &bin);
  r.conv(d + ".conv_in", c.vae_z_channels, bin, 3);
  reg_vae_res(r, d + ".mid.block_1", bin, bin);
  r.norm(d + ".mid.attn_1.norm", bin);
  for (const char* n : {"q", "k", "v", "proj_out"}) r.conv(d + ".mid.attn_1." + n, bin, bin, 1);
  reg_vae_res(r, d + ".mid.block_2", bin, bin);
  int last = bin;
  for (auto& L : levels) {
    for (size_t j = 0; j < L.blocks.size(); ++j) {
      reg_vae_res(r, d + ".up." + std::to_string(L.level) + ".block." + std::to_string(j), L.blocks[j].first, L.blocks[j].second);
      last = L.blocks[j].second;
    }
    if (L.up) r.conv(d + ".up." + std::to_string(L.level) + ".upsample.conv", last, last, 3);
  }
  r.norm(d + ".norm_out", last);
  r.conv(d + ".conv_out", last, c.vae_out_ch, 3, (c.vae_out_ch + 3) / 4 * 4);
  e->wslab_bytes = align_up(r.size, 256);
}

// ------------------------------------------------------------------------------------------------
// program builder
// ------------------------------------------------------------------------------------------------
struct RowStats {                     // per-row (sum, sumsq) partials of a [rows][c] tensor: fp32 [rows][ld][2]
  float* p = nullptr;
  int ld = 0, strips = 0, c = 0;
};

struct ConvOpts {                     // conv / gemm options
  const float* bias2 = nullptr; int ld_bias2 = 0;
  const float* const* bias2_cur = nullptr; const int* ld_bias2_cur = nullptr; int bias2_off = 0;    // read at launch time (time embedding)
  const T* res = nullptr;
  int act = 0;
  const float* scale_host = nullptr;   // read at launch time (control scales)
  const T* out = nullptr;        // write into this view instead of allocating
  int cout_store = -1;           // stored output channels (>= logical, padded rows of W are zero)
  RowStats* stats = nullptr;     // producer: emit per-row (sum, sumsq) partials of the stored values into this buffer
  const RowStats* ln = nullptr;  // consumer: LayerNorm folded into this GEMM, statistics of x from *ln, row sums ln_s
  const float* ln_s = nullptr;
  bool gn_next = false;          // the output feeds a GroupNorm(32): let the epilogue emit its partial statistics when the plan can
  // GroupNorm(32) (+ SiLU) of the conv's INPUT applied inside the conv kernel (Builder::gn_conv decides): the partials of x, gamma / beta
  const float* gn_in = nullptr; int gn_in_slots = 0; const float* gn_gamma = nullptr; const float* gn_beta = nullptr;
  float gn_in_eps = 1e-5f; int gn_in_silu = 0;
  // conv(): stream this [cout][k*k*x.c] matrix / bias instead of the tensors registered under `name` (a composed Linear)
  const f16* w_ovr = nullptr; const float* b_ovr = nullptr;
};

struct Builder {
  typedef ConvOpts CO;
  Engine* e;
  Arena* arena;
  bool dry;
  char* base = nullptr;      // device base of `arena`
  int ws_sel = 0;            // 0: main-stream workspaces, 1: side-stream (ControlNet) workspaces
  Program* prog = nullptr;
  size_t max_splitk = 0, max_gn = 0;
  std::string err;

  T alloc(int n, int h, int w, int c) {
    T t;
    t.n = n; t.h = h; t.w = w; t.c = c; t.ld = c;
    t.off = arena->alloc((size_t)n * h * w * c * 2);
    t.p = reinterpret_cast<f16*>(base + t.off);
    return t;
  }
  T alloc2d(int rows, int c) { return alloc(1, 1, rows, c); }
  void release(T& t) {
    if (t.off != (size_t)-1) arena->release(t.off);
    t.off = (size_t)-1;
    if (t.gnp_off != (size_t)-1) arena->release(t.gnp_off);
    t.gnp_off = (size_t)-1;
    t.gnp = nullptr;
  }
  // producer side of the GroupNorm fusion: when the plan of p can, give it a partials buffer and record it on the output tensor
  static bool gn_from_producer() { static const bool on = [] { const char* v = getenv("SDEO_GN_PRODUCER_STATS"); return !v || atoi(v) != 0; }(); return on; }
  // The buffer is reserved whatever the plan (sized for the smallest tile: 32 rows per entry), so that the arena plan does not depend
  // on plans measured between the planning pass and the build pass (SDEO_AUTOTUNE); whether the launch emits is decided in
  // launch_conv, after the plan of the shape is final.
  void reserve_gn_partials(const ConvGemm& p, T& y) {
    if (!gn_from_producer() || p.N % 32) return;
    const int hw = p.Ho * p.Wo;
    y.gnp_off = arena->alloc((size_t)p.B * ((hw + 31) / 32) * 32 * 2 * sizeof(float));
    y.gnp = nullptr;                       // set by launch_conv when the plan emits
    y.gn_slots = 0;
  }
  void push(Op op, const char* key = "elementwise", double flops = 0, double bytes = 0, const std::string& tag = std::string()) {
    op.key = key; op.flops = flops; op.bytes = bytes; op.tag = tag;
    if (!dry) prog->push_back(std::move(op));
  }

  const WEntry* W(const std::string& name) {
    auto it = e->windex.find(name);
    if (it == e->windex.end()) { if (err.empty()) err = "unknown weight " + name; return nullptr; }
    return &e->weights[it->second];
  }
  const f16* wptr(const std::string& name) { auto w = W(name); return w ? reinterpret_cast<const f16*>(e->wslab + w->off) : nullptr; }
  const float* vptr(const std::string& name) { auto w = W(name); return w ? reinterpret_cast<const float*>(e->wslab + w->off) : nullptr; }
  const f16* named_w(const std::string& name) { return reinterpret_cast<const f16*>(e->wslab + e->named_off.at(name)); }
  const float* named_v(const std::string& name) { return reinterpret_cast<const float*>(e->wslab + e->named_off.at(name)); }


  void launch_conv(ConvGemm p, const float* scale_host, RowStats* stats = nullptr, T* gn_y = nullptr, const ConvOpts* lo = nullptr) {
    if (e->act_bits == 8 && e->mxslab && p.M >= e->mx_min_rows && p.R == 1 && p.S == 1 && p.stride == 1 && !p.ups && p.K % 128 == 0 &&
        p.K == p.Cin && p.ldx % 16 == 0 && !p.bias_per_row && p.y && !p.y32) {
      // block-scaled fp8 on both sides: pack the activations (one launch), run the GEMM on the fp8 MFMA
      auto it = e->mxindex.find((size_t)(reinterpret_cast<const char*>(p.w) - e->wslab));
      if (it != e->mxindex.end() && it->second.cols == p.ldw && p.N <= it->second.rows) {
        T xq = alloc2d(p.M, p.K / 2), xs = alloc2d(p.M, (p.K / 32 + 15) / 16 * 8);      // bytes: M x K codes, M x roundup(K/32, 16) scales
        uint8_t* q = reinterpret_cast<uint8_t*>(xq.p);
        uint8_t* sc = reinterpret_cast<uint8_t*>(xs.p);
        const int lds = (p.K / 32 + 15) / 16 * 16;
        const f16* xp = p.x; const int M_ = p.M, K_ = p.K, ldx_ = p.ldx;
        push([=](hipStream_t s) { return quantize_mx(q, sc, xp, M_, K_, ldx_, K_, lds, s); }, "quantize_mx", 0, 3.0 * M_ * K_,
             "rows" + std::to_string(M_) + " C" + std::to_string(K_));
        p.x = reinterpret_cast<const f16*>(q); p.ldx = p.K;
        p.w = reinterpret_cast<const f16*>(e->mxslab + it->second.q_off); p.ldw = it->second.cols;
        p.mx_sx = sc; p.mx_ldsx = lds;
        p.mx_sw = reinterpret_cast<const uint8_t*>(e->mxslab + it->second.s_off); p.mx_ldsw = it->second.cols / 32;
        if (!dry) ++e->mx_launches;
        mx_tmp.push_back(xq); mx_tmp.push_back(xs);
      }
    }
    if (!p.mx_sx && e->weight_bits == 8 && p.M <= 512 && p.Cin % 64 == 0 && !p.ups && !p.bias_per_row && !conv_gemm_plan_is_halo(p)) {
      // weight-bound shapes stream the fp8 copy of their matrix (same numbers: the fp16 copy holds the dequantised values); where
      // the measured fp16 plan is a halo-reuse 3x3 kernel (activation-bound: M = 512 at long K) that kernel keeps the job
      auto it = e->qindex.find((size_t)(reinterpret_cast<const char*>(p.w) - e->wslab));
      if (it != e->qindex.end()) {
        const QRegion& q = e->qregions[it->second];
        if (q.cols == p.ldw && p.N <= q.rows) {
          p.w = reinterpret_cast<const f16*>(e->q8slab + q.q_off);
          p.wscale = reinterpret_cast<const float*>(e->q8slab + q.s_off);
        }
      }
    }
    max_splitk = std::max(max_splitk, e->autotune ? conv_gemm_autotune_workspace_bytes(p) : conv_gemm_workspace_bytes(p));
    if (!dry && e->autotune) {
      ConvGemm q = p;
      q.workspace = ws_sel ? e->splitk_ws2 : e->splitk_ws;
      q.workspace_bytes = e->splitk_ws_bytes;
      if (conv_gemm_autotune(q, 0) && err.empty()) err = std::string("autotune failed: ") + sdeo_last_error();
    }
    if (gn_y && gn_y->gnp_off != (size_t)-1) {      // the plan of this shape is final now: emit the GroupNorm partials if it can
      const int cpg = p.N / 32, slots = conv_gemm_gn_slots(p, cpg);
      if (slots > 0) {
        gn_y->gnp = reinterpret_cast<float*>(base + gn_y->gnp_off);
        gn_y->gn_slots = slots;
        p.gn_out = gn_y->gnp; p.gn_cpg = cpg; p.gn_slots = slots; p.gn_groups = 32;
      }
    }
    if (!dry && getenv("SDEO_DUMP_GEMM"))   // shape census for tools/tune_gemm.py
      fprintf(stderr, "SDEO_GEMM %d %d %d %d %d %d %d %d %d %d %s\n", p.M, p.N, p.K, p.Cin, p.R, p.stride, p.ups, p.B, p.Hi, p.Wi,
              conv_gemm_kernel_name(p));
    bool stats_by_kernel = false;
    if (stats) {
      // the epilogue of an unsplit plan writes one partial per strip; a split-K plan leaves them to a row_stats launch
      stats->c = p.N;
      stats->strips = conv_gemm_stats_strips(p);
      if (stats->strips > 0 && stats->strips <= stats->ld) { p.stats_out = stats->p; p.stats_ld = stats->ld; }
      else { stats->strips = 1; stats_by_kernel = true; }
    }
    Engine* eng = e;
    const int sel = ws_sel;
    const float* const* b2cur = lo ? lo->bias2_cur : nullptr;
    const int* b2ld = lo ? lo->ld_bias2_cur : nullptr;
    const int b2off = lo ? lo->bias2_off : 0;
    push([p, scale_host, eng, sel, b2cur, b2ld, b2off](hipStream_t s) mutable {
      p.workspace = sel ? eng->splitk_ws2 : eng->splitk_ws;
      p.workspace_bytes = eng->splitk_ws_bytes;
      if (scale_host) p.scale = *scale_host;
      if (b2cur) { p.bias2 = *b2cur + b2off; p.ld_bias2 = *b2ld; }
      return conv_gemm(p, s);
    }, conv_gemm_kernel_name(p), 2.0 * p.M * p.N * p.K,
       2.0 * ((double)p.M * p.Cin * (p.R * p.S > 1 ? 1 : 1) + (double)p.N * p.K + (double)p.M * p.N),
       "M" + std::to_string(p.M) + " N" + std::to_string(p.N) + " K" + std::to_string(p.K) + " R" + std::to_string(p.R) + " s" +
           std::to_string(p.stride) + " u" + std::to_string(p.ups));
    for (auto& t : mx_tmp) release(t);       // the packed activations live for this one launch
    mx_tmp.clear();
    if (stats_by_kernel) {
      float* sp = stats->p; const int ld = stats->ld, rows = p.M, C = p.N, ldy = p.ldy; const f16* y = p.y;
      push([=](hipStream_t s) { return row_stats(sp, ld, y, ldy, rows, C, s); }, "row_stats", 0, 2.0 * rows * C,
           "rows" + std::to_string(rows) + " C" + std::to_string(C));
    }
  }

  std::vector<T> mx_tmp;
  RowStats alloc_stats(int rows, int c) {
    RowStats st;
    st.ld = std::max(1, (c + 31) / 32);          // narrowest epilogue strip is 32 columns
    st.c = c;
    T raw = alloc2d(rows, st.ld * 4);            // rows * ld * 2 floats in an fp16-typed arena block
    st.p = reinterpret_cast<float*>(raw.p);
    stats_blocks.push_back(raw);
    return st;
  }
  void release_stats() {                         // statistics buffers live until the end of their transformer block
    for (auto& t : stats_blocks) release(t);
    stats_blocks.clear();
  }
  std::vector<T> stats_blocks;

  static void set_ln(ConvGemm& p, const ConvOpts& o) {
    if (!o.ln) return;
    p.ln_stats = o.ln->p; p.ln_s = o.ln_s; p.ln_strips = o.ln->strips; p.ln_ld = o.ln->ld; p.ln_c = o.ln->c; p.ln_eps = 1e-5f;
  }

  // conv on an image view; weights by name (".weight"/".bias" appended)
  T conv(const T& x, const std::string& name, int cout, int k, int stride, int ups, const CO& o = CO()) {
    const WEntry* w = o.w_ovr ? nullptr : W(name + ".weight");
    ConvGemm p;
    const int pad = k / 2;
    const int hv = ups ? 2 * x.h : x.h, wv = ups ? 2 * x.w : x.w;
    const int ho = (hv + 2 * pad - k) / stride + 1, wo = (wv + 2 * pad - k) / stride + 1;
    const int cs = o.cout_store > 0 ? o.cout_store : cout;
    T y = o.out ? *o.out : alloc(x.n, ho, wo, cs);
    if (w && w->ipad != x.c && err.empty()) err = "conv " + name + ": input has " + std::to_string(x.c) + " channels, weight expects " + std::to_string(w->ipad);
    p.x = x.p; p.y = y.p;
    if (o.w_ovr) { p.w = o.w_ovr; p.bias = o.b_ovr; } else { p.w = wptr(name + ".weight"); p.bias = vptr(name + ".bias"); }
    p.bias2 = o.bias2; p.ld_bias2 = o.ld_bias2;
    if (o.res) { p.res = o.res->p; p.ldres = o.res->ld; }
    p.B = x.n; p.Hi = x.h; p.Wi = x.w; p.Cin = x.c; p.Ho = ho; p.Wo = wo; p.R = p.S = k; p.stride = stride; p.pad = pad; p.ups = ups;
    p.M = x.n * ho * wo; p.N = cs; p.K = k * k * x.c;
    p.ldx = x.ld; p.ldw = p.K; p.ldy = y.ld; p.act = o.act;
    if (o.gn_in) {
      p.gn_in = o.gn_in; p.gn_in_slots = o.gn_in_slots; p.gn_gamma = o.gn_gamma; p.gn_beta = o.gn_beta; p.gn_in_eps = o.gn_in_eps;
      p.gn_in_silu = o.gn_in_silu;
    }
    const bool want_gn = o.gn_next && !o.out && !o.scale_host && cs == y.c;
    if (want_gn) reserve_gn_partials(p, y);
    launch_conv(p, o.scale_host, o.stats, want_gn ? &y : nullptr, &o);
    T r = y;
    if (o.out) r.off = (size_t)-1;
    return r;
  }

  // y[rows][n] = x[rows][k] . w[n][k]^T (+bias)(+res)
  T gemm(const T& x, const f16* w, int ldw, int n, const float* bias, const CO& o = CO(), float* out32 = nullptr, int ld32 = 0) {
    ConvGemm p;
    const int rows = x.rows();
    T y;
    if (!out32) y = o.out ? *o.out : alloc(x.n, x.h, x.w, n);
    p.x = x.p; p.w = w; p.bias = bias;
    if (out32) { p.y32 = out32; p.ldy = ld32; } else { p.y = y.p; p.ldy = y.ld; }
    if (o.res) { p.res = o.res->p; p.ldres = o.res->ld; }
    p.B = rows; p.Cin = x.c; p.M = rows; p.N = n; p.K = x.c;
    p.ldx = x.ld; p.ldw = ldw; p.act = o.act;
    set_ln(p, o);
    if (o.ln && o.ln->c != x.c && err.empty()) err = "LayerNorm statistics of a " + std::to_string(o.ln->c) + "-channel tensor fed to K = " + std::to_string(x.c);
    launch_conv(p, o.scale_host, o.stats);
    if (o.out) y.off = (size_t)-1;
    return y;
  }

  // yt[c][rows] = w[c][k] . x[rows][k]^T (+bias per row): the transposed projection (V^T)
  T gemm_t(const T& x, const f16* w, int ldw, int c, const float* bias_rows) {
    ConvGemm p;
    const int rows = x.rows();
    T y = alloc2d(c, rows);
    p.x = w; p.w = x.p; p.y = y.p; p.bias = bias_rows; p.bias_per_row = bias_rows ? 1 : 0;
    p.B = c; p.Cin = x.c; p.M = c; p.N = rows; p.K = x.c;
    p.ldx = ldw; p.ldw = x.ld; p.ldy = rows;
    launch_conv(p, nullptr);
    return y;
  }

  T gn(const T& x, const std::string& name, float eps, int silu_, const T* out = nullptr) {
    T y = out ? *out : alloc(x.n, x.h, x.w, x.c);
    const float* g = vptr(name + ".weight");
    const float* b = vptr(name + ".bias");
    const int B = x.n, HW = x.h * x.w, C = x.c;
    max_gn = std::max(max_gn, (size_t)B * gn_chunks(HW) * 32 * 2 * sizeof(float));
    Engine* eng = e;
    const int sel = ws_sel;
    const f16* xp = x.p; f16* yp = y.p; const int ldx = x.ld, ldy = y.ld;
    const GnArgs gargs{yp, xp, g, b, nullptr, ldy, ldx, B, HW, C, 32, eps, silu_};
    if (x.gnp && !groupnorm_is_single_launch(gargs)) {
      // the statistics came out of the producer's epilogue: one launch (normalise) instead of two
      GnArgs ga = gargs;
      ga.ext_partials = x.gnp; ga.ext_nsc = x.gn_slots;
      max_gn = std::max(max_gn, (size_t)B * 32 * 2 * sizeof(float));
      push([=](hipStream_t s) mutable { ga.partials = sel ? eng->gn_ws2 : eng->gn_ws; return groupnorm_nhwc(ga, s); }, "groupnorm", 0,
           2.0 * 2.0 * B * HW * C, "C" + std::to_string(C) + " HW" + std::to_string(HW) + " apply");
      if (out) y.off = (size_t)-1;
      return y;
    }
    push([=](hipStream_t s) { return groupnorm_nhwc(yp, ldy, xp, ldx, g, b, B, HW, C, 32, eps, silu_, sel ? eng->gn_ws2 : eng->gn_ws, s); }, "groupnorm", 0,
         3.0 * 2.0 * B * HW * C, "C" + std::to_string(C) + " HW" + std::to_string(HW));
    if (out) y.off = (size_t)-1;
    return y;
  }

  // conv3x3(act(GroupNorm(x))) (`openaimodel.py:255-275`, `model.py:129-149`).  When x carries its producer's partials and the conv's
  // plan is a halo-reuse kernel with LDS left for the (a, b) table, the conv applies the GroupNorm itself (KP::gn_in): no GroupNorm
  // launch, no normalised copy.  Otherwise GroupNorm launch(es) + conv.
  // default OFF since the end of round 3: with the launch prologues shortened (kernel-argument fetch) the 21 GroupNorm launches it
  // removes cost less than the ~6 us it adds to each conv (three same-box alternations: 5.931 vs 5.876 ms per step, 8.13 vs 8.22 images/s)
  static bool gn_in_enabled() { static const bool on = [] { const char* v = getenv("SDEO_GN_IN_CONV"); return v && atoi(v) != 0; }(); return on; }
  T gn_conv(const T& x, const std::string& gn_name, float eps, int silu_, const std::string& conv_name, int cout, CO o) {
    if (gn_in_enabled() && !e->autotune && x.gnp && x.gn_slots > 0 && x.c % 32 == 0) {
      ConvGemm q;
      const int cs = o.cout_store > 0 ? o.cout_store : cout;
      q.B = x.n; q.Hi = x.h; q.Wi = x.w; q.Cin = x.c; q.Ho = x.h; q.Wo = x.w; q.R = q.S = 3; q.stride = 1; q.pad = 1;
      q.M = x.n * x.h * x.w; q.N = cs; q.K = 9 * x.c; q.ldx = x.ld; q.ldw = q.K; q.ldy = cs; q.act = o.act;
      if (x.ld == x.c && conv_gemm_gn_in_ok(q)) {
        const float* part = x.gnp;
        int slots = x.gn_slots;
        T folded;
        if (slots > 128) {               // large images (VAE): fold the entries of a group first, the conv's prologue sums few
          folded = alloc2d(x.n, 32 * 2 * 2);            // [n][1][32][2] floats
          float* fp = reinterpret_cast<float*>(folded.p);
          const int B = x.n, ns = slots;
          push([=](hipStream_t s) { return groupnorm_fold_partials(fp, part, B, ns, 32, s); }, "groupnorm", 0, 0, "fold " + std::to_string(ns));
          part = fp;
          slots = 1;
        }
        o.gn_in = part; o.gn_in_slots = slots; o.gn_gamma = vptr(gn_name + ".weight"); o.gn_beta = vptr(gn_name + ".bias");
        o.gn_in_eps = eps; o.gn_in_silu = silu_;
        T y = conv(x, conv_name, cout, 3, 1, 0, o);
        if (slots == 1 && folded.off != (size_t)-1) release(folded);
        ++gn_in_fused;
        return y;
      }
    }
    T t = gn(x, gn_name, eps, silu_);
    T y = conv(t, conv_name, cout, 3, 1, 0, o);
    release(t);
    return y;
  }
  int gn_in_fused = 0;

  T ln(const T& x, const std::string& name) {
    T y = alloc(x.n, x.h, x.w, x.c);
    const float* g = vptr(name + ".weight");
    const float* b = vptr(name + ".bias");
    const f16* xp = x.p; f16* yp = y.p; const int ldx = x.ld, ldy = y.ld, rows = x.rows(), C = x.c;
    push([=](hipStream_t s) { return layernorm(yp, ldy, xp, ldx, g, b, rows, C, 1e-5f, s); }, "layernorm", 0, 4.0 * rows * C,
         "rows" + std::to_string(rows) + " C" + std::to_string(C));
    return y;
  }

  void attn(const T& o, const f16* q, int ldq, const f16* k, int ldk, const f16* v, int ldv, int B, int H, int Tq, int Tk, int TkS, int TkSv, int d) {
    f16* op = o.p; const int ldo = o.ld;
    const float scale = 1.0f / sqrtf((float)d);
    push([=](hipStream_t s) { return attention(op, ldo, q, ldq, k, ldk, v, ldv, B, H, Tq, Tk, TkS, TkSv, d, scale, s); }, "attention",
         4.0 * B * H * (double)Tq * Tk * d, 2.0 * B * H * d * (2.0 * Tq + 2.0 * Tk),
         "Tq" + std::to_string(Tq) + " Tk" + std::to_string(Tk) + " d" + std::to_string(d));
  }
};

// ResBlock._forward (`openaimodel.py:255-275`)
static T build_res(Builder& b, const std::string& ns, const Blk& blk, const T& x, int net, const T* out = nullptr) {
  const std::string p = ns + blk.name;
  Builder::CO o1;
  o1.bias2_off = b.e->emb_row.at(p);
  o1.bias2 = b.e->emb_all[net] + o1.bias2_off;     // what planning / autotune see; the launch reads emb_cur / emb_ld_cur
  o1.ld_bias2 = b.e->emb_total[net];
  o1.bias2_cur = &b.e->emb_cur[net];
  o1.ld_bias2_cur = &b.e->emb_ld_cur[net];
  o1.gn_next = true;
  T h1 = b.gn_conv(x, p + ".in_layers.0", 1e-5f, 1, p + ".in_layers.2", blk.cout, o1);
  T skip;
  const T* res = &x;
  if (blk.cin != blk.cout) {
    skip = b.conv(x, p + ".skip_connection", blk.cout, 1, 1, 0);
    res = &skip;
  }
  Builder::CO o2;
  o2.res = res;
  o2.out = out;
  o2.gn_next = true;
  T y = b.gn_conv(h1, p + ".out_layers.0", 1e-5f, 1, p + ".out_layers.3", blk.cout, o2);
  b.release(h1);
  if (blk.cin != blk.cout) b.release(skip);
  return y;
}

struct CtxKV { T kv; };   // cross-attention K | V of one attention block: [N*TkS][2C] (K in columns 0..C-1, V in C..2C-1)

// SpatialTransformer.forward + BasicTransformerBlock._forward (`attention.py:381-385,431-450`).  Each pre-LN sub-block
// x + f(LN(x)) runs as: [GEMM that writes x also writes x's per-row statistics] -> [GEMM of f's first Linear on the RAW x with
// LN folded in] -> ... ; see the file comment.
static T build_attn(Builder& b, const std::string& ns, const Blk& blk, const T& x, const CtxKV& kv, const T* out = nullptr) {
  const sdeo_config& c = b.e->cfg;
  const std::string p = ns + blk.name;
  const std::string t = p + ".transformer_blocks.0";
  const int C = blk.cin, H = c.num_heads, d = C / H, N = x.n, Tq = x.h * x.w;
  const int TkS = round8(c.context_len);
  T g = b.gn(x, p + ".norm", 1e-6f, 0);
  RowStats st0 = b.alloc_stats(x.rows(), C), st1 = b.alloc_stats(x.rows(), C), st2 = b.alloc_stats(x.rows(), C);
  Builder::CO pi; pi.stats = &st0;
  T tok = b.conv(g, p + ".proj_in", C, 1, 1, 0, pi);
  b.release(g);
  // attn1 (self): q | k | v = LN1(tok) [Wq; Wk; Wv]^T in one GEMM
  Builder::CO l1; l1.ln = &st0; l1.ln_s = b.named_v(t + ".attn1.qkv_ln.s");
  T qkv = b.gemm(tok, b.named_w(t + ".attn1.qkv_ln.w"), C, 3 * C, b.named_v(t + ".attn1.qkv_ln.b"), l1);
  T o1 = b.alloc(x.n, x.h, x.w, C);
  b.attn(o1, qkv.p, 3 * C, qkv.p + C, 3 * C, qkv.p + 2 * C, 3 * C, N, H, Tq, Tq, Tq, Tq, d);
  b.release(qkv);
  Builder::CO r1; r1.res = &tok; r1.stats = &st1;
  T tok1 = b.gemm(o1, b.wptr(t + ".attn1.to_out.0.weight"), C, C, b.vptr(t + ".attn1.to_out.0.bias"), r1);
  b.release(o1);
  b.release(tok);
  // attn2 (cross, K | V precomputed from the context)
  Builder::CO l2; l2.ln = &st1; l2.ln_s = b.named_v(t + ".attn2.q_ln.s");
  T q2 = b.gemm(tok1, b.named_w(t + ".attn2.q_ln.w"), C, C, b.named_v(t + ".attn2.q_ln.b"), l2);
  T o2 = b.alloc(x.n, x.h, x.w, C);
  b.attn(o2, q2.p, C, kv.kv.p, 2 * C, kv.kv.p + C, 2 * C, N, H, Tq, c.context_len, TkS, TkS, d);
  b.release(q2);
  // ff.net.2 and proj_out are two Linear maps with only the residual add between them: composed at finalisation into ONE [C][5C]
  // matrix over the row-concatenated operand [GEGLU output (4C) | tok2 (C)] (ComposeJob), so attn2.to_out writes tok2 into the last C
  // columns of that operand, the GEGLU GEMM writes the first 4C, and one GEMM replaces two launches (SDEO_COMPOSE_FF=0: the two-launch form)
  static const bool compose = [] { const char* v = getenv("SDEO_COMPOSE_FF"); return !v || atoi(v) != 0; }();
  T cat, ggv, tok2v;
  if (compose) {
    cat = b.alloc(x.n, x.h, x.w, 5 * C);
    ggv = cat; ggv.c = 4 * C; ggv.off = (size_t)-1;
    tok2v = cat; tok2v.p = cat.p + 4 * C; tok2v.c = C; tok2v.off = (size_t)-1;
  }
  Builder::CO r2; r2.res = &tok1; r2.stats = &st2;
  if (compose) r2.out = &tok2v;
  T tok2 = b.gemm(o2, b.wptr(t + ".attn2.to_out.0.weight"), C, C, b.vptr(t + ".attn2.to_out.0.bias"), r2);
  b.release(o2);
  b.release(tok1);
  // GEGLU feed-forward: LN3 + ff.net.0.proj + GEGLU in one launch (act 3: value * gelu(gate) in the GEMM epilogue, 4C columns out)
  T gg = compose ? ggv : b.alloc(x.n, x.h, x.w, 4 * C);
  {
    Builder::CO og; og.act = 3; og.out = &gg; og.ln = &st2; og.ln_s = b.named_v(t + ".ff1_ln.s");
    b.gemm(tok2, b.named_w(t + ".ff1_ln.w"), C, 8 * C, b.named_v(t + ".ff1_ln.b"), og);
  }
  T y;
  if (compose) {
    b.release_stats();
    Builder::CO ro; ro.res = &x; ro.out = out; ro.gn_next = true;
    ro.w_ovr = b.named_w(t + ".ffproj.w"); ro.b_ovr = b.named_v(t + ".ffproj.b");
    y = b.conv(cat, p + ".proj_out", C, 1, 1, 0, ro);
    b.release(cat);
  } else {
    Builder::CO r3; r3.res = &tok2;
    T tok3 = b.gemm(gg, b.wptr(t + ".ff.net.2.weight"), 4 * C, C, b.vptr(t + ".ff.net.2.bias"), r3);
    b.release(gg);
    b.release(tok2);
    b.release_stats();
    Builder::CO ro; ro.res = &x; ro.out = out; ro.gn_next = true;
    y = b.conv(tok3, p + ".proj_out", C, 1, 1, 0, ro);
    b.release(tok3);
  }
  return y;
}

// time_embed MLP + stacked emb_layers projection: int64 t[rows] -> fp32 out[rows][emb_total[net]]
static void build_time_embed(Builder& b, const std::string& ns, int net, int rows, const int64_t* tp, float* out) {
  const sdeo_config& c = b.e->cfg;
  const int mc = c.model_channels, emb = 4 * mc, total = b.e->emb_total[net];
  T te = b.alloc2d(rows, mc);
  {
    f16* o = te.p;
    b.push([=](hipStream_t s) { return timestep_embedding(o, tp, rows, mc, s); });
  }
  Builder::CO a; a.act = 1;
  T e1 = b.gemm(te, b.wptr(ns + "time_embed.0.weight"), mc, emb, b.vptr(ns + "time_embed.0.bias"), a);
  b.release(te);
  // every consumer of emb applies SiLU first (emb_layers = SiLU -> Linear), so store SiLU(emb)
  T e2 = b.gemm(e1, b.wptr(ns + "time_embed.2.weight"), emb, emb, b.vptr(ns + "time_embed.2.bias"), a);
  b.release(e1);
  b.gemm(e2, b.named_w(ns + "emb_all.weight"), emb, total, b.named_v(ns + "emb_all.bias"), Builder::CO(), out, total);
  b.release(e2);
}

static std::vector<const Blk*> attn_blocks(const UPlan& p) {
  std::vector<const Blk*> v;
  auto acc = [&](const std::vector<Blk>& l) { for (auto& b : l) if (b.kind == B_ATTN) v.push_back(&b); };
  for (auto& l : p.in) acc(l);
  acc(p.mid);
  for (auto& l : p.out) acc(l);
  return v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// configure: plan the arena and build all programs
// ------------------------------------------------------------------------------------------------
namespace {

struct Built {
  std::unordered_map<std::string, CtxKV> kv[2];   // per net: attn block name -> cached K / V^T
  T ctx16;
  T hint_feat;
};

static int run(Engine* e, const Program& p, hipStream_t s, bool skip_zero_convs = false) {
  if (!e->profiling) {
    for (auto& op : p) {
      if (skip_zero_convs && op.zero_conv) continue;
      if (int rc = op(s)) return rc;
    }
    return 0;
  }
  static const bool detail = [] { const char* v = getenv("SDEO_PROFILE_DETAIL"); return v && atoi(v) != 0; }();
  for (auto& op : p) {
    if (skip_zero_convs && op.zero_conv) continue;
    ProfRec r{detail && !op.tag.empty() ? std::string(op.key) + " | " + op.tag : std::string(op.key), op.flops, op.bytes, nullptr, nullptr};
    SDEO_HIP(hipEventCreate(&r.a));
    SDEO_HIP(hipEventCreate(&r.b));
    SDEO_HIP(hipEventRecord(r.a, s));
    if (int rc = op(s)) return rc;
    SDEO_HIP(hipEventRecord(r.b, s));
    e->prof.push_back(r);
  }
  return 0;
}

static void build_all(Engine* e, Arena& arena, Arena& arena2, bool dry, size_t* max_splitk, size_t* max_gn, std::string* err) {
  const sdeo_config& c = e->cfg;
  const int N = e->N, h = e->lh, w = e->lw;
  Builder b{e, &arena, dry};
  b.base = e->arena;
  Built bt;
  const int TkS = round8(c.context_len);

  // ---- persistent tensors first (never released): controls, context, cached K/V^T, hint features
  for (size_t i = 0; i < e->cplan.in_ch.size(); ++i) {
    const int ds = e->cplan.in_ds[i];
    e->ctrl[i] = b.alloc(N, h / ds, w / ds, e->cplan.in_ch[i]);
  }
  {
    const int ds = e->cplan.in_ds.back();
    e->ctrl[e->cplan.in_ch.size()] = b.alloc(N, h / ds, w / ds, e->cplan.in_ch.back());
  }
  bt.ctx16 = b.alloc2d(N * TkS, c.context_dim);
  bt.hint_feat = b.alloc(N, h, w, c.model_channels);
  e->x0 = b.alloc(N, h, w, round8(c.in_channels));
  e->eps16 = b.alloc(N, h, w, 4 * ((c.out_channels + 3) / 4));
  const char* nss[2] = {NS_UNET, NS_CN};
  const UPlan* plans[2] = {&e->uplan, &e->cplan};
  for (int net = 0; net < 2; ++net)
    for (const Blk* ab : attn_blocks(*plans[net])) {
      CtxKV kv;
      kv.kv = b.alloc2d(N * TkS, 2 * ab->cin);
      bt.kv[net][std::string(nss[net]) + ab->name] = kv;
    }

  // ---- small programs (their temporaries come and go: only after every persistent tensor has its place)
  {   // latent in: NCHW fp32 -> NHWC fp16, once for both networks
    b.prog = &e->p_x0;
    f16* o = e->x0.p; const float* in = e->in_x; const int Cc = c.in_channels, ld = e->x0.ld, HW = h * w;
    b.push([=](hipStream_t s) { return nchw_f32_to_nhwc_f16(o, ld, in, N, Cc, HW, 1.0f, s); });
    b.prog = &e->p_eps_export;
    float* eo = e->out_eps; const f16* ein = e->eps16.p; const int eld = e->eps16.ld, Ce = c.out_channels;
    b.push([=](hipStream_t s) { return nhwc_f16_to_nchw_f32(eo, ein, eld, N, Ce, HW, 1.0f, s); });
  }
  {   // time embedding: per forward (rows = N, t from in_t) and per schedule (rows = kTabRows, t from tab_t), both networks
    // (the ControlNet's runs on the side stream beside the UNet encoder: its temporaries and split-K workspace are the side stream's)
    const char* tns[2] = {NS_UNET, NS_CN};
    for (int net = 0; net < 2; ++net) {
      b.prog = &e->p_temb[net];
      if (net == 1) { b.arena = &arena2; b.base = e->arena2; b.ws_sel = 1; }
      build_time_embed(b, tns[net], net, N, e->in_t, e->emb_all[net]);
      if (net == 1) { b.arena = &arena; b.base = e->arena; b.ws_sel = 0; }
    }
    b.prog = &e->p_temb_tab;
    for (int net = 0; net < 2; ++net) build_time_embed(b, tns[net], net, Engine::kTabRows, e->tab_t, e->temb_tab[net]);
  }

  // ---- context programs: fp32 [N][77][768] -> fp16 padded; K | V = ctx [Wk; Wv]^T per attn block, one GEMM each
  for (int net = 0; net < 2; ++net) {
    b.prog = net == 0 ? &e->p_ctx_unet : &e->p_ctx_cn;
    {
      f16* o = bt.ctx16.p; const float* in = e->in_ctx; const int T_ = c.context_len, Cd = c.context_dim;
      b.push([=](hipStream_t s) { return pad_rows_f32_to_f16(o, in, N, T_, TkS, Cd, s); });
    }
    for (const Blk* ab : attn_blocks(*plans[net])) {
      const std::string t = std::string(nss[net]) + ab->name + ".transformer_blocks.0";
      const CtxKV& kv = bt.kv[net][std::string(nss[net]) + ab->name];
      Builder::CO o; o.out = &kv.kv;
      b.gemm(bt.ctx16, b.named_w(t + ".attn2.to_kv"), c.context_dim, 2 * ab->cin, nullptr, o);
    }
  }

  // ---- hint program (`cldm/cldm.py:147-163,288`): 8 conv3x3, SiLU between, cached across steps
  {
    b.prog = &e->p_hint;
    T x = b.alloc(N, 8 * h, 8 * w, round8(c.hint_channels));
    {
      f16* o = x.p; const float* in = e->in_hint; const int Cc = c.hint_channels, ld = x.ld, HW = 64 * h * w;
      b.push([=](hipStream_t s) { return nchw_f32_to_nhwc_f16(o, ld, in, N, Cc, HW, 1.0f, s); });
    }
    for (size_t i = 0; i < e->hconvs.size(); ++i) {
      const HintConv& hc = e->hconvs[i];
      Builder::CO o;
      o.act = i + 1 < e->hconvs.size() ? 1 : 0;
      o.cout_store = round8(hc.cout);
      if (i + 1 == e->hconvs.size()) o.out = &bt.hint_feat;
      T y = b.conv(x, std::string(NS_CN) + hc.name, hc.cout, 3, hc.stride, 0, o);
      b.release(x);
      x = y;
    }
  }

  auto run_blocks = [&](const std::string& ns, const std::vector<Blk>& blocks, T x, bool release_in, int net, const T* final_out) -> T {
    for (size_t i = 0; i < blocks.size(); ++i) {
      const Blk& blk = blocks[i];
      const T* out = (i + 1 == blocks.size()) ? final_out : nullptr;
      T y;
      switch (blk.kind) {
        case B_CONV_IN: { Builder::CO o; o.out = out; o.gn_next = true; y = b.conv(x, ns + blk.name, blk.cout, 3, 1, 0, o); break; }
        case B_RES: y = build_res(b, ns, blk, x, net, out); break;
        case B_ATTN: y = build_attn(b, ns, blk, x, bt.kv[net].at(ns + blk.name), out); break;
        case B_DOWN: { Builder::CO o; o.out = out; o.gn_next = true; y = b.conv(x, ns + blk.name + ".op", blk.cout, 3, 2, 0, o); break; }
        case B_UP: { Builder::CO o; o.out = out; o.gn_next = true; y = b.conv(x, ns + blk.name + ".conv", blk.cout, 3, 1, 1, o); break; }
      }
      if (release_in || i > 0) b.release(x);
      x = y;
    }
    return x;
  };

  // ---- ControlNet program (`cldm/cldm.py:284-305`)
  {
    b.prog = &e->p_cn;
    b.arena = &arena2; b.base = e->arena2; b.ws_sel = 1;
    const std::string ns = NS_CN;
    // The block outputs stay allocated (cn_h): sdeo_apply_model / sdeo_ddim_step skip the zero convs here and let the UNet decoder apply
    // them with the skip connection as the residual operand (p_unet_dec_fused); sdeo_controlnet_forward runs them here (13 controls out).
    auto zero_conv = [&](const T& x, const std::string& name, int cout, const T* out) {
      const size_t first = e->p_cn.size();
      Builder::CO zo; zo.out = out;
      b.conv(x, name, cout, 1, 1, 0, zo);
      for (size_t k = first; k < e->p_cn.size(); ++k) e->p_cn[k].zero_conv = true;
    };
    T hcur = e->x0;
    for (size_t i = 0; i < e->cplan.in.size(); ++i) {
      T y;
      if (i == 0) {
        // input_blocks.0 conv, then h += guided_hint (residual epilogue)
        Builder::CO o; o.res = &bt.hint_feat; o.gn_next = true;
        y = b.conv(hcur, ns + e->cplan.in[0][0].name, c.model_channels, 3, 1, 0, o);
      } else {
        y = run_blocks(ns, e->cplan.in[i], hcur, false, 1, nullptr);
      }
      hcur = y;
      e->cn_h[i] = y;
      zero_conv(hcur, ns + "zero_convs." + std::to_string(i) + ".0", e->cplan.in_ch[i], &e->ctrl[i]);
    }
    T m = run_blocks(ns, e->cplan.mid, hcur, false, 1, nullptr);
    e->cn_h[e->cplan.in.size()] = m;
    zero_conv(m, ns + "middle_block_out.0", e->cplan.in_ch.back(), &e->ctrl[e->cplan.in.size()]);
    b.arena = &arena; b.base = e->arena; b.ws_sel = 0;
  }
  const int nctrl = (int)e->cplan.in.size() + 1;

  // ---- control export (fp16 NHWC -> fp32 NCHW boundary buffers) and import
  {
    e->ctrl_elems.clear();
    for (int i = 0; i < nctrl; ++i) {
      const T& t = e->ctrl[i];
      e->ctrl_elems.push_back((size_t)t.n * t.c * t.h * t.w);
      if (!dry) {
        float* o = e->out_ctrl[i]; const f16* in = t.p; const int ld = t.ld, Cc = t.c, HW = t.h * t.w;
        e->p_cn_export.push_back([=](hipStream_t s) { return nhwc_f16_to_nchw_f32(o, in, ld, N, Cc, HW, 1.0f, s); });
        const float* ci = e->in_ctrl[i]; f16* co = t.p;
        e->p_ctrl_import.push_back([=](hipStream_t s) { return nchw_f32_to_nhwc_f16(co, ld, ci, N, Cc, HW, 1.0f, s); });
      }
    }
  }

  // ---- UNet programs (`cldm/cldm.py:22-45`), with and without control
  for (int variant = 0; variant < 2; ++variant) {
    const bool with_ctrl = variant == 0;
    b.prog = with_ctrl ? &e->p_unet_enc : &e->p_unet_noctrl;
    const std::string ns = NS_UNET;
    std::vector<T> hs;
    T hcur = e->x0;
    for (size_t i = 0; i < e->uplan.in.size(); ++i) {
      T y = run_blocks(ns, e->uplan.in[i], hcur, false, 0, nullptr);
      hs.push_back(y);
      hcur = y;
    }
    // middle block; its output goes straight into the first concat buffer
    auto cat_for = [&](int c_h, const T& skip) { return b.alloc(skip.n, skip.h, skip.w, c_h + skip.c); };
    T cat0 = cat_for(e->uplan.mid.back().cout, hs.back());
    T view0 = cat0; view0.c = e->uplan.mid.back().cout; view0.off = (size_t)-1;
    run_blocks(ns, e->uplan.mid, hcur, false, 0, &view0);
    // The decoder (`openaimodel.py:797-801` with `cldm/cldm.py:33-41`): h = cat([h, hs.pop() + control.pop()]).  Three forms of the same
    // program: no control; controls given as tensors (the 13-tensor boundary: one add per control); controls applied as the ControlNet's
    // zero convs themselves, out = scale * zero_conv(cn_h) + skip written straight into the concat buffer (no add launches, no fp16
    // round trip of the control).  The latter two start from the same encoder state, so the arena is rewound between them.
    enum { D_NOCTRL, D_ADD, D_FUSED };
    auto build_decoder = [&](int mode, std::vector<T> hs, T cat, T view) {
      int ci = nctrl - 1;
      if (mode == D_ADD) {   // h += control.pop()
        f16* yp = view.p; const int ld = view.ld, rows = view.rows(), Cc = view.c;
        const f16* cp = e->ctrl[ci].p; const int ldc = e->ctrl[ci].ld; const float* sc = &e->scales[ci];
        b.push([=](hipStream_t s) { return add_scaled(yp, ld, yp, ld, cp, ldc, *sc, rows, Cc, s); });
      } else if (mode == D_FUSED) {
        Builder::CO zo; zo.out = &view; zo.res = &view; zo.scale_host = &e->eff_scales[ci];
        b.conv(e->cn_h[ci], std::string(NS_CN) + "middle_block_out.0", view.c, 1, 1, 0, zo);
      }
      --ci;
      for (size_t oi = 0; oi < e->uplan.out.size(); ++oi) {
        // second half of the concat buffer: hs.pop() (+ control.pop() unless only_mid_control)
        T skip = hs.back();
        hs.pop_back();
        const int c_h = cat.c - skip.c;
        if (mode == D_FUSED) {
          T half = cat; half.p = cat.p + c_h; half.c = skip.c; half.off = (size_t)-1;
          Builder::CO zo; zo.out = &half; zo.res = &skip; zo.scale_host = &e->eff_scales[ci];
          b.conv(e->cn_h[ci], std::string(NS_CN) + "zero_convs." + std::to_string(ci) + ".0", skip.c, 1, 1, 0, zo);
        } else {
          f16* yp = cat.p + c_h; const int ld = cat.ld, rows = cat.rows(), Cc = skip.c;
          const f16* ap = skip.p; const int lda = skip.ld;
          const f16* cp = mode == D_ADD ? e->ctrl[ci].p : nullptr; const int ldc = mode == D_ADD ? e->ctrl[ci].ld : 0;
          const float* sc = &e->scales[ci]; const int* om = &e->only_mid;
          b.push([=](hipStream_t s) { return add_scaled(yp, ld, ap, lda, (*om) ? nullptr : cp, ldc, *sc, rows, Cc, s); });
        }
        --ci;
        b.release(skip);
        const std::vector<Blk>& blocks = e->uplan.out[oi];
        if (oi + 1 < e->uplan.out.size()) {
          const T& nskip = hs.back();
          const int c_hn = blocks.back().cout;
          const int up = blocks.back().kind == B_UP ? 2 : 1;
          T ncat = b.alloc(cat.n, cat.h * up, cat.w * up, c_hn + nskip.c);
          T nview = ncat; nview.c = c_hn; nview.off = (size_t)-1;
          run_blocks(ns, blocks, cat, true, 0, &nview);
          cat = ncat;
        } else {
          T y = run_blocks(ns, blocks, cat, true, 0, nullptr);
          Builder::CO oo; oo.cout_store = 4 * ((c.out_channels + 3) / 4);
          oo.out = &e->eps16;
          b.gn_conv(y, ns + "out.0", 1e-5f, 1, ns + "out.2", c.out_channels, oo);
          b.release(y);
        }
      }
    };
    if (!with_ctrl) {
      build_decoder(D_NOCTRL, hs, cat0, view0);
    } else {
      const Arena after_encoder = arena;          // everything below needs the controls: runs after the join
      b.prog = &e->p_unet_dec;
      build_decoder(D_ADD, hs, cat0, view0);
      const size_t peak_add = arena.peak;
      arena = after_encoder;
      arena.peak = std::max(arena.peak, peak_add);
      b.prog = &e->p_unet_dec_fused;
      build_decoder(D_FUSED, hs, cat0, view0);
    }
  }

  // ---- VAE decode program, batch 1 (`model.py:619-652`; decode_first_stage wrapper: z/scale_factor ->
  //      post_quant_conv -> Decoder; the wrapper itself is absent from the reference tree)
  {
    b.prog = &e->p_vae;
    const std::string d = std::string(NS_VAE) + "decoder";
    T z = b.alloc(1, h, w, round8(c.vae_z_channels));
    {
      f16* o = z.p; const float* in = e->vae_in; const int Cc = c.vae_z_channels, ld = z.ld, HW = h * w;
      const float sc = 1.0f / c.vae_scale_factor;
      b.push([=](hipStream_t s) { return nchw_f32_to_nhwc_f16(o, ld, in, 1, Cc, HW, sc, s); });
    }
    Builder::CO pq; pq.cout_store = round8(c.vae_z_channels);
    T z2 = b.conv(z, std::string(NS_VAE) + "post_quant_conv", c.vae_z_channels, 1, 1, 0, pq);
    b.release(z);
    int bin = 0;
    auto levels = vae_levels(c, &bin);
    Builder::CO gnx; gnx.gn_next = true;       // every conv of the decoder below feeds a GroupNorm
    T hcur = b.conv(z2, d + ".conv_in", bin, 3, 1, 0, gnx);
    b.release(z2);
    auto vres = [&](const std::string& p, const T& x, int cin, int cout) {
      T h1 = b.gn_conv(x, p + ".norm1", 1e-6f, 1, p + ".conv1", cout, gnx);
      T sk; const T* res = &x;
      if (cin != cout) { sk = b.conv(x, p + ".nin_shortcut", cout, 1, 1, 0); res = &sk; }
      Builder::CO o; o.res = res; o.gn_next = true;
      T y = b.gn_conv(h1, p + ".norm2", 1e-6f, 1, p + ".conv2", cout, o);
      b.release(h1);
      if (cin != cout) b.release(sk);
      return y;
    };
    T y = vres(d + ".mid.block_1", hcur, bin, bin);
    b.release(hcur);
    hcur = y;
    {   // AttnBlock (`model.py:179-203`): single head of `bin` channels over h*w tokens
      const std::string p = d + ".mid.attn_1";
      const int Tn = h * w;
      T g = b.gn(hcur, p + ".norm", 1e-6f, 0);
      T q = b.conv(g, p + ".q", bin, 1, 1, 0);
      T k = b.conv(g, p + ".k", bin, 1, 1, 0);
      T o;
      if ((bin % 8 == 0 && bin <= 160) || bin == 256 || bin == 512) {
        // flash attention (d = 512 on the wide-head kernel: the four waves of a workgroup split the channels)
        T v = b.conv(g, p + ".v", bin, 1, 1, 0);
        b.release(g);
        o = b.alloc(1, h, w, bin);
        b.attn(o, q.p, q.ld, k.p, k.ld, v.p, v.ld, 1, 1, Tn, Tn, Tn, Tn, bin);
        b.release(q);
        b.release(k);
        b.release(v);
      } else {
        // other widths: scores materialised in fp32, row softmax, second GEMM
        T vt = b.gemm_t(g, b.wptr(p + ".v.weight"), bin, bin, b.vptr(p + ".v.bias"));
        b.release(g);
        T s32 = b.alloc2d(Tn, Tn * 2);
        float* sp = reinterpret_cast<float*>(s32.p);
        b.gemm(q, k.p, k.ld, Tn, nullptr, Builder::CO(), sp, Tn);
        b.release(q);
        b.release(k);
        T pr = b.alloc2d(Tn, Tn);
        {
          f16* pp = pr.p; const float sc = 1.0f / sqrtf((float)bin);
          b.push([=](hipStream_t s) { return softmax_rows(pp, Tn, sp, Tn, Tn, Tn, sc, s); });
        }
        b.release(s32);
        o = b.gemm(pr, vt.p, vt.ld, bin, nullptr);
        o.n = 1; o.h = h; o.w = w;
        b.release(pr);
        b.release(vt);
      }
      Builder::CO ro; ro.res = &hcur; ro.gn_next = true;
      T yo = b.conv(o, p + ".proj_out", bin, 1, 1, 0, ro);
      b.release(o);
      b.release(hcur);
      hcur = yo;
    }
    y = vres(d + ".mid.block_2", hcur, bin, bin);
    b.release(hcur);
    hcur = y;
    int last = bin;
    for (auto& L : levels) {
      for (size_t j = 0; j < L.blocks.size(); ++j) {
        y = vres(d + ".up." + std::to_string(L.level) + ".block." + std::to_string(j), hcur, L.blocks[j].first, L.blocks[j].second);
        b.release(hcur);
        hcur = y;
        last = L.blocks[j].second;
      }
      if (L.up) {
        y = b.conv(hcur, d + ".up." + std::to_string(L.level) + ".upsample.conv", last, 3, 1, 1, gnx);
        b.release(hcur);
        hcur = y;
      }
    }
    Builder::CO oo; oo.cout_store = 4 * ((c.vae_out_ch + 3) / 4);
    T img = b.gn_conv(hcur, d + ".norm_out", 1e-6f, 1, d + ".conv_out", c.vae_out_ch, oo);
    b.release(hcur);
    {
      float* o = e->vae_out; uint8_t* u8 = e->vae_u8; const f16* in = img.p; const int ld = img.ld, Cc = c.vae_out_ch, HW = 64 * h * w;
      b.push([=](hipStream_t s) {
        if (int rc = nhwc_f16_to_nchw_f32(o, in, ld, 1, Cc, HW, 1.0f, s)) return rc;
        return nhwc_f16_to_nhwc_u8(u8, in, ld, HW, Cc, s);
      });
    }
    b.release(img);
  }
  *max_splitk = b.max_splitk;
  *max_gn = b.max_gn;
  *err = b.err;
}

template <typename Tp>
static int dev_alloc(Engine* e, Tp** p, size_t bytes) {
  void* v = nullptr;
  SDEO_HIP(hipMalloc(&v, bytes < 256 ? 256 : bytes));
  e->extra_allocs.push_back(v);
  e->device_bytes += bytes;
  *p = reinterpret_cast<Tp*>(v);
  return 0;
}

static void free_configured(Engine* e) {
  for (void* v : e->extra_allocs) (void)hipFree(v);
  e->extra_allocs.clear();
  if (e->arena) (void)hipFree(e->arena);
  e->arena = nullptr;
  if (e->arena2) (void)hipFree(e->arena2);
  e->arena2 = nullptr;
  for (Program* p : {&e->p_hint, &e->p_ctx_cn, &e->p_ctx_unet, &e->p_cn, &e->p_cn_export, &e->p_ctrl_import, &e->p_unet_enc,
                     &e->p_unet_dec, &e->p_unet_noctrl, &e->p_vae, &e->p_temb[0], &e->p_temb[1], &e->p_temb_tab, &e->p_x0, &e->p_eps_export, &e->p_unet_dec_fused})
    p->clear();
  e->tab_count = 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C entry points
// ------------------------------------------------------------------------------------------------
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

int sdeo_create(const sdeo_config* cfg, sdeo_handle* out) {
  SDEO_CHECK(cfg && out, "sdeo_create: null argument");
  SDEO_CHECK(cfg->num_levels >= 1 && cfg->num_levels <= 8 && cfg->vae_num_levels >= 1 && cfg->vae_num_levels <= 8,
             "sdeo_create: bad level count");
  SDEO_CHECK(cfg->model_channels % 32 == 0 && cfg->vae_ch % 32 == 0, "sdeo_create: channels must be multiples of 32 (GroupNorm)");
  SDEO_CHECK(cfg->context_dim % 8 == 0, "sdeo_create: context_dim must be a multiple of 8");
  SDEO_CHECK((cfg->model_channels / cfg->num_heads) % 8 == 0, "sdeo_create: head dim must be a multiple of 8");
  std::unique_ptr<Engine> e(new Engine());
  e->cfg = *cfg;
  e->uplan = make_uplan(*cfg, true);
  e->cplan = make_uplan(*cfg, false);
  SDEO_CHECK(e->cplan.in.size() + 1 <= 13, "sdeo_create: more than 13 control tensors");
  e->hconvs = hint_convs(*cfg);
  if (const char* at = getenv("SDEO_AUTOTUNE")) e->autotune = atoi(at) != 0;
  if (const char* ov = getenv("SDEO_OVERLAP")) e->overlap = atoi(ov) != 0;
  SDEO_HIP(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
  SDEO_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  SDEO_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
  for (int i = 0; i < 13; ++i) e->scales[i] = 1.0f;
  build_registry(e.get());
  SDEO_HIP(hipMalloc((void**)&e->wslab, e->wslab_bytes));
  SDEO_HIP(hipMemset(e->wslab, 0, e->wslab_bytes));
  size_t mx = 0;
  for (auto& w : e->weights) {
    size_t n = 1;
    for (int i = 0; i < w.ndim; ++i) n *= (size_t)w.dims[i];
    mx = std::max(mx, n);
  }
  e->stage_bytes = mx * sizeof(float);
  SDEO_HIP(hipMalloc((void**)&e->stage, e->stage_bytes));
  e->device_bytes = e->wslab_bytes;
  *out = e.release();
  return 0;
}

int sdeo_destroy(sdeo_handle h) {
  if (!h) return 0;
  free_configured(h);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->wslab) (void)hipFree(h->wslab);
  if (h->q8slab) (void)hipFree(h->q8slab);
  if (h->mxslab) (void)hipFree(h->mxslab);
  if (h->stage) (void)hipFree(h->stage);
  delete h;
  return 0;
}

int sdeo_num_weights(sdeo_handle h) { return h ? (int)h->weights.size() : 0; }

int sdeo_weight_info(sdeo_handle h, int i, const char** name, int64_t dims[4], int* ndim) {
  SDEO_CHECK(h && i >= 0 && i < (int)h->weights.size(), "sdeo_weight_info: bad index");
  const WEntry& w = h->weights[i];
  if (name) *name = w.name.c_str();
  if (ndim) *ndim = w.ndim;
  if (dims) for (int k = 0; k < 4; ++k) dims[k] = k < w.ndim ? w.dims[k] : 1;
  return 0;
}

int sdeo_load_weight(sdeo_handle h, const char* name, const float* host_data, const int64_t* dims, int ndim, int strict) {
  SDEO_CHECK(h && name && host_data && dims, "sdeo_load_weight: null argument");
  auto it = h->windex.find(name);
  if (it == h->windex.end()) {
    if (strict) return fail("sdeo_load_weight: unexpected tensor '%s'", name);
    return 0;
  }
  WEntry& w = h->weights[it->second];
  SDEO_CHECK(ndim == w.ndim, "sdeo_load_weight: %s has %d dims, expected %d", name, ndim, w.ndim);
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    SDEO_CHECK(dims[i] == w.dims[i], "sdeo_load_weight: %s dim %d is %lld, expected %lld", name, i, (long long)dims[i],
               (long long)w.dims[i]);
    n *= (size_t)dims[i];
  }
  if (h->stage == nullptr) {
    SDEO_HIP(hipMalloc((void**)&h->stage, h->stage_bytes));
  }
  SDEO_HIP(hipMemcpy(h->stage, host_data, n * sizeof(float), hipMemcpyDefault));
  void* dst = h->wslab + w.off;
  int rc = 0;
  switch (w.kind) {
    case W_CONV:
      rc = oihw_f32_to_ohwi_f16((f16*)dst, h->stage, (int)w.dims[0], (int)w.dims[1], (int)w.dims[2], (int)w.dims[3], w.ipad, 0);
      break;
    case W_LINEAR: rc = f32_to_f16((f16*)dst, h->stage, (int64_t)n, 0); break;
    case W_VEC: SDEO_HIP(hipMemcpy(dst, h->stage, n * sizeof(float), hipMemcpyDeviceToDevice)); break;
    case W_GEGLU_W: rc = geglu_interleave_f32_to_f16((f16*)dst, h->stage, (int)(w.dims[0] / 2), (int)w.dims[1], 0); break;
    case W_GEGLU_B: rc = geglu_interleave_f32((float*)dst, h->stage, (int)(w.dims[0] / 2), 0); break;
  }
  if (rc) return rc;
  SDEO_HIP(hipDeviceSynchronize());
  w.loaded = true;
  return 0;
}

int sdeo_finalize_weights(sdeo_handle h) {
  SDEO_CHECK(h, "sdeo_finalize_weights: null handle");
  std::string missing;
  int nmiss = 0;
  for (auto& w : h->weights)
    if (!w.loaded) {
      if (nmiss < 5) missing += (nmiss ? ", " : "") + w.name;
      ++nmiss;
    }
  SDEO_CHECK(nmiss == 0, "sdeo_finalize_weights: %d tensors missing (%s%s)", nmiss, missing.c_str(), nmiss > 5 ? ", ..." : "");
  if (h->stage) { (void)hipFree(h->stage); h->stage = nullptr; }
  // LayerNorm-folded copies of the Linear layers that consume a LayerNorm (rebuilt on every finalize, from the raw tensors)
  for (const FoldJob& f : h->folds) {
    auto vec = [&](const std::string& n) -> const float* {
      return n.empty() ? nullptr : reinterpret_cast<const float*>(h->wslab + h->weights[h->windex.at(n)].off);
    };
    if (int rc = fold_layernorm(reinterpret_cast<f16*>(h->wslab + f.w_out), reinterpret_cast<float*>(h->wslab + f.s_out),
                                reinterpret_cast<float*>(h->wslab + f.b_out), reinterpret_cast<const f16*>(h->wslab + f.w_in),
                                vec(f.gamma), vec(f.beta), vec(f.bias), f.rows, f.C, 0))
      return rc;
  }
  if (h->weight_bits == 8) {
    // fp8 pack: codes + per-row scales in a slab of their own; the fp16 copies become the dequantised values
    if (!h->q8slab) {
      size_t sz = 0;
      for (QRegion& q : h->qregions) {
        q.q_off = align_up(sz, 256); sz = q.q_off + (size_t)q.rows * q.cols;
        q.s_off = align_up(sz, 256); sz = q.s_off + (size_t)q.rows * 4;
      }
      h->q8_bytes = align_up(sz, 256);
      SDEO_HIP(hipMalloc((void**)&h->q8slab, h->q8_bytes));
      h->device_bytes += h->q8_bytes;
    }
    for (const QRegion& q : h->qregions)
      if (int rc = quantize_fp8_rows(reinterpret_cast<uint8_t*>(h->q8slab + q.q_off), reinterpret_cast<float*>(h->q8slab + q.s_off),
                                     reinterpret_cast<f16*>(h->wslab + q.off), q.rows, q.cols, q.cols, q.cols, 0))
        return rc;
    for (const FoldJob& f : h->folds)          // the row sums of the LayerNorm fold must be those of the re-quantised matrix
      if (int rc = row_sums_f16(reinterpret_cast<float*>(h->wslab + f.s_out), reinterpret_cast<const f16*>(h->wslab + f.w_out), f.rows, f.C, 0))
        return rc;
  }
  // ff.net.2 x proj_out products, from the values the fp16 copies hold NOW (the dequantised ones when weight_bits == 8)
  for (const ComposeJob& cj : h->composes) {
    auto off = [&](const std::string& n) { return h->wslab + h->weights[h->windex.at(n)].off; };
    if (int rc = compose_proj(reinterpret_cast<f16*>(h->wslab + cj.w_out), reinterpret_cast<float*>(h->wslab + cj.b_out),
                              reinterpret_cast<const f16*>(off(cj.wp)), reinterpret_cast<const float*>(off(cj.bp)),
                              reinterpret_cast<const f16*>(off(cj.w2)), reinterpret_cast<const float*>(off(cj.b2)), cj.C, 4 * cj.C, 0))
      return rc;
  }
  if (h->act_bits == 8) {
    // block-scaled packs of every Linear / conv1x1 matrix whose K is a multiple of 128 (from the values the fp16 copies hold now,
    // i.e. after the per-row fp8 rounding when weight_bits == 8)
    if (!h->mxslab) {
      size_t sz = 0;
      for (const QRegion& q : h->qregions) {
        if (q.cols % 128) continue;
        sdeo_handle_s::MxRegion m{};
        m.rows = q.rows; m.cols = q.cols;
        m.q_off = align_up(sz, 256); sz = m.q_off + (size_t)q.rows * q.cols;
        m.s_off = align_up(sz, 256); sz = m.s_off + (size_t)q.rows * (q.cols / 32);
        h->mxindex[q.off] = m;
      }
      h->mx_bytes = align_up(sz, 256);
      SDEO_HIP(hipMalloc((void**)&h->mxslab, h->mx_bytes < 256 ? 256 : h->mx_bytes));
      h->device_bytes += h->mx_bytes;
    }
    for (auto& kv : h->mxindex)
      if (int rc = quantize_mx(reinterpret_cast<uint8_t*>(h->mxslab + kv.second.q_off), reinterpret_cast<uint8_t*>(h->mxslab + kv.second.s_off),
                               reinterpret_cast<const f16*>(h->wslab + kv.first), kv.second.rows, kv.second.cols, kv.second.cols, kv.second.cols,
                               kv.second.cols / 32, 0))
        return rc;
  }
  SDEO_HIP(hipDeviceSynchronize());
  h->finalized = true;
  return 0;
}

int sdeo_set_activation_precision(sdeo_handle h, int bits, int min_rows) {
  SDEO_CHECK(h, "sdeo_set_activation_precision: null handle");
  SDEO_CHECK(bits == 16 || bits == 8, "sdeo_set_activation_precision: %d bits unsupported (16 or 8)", bits);
  SDEO_CHECK(!h->finalized && !h->arena, "sdeo_set_activation_precision: call it before sdeo_finalize_weights / sdeo_configure");
  h->act_bits = bits;
  h->mx_min_rows = min_rows > 0 ? min_rows : 2048;
  return 0;
}
int sdeo_debug_mx_launches(sdeo_handle h) { return h ? h->mx_launches : -1; }

int sdeo_set_weight_precision(sdeo_handle h, int bits) {
  SDEO_CHECK(h, "sdeo_set_weight_precision: null handle");
  SDEO_CHECK(bits == 16 || bits == 8, "sdeo_set_weight_precision: %d bits unsupported (16 or 8)", bits);
  SDEO_CHECK(!h->finalized && !h->arena, "sdeo_set_weight_precision: call it before sdeo_finalize_weights / sdeo_configure");
  h->weight_bits = bits;
  return 0;
}

int sdeo_configure(sdeo_handle h, int n, int latent_h, int latent_w) {
  SDEO_CHECK(h, "sdeo_configure: null handle");
  SDEO_CHECK(n >= 1 && n <= 64, "sdeo_configure: n=%d out of range", n);
  const int maxds = 1 << (h->cfg.num_levels - 1);
  SDEO_CHECK(latent_h >= maxds && latent_w >= maxds && latent_h % maxds == 0 && latent_w % maxds == 0,
             "sdeo_configure: latent %dx%d must be a positive multiple of %d", latent_h, latent_w, maxds);
  SDEO_CHECK(h->weight_bits != 8 || h->q8slab, "sdeo_configure: fp8 weights are packed by sdeo_finalize_weights: call it first");
  free_configured(h);
  h->device_bytes = h->wslab_bytes + h->q8_bytes + h->mx_bytes;
  h->mx_launches = 0;
  h->N = n; h->lh = latent_h; h->lw = latent_w;
  const sdeo_config& c = h->cfg;
  const size_t px = (size_t)latent_h * latent_w;
  // boundary buffers
  if (int rc = dev_alloc(h, &h->in_x, (size_t)n * c.in_channels * px * 4)) return rc;
  if (int rc = dev_alloc(h, &h->in_hint, (size_t)n * c.hint_channels * px * 64 * 4)) return rc;
  if (int rc = dev_alloc(h, &h->in_ctx, (size_t)n * c.context_len * c.context_dim * 4)) return rc;
  if (int rc = dev_alloc(h, &h->in_t, (size_t)n * 8)) return rc;
  if (int rc = dev_alloc(h, &h->tab_t, (size_t)sdeo_handle_s::kTabRows * 8)) return rc;
  SDEO_HIP(hipMemset(h->tab_t, 0, (size_t)sdeo_handle_s::kTabRows * 8));
  for (int net = 0; net < 2; ++net) {
    if (int rc = dev_alloc(h, &h->emb_all[net], (size_t)n * h->emb_total[net] * 4)) return rc;
    if (int rc = dev_alloc(h, &h->temb_tab[net], (size_t)sdeo_handle_s::kTabRows * h->emb_total[net] * 4)) return rc;
  }
  if (int rc = dev_alloc(h, &h->out_eps, (size_t)n * c.out_channels * px * 4)) return rc;
  if (int rc = dev_alloc(h, &h->vae_in, (size_t)c.vae_z_channels * px * 4)) return rc;
  if (int rc = dev_alloc(h, &h->vae_out, (size_t)c.vae_out_ch * px * 64 * 4)) return rc;
  if (int rc = dev_alloc(h, &h->vae_u8, (size_t)c.vae_out_ch * px * 64)) return rc;
  const int nctrl = (int)h->cplan.in.size() + 1;
  for (int i = 0; i < nctrl; ++i) {
    const int idx = i < (int)h->cplan.in.size() ? i : (int)h->cplan.in.size() - 1;
    const int ds = h->cplan.in_ds[idx];
    const size_t elems = (size_t)n * h->cplan.in_ch[idx] * (latent_h / ds) * (latent_w / ds);
    if (int rc = dev_alloc(h, &h->in_ctrl[i], elems * 4)) return rc;
    if (int rc = dev_alloc(h, &h->out_ctrl[i], elems * 4)) return rc;
  }
  // pass 1: plan the arena (no launches recorded), pass 2: build for real
  size_t ms = 0, mg = 0;
  std::string err;
  {
    Arena a, a2;
    build_all(h, a, a2, true, &ms, &mg, &err);
    SDEO_CHECK(err.empty(), "sdeo_configure: %s", err.c_str());
    h->arena_bytes = align_up(a.peak, 256);
    h->arena2_bytes = align_up(a2.peak, 256);
  }
  SDEO_HIP(hipMalloc((void**)&h->arena, h->arena_bytes));
  SDEO_HIP(hipMemset(h->arena, 0, h->arena_bytes));
  SDEO_HIP(hipMalloc((void**)&h->arena2, h->arena2_bytes));
  SDEO_HIP(hipMemset(h->arena2, 0, h->arena2_bytes));
  h->device_bytes += h->arena_bytes + h->arena2_bytes;
  if (int rc = dev_alloc(h, &h->splitk_ws, ms)) return rc;
  if (int rc = dev_alloc(h, &h->splitk_ws2, ms)) return rc;
  h->splitk_ws_bytes = ms;
  if (int rc = dev_alloc(h, &h->gn_ws, mg)) return rc;
  if (int rc = dev_alloc(h, &h->gn_ws2, mg)) return rc;
  h->gn_ws_bytes = mg;
  {
    Arena a, a2;
    build_all(h, a, a2, false, &ms, &mg, &err);
    SDEO_CHECK(err.empty(), "sdeo_configure: %s", err.c_str());
    SDEO_CHECK(align_up(a.peak, 256) == h->arena_bytes && align_up(a2.peak, 256) == h->arena2_bytes,
               "sdeo_configure: arena plan not reproducible");
  }
  SDEO_HIP(hipDeviceSynchronize());      // autotune launches are done before the first real forward
  return 0;
}

static int copy_in(void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (dst == src) return 0;
  SDEO_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
  return 0;
}

static int stage_inputs(sdeo_handle h, const float* x, const float* hint, const int64_t* t, const float* ctx, int hint_is_new,
                        int ctx_for_net, hipStream_t s) {
  const sdeo_config& c = h->cfg;
  const size_t px = (size_t)h->lh * h->lw;
  if (x) if (int rc = copy_in(h->in_x, x, (size_t)h->N * c.in_channels * px * 4, s)) return rc;
  if (t) if (int rc = copy_in(h->in_t, t, (size_t)h->N * 8, s)) return rc;
  if (hint && hint_is_new) {
    if (int rc = copy_in(h->in_hint, hint, (size_t)h->N * c.hint_channels * px * 64 * 4, s)) return rc;
    if (int rc = run(h, h->p_hint, s)) return rc;
  }
  if (ctx) {
    if (int rc = copy_in(h->in_ctx, ctx, (size_t)h->N * c.context_len * c.context_dim * 4, s)) return rc;
    if (ctx_for_net & 1) if (int rc = run(h, h->p_ctx_unet, s)) return rc;
    if (ctx_for_net & 2) if (int rc = run(h, h->p_ctx_cn, s)) return rc;
  }
  return 0;
}

#define REQUIRE_READY(h)                                                                     \
  SDEO_CHECK(h, "null handle");                                                              \
  SDEO_CHECK(h->finalized, "weights not finalized (call sdeo_finalize_weights)");            \
  SDEO_CHECK(h->arena, "not configured (call sdeo_configure)")

// time embedding of this forward: row `row` of the schedule table (>= 0; every image of the batch at that timestep), or in_t via p_temb
static int select_time(sdeo_handle h, int flags, const int64_t* timesteps, const char* who, int* row_out) {
  const int row = (flags & 8) ? (flags >> 8) : -1;
  SDEO_CHECK(row >= 0 || timesteps, "%s: timesteps required", who);
  SDEO_CHECK(row < h->tab_count, "%s: timestep row %d, but the table holds %d (sdeo_set_timestep_table)", who, row, h->tab_count);
  for (int net = 0; net < 2; ++net) {
    h->emb_cur[net] = row >= 0 ? h->temb_tab[net] + (size_t)row * h->emb_total[net] : h->emb_all[net];
    h->emb_ld_cur[net] = row >= 0 ? 0 : h->emb_total[net];
  }
  *row_out = row;
  return 0;
}

// only_mid_control (`cldm/cldm.py:35-41`): the twelve skip controls are dropped, the middle one stays
static void set_effective_scales(sdeo_handle h) {
  const int mid = (int)h->cplan.in.size();
  for (int i = 0; i < 13; ++i) h->eff_scales[i] = (h->only_mid && i != mid) ? 0.0f : h->scales[i];
}

// ControlNet || UNet encoder, join, UNet decoder: eps16 holds the result.  The latent is already in x0.
static int run_step_programs(sdeo_handle h, bool no_control, bool time_from_table, hipStream_t s) {
  if (no_control) {
    if (!time_from_table) if (int rc = run(h, h->p_temb[0], s)) return rc;
    return run(h, h->p_unet_noctrl, s);
  }
  if (h->overlap && !h->profiling) {
    // fork: ControlNet on the side stream, UNet encoder + middle block on the caller's stream (capturable)
    SDEO_HIP(hipEventRecord(h->ev_fork, s));
    SDEO_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    if (!time_from_table) if (int rc = run(h, h->p_temb[1], h->side)) return rc;
    if (int rc = run(h, h->p_cn, h->side, true)) return rc;
    SDEO_HIP(hipEventRecord(h->ev_join, h->side));
    if (!time_from_table) if (int rc = run(h, h->p_temb[0], s)) return rc;
    if (int rc = run(h, h->p_unet_enc, s)) return rc;
    SDEO_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
  } else {
    if (!time_from_table) {
      if (int rc = run(h, h->p_temb[1], s)) return rc;
      if (int rc = run(h, h->p_temb[0], s)) return rc;
    }
    if (int rc = run(h, h->p_cn, s, true)) return rc;
    if (int rc = run(h, h->p_unet_enc, s)) return rc;
  }
  return run(h, h->p_unet_dec_fused, s);      // applies the zero convs itself (skipped above)
}

int sdeo_controlnet_forward(sdeo_handle h, const float* x_noisy, const float* hint, const int64_t* timesteps,
                            const float* context, float* const* controls, int flags, void* stream) {
  REQUIRE_READY(h);
  SDEO_CHECK(x_noisy && controls, "sdeo_controlnet_forward: null argument");
  hipStream_t s = S(stream);
  const int hint_new = !(flags & 1), ctx_new = !(flags & 2);
  SDEO_CHECK(!hint_new || hint, "sdeo_controlnet_forward: hint required");
  SDEO_CHECK(!ctx_new || context, "sdeo_controlnet_forward: context required");
  int trow = -1;
  if (int rc = select_time(h, flags, timesteps, "sdeo_controlnet_forward", &trow)) return rc;
  if (int rc = stage_inputs(h, x_noisy, hint, trow < 0 ? timesteps : nullptr, ctx_new ? context : nullptr, hint_new, 2, s)) return rc;
  if (int rc = run(h, h->p_x0, s)) return rc;
  if (trow < 0) if (int rc = run(h, h->p_temb[1], s)) return rc;
  if (int rc = run(h, h->p_cn, s)) return rc;
  if (int rc = run(h, h->p_cn_export, s)) return rc;
  for (size_t i = 0; i < h->ctrl_elems.size(); ++i)
    if (controls[i]) if (int rc = copy_in(controls[i], h->out_ctrl[i], h->ctrl_elems[i] * 4, s)) return rc;
  return 0;
}

int sdeo_unet_forward(sdeo_handle h, const float* x_noisy, const int64_t* timesteps, const float* context,
                      const float* const* controls, const float* host_control_scales, int only_mid_control, float* eps,
                      int flags, void* stream) {
  REQUIRE_READY(h);
  SDEO_CHECK(x_noisy && eps, "sdeo_unet_forward: null argument");
  hipStream_t s = S(stream);
  const int ctx_new = !(flags & 2);
  SDEO_CHECK(!ctx_new || context, "sdeo_unet_forward: context required");
  int trow = -1;
  if (int rc = select_time(h, flags, timesteps, "sdeo_unet_forward", &trow)) return rc;
  if (int rc = stage_inputs(h, x_noisy, nullptr, trow < 0 ? timesteps : nullptr, ctx_new ? context : nullptr, 0, 1, s)) return rc;
  h->only_mid = only_mid_control;
  for (int i = 0; i < 13; ++i) h->scales[i] = host_control_scales ? host_control_scales[i] : 1.0f;
  if (int rc = run(h, h->p_x0, s)) return rc;
  if (trow < 0) if (int rc = run(h, h->p_temb[0], s)) return rc;
  if (controls) {
    for (size_t i = 0; i < h->ctrl_elems.size(); ++i) {
      SDEO_CHECK(controls[i], "sdeo_unet_forward: control %zu is null", i);
      if (int rc = copy_in(h->in_ctrl[i], controls[i], h->ctrl_elems[i] * 4, s)) return rc;
    }
    if (int rc = run(h, h->p_ctrl_import, s)) return rc;
    if (int rc = run(h, h->p_unet_enc, s)) return rc;
    if (int rc = run(h, h->p_unet_dec, s)) return rc;
  } else {
    if (int rc = run(h, h->p_unet_noctrl, s)) return rc;
  }
  if (int rc = run(h, h->p_eps_export, s)) return rc;
  return copy_in(eps, h->out_eps, (size_t)h->N * h->cfg.out_channels * h->lh * h->lw * 4, s);
}

int sdeo_apply_model(sdeo_handle h, const float* x_noisy, const float* hint, const int64_t* timesteps, const float* context,
                     const float* host_control_scales, int only_mid_control, int flags, float* eps, void* stream) {
  REQUIRE_READY(h);
  SDEO_CHECK(x_noisy && eps, "sdeo_apply_model: null argument");
  hipStream_t s = S(stream);
  const int hint_new = !(flags & 1), ctx_new = !(flags & 2), no_control = (flags & 4) != 0;
  SDEO_CHECK(!ctx_new || context, "sdeo_apply_model: context required");
  SDEO_CHECK(no_control || !hint_new || hint, "sdeo_apply_model: hint required");
  int trow = -1;
  if (int rc = select_time(h, flags, timesteps, "sdeo_apply_model", &trow)) return rc;
  if (int rc = stage_inputs(h, x_noisy, no_control ? nullptr : hint, trow < 0 ? timesteps : nullptr, ctx_new ? context : nullptr, hint_new,
                            no_control ? 1 : 3, s))
    return rc;
  h->only_mid = only_mid_control;
  for (int i = 0; i < 13; ++i) h->scales[i] = host_control_scales ? host_control_scales[i] : 1.0f;
  set_effective_scales(h);
  if (int rc = run(h, h->p_x0, s)) return rc;
  if (int rc = run_step_programs(h, no_control, trow >= 0, s)) return rc;
  if (int rc = run(h, h->p_eps_export, s)) return rc;
  return copy_in(eps, h->out_eps, (size_t)h->N * h->cfg.out_channels * h->lh * h->lw * 4, s);
}

int sdeo_set_timestep_table(sdeo_handle h, const int64_t* host_timesteps, int count, void* stream) {
  REQUIRE_READY(h);
  SDEO_CHECK(host_timesteps && count >= 1 && count <= sdeo_handle_s::kTabRows, "sdeo_set_timestep_table: 1..%d timesteps (got %d)",
             sdeo_handle_s::kTabRows, count);
  hipStream_t s = S(stream);
  int64_t padded[sdeo_handle_s::kTabRows] = {0};
  for (int i = 0; i < count; ++i) padded[i] = host_timesteps[i];
  h->tab_count = 0;
  SDEO_HIP(hipMemcpyAsync(h->tab_t, padded, sizeof(padded), hipMemcpyHostToDevice, s));
  SDEO_HIP(hipStreamSynchronize(s));          // `padded` is a stack array; this call is made once per schedule, outside any capture
  if (int rc = run(h, h->p_temb_tab, s)) return rc;
  h->tab_count = count;
  return 0;
}

int sdeo_ddim_step(sdeo_handle h, float* x, float* pred_x0, int table_row, float cfg_scale, float a_t, float a_prev,
                   float sqrt_one_minus_at, const float* host_control_scales, int only_mid_control, int flags, void* stream) {
  REQUIRE_READY(h);
  SDEO_CHECK(x, "sdeo_ddim_step: null latent");
  SDEO_CHECK(h->cfg.in_channels == h->cfg.out_channels, "sdeo_ddim_step: eps and latent must have the same channel count");
  SDEO_CHECK(h->N % 2 == 0, "sdeo_ddim_step: configured for %d images; the CFG pair needs an even count (n = 2 x latents)", h->N);
  SDEO_CHECK(table_row >= 0 && table_row < h->tab_count, "sdeo_ddim_step: timestep row %d, but the table holds %d (sdeo_set_timestep_table)",
             table_row, h->tab_count);
  hipStream_t s = S(stream);
  const sdeo_config& c = h->cfg;
  const int b = h->N / 2, HW = h->lh * h->lw;
  int trow = -1;
  if (int rc = select_time(h, 8 | (table_row << 8), nullptr, "sdeo_ddim_step", &trow)) return rc;
  h->only_mid = only_mid_control;
  for (int i = 0; i < 13; ++i) h->scales[i] = host_control_scales ? host_control_scales[i] : 1.0f;
  set_effective_scales(h);
  if (!(flags & 16))
    if (int rc = latent_pair_to_nhwc(h->x0.p, h->x0.ld, x, b, c.in_channels, HW, s)) return rc;
  if (int rc = run_step_programs(h, false, true, s)) return rc;
  return cfg_ddim_pair(x, pred_x0, h->eps16.p, h->eps16.ld, h->x0.p, h->x0.ld, b, c.out_channels, HW, cfg_scale, a_t, a_prev,
                       sqrt_one_minus_at, s);
}

int sdeo_vae_decode(sdeo_handle h, const float* z, int n, float* images, uint8_t* images_u8, void* stream) {
  REQUIRE_READY(h);
  SDEO_CHECK(z && n >= 1 && (images || images_u8), "sdeo_vae_decode: bad argument");
  hipStream_t s = S(stream);
  const sdeo_config& c = h->cfg;
  const size_t px = (size_t)h->lh * h->lw;
  for (int i = 0; i < n; ++i) {
    if (int rc = copy_in(h->vae_in, z + (size_t)i * c.vae_z_channels * px, (size_t)c.vae_z_channels * px * 4, s)) return rc;
    if (int rc = run(h, h->p_vae, s)) return rc;
    if (images)
      if (int rc = copy_in(images + (size_t)i * c.vae_out_ch * px * 64, h->vae_out, (size_t)c.vae_out_ch * px * 64 * 4, s)) return rc;
    if (images_u8)
      if (int rc = copy_in(images_u8 + (size_t)i * c.vae_out_ch * px * 64, h->vae_u8, (size_t)c.vae_out_ch * px * 64, s)) return rc;
  }
  return 0;
}

size_t sdeo_device_bytes(sdeo_handle h) { return h ? h->device_bytes : 0; }

int sdeo_profile_begin(sdeo_handle h) {
  SDEO_CHECK(h, "sdeo_profile_begin: null handle");
  for (auto& r : h->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  h->prof.clear();
  h->profiling = true;
  return 0;
}

const char* sdeo_profile_end(sdeo_handle h) {
  if (!h) return "";
  h->profiling = false;
  (void)hipDeviceSynchronize();
  struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::vector<std::pair<std::string, Agg>> aggs;
  for (auto& r : h->prof) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
    size_t i = 0;
    for (; i < aggs.size(); ++i) if (aggs[i].first == r.key) break;
    if (i == aggs.size()) aggs.push_back({r.key, Agg()});
    aggs[i].second.n += 1; aggs[i].second.ms += ms; aggs[i].second.flops += r.flops; aggs[i].second.bytes += r.bytes;
  }
  h->prof.clear();
  std::string out = "[";
  char buf[768];
  for (size_t i = 0; i < aggs.size(); ++i) {
    snprintf(buf, sizeof(buf), "%s{\"kernel\": \"%s\", \"launches\": %ld, \"total_ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
             i ? ", " : "", aggs[i].first.c_str(), aggs[i].second.n, aggs[i].second.ms, aggs[i].second.flops, aggs[i].second.bytes);
    out += buf;
  }
  out += "]";
  h->prof_report = out;
  return h->prof_report.c_str();
}

}  // extern "C"
