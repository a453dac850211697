"""Python owner of one libsdeo handle (one per process / GPU): configuration, weight upload and the
net-level calls.  Device memory for inputs/outputs is borrowed from torch; all arithmetic is in libsdeo."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from . import _lib, spec as S
from ._lib import SdeoConfig, check, cur_stream, ptr

HINT_CACHED, CONTEXT_CACHED, NO_CONTROL = 1, 2, 4
STEP_LATENT_STAGED = 16


def TIMESTEP_ROW(i: int) -> int:
    """SDEO_TIMESTEP_ROW(i) of include/sdeo.h: run at row i of the table set with `set_timestep_table`"""
    return 8 | (int(i) << 8)



def make_config(ucfg: S.UNetConfig = S.UNET_SD15, vcfg: S.VAEConfig = S.VAE_SD15) -> SdeoConfig:
    c = SdeoConfig()
    c.in_channels, c.out_channels, c.hint_channels = ucfg.in_channels, ucfg.out_channels, ucfg.hint_channels
    c.model_channels, c.num_res_blocks = ucfg.model_channels, ucfg.num_res_blocks
    for i, m in enumerate(ucfg.channel_mult):
        c.channel_mult[i] = m
    c.num_levels = len(ucfg.channel_mult)
    for i, a in enumerate(ucfg.attention_resolutions):
        c.attention_resolutions[i] = a
    c.num_attention_resolutions = len(ucfg.attention_resolutions)
    c.num_heads, c.context_dim, c.context_len = ucfg.num_heads, ucfg.context_dim, ucfg.context_len
    c.vae_ch, c.vae_out_ch = vcfg.ch, vcfg.out_ch
    for i, m in enumerate(vcfg.ch_mult):
        c.vae_ch_mult[i] = m
    c.vae_num_levels, c.vae_num_res_blocks, c.vae_z_channels = len(vcfg.ch_mult), vcfg.num_res_blocks, vcfg.z_channels
    c.vae_scale_factor = vcfg.scale_factor
    return c


class SdeoRuntime:
    """create -> load_state_dict -> configure(n, h, w) -> controlnet / unet / apply_model / vae_decode."""

    def __init__(self, ucfg: S.UNetConfig = S.UNET_SD15, vcfg: S.VAEConfig = S.VAE_SD15, device: Optional[torch.device] = None,
                 weight_bits: int = 16, act_bits: int = 16, mx_min_rows: int = 0):
        """weight_bits = 8: the UNet / ControlNet matrices are packed to fp8 (OCP e4m3fn, per-output-channel power-of-two scale) when
        the weights are finalised (BASELINE configs[4]; the reference's precision switch is `onnx2trt_static_plugin.py:40-42`)."""
        if not torch.cuda.is_available():
            raise _lib.SdeoError("SdeoRuntime needs a HIP device (there is no CPU fallback)")
        self.lib = _lib.load()
        self.ucfg, self.vcfg = ucfg, vcfg
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.handle = C.c_void_p()
        self._cfg = make_config(ucfg, vcfg)
        check(self.lib.sdeo_create(C.byref(self._cfg), C.byref(self.handle)), "sdeo_create")
        self.weight_bits = int(weight_bits)
        if self.weight_bits != 16:
            check(self.lib.sdeo_set_weight_precision(self.handle, C.c_int(self.weight_bits)), "set_weight_precision")
        # act_bits = 8: the GEMMs of >= mx_min_rows rows (default 2048) run block-scaled fp8 x fp8 on the fp8 MFMA (sdeo.h)
        self.act_bits = int(act_bits)
        if self.act_bits != 16:
            check(self.lib.sdeo_set_activation_precision(self.handle, C.c_int(self.act_bits), C.c_int(int(mx_min_rows))), "set_activation_precision")
        self.n = self.h = self.w = 0
        self.n_controls = 13
        # bumped by every sdeo_configure: the library frees and re-plans its arenas / boundary buffers there, so a hipGraph
        # captured under an older generation points into freed memory and must be re-captured, never replayed
        self.generation = 0

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.sdeo_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    def expected_weights(self) -> Dict[str, tuple]:
        out = {}
        name = C.c_char_p()
        dims = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(self.lib.sdeo_num_weights(self.handle)):
            check(self.lib.sdeo_weight_info(self.handle, C.c_int(i), C.byref(name), dims, C.byref(nd)), "weight_info")
            out[name.value.decode()] = tuple(int(dims[k]) for k in range(nd.value))
        return out

    def load_tensor(self, name: str, t: torch.Tensor, strict: bool = True):
        t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
        dims = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        check(self.lib.sdeo_load_weight(self.handle, name.encode(), C.c_void_p(t.data_ptr()), dims, C.c_int(t.dim()),
                                        C.c_int(int(strict))), f"load_weight({name})")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """`sd` uses the reference checkpoint names (model.diffusion_model.*, control_model.*, first_stage_model.*).
        Tensors the hot path does not use (CLIP, VAE encoder, EMA ...) are ignored unless strict."""
        for k, v in sd.items():
            self.load_tensor(k, v, strict)
        check(self.lib.sdeo_finalize_weights(self.handle), "finalize_weights")

    def load_synthetic(self, seed: int = 0):
        """Seeded synthetic weights, generated tensor by tensor (never holds the full fp32 model on the host)."""
        for name, shape in self.expected_weights().items():
            self.load_tensor(name, S.synth_tensor(name, shape, seed))
        check(self.lib.sdeo_finalize_weights(self.handle), "finalize_weights")

    def load_synthetic_device(self, seed: int = 0):
        """Synthetic weights drawn on the GPU (device generator) -- fast path for bench.py; NOT reproducible on the
        CPU, so parity tests use load_synthetic instead.  Same distributions as spec.synth_tensor."""
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        for name, shape in self.expected_weights().items():
            leaf = name.rsplit(".", 1)[-1]
            if len(shape) == 1:
                is_norm = any(t in name for t in (".norm", "in_layers.0", "out_layers.0", "out.0", "norm_out"))
                t = torch.randn(shape, generator=g, device=self.device)
                t = 1.0 + 0.1 * t if (leaf == "weight" and is_norm) else 0.02 * t
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                t = torch.randn(shape, generator=g, device=self.device) * (1.0 / fan_in) ** 0.5
            t = t.contiguous()
            dims = (C.c_int64 * len(shape))(*shape)
            check(self.lib.sdeo_load_weight(self.handle, name.encode(), C.c_void_p(t.data_ptr()), dims, C.c_int(len(shape)),
                                            C.c_int(1)), f"load_weight({name})")
        check(self.lib.sdeo_finalize_weights(self.handle), "finalize_weights")

    # ---------------------------------------------------------------- profiling
    def profile_begin(self):
        check(self.lib.sdeo_profile_begin(self.handle), "profile_begin")

    def profile_end(self):
        import json
        self.lib.sdeo_profile_end.restype = C.c_char_p
        self.lib.sdeo_profile_end.argtypes = [C.c_void_p]
        return json.loads(self.lib.sdeo_profile_end(self.handle).decode())

    # ---------------------------------------------------------------- shapes
    def configure(self, n: int, h: int, w: int):
        if (n, h, w) != (self.n, self.h, self.w):
            self._graph_key = None
            self._graphs = None
            self.generation += 1
            self.n = self.h = self.w = 0            # a failed configure leaves the handle unconfigured
            check(self.lib.sdeo_configure(self.handle, C.c_int(n), C.c_int(h), C.c_int(w)), "configure")
            self.n, self.h, self.w = n, h, w
        return self

    def control_shapes(self) -> List[tuple]:
        plan = S.unet_plan(self.ucfg, with_decoder=False)
        shp = [(self.n, c, self.h // ds, self.w // ds) for c, ds in zip(plan.input_block_chans, plan.input_block_ds)]
        shp.append(shp[-1])
        return shp

    def device_bytes(self) -> int:
        return int(self.lib.sdeo_device_bytes(self.handle))

    # ---------------------------------------------------------------- forward calls
    def _f32(self, t, shape=None):
        if t is None:
            return None
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise _lib.SdeoError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def _t64(self, t):
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        if tuple(t.shape) != (self.n,):
            raise _lib.SdeoError(f"timesteps must have shape ({self.n},)")
        return t

    def _scales(self, scales):
        if scales is None:
            return None
        return (C.c_float * 13)(*[float(s) for s in list(scales) + [1.0] * (13 - len(scales))])

    def controlnet(self, x, hint, t, ctx, flags: int = 0, outs: Optional[Sequence[torch.Tensor]] = None):
        u = self.ucfg
        x = self._f32(x, (self.n, u.in_channels, self.h, self.w))
        hint = self._f32(hint, (self.n, u.hint_channels, 8 * self.h, 8 * self.w)) if hint is not None else None
        ctx = self._f32(ctx, (self.n, u.context_len, u.context_dim)) if ctx is not None else None
        t = self._t64(t)
        if outs is None:
            outs = [torch.empty(s, dtype=torch.float32, device=self.device) for s in self.control_shapes()]
        arr = (C.c_void_p * 13)(*[o.data_ptr() for o in outs])
        check(self.lib.sdeo_controlnet_forward(self.handle, ptr(x), ptr(hint), ptr(t), ptr(ctx), arr, C.c_int(flags),
                                               cur_stream()), "controlnet_forward")
        return list(outs)

    def unet(self, x, t, ctx, control=None, scales=None, only_mid_control=False, flags: int = 0, out=None):
        u = self.ucfg
        x = self._f32(x, (self.n, u.in_channels, self.h, self.w))
        ctx = self._f32(ctx, (self.n, u.context_len, u.context_dim)) if ctx is not None else None
        t = self._t64(t)
        arr = None
        keep = None
        if control is not None:
            keep = [self._f32(c, s) for c, s in zip(control, self.control_shapes())]
            arr = (C.c_void_p * 13)(*[c.data_ptr() for c in keep])
        eps = out if out is not None else torch.empty((self.n, u.out_channels, self.h, self.w), dtype=torch.float32,
                                                      device=self.device)
        check(self.lib.sdeo_unet_forward(self.handle, ptr(x), ptr(t), ptr(ctx), arr, self._scales(scales),
                                         C.c_int(int(only_mid_control)), ptr(eps), C.c_int(flags), cur_stream()), "unet_forward")
        return eps

    def apply_model(self, x, hint, t, ctx, scales=None, only_mid_control=False, flags: int = 0, out=None):
        u = self.ucfg
        x = self._f32(x, (self.n, u.in_channels, self.h, self.w))
        hint = self._f32(hint, (self.n, u.hint_channels, 8 * self.h, 8 * self.w)) if hint is not None else None
        ctx = self._f32(ctx, (self.n, u.context_len, u.context_dim)) if ctx is not None else None
        t = self._t64(t) if t is not None else None          # None: flags carry TIMESTEP_ROW(i)
        if hint is None and not (flags & HINT_CACHED):
            flags |= NO_CONTROL
        eps = out if out is not None else torch.empty((self.n, u.out_channels, self.h, self.w), dtype=torch.float32,
                                                      device=self.device)
        check(self.lib.sdeo_apply_model(self.handle, ptr(x), ptr(hint), ptr(t), ptr(ctx), self._scales(scales),
                                        C.c_int(int(only_mid_control)), C.c_int(flags), ptr(eps), cur_stream()), "apply_model")
        return eps

    def apply_model_graphed(self, x, t, scales=None, only_mid_control=False):
        """apply_model with the hint block and context K/V cached, replayed from a hipGraph (the reference captures its
        TensorRT engines the same way, `Engine.py:139-152`).  The graph holds both streams of the step (ControlNet on the
        side stream, UNet encoder on the main one), so the GPU sees the whole fork/join at once.  Inputs are copied into
        fixed device buffers; the returned eps tensor is owned by the runtime (valid until the next call)."""
        key = (self.generation, self.n, self.h, self.w, tuple(float(s) for s in (scales or [])), bool(only_mid_control))
        if getattr(self, "_graph_key", None) != key:
            u = self.ucfg
            self._gx = torch.zeros((self.n, u.in_channels, self.h, self.w), dtype=torch.float32, device=self.device)
            self._gt = torch.zeros((self.n,), dtype=torch.int64, device=self.device)
            self._geps = torch.zeros((self.n, u.out_channels, self.h, self.w), dtype=torch.float32, device=self.device)
            self._gx.copy_(x)
            self._gt.copy_(t)
            flags = HINT_CACHED | CONTEXT_CACHED
            self.apply_model(self._gx, None, self._gt, None, scales, only_mid_control, flags, self._geps)   # warm-up, eager
            torch.cuda.synchronize(self.device)
            # (measured, tools/step_replay.py: two instances of the capture replayed alternately are SLOWER, 6.91 vs 6.78 ms per
            # step; the host enqueues a replay in 1.6 ms, so the step is GPU-bound and one instance is enough)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.apply_model(self._gx, None, self._gt, None, scales, only_mid_control, flags, self._geps)
            self._graphs, self._graph_key = [g], key
        self._gx.copy_(x)
        self._gt.copy_(t)
        self._graphs[0].replay()
        return self._geps

    def set_timestep_table(self, timesteps) -> int:
        """Hand the sampler's schedule (a host sequence of ints, in the order the loop visits them) to the library: the time
        embeddings of both networks for every step are computed once (`sdeo_set_timestep_table`).  Returns the number of rows."""
        ts = [int(t) for t in timesteps]
        arr = (C.c_int64 * len(ts))(*ts)
        check(self.lib.sdeo_set_timestep_table(self.handle, arr, C.c_int(len(ts)), cur_stream()), "set_timestep_table")
        self._table_key = (self.generation, tuple(ts))      # whose schedule the table holds (captured graphs read it by address)
        return len(ts)

    def ddim_step(self, x, pred_x0, row: int, cfg_scale: float, a_t: float, a_prev: float, sqrt_one_minus_at: float, scales=None,
                  only_mid_control: bool = False, staged: bool = False):
        """`sdeo_ddim_step`: one eta = 0 DDIM step of the CFG pair; x (b,4,h,w) fp32 contiguous is updated in place."""
        assert x.is_contiguous() and x.dtype == torch.float32 and 2 * x.shape[0] == self.n
        assert pred_x0 is None or (pred_x0.is_contiguous() and pred_x0.dtype == torch.float32 and pred_x0.shape == x.shape)
        check(self.lib.sdeo_ddim_step(self.handle, ptr(x), ptr(pred_x0), C.c_int(int(row)), C.c_float(cfg_scale), C.c_float(a_t),
                                      C.c_float(a_prev), C.c_float(sqrt_one_minus_at), self._scales(scales),
                                      C.c_int(int(only_mid_control)), C.c_int(STEP_LATENT_STAGED if staged else 0), cur_stream()),
              "ddim_step")
        return x

    def vae_decode(self, z, want_u8: bool = False):
        """z (b,4,h,w) latents (sampler output) -> images (b,3,8h,8w) fp32 in [-1,1] (+ optional NHWC uint8)."""
        v = self.vcfg
        z = self._f32(z)
        b = z.shape[0]
        if tuple(z.shape[1:]) != (v.z_channels, self.h, self.w):
            raise _lib.SdeoError(f"latent shape {tuple(z.shape)} does not match the configured {self.h}x{self.w}")
        img = torch.empty((b, v.out_ch, 8 * self.h, 8 * self.w), dtype=torch.float32, device=self.device)
        u8 = torch.empty((b, 8 * self.h, 8 * self.w, v.out_ch), dtype=torch.uint8, device=self.device) if want_u8 else None
        check(self.lib.sdeo_vae_decode(self.handle, ptr(z), C.c_int(b), ptr(img), ptr(u8), cur_stream()), "vae_decode")
        return (img, u8) if want_u8 else img


class ClipRuntime:
    """CLIP text transformer on the HIP path: create -> load_state_dict -> configure(batch) ->
    encode(tokens).  Mirrors what `FrozenCLIPEmbedder.forward` does after tokenisation
    (`ldm/modules/encoders/modules.py:126-131`: `self.transformer(input_ids=tokens).last_hidden_state`)."""

    def __init__(self, cfg: S.ClipConfig = S.CLIP_SD15, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise _lib.SdeoError("ClipRuntime needs a HIP device (there is no CPU fallback)")
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.handle = C.c_void_p()
        self._cfg = _lib.SdeoClipConfig(cfg.vocab, cfg.positions, cfg.width, cfg.layers, cfg.heads, cfg.ffn)
        check(self.lib.sdeo_clip_create(C.byref(self._cfg), C.byref(self.handle)), "sdeo_clip_create")
        self.batch = 0
        self.generation = 0          # bumped by every re-plan of the activation buffers (see SdeoRuntime.generation)

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.sdeo_clip_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    def expected_weights(self) -> Dict[str, tuple]:
        out = {}
        name = C.c_char_p()
        dims = (C.c_int64 * 2)()
        nd = C.c_int()
        for i in range(self.lib.sdeo_clip_num_weights(self.handle)):
            check(self.lib.sdeo_clip_weight_info(self.handle, C.c_int(i), C.byref(name), dims, C.byref(nd)), "clip_weight_info")
            out[name.value.decode()] = tuple(int(dims[k]) for k in range(nd.value))
        return out

    def load_tensor(self, name: str, t: torch.Tensor, strict: bool = True):
        t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
        dims = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        check(self.lib.sdeo_clip_load_weight(self.handle, name.encode(), C.c_void_p(t.data_ptr()), dims, C.c_int(t.dim()),
                                             C.c_int(int(strict))), f"clip_load_weight({name})")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """Accepts HuggingFace names (`text_model.*`, or bare), or the SD checkpoint's `cond_stage_model.transformer.text_model.*`;
        anything else (e.g. `position_ids`, the UNet tensors of a full checkpoint) is ignored unless strict."""
        for k, v in sd.items():
            if not torch.is_floating_point(v):
                continue
            self.load_tensor(k, v, strict)
        check(self.lib.sdeo_clip_finalize_weights(self.handle), "clip_finalize_weights")
        return self

    def load_synthetic(self, seed: int = 0):
        for name, shape in self.expected_weights().items():
            self.load_tensor(name, S.synth_tensor(S.NS_CLIP + name, shape, seed))
        check(self.lib.sdeo_clip_finalize_weights(self.handle), "clip_finalize_weights")
        return self

    def configure(self, batch: int):
        if batch != self.batch:
            self.generation += 1
            self.batch = 0
            check(self.lib.sdeo_clip_configure(self.handle, C.c_int(batch)), "sdeo_clip_configure")
            self.batch = batch
        return self

    def encode(self, tokens: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tokens: integer [batch, positions] -> fp32 [batch, positions, width] on the device (into `out` when given)."""
        if tokens.dim() != 2 or tokens.shape[1] != self.cfg.positions:
            raise ValueError(f"tokens must be [batch, {self.cfg.positions}], got {tuple(tokens.shape)}")
        if tokens.shape[0] != self.batch:
            self.configure(int(tokens.shape[0]))
        tok = tokens.to(device=self.device, dtype=torch.int32).contiguous()
        shape = (self.batch, self.cfg.positions, self.cfg.width)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise _lib.SdeoError(f"encode: out must be a contiguous fp32 tensor of shape {shape}")
        check(self.lib.sdeo_clip_encode(self.handle, ptr(tok), C.c_int(self.batch), ptr(out), cur_stream()), "sdeo_clip_encode")
        return out

    def device_bytes(self) -> int:
        return int(self.lib.sdeo_clip_device_bytes(self.handle))
