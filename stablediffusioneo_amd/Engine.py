"""`Engine.py` of the reference (TensorRT context + torch buffers + CUDA graph, `Engine.py:45-161`) re-seated on
libsdeo: same class name, methods, attributes, tensor names / binding order and error behaviour, so the callers
(`cldm_trt/ddim_hacked.py:25-44,140-169`, `cldm_trt/cldm.py:321-384`, `trt_check.py`) keep working.

`engine_path` used to name a serialized TensorRT plan; here only its basename matters: it selects which network of
the shared libsdeo handle this Engine drives ("ControlNet*", "ControlledUnet*", "Decoder*", "CLIP*" as in
`trt_check.py:4` / `ldm_trt/modules/encoders/modules.py:112`).  The weights come from `Engine.weights_source` (a state
dict, a checkpoint path, or "synthetic:<seed>"), loaded once per process.

    native seam:  context.set_tensor_address(...) + context.execute_async_v3(stream)   (`Engine.py:136-157`)
    becomes:      sdeo_controlnet_forward / sdeo_unet_forward / sdeo_vae_decode(handle, device pointers, stream)
    CUDA graph:   cudaStreamBeginCapture / cudaGraphLaunch (`:139-152`) -> torch.cuda.CUDAGraph (hipGraph) around the same call
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch

from . import spec as S
from ._lib import SdeoError
from .runtime import ClipRuntime, SdeoRuntime

_shared = {}


def shared_runtime(ucfg=S.UNET_SD15, vcfg=S.VAE_SD15, source=None) -> SdeoRuntime:
    """One libsdeo handle per process and GPU (weights are shared by the ControlNet / UNet / Decoder engines)."""
    key = (torch.cuda.current_device(), ucfg, vcfg)
    if key not in _shared:
        rt = SdeoRuntime(ucfg, vcfg)
        src = source if source is not None else Engine.weights_source
        if isinstance(src, dict):
            rt.load_state_dict(src)
        elif isinstance(src, str) and src.startswith("synthetic"):
            rt.load_synthetic(int(src.split(":")[1]) if ":" in src else 0)
        elif isinstance(src, str):
            from .cldm.model import load_state_dict
            rt.load_state_dict(load_state_dict(src))
        else:
            raise SdeoError("Engine.weights_source is not set (state dict, checkpoint path or 'synthetic:<seed>')")
        _shared[key] = rt
    return _shared[key]


def shared_clip_runtime(ccfg=S.CLIP_SD15, source=None) -> ClipRuntime:
    """One CLIP text-encoder handle per process and GPU (`sdeo_clip_*`), weights from the same source as the networks: a state
    dict / checkpoint holding `cond_stage_model.transformer.text_model.*` (or bare HuggingFace names), or "synthetic:<seed>"."""
    key = (torch.cuda.current_device(), ccfg)
    if key not in _shared:
        rt = ClipRuntime(ccfg)
        src = source if source is not None else Engine.weights_source
        if isinstance(src, dict):
            rt.load_state_dict({k: v for k, v in src.items() if "text_model." in k or not k.startswith(("model.", "control_model.", "first_stage_model."))})
        elif isinstance(src, str) and src.startswith("synthetic"):
            rt.load_synthetic(int(src.split(":")[1]) if ":" in src else 0)
        elif isinstance(src, str):
            from .cldm.model import load_state_dict
            rt.load_state_dict({k: v for k, v in load_state_dict(src).items() if "text_model." in k})
        else:
            raise SdeoError("Engine.weights_source is not set (state dict, checkpoint path or 'synthetic:<seed>')")
        _shared[key] = rt
    return _shared[key]


class Engine():
    weights_source = os.environ.get("SDEO_WEIGHTS")     # process-wide default
    unet_config = S.UNET_SD15
    vae_config = S.VAE_SD15
    clip_config = S.CLIP_SD15

    def __init__(self, engine_path):
        self.engine_path = engine_path
        self.engine = None
        self.context = None
        self.buffers = OrderedDict()
        self.tensors = OrderedDict()
        self.latent_h = 32
        self.latent_w = 48
        self.batch_size = 1
        self.cuda_graph_instance = None
        self._graph_generation = None
        # read by controlunet_model_shape_dict (`Engine.py:72-77`; the reference never sets them, so its helper raises
        # AttributeError there -- here they carry the values the export script uses, `export_onnx_all.py:193-196`)
        self.unet_dim = 4
        self.text_maxlen = 77
        self.embedding_dim = 768
        name = os.path.basename(str(engine_path)).lower()
        if "clip" in name:
            self.kind = "clip"
        elif "controlnet" in name or "control_net" in name:
            self.kind = "controlnet"
        elif "unet" in name:
            self.kind = "unet"
        elif "decoder" in name or "vae" in name:
            self.kind = "decoder"
        else:
            self.kind = "unsupported"

    # ---- shape helpers (`Engine.py:67-91`)
    def clip_model_shape_dict(self, batch_size, text_maxlen, embedding_dim):
        return {"input_ids": (batch_size, text_maxlen), "last_hidden_state": (batch_size, text_maxlen, embedding_dim)}

    @property
    def latent_height(self):
        return self.latent_h

    @property
    def latent_width(self):
        return self.latent_w

    def controlunet_model_shape_dict(self):
        """`Engine.py:72-77` (the fused CFG pair: 2 x batch)."""
        return {"sample": (2 * self.batch_size, self.unet_dim, self.latent_height, self.latent_width),
                "encoder_hidden_states": (2 * self.batch_size, self.text_maxlen, self.embedding_dim),
                "latent": (2 * self.batch_size, 4, self.latent_height, self.latent_width)}

    def control_model_shape_dict(self):
        return {"x_noisy": (self.batch_size, 4, self.latent_h, self.latent_w)}

    def decoder_model_shape_dict(self):
        return {"latent": (self.batch_size, 4, self.latent_h, self.latent_w),
                "images": (self.batch_size, 3, self.latent_h * 8, self.latent_w * 8)}

    def load(self):
        if self.kind == "unsupported":
            raise SdeoError(f"{self.engine_path}: the engine name selects the network and must contain one of "
                            f"'CLIP', 'ControlNet', 'Unet', 'Decoder' / 'VAE' (the reference's plan names, `cldm_trt/ddim_hacked.py:25-44`)")
        print(f"Loading libsdeo engine: {self.engine_path} ({self.kind})")
        if self.kind == "clip":
            self.engine = shared_clip_runtime(self.clip_config)
        else:
            self.engine = shared_runtime(self.unet_config, self.vae_config)
        return self

    def activate(self, reuse_device_memory=None):
        self.context = self.engine     # the handle is its own execution context
        return self

    def allocate_buffers(self, shape_dict=None, device="cuda"):
        rt = self.engine
        self.cuda_graph_instance = None      # a graph captured over the previous buffers must not be replayed
        if self.kind == "clip":              # `Engine.py:67-71`: input_ids int32 -> last_hidden_state
            cc = rt.cfg
            n, tlen = (shape_dict or {}).get("input_ids", (self.batch_size, cc.positions))
            if tlen != cc.positions:
                raise SdeoError(f"CLIP engine: text_maxlen {tlen} != {cc.positions}")
            self.batch_size = n
            rt.configure(n)
            t = OrderedDict()
            t["input_ids"] = torch.zeros((n, cc.positions), dtype=torch.int32, device=device)
            t["last_hidden_state"] = torch.empty((n, cc.positions, cc.width), dtype=torch.float32, device=device)
            self.tensors = t
            return self
        u, v = rt.ucfg, rt.vcfg
        key = "latent" if self.kind == "decoder" else "x_noisy"
        n, _, h, w = (shape_dict or {}).get(key, (self.batch_size, 4, self.latent_h, self.latent_w))
        self.batch_size, self.latent_h, self.latent_w = n, h, w
        if self.kind != "decoder":
            rt.configure(n, h, w)
        elif (rt.h, rt.w) != (h, w):
            rt.configure(max(rt.n, 1), h, w)
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=device)
        t = OrderedDict()
        if self.kind == "controlnet":       # binding order of ControlNet.onnx (`export_onnx_all.py:199-200`)
            t["x_noisy"] = f32(n, u.in_channels, h, w)
            t["hint"] = f32(n, u.hint_channels, 8 * h, 8 * w)
            t["timestep"] = torch.empty((n,), dtype=torch.int32, device=device)
            t["context"] = f32(n, u.context_len, u.context_dim)
            for i, s in enumerate(rt.control_shapes()):    # positions 4..16 (`cldm_trt/ddim_hacked.py:144-149`)
                t[f"control{i}"] = f32(*s)
        elif self.kind == "unet":           # `export_onnx_all.py:258-262`
            t["x_noisy"] = f32(n, u.in_channels, h, w)
            t["timestep"] = torch.empty((n,), dtype=torch.int32, device=device)
            t["context"] = f32(n, u.context_len, u.context_dim)
            for i, s in enumerate(rt.control_shapes()):
                t[f"control{i}"] = f32(*s)
            t["latent"] = f32(n, u.out_channels, h, w)
        else:                               # `Engine.py:82-86`
            t["latent"] = f32(n, v.z_channels, h, w)
            t["images"] = f32(n, v.out_ch, 8 * h, 8 * w)
        self.tensors = t
        self._t64 = torch.empty((n,), dtype=torch.int64, device=device)
        return self

    def get_engine_infor(self):
        nin = {"controlnet": 4, "unet": 16, "decoder": 1, "clip": 1}[self.kind]
        names = list(self.tensors)
        print("libsdeo engine infors -----------------")
        print("engin nInput: ", nin, ", Input shape: ", {k: tuple(self.tensors[k].shape) for k in names[:nin]})
        print("engin nOutput: ", len(names) - nin, ", Outpu shape: ", {k: tuple(self.tensors[k].shape) for k in names[nin:]})

    def _generation(self):
        return getattr(self.engine, "generation", 0)

    def _bind(self):
        """Make the shared handle's problem size this engine's (several Engine objects share one handle: another engine's
        allocate_buffers may have re-planned it).  A re-plan bumps the handle's generation, which invalidates captured graphs."""
        rt = self.engine
        if self.kind == "clip":
            if rt.batch != self.batch_size:
                rt.configure(self.batch_size)
        elif self.kind == "decoder":
            if (rt.h, rt.w) != (self.latent_h, self.latent_w) or rt.n < 1:
                rt.configure(max(rt.n, 1), self.latent_h, self.latent_w)
        else:
            rt.configure(self.batch_size, self.latent_h, self.latent_w)

    def _execute(self):
        rt, t = self.engine, self.tensors
        if self.kind == "clip":
            rt.encode(t["input_ids"], out=t["last_hidden_state"])
        elif self.kind == "controlnet":
            self._t64.copy_(t["timestep"])
            rt.controlnet(t["x_noisy"], t["hint"], self._t64, t["context"], outs=[t[f"control{i}"] for i in range(13)])
        elif self.kind == "unet":
            self._t64.copy_(t["timestep"])
            rt.unet(t["x_noisy"], self._t64, t["context"], control=[t[f"control{i}"] for i in range(13)], out=t["latent"])
        else:
            t["images"].copy_(rt.vae_decode(t["latent"] * rt.vcfg.scale_factor))

    def infer(self, feed_dict, stream=None, use_cuda_graph=False):
        """Copy-in, execute (or graph-launch), return the engine-owned tensor dict (`Engine.py:131-161`): callers must
        `.clone()` what they keep.  Failures raise ValueError("ERROR: inference failed.") like the reference."""
        ts = None
        if stream is not None:
            p = getattr(stream, "ptr", None) or getattr(stream, "cuda_stream", None)
            ts = torch.cuda.ExternalStream(int(p)) if p else None
        ctx = torch.cuda.stream(ts) if ts is not None else torch.cuda.stream(torch.cuda.current_stream())
        try:
            with ctx:
                for name, buf in feed_dict.items():
                    self.tensors[name].copy_(buf)
                self._bind()
                if self.cuda_graph_instance is not None and self._graph_generation != self._generation():
                    self.cuda_graph_instance = None      # the handle was re-planned since the capture: its kernels point into freed memory
                if use_cuda_graph:
                    if self.cuda_graph_instance is not None:
                        self.cuda_graph_instance.replay()
                        torch.cuda.current_stream().synchronize()
                    else:
                        self._execute()                      # eager run before capture, as the reference does
                        torch.cuda.current_stream().synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=torch.cuda.current_stream() if ts is not None else None):
                            self._execute()
                        self.cuda_graph_instance = g
                        self._graph_generation = self._generation()
                else:
                    self._execute()
        except SdeoError as e:
            raise ValueError(f"ERROR: inference failed. ({e})")
        return self.tensors
