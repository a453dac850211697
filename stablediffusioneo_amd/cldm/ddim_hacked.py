"""DDIM sampler with the reference's surface (`cldm/ddim_hacked.py:10-317` and the TensorRT variant's
`sample_simple`, `cldm_trt/ddim_hacked.py:88-197`): same method names, arguments, return values and schedule
arithmetic; the per-step work is two C-ABI calls (sdeo_apply_model on the fused CFG pair + sdeo_cfg_ddim_step).

Differences that are deliberate and documented:
  * the conditional and unconditional passes of classifier-free guidance (`:190-191`, two sequential
    apply_model calls) run as ONE batch-2B pass when both use the same hint; GroupNorm / LayerNorm / attention are
    per-sample, so this is the same arithmetic;
  * the hint block and the cross-attention K/V projections are computed at the first step and reused
    (they do not depend on the timestep or on x);
  * buffers are not forced onto "cuda" by name (`:17-21`), they follow the model's device.
"""
from __future__ import annotations

import numpy as np
import torch

import os

from .. import ops
from ..runtime import CONTEXT_CACHED, HINT_CACHED

# replay the per-step apply_model from a hipGraph once hint / context are cached (SDEO_GRAPH=0: eager launches)
USE_GRAPH = os.environ.get("SDEO_GRAPH", "1") != "0"
# ... and replay steps 2..S of a deterministic (eta = 0) fused-CFG loop as ONE hipGraph (SDEO_LOOP_GRAPH=0: one replay per step).
# Measured on MI355X (tools/step_replay.py): the per-step replay costs 6.99 ms per step inside the sampler loop against 6.78 ms
# for back-to-back replays; the small kernels between two replays (cat, fill, copies, CFG / DDIM update) are what is in between.
USE_LOOP_GRAPH = os.environ.get("SDEO_LOOP_GRAPH", "1") != "0"         # needs USE_GRAPH as well (checked per call)
# DDIM steps per captured graph (0 = all remaining steps in ONE graph).  The rocprofv3 timeline of the one-graph loop (13.6 k kernel
# nodes) shows seven ~0.7 ms chip-wide stalls per image, but they belong to the profiler: measured without it on one box, one graph
# 6.63 ms per step, graphs of 4 / 2 / 1 steps 6.73 / 6.76 / 6.74 ms.
LOOP_GRAPH_STEPS = max(0, int(os.environ.get("SDEO_LOOP_GRAPH_STEPS", "0")))


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """`ldm/modules/diffusionmodules/util.py:46-60`."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = ddim_timesteps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """`util.py:63-74`."""
    alphacums = np.asarray(alphacums)
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule {sigmas}")
    return sigmas, alphas, alphas_prev


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self._cache_key = None

    def register_buffer(self, name, attr):
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """`cldm/ddim_hacked.py:23-52`.  The buffers depend only on (steps, discretisation, eta) and the model's schedule, so a repeat
        call with the same arguments keeps them: the reference rebuilds them for every image, which here would cost a stream
        synchronisation (the pageable host -> device copies below) and leave the GPU idle between two images."""
        # ... and the model's schedule: the tensor object, its in-place version counter and its storage identify what the buffers
        # were computed from (a re-registered or reloaded schedule on the same model object must not reuse stale alphas / sigmas)
        ac = self.model.alphas_cumprod
        fp = (id(ac), int(getattr(ac, "_version", 0)), int(ac.data_ptr()) if isinstance(ac, torch.Tensor) else 0, tuple(ac.shape))
        key = (int(ddim_num_steps), str(ddim_discretize), float(ddim_eta), id(self.model), self.ddpm_num_timesteps, fp)
        if getattr(self, "_schedule_key", None) == key:
            return
        self._schedule_key = None
        self._make_schedule(ddim_num_steps, ddim_discretize, ddim_eta, verbose)
        self._schedule_key = key

    def _make_schedule(self, ddim_num_steps, ddim_discretize, ddim_eta, verbose):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        f32 = lambda x: x.clone().detach().to(torch.float32).to(self.model.device)
        ac = alphas_cumprod.detach().cpu()
        self.register_buffer("betas", f32(self.model.betas))
        self.register_buffer("alphas_cumprod", f32(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", f32(self.model.alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", f32(torch.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(torch.sqrt(1. - ac)))
        self.register_buffer("log_one_minus_alphas_cumprod", f32(torch.log(1. - ac)))
        self.register_buffer("sqrt_recip_alphas_cumprod", f32(torch.sqrt(1. / ac)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(torch.sqrt(1. / ac - 1)))
        sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(alphacums=ac.numpy(), ddim_timesteps=self.ddim_timesteps,
                                                                    eta=ddim_eta, verbose=verbose)
        self._sigmas_all_zero = bool(float(np.abs(np.asarray(sigmas)).max()) == 0.0)      # host-side, once per schedule (loop-graph gate)
        self.register_buffer("ddim_sigmas", sigmas)
        self.register_buffer("ddim_alphas", alphas)
        self.register_buffer("ddim_alphas_prev", alphas_prev)
        self.register_buffer("ddim_sqrt_one_minus_alphas", np.sqrt(1. - alphas))
        sig_orig = ddim_eta * torch.sqrt((1 - self.alphas_cumprod_prev) / (1 - self.alphas_cumprod) *
                                         (1 - self.alphas_cumprod / self.alphas_cumprod_prev))
        self.register_buffer("ddim_sigmas_for_original_num_steps", sig_orig)

    # ------------------------------------------------------------------------------------------ sample
    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None, **kwargs):
        """`cldm/ddim_hacked.py:54-120`."""
        if conditioning is not None and isinstance(conditioning, dict):
            ctmp = conditioning[list(conditioning.keys())[0]]
            while isinstance(ctmp, list):
                ctmp = ctmp[0]
            if ctmp is not None and ctmp.shape[0] != batch_size:
                print(f"Warning: Got {ctmp.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f"Data shape for DDIM sampling is {size}, eta {eta}")
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature, score_corrector=score_corrector,
                                  corrector_kwargs=corrector_kwargs, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning,
                                  dynamic_threshold=dynamic_threshold, ucg_schedule=ucg_schedule)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, dynamic_threshold=None, ucg_schedule=None):
        """`cldm/ddim_hacked.py:122-178`."""
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        if timesteps is None:
            timesteps = self.ddpm_num_timesteps if ddim_use_original_steps else self.ddim_timesteps
        elif not ddim_use_original_steps:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = list(reversed(range(0, timesteps))) if ddim_use_original_steps else np.flip(timesteps)
        total_steps = timesteps if ddim_use_original_steps else timesteps.shape[0]
        self._cache_key = None
        if (mask is None and callback is None and img_callback is None and ucg_schedule is None and not ddim_use_original_steps
                and score_corrector is None and not quantize_denoised and dynamic_threshold is None
                and self._loop_graph_ok(img, cond, unconditional_conditioning, unconditional_guidance_scale, total_steps)):
            return self._loop_graphed(img, cond, unconditional_conditioning, unconditional_guidance_scale, time_range,
                                      total_steps, log_every_t, intermediates)
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                img_orig = self.model.q_sample(x0, ts)
                img = img_orig * mask + (1. - mask) * img
            if ucg_schedule is not None:
                assert len(ucg_schedule) == len(time_range)
                unconditional_guidance_scale = ucg_schedule[i]
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, use_original_steps=ddim_use_original_steps,
                                              quantize_denoised=quantize_denoised, temperature=temperature,
                                              noise_dropout=noise_dropout, score_corrector=score_corrector,
                                              corrector_kwargs=corrector_kwargs,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning,
                                              dynamic_threshold=dynamic_threshold)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    # ------------------------------------------------------------------------------------------ whole-loop graph
    def _fusable(self, c, uc, scale):
        m = self.model
        return (uc is not None and scale != 1. and hasattr(m, "rt") and isinstance(c, dict) and isinstance(uc, dict)
                and c.get("c_concat") is not None and uc.get("c_concat") is not None)

    def _loop_graph_ok(self, img, c, uc, scale, total_steps):
        """Steps 2..S as one graph: only the plain canny2image loop qualifies -- fused CFG pair, eta = 0 (no noise drawn inside
        the loop), eps-parameterisation, no mask / callbacks / correctors (the callers check those)."""
        if not (USE_GRAPH and USE_LOOP_GRAPH and img.is_cuda and total_steps >= 2 and self._fusable(c, uc, scale)):
            return False
        if self.model.parameterization != "eps":
            return False
        return bool(getattr(self, "_sigmas_all_zero", False))      # computed with the schedule: no device -> host copy per image

    def _loop_graphed(self, img, c, uc, scale, time_range, total_steps, log_every_t, intermediates):
        """The loop of `cldm/ddim_hacked.py:137-160` with step 1 run eagerly (it computes the hint block and the context K / V of
        THIS image into the runtime's caches) and steps 2..S replayed from one captured graph.  The graph reads the latent from a
        fixed buffer and the conditioning from those caches, so it is reused for every image of the same shape, step count,
        guidance scale and control scales; per-step constants (alpha_t, ...) are baked in at capture."""
        m, rt = self.model, self.model.rt
        b = img.shape[0]
        dev = img.device
        ts = torch.full((b,), int(time_range[0]), device=dev, dtype=torch.long)
        img, pred_x0 = self.p_sample_ddim(img, c, ts, index=total_steps - 1, unconditional_guidance_scale=scale,
                                          unconditional_conditioning=uc)
        intermediates["x_inter"].append(img)             # index == total_steps - 1
        intermediates["pred_x0"].append(pred_x0)
        key = (rt.generation, tuple(img.shape), tuple(int(t) for t in time_range), float(scale), int(log_every_t),
               tuple(float(v) for v in m.control_scales), bool(m.only_mid_control), getattr(self, "_schedule_key", None), LOOP_GRAPH_STEPS)
        if getattr(rt, "_table_key", None) != (rt.generation, tuple(int(t) for t in time_range)):
            rt.set_timestep_table(time_range)            # another schedule used the runtime since (the graphs read the table by address)
        if getattr(self, "_loop_key", None) != key:
            self._loop_key = None
            self._loop_x = torch.empty_like(img)
            self._loop_x.copy_(img)
            a_t, a_p, s1m = self.ddim_alphas, self.ddim_alphas_prev, self.ddim_sqrt_one_minus_alphas
            # the schedule is known: the time embeddings of all steps were computed once above (row i = step i of the loop) and each step is
            # ONE library call (sdeo_ddim_step: apply_model on [x; x] + CFG + DDIM update, x updated in place in the fixed buffer)
            self._loop_pred = torch.empty_like(img)
            torch.cuda.synchronize(dev)
            graphs, kept_x, kept_p = [], [], []
            per_graph = LOOP_GRAPH_STEPS if LOOP_GRAPH_STEPS > 0 else total_steps
            for first in range(1, total_steps, per_graph):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for i in range(first, min(first + per_graph, total_steps)):
                        index = total_steps - i - 1
                        rt.ddim_step(self._loop_x, self._loop_pred, i, scale, float(a_t[index]), float(a_p[index]), float(s1m[index]),
                                     m.control_scales, m.only_mid_control, staged=i > first)
                        if index % log_every_t == 0:
                            kept_x.append(self._loop_x.clone())
                            kept_p.append(self._loop_pred.clone())
                graphs.append(g)
            self._loop_graphs, self._loop_kept = graphs, (kept_x, kept_p)
            self._loop_key = key
        else:
            self._loop_x.copy_(img)
        for g in self._loop_graphs:
            g.replay()
        # the graphs' tensors are overwritten by the next replay: hand out copies
        intermediates["x_inter"].extend(t.clone() for t in self._loop_kept[0])
        intermediates["pred_x0"].extend(t.clone() for t in self._loop_kept[1])
        return self._loop_x.clone(), intermediates

    # ------------------------------------------------------------------------------------------ one step
    def _eps_pair(self, x, c, t, uc, scale):
        """eps for (cond, uncond); returns (eps_c, eps_u) with eps_u None when guidance is off."""
        m = self.model
        if uc is None or scale == 1.:
            return m.apply_model(x, t, c), None
        fusable = (hasattr(m, "rt") and isinstance(c, dict) and isinstance(uc, dict) and c.get("c_concat") is not None
                   and uc.get("c_concat") is not None)
        if not fusable:
            # e.g. guess mode: the unconditional branch runs without ControlNet (`canny2image_torch.py:48`)
            return m.apply_model(x, t, c), m.apply_model(x, t, uc)
        b = x.shape[0]

        # identity of the CALLER's conditioning tensors (storage, in-place version, shape): a key built from the result of
        # torch.cat would compare addresses of temporaries, which match from step to step only by allocator luck
        def ident(ts):
            return tuple((v.data_ptr(), v._version, tuple(v.shape)) for v in ts)

        key = (ident(c["c_concat"]), ident(uc["c_concat"]), ident(c["c_crossattn"]), ident(uc["c_crossattn"]), tuple(x.shape))
        if key == self._cache_key:
            rt = m.rt.configure(2 * b, x.shape[2], x.shape[3])
            if USE_GRAPH:
                eps2 = rt.apply_model_graphed(torch.cat([x, x]), torch.cat([t, t]), m.control_scales, m.only_mid_control)
            else:
                eps2 = rt.apply_model(torch.cat([x, x]), None, torch.cat([t, t]), None, m.control_scales,
                                      m.only_mid_control, HINT_CACHED | CONTEXT_CACHED)
        else:
            hint_c, hint_u = torch.cat(c["c_concat"], 1), torch.cat(uc["c_concat"], 1)
            ctx_c, ctx_u = torch.cat(c["c_crossattn"], 1), torch.cat(uc["c_crossattn"], 1)
            pair = {"c_concat": [torch.cat([hint_c, hint_u])], "c_crossattn": [torch.cat([ctx_c, ctx_u])]}
            eps2 = m.apply_model(torch.cat([x, x]), torch.cat([t, t]), pair)
            self._cache_key = key
        return eps2[:b], eps2[b:]

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None):
        """`cldm/ddim_hacked.py:180-231` (eps-parameterisation; v-prediction, score correctors, x0 quantisation and
        dynamic thresholding are not on the canny2image path and raise)."""
        if self.model.parameterization != "eps":
            raise NotImplementedError("only eps-parameterisation is on the canny2image path")
        if score_corrector is not None or quantize_denoised or dynamic_threshold is not None:
            raise NotImplementedError("score_corrector / quantize_denoised / dynamic_threshold are off the hot path")
        eps_c, eps_u = self._eps_pair(x, c, t, unconditional_conditioning, unconditional_guidance_scale)
        alphas = self.model.alphas_cumprod if use_original_steps else self.ddim_alphas
        alphas_prev = self.model.alphas_cumprod_prev if use_original_steps else self.ddim_alphas_prev
        sqrt_1m = self.model.sqrt_one_minus_alphas_cumprod if use_original_steps else self.ddim_sqrt_one_minus_alphas
        sigmas = self.ddim_sigmas_for_original_num_steps if use_original_steps else self.ddim_sigmas
        a_t, a_prev = float(alphas[index]), float(alphas_prev[index])
        sigma_t, s1m = float(sigmas[index]), float(sqrt_1m[index])
        noise = None
        if sigma_t != 0.:
            shape = (1, *x.shape[1:]) if repeat_noise else x.shape
            noise = torch.randn(shape, device=x.device).expand(x.shape).contiguous() * temperature
            if noise_dropout > 0.:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        x_prev, pred_x0 = ops.cfg_ddim_step(x.contiguous(), eps_c.contiguous(), None if eps_u is None else eps_u.contiguous(),
                                            unconditional_guidance_scale, a_t, a_prev, sigma_t, s1m, noise=noise)
        return x_prev, pred_x0

    # ------------------------------------------------------------------------------------------ TRT-variant entry
    @torch.no_grad()
    def sample_simple(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
                      img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
                      score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, dynamic_threshold=None,
                      ucg_schedule=None, **kwargs):
        """`cldm_trt/ddim_hacked.py:88-197`: sample + ddim_sampling + p_sample_ddim flattened, eps-parameterisation,
        no mask / corrector paths.  (`canny2image_TRT.py:80` calls this.)"""
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        device = self.model.device
        img = torch.randn((batch_size, C, H, W), device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        time_range = np.flip(self.ddim_timesteps)
        total_steps = self.ddim_timesteps.shape[0]
        self._cache_key = None
        if self._loop_graph_ok(img, conditioning, unconditional_conditioning, unconditional_guidance_scale, total_steps):
            return self._loop_graphed(img, conditioning, unconditional_conditioning, unconditional_guidance_scale, time_range,
                                      total_steps, log_every_t, intermediates)
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((batch_size,), int(step), device=device, dtype=torch.long)
            img, pred_x0 = self.p_sample_ddim(img, conditioning, ts, index=index, temperature=temperature,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    # ------------------------------------------------------------------------------------------ adjacent features (F4)
    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """`cldm/ddim_hacked.py:281-295`."""
        if use_original_steps:
            sqrt_ac, sqrt_1m = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        else:
            sqrt_ac = torch.sqrt(torch.as_tensor(self.ddim_alphas, dtype=torch.float32, device=x0.device))
            sqrt_1m = torch.as_tensor(self.ddim_sqrt_one_minus_alphas, dtype=torch.float32, device=x0.device)
        if noise is None:
            noise = torch.randn_like(x0)
        ext = lambda a: a.to(x0.device)[t].reshape(-1, *([1] * (x0.dim() - 1)))
        return ext(sqrt_ac) * x0 + ext(sqrt_1m) * noise

    @torch.no_grad()
    def encode(self, x0, c, t_enc, use_original_steps=False, return_intermediates=None, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, callback=None):
        """DDIM inversion, `cldm/ddim_hacked.py:233-279`: t_enc deterministic steps from x0 towards noise.  (With guidance the
        reference concatenates the two conditionings as tensors, `:257-259`, which only works for tensor conditionings; here the
        pair goes through the same fused cond / uncond pass as sampling.)"""
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        assert t_enc <= timesteps.shape[0]
        num_steps = t_enc
        if use_original_steps:
            alphas_next = self.model.alphas_cumprod[:num_steps].double().cpu().numpy()
            alphas = self.model.alphas_cumprod_prev[:num_steps].double().cpu().numpy()
        else:
            alphas_next = np.asarray(self.ddim_alphas[:num_steps], dtype=np.float64)
            alphas = np.asarray(self.ddim_alphas_prev[:num_steps], dtype=np.float64)
        x_next = x0.to(device=self.model.device, dtype=torch.float32)
        intermediates, inter_steps = [], []
        self._cache_key = None
        for i in range(num_steps):
            t = torch.full((x0.shape[0],), int(timesteps[i]), device=self.model.device, dtype=torch.long)
            if unconditional_guidance_scale == 1.:
                noise_pred = self.model.apply_model(x_next, t, c)
            else:
                assert unconditional_conditioning is not None
                e_c, e_u = self._eps_pair(x_next, c, t, unconditional_conditioning, unconditional_guidance_scale)
                noise_pred = e_u + unconditional_guidance_scale * (e_c - e_u)
            xt_weighted = float(np.sqrt(alphas_next[i] / alphas[i])) * x_next
            weighted_noise_pred = float(np.sqrt(alphas_next[i]) * (np.sqrt(1 / alphas_next[i] - 1) - np.sqrt(1 / alphas[i] - 1))) * noise_pred
            x_next = xt_weighted + weighted_noise_pred
            if return_intermediates and i % (num_steps // return_intermediates) == 0 and i < num_steps - 1:
                intermediates.append(x_next)
                inter_steps.append(i)
            elif return_intermediates and i >= num_steps - 2:
                intermediates.append(x_next)
                inter_steps.append(i)
            if callback:
                callback(i)
        out = {"x_encoded": x_next, "intermediate_steps": inter_steps}
        if return_intermediates:
            out.update({"intermediates": intermediates})
        return x_next, out

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        """`cldm/ddim_hacked.py:297-317`."""
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        timesteps = timesteps[:t_start]
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        x_dec = x_latent
        self._cache_key = None
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=index, use_original_steps=use_original_steps,
                                          unconditional_guidance_scale=unconditional_guidance_scale,
                                          unconditional_conditioning=unconditional_conditioning)
            if callback:
                callback(i)
        return x_dec
