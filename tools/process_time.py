"""End-to-end time of hackathon.process (the reference's entry point, `canny2image_torch.py:26-70`) on a 512x512 picture:
resize + Canny + control tensor, CLIP text encoder for prompt and negative prompt, 20 DDIM steps, VAE decode, image to the host.
    python tools/process_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stablediffusioneo_amd.canny2image import hackathon

h = hackathon()
h.initialize(weights="synthetic:0", config="sd15", text_encoder="clip")
rng = np.random.default_rng(0)
img = (rng.random((512, 512, 3)) * 255).astype(np.uint8)
args = dict(prompt="a bird", a_prompt="best quality, extremely detailed", n_prompt="longbody, lowres, bad anatomy", num_samples=1,
            image_resolution=512, ddim_steps=20, guess_mode=False, strength=1.0, scale=9.0, seed=2946901, eta=0.0,
            low_threshold=100, high_threshold=200)
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = h.process(img, **args)
    t1 = time.perf_counter()
    print(f"process() call {i}: {(t1 - t0) * 1e3:.1f} ms  -> {out[0].shape} {out[0].dtype}", flush=True)
# where the time goes in one more call (host timers around the stages, each ended by a synchronise)
import stablediffusioneo_amd.canny2image as c2i
m = h.model
def stage(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"  {name}: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True); return r
im = stage("resize_image + HWC3", lambda: c2i.resize_image(c2i.HWC3(img), 512))
ctrl = stage("canny + control tensor", lambda: h.apply_canny.control_hint(im, 100, 200).to(m.device)[None].contiguous())
c = stage("CLIP prompt", lambda: m.get_learned_conditioning(["a bird, best quality, extremely detailed"]))
u = stage("CLIP negative prompt", lambda: m.get_learned_conditioning(["longbody, lowres, bad anatomy"]))
z = stage("20 DDIM steps", lambda: h.ddim_sampler.sample(20, 1, (4, 64, 64), {"c_concat": [ctrl], "c_crossattn": [c]}, verbose=False, eta=0.0,
                                                            unconditional_guidance_scale=9.0, unconditional_conditioning={"c_concat": [ctrl], "c_crossattn": [u]})[0])
stage("VAE decode + image to host", lambda: m.decode_first_stage_uint8(z).cpu().numpy())
