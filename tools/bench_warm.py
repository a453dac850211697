"""Per-image wall time of the bench's pipeline from a cold start (does the first timed image still pay one-off costs?):
    python tools/bench_warm.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import spec as S
from stablediffusioneo_amd.cldm.cldm import ControlLDM
from stablediffusioneo_amd.cldm.ddim_hacked import DDIMSampler
from stablediffusioneo_amd.runtime import SdeoRuntime
from tests.common import X_T_SEED, make_hint, randn
dev = torch.device("cuda", 0)
rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, device=dev); rt.load_synthetic_device(0)
m = ControlLDM(rt); sampler = DDIMSampler(m)
hint = make_hint(1, 512, 512).to(dev)
cond = {"c_concat": [hint], "c_crossattn": [randn((1, 77, 768), 1).to(dev)]}
unc = {"c_concat": [hint], "c_crossattn": [randn((1, 77, 768), 2).to(dev)]}
xs = [randn((1, 4, 64, 64), X_T_SEED + i).to(dev) for i in range(6)]
torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter()
    z, _ = sampler.sample(20, 1, (4, 64, 64), cond, verbose=False, eta=0.0, unconditional_guidance_scale=9.0, unconditional_conditioning=unc, x_T=xs[i])
    t1 = time.perf_counter()
    img = m.decode_first_stage_uint8(z)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"image {i}: host returns from sample() after {(t1 - t0) * 1e3:7.1f} ms, image done after {(t2 - t0) * 1e3:7.1f} ms", flush=True)
