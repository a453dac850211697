"""Throughput with K independent batch-1 requests in flight on one GPU (one host thread, K handles, K streams,
DDIM steps interleaved round-robin so the GPU always has K independent kernel streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stablediffusioneo_amd import spec as S
from stablediffusioneo_amd.cldm.cldm import ControlLDM
from stablediffusioneo_amd.cldm.ddim_hacked import DDIMSampler
from stablediffusioneo_amd.runtime import SdeoRuntime
from tests.common import make_hint, randn

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
hint = make_hint(1, 512, 512).to(dev); cc = randn((1, 77, 768), 1).to(dev); cu = randn((1, 77, 768), 2).to(dev)
cond = {"c_concat": [hint], "c_crossattn": [cc]}; unc = {"c_concat": [hint], "c_crossattn": [cu]}
pipes = []
for k in range(K):
    rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, device=dev); rt.load_synthetic_device(0)
    m = ControlLDM(rt); s = DDIMSampler(m); s.make_schedule(20, verbose=False)
    pipes.append((m, s, torch.cuda.Stream()))

def run_round(seed):
    imgs = [randn((1, 4, 64, 64), seed + k).to(dev) for k in range(K)]
    ts = np.flip(pipes[0][1].ddim_timesteps)
    for (m, s, st) in pipes:
        s._cache_key = None
    for i, step in enumerate(ts):
        index = len(ts) - i - 1
        for k, (m, s, st) in enumerate(pipes):
            with torch.cuda.stream(st):
                t = torch.full((1,), int(step), device=dev, dtype=torch.long)
                imgs[k], _ = s.p_sample_ddim(imgs[k], cond, t, index=index, unconditional_guidance_scale=9.0,
                                             unconditional_conditioning=unc)
    for k, (m, s, st) in enumerate(pipes):
        with torch.cuda.stream(st):
            m.decode_first_stage_uint8(imgs[k])

run_round(0); torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(ROUNDS):
    run_round(100 * r)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"inflight K={K}: {K * ROUNDS / dt:.3f} images/s ({dt / ROUNDS * 1e3:.1f} ms per round of {K})")
