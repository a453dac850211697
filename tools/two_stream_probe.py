"""Probe (measurement only): one fused N=2 CFG pass per DDIM step vs the cond / uncond halves as two independent N=1 passes
on two HIP streams (two handles, each with its own weight replica, arena and hipGraph).  Prints ms per DDIM step of each
arrangement at 512x512 (apply_model only, hint/context cached)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import spec as S
from stablediffusioneo_amd.runtime import SdeoRuntime, HINT_CACHED, CONTEXT_CACHED
from tests.common import make_hint, randn

dev = torch.device("cuda", 0)
h = w = 64
hint = make_hint(1, 512, 512).to(dev)
ctx = [randn((1, 77, 768), 1).to(dev), randn((1, 77, 768), 2).to(dev)]
x = randn((1, 4, h, w), 7).to(dev)
t = torch.tensor([501], dtype=torch.long, device=dev)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


# fused pair
rt2 = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, device=dev)
rt2.load_synthetic_device(0)
rt2.configure(2, h, w)
rt2.apply_model(torch.cat([x, x]), torch.cat([hint, hint]), torch.cat([t, t]), torch.cat(ctx), None, False, 0)
ms_fused = timed(lambda: rt2.apply_model_graphed(torch.cat([x, x]), torch.cat([t, t])))
print(f"fused N=2, one handle, 2 streams inside (ControlNet || UNet encoder): {ms_fused:.3f} ms / step", flush=True)

# two independent halves
rts, streams = [], [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for i in range(2):
    r = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, device=dev)
    r.load_synthetic_device(0)
    r.configure(1, h, w)
    with torch.cuda.stream(streams[i]):
        r.apply_model(x, hint, t, ctx[i], None, False, 0)
        r.apply_model_graphed(x, t)          # capture on this stream
    rts.append(r)
torch.cuda.synchronize()


def halves():
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            rts[i].apply_model_graphed(x, t)


ms_split = timed(halves)
print(f"cond / uncond as two N=1 handles on two streams (4 streams in flight): {ms_split:.3f} ms / step", flush=True)
ms_one = timed(lambda: rts[0].apply_model_graphed(x, t))
print(f"one N=1 half alone: {ms_one:.3f} ms / step", flush=True)
