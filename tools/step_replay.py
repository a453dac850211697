"""How long does one DDIM step's device work take when NOTHING but the step is enqueued?  (MI355X, 512x512, CFG pair)

    python tools/step_replay.py [reps]

Times `reps` back-to-back hipGraph replays of apply_model (hint block and context K / V cached), the same number of eager calls,
and the sampler's own loop (graph replay + input copies + CFG / DDIM update per step), with HIP events on the current stream.
The difference between the first and the last is what the per-step host work and the small torch kernels of the sampler cost."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                        # noqa: E402
from stablediffusioneo_amd import spec as S                        # noqa: E402
from stablediffusioneo_amd.cldm.cldm import ControlLDM             # noqa: E402
from stablediffusioneo_amd.cldm.ddim_hacked import DDIMSampler     # noqa: E402
from stablediffusioneo_amd.runtime import SdeoRuntime, HINT_CACHED, CONTEXT_CACHED   # noqa: E402
from tests.common import X_T_SEED, make_hint, randn                # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, device=dev)
rt.load_synthetic_device(0)
m = ControlLDM(rt)
h = w = 64
hint = make_hint(1, 512, 512).to(dev)
cond = {"c_concat": [hint], "c_crossattn": [randn((1, 77, 768), 1).to(dev)]}
unc = {"c_concat": [hint], "c_crossattn": [randn((1, 77, 768), 2).to(dev)]}
x_T = randn((1, 4, h, w), X_T_SEED).to(dev)
sampler = DDIMSampler(m)


def timed(fn, n):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n, host / n * 1e3


# the sampler's loop (also stages hint + context and builds the graph)
for _ in range(2):
    z, _ = sampler.sample(20, 1, (4, h, w), cond, verbose=False, eta=0.0, unconditional_guidance_scale=9.0,
                          unconditional_conditioning=unc, x_T=x_T)
ms, host = timed(lambda: sampler.sample(20, 1, (4, h, w), cond, verbose=False, eta=0.0, unconditional_guidance_scale=9.0,
                                        unconditional_conditioning=unc, x_T=x_T), 3)
print(f"sampler loop: {ms / 20:.3f} ms per step (host enqueue {host / 20:.3f} ms per step)")
x2 = torch.cat([x_T, x_T])
t2 = torch.full((2,), 500, device=dev, dtype=torch.long)
rt.configure(2, h, w)
rt.apply_model_graphed(x2, t2, m.control_scales, m.only_mid_control)
ms, host = timed(lambda: rt.apply_model_graphed(x2, t2, m.control_scales, m.only_mid_control), reps)
print(f"graph replay + 2 input copies, back to back: {ms:.3f} ms per step (host {host:.3f} ms)")
g = rt._graphs
ms, host = timed(lambda: g[0].replay(), reps)
print(f"graph replay only (one instance): {ms:.3f} ms per step (host {host:.3f} ms)")
eps = torch.empty((2, 4, h, w), device=dev)
ms, host = timed(lambda: rt.apply_model(x2, None, t2, None, m.control_scales, m.only_mid_control, HINT_CACHED | CONTEXT_CACHED, eps), reps)
print(f"eager, two streams: {ms:.3f} ms per step (host {host:.3f} ms)")

# host cost of ONE enqueue into an empty queue (is the step launch-bound on the host, or does the host only wait for the GPU?)
for name, fn in (("graph replay", lambda: g[0].replay()),
                 ("eager, two streams", lambda: rt.apply_model(x2, None, t2, None, m.control_scales, m.only_mid_control, HINT_CACHED | CONTEXT_CACHED, eps))):
    hs, ds = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2_ = time.perf_counter()
        hs.append((t1 - t0) * 1e3); ds.append((t2_ - t0) * 1e3)
    print(f"{name}: one call into an idle queue: host returns after {min(hs):.3f} ms, device done after {min(ds):.3f} ms")
