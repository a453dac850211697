import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stablediffusioneo_amd import spec as S
from stablediffusioneo_amd.runtime import SdeoRuntime
from tests.common import make_inputs
g = np.load("tests/golden/tiny_nets.npz")
rt = SdeoRuntime(S.UNET_TINY, S.VAE_TINY); rt.load_synthetic(0)
for n, h, w, t in [(2, 16, 16, [801, 1]), (1, 8, 24, [401])]:
    tag = f"n{n}_{h}x{w}"
    rt.configure(n, h, w)
    x, ctx, hint = make_inputs(n, h, w, ctx_dim=S.UNET_TINY.context_dim)
    tt = torch.tensor(t, dtype=torch.long)
    def err(a, name):
        r = g[name]; a = a.float().cpu().numpy()
        return float(np.abs(a - r).max() / np.abs(r).max())
    e = rt.unet(x, tt, ctx, control=None)
    print(tag, "nocontrol", err(e, f"{tag}.eps_nocontrol"))
    e = rt.unet(x, tt, ctx, control=None)
    print(tag, "nocontrol again", err(e, f"{tag}.eps_nocontrol"))
    c = rt.controlnet(x, hint, tt, ctx)
    print(tag, "controls", max(err(ci, f"{tag}.control{i}") for i, ci in enumerate(c)))
    e = rt.unet(x, tt, ctx, control=[torch.tensor(g[f"{tag}.control{i}"]) for i in range(13)])
    print(tag, "unet golden controls", err(e, f"{tag}.eps"))
    e = rt.apply_model(x, hint, tt, ctx, scales=[1.0] * 13)
    print(tag, "apply_model", err(e, f"{tag}.eps"))
