"""Per-shape device-time breakdown of one image of the benchmark workload (GPU box only).

    SDEO_PROFILE_DETAIL=1 python tools/shape_profile.py [--res 512] [--ddim-steps 20] [--top 60]

Uses the in-library HIP-event profiler (sdeo_profile_begin/end; eager launches, one stream) with the shape tag of
every launch appended to the kernel key, so each row is one (kernel, problem shape): launches, total ms per image,
us per launch, TFLOP/s or GB/s.  Rows are sorted by total time."""
import argparse
import os
import sys

os.environ.setdefault("SDEO_PROFILE_DETAIL", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from stablediffusioneo_amd import spec as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--top", type=int, default=80)
    a = ap.parse_args()
    import stablediffusioneo_amd.cldm.ddim_hacked as dh
    from stablediffusioneo_amd.cldm.cldm import ControlLDM
    from stablediffusioneo_amd.runtime import SdeoRuntime
    from tests.common import X_T_SEED, make_hint, randn
    dev = torch.device("cuda", 0)
    rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, device=dev)
    rt.load_synthetic_device(0)
    model = ControlLDM(rt)
    sampler = dh.DDIMSampler(model)
    h = w = a.res // 8
    hint = make_hint(1, a.res, a.res).to(dev)
    ctx_c = randn((1, 77, 768), 1).to(dev)
    ctx_u = randn((1, 77, 768), 2).to(dev)
    cond = {"c_concat": [hint], "c_crossattn": [ctx_c]}
    unc = {"c_concat": [hint], "c_crossattn": [ctx_u]}

    def loop():
        x_T = randn((1, 4, h, w), X_T_SEED).to(dev)
        z, _ = sampler.sample(a.ddim_steps, 1, (4, h, w), cond, verbose=False, eta=0.0, unconditional_guidance_scale=9.0,
                              unconditional_conditioning=unc, x_T=x_T)
        return z

    z = loop()
    model.decode_first_stage_uint8(z)
    torch.cuda.synchronize()
    dh.USE_GRAPH = False
    for name, fn in (("ddim loop", loop), ("vae decode", lambda: model.decode_first_stage_uint8(z))):
        rt.profile_begin()
        fn()
        prof = rt.profile_end()
        tot = sum(k["total_ms"] for k in prof)
        print(f"== {name}: {tot:.2f} ms of launches ({sum(k['launches'] for k in prof)} launches)")
        for k in sorted(prof, key=lambda k: -k["total_ms"])[:a.top]:
            us = k["total_ms"] * 1e3 / k["launches"]
            rate = f"{k['flops'] / k['total_ms'] / 1e9:7.0f} TF" if k["flops"] > 0 else (
                f"{k['bytes'] / k['total_ms'] / 1e6:7.0f} GB/s" if k["bytes"] > 0 else "")
            print(f"{k['total_ms']:8.2f} ms {100 * k['total_ms'] / tot:5.1f}%  x{k['launches']:5d} {us:8.1f} us  {rate:>12s}  {k['kernel']}")


if __name__ == "__main__":
    main()
