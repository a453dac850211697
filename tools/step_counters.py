"""Driver for the whole-step rocprofv3 counter passes (HBM bytes / MFMA-busy per DDIM step), eager and single-stream:

    rocprofv3 --pmc FETCH_SIZE --kernel-include-regex sdeo -d <dir> -o p -- python3 tools/step_counters.py [res] [vae]

Three sampler runs separated by a marker kernel (sdeo::silu_kernel on 64 elements, launched nowhere else): a warm-up of 2 DDIM
steps, then 2 steps, then 6 steps (ControlNet + UNet on the fused CFG pair + CFG / DDIM update; each run computes the hint
block and the context K / V once).  One DDIM step = (segment of 6 - segment of 2) / 4, which cancels the once-per-image work.
With `vae` a fourth segment holds one VAE decode.  NO hipGraph replay and NO side stream here (a whole-bench counter pass
under graph replay + the two-stream fork / join died inside rocprofv3's dispatch interception in round 1; see
profiles/README.md).  tools/step_counters_summary.py turns the rocpd databases into profiles/*_step_counters.json."""
import os
import sys

# SDEO_STEPCTR_GRAPH / SDEO_STEPCTR_OVERLAP = 1 switch ONE of the two back on (tools/gpu_session.sh pmc_graph_only / pmc_overlap_only:
# which of them the round-1 rocprofv3 crash needs)
os.environ["SDEO_GRAPH"] = os.environ.get("SDEO_STEPCTR_GRAPH", "0")
os.environ["SDEO_OVERLAP"] = os.environ.get("SDEO_STEPCTR_OVERLAP", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                        # noqa: E402
from stablediffusioneo_amd import spec as S                        # noqa: E402
from stablediffusioneo_amd.cldm.cldm import ControlLDM             # noqa: E402
from stablediffusioneo_amd.cldm.ddim_hacked import DDIMSampler     # noqa: E402
from stablediffusioneo_amd.runtime import SdeoRuntime              # noqa: E402
from tests.common import X_T_SEED, make_hint, randn                # noqa: E402

import ctypes as C                                                 # noqa: E402
from stablediffusioneo_amd import _lib                             # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with_vae = "vae" in sys.argv[2:]
dev = torch.device("cuda", 0)
rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, device=dev)
rt.load_synthetic_device(0)
m = ControlLDM(rt)
h = w = res // 8
hint = make_hint(1, res, res).to(dev)
cond = {"c_concat": [hint], "c_crossattn": [randn((1, 77, 768), 1).to(dev)]}
unc = {"c_concat": [hint], "c_crossattn": [randn((1, 77, 768), 2).to(dev)]}
x_T = randn((1, 4, h, w), X_T_SEED).to(dev)
lib = _lib.load()
mk_x = torch.zeros(64, dtype=torch.float16, device=dev)
mk_y = torch.zeros(64, dtype=torch.float16, device=dev)


def marker():
    lib.sdeo_debug_silu(C.c_void_p(mk_y.data_ptr()), C.c_void_p(mk_x.data_ptr()), C.c_int64(64), C.c_void_p(torch.cuda.current_stream().cuda_stream))


z = None
short = "short" in sys.argv[2:]
for steps in ((2, 4) if short else (2, 2, 6)):
    z, _ = DDIMSampler(m).sample(steps, 1, (4, h, w), cond, verbose=False, eta=0.0, unconditional_guidance_scale=9.0,
                                 unconditional_conditioning=unc, x_T=x_T)
    torch.cuda.synchronize()
    marker()
if with_vae:
    m.decode_first_stage_uint8(z)
    marker()
torch.cuda.synchronize()
print(f"STEP_COUNTERS res={res} segments=2,2,6 vae={int(with_vae)} finite={bool(torch.isfinite(z).all())}", flush=True)
