"""How much throughput do K independent batch-1 pipelines (K handles, K host threads, K streams) get on one GPU?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import spec as S
from stablediffusioneo_amd.cldm.cldm import ControlLDM
from stablediffusioneo_amd.cldm.ddim_hacked import DDIMSampler
from stablediffusioneo_amd.runtime import SdeoRuntime
from tests.common import make_hint, randn

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
IM = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
pipes = []
for k in range(K):
    rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, device=dev); rt.load_synthetic_device(0)
    m = ControlLDM(rt); pipes.append((m, DDIMSampler(m), torch.cuda.Stream()))
hint = make_hint(1, 512, 512).to(dev); cc = randn((1, 77, 768), 1).to(dev); cu = randn((1, 77, 768), 2).to(dev)
cond = {"c_concat": [hint], "c_crossattn": [cc]}; unc = {"c_concat": [hint], "c_crossattn": [cu]}

def work(k, n):
    m, s, st = pipes[k]
    with torch.cuda.stream(st):
        for i in range(n):
            z, _ = s.sample(20, 1, (4, 64, 64), cond, verbose=False, eta=0.0, unconditional_guidance_scale=9.0,
                            unconditional_conditioning=unc, x_T=randn((1, 4, 64, 64), 7 + i).to(dev))
            m.decode_first_stage_uint8(z)
        st.synchronize()

for k in range(K):
    work(k, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(k, IM)) for k in range(K)]
[t.start() for t in ths]; [t.join() for t in ths]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"K={K}: {K * IM / dt:.3f} images/s ({dt / IM * 1e3:.1f} ms per image per pipeline)")
