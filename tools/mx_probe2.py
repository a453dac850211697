import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops
from oracle import fp8_quant as Q
dev = "cuda"
torch.manual_seed(0)
M = N = K = 128
def dec(c): return c.view(torch.float8_e4m3fn).float()
def run(xq, xs, wq, ws): return ops.gemm_mx(xq.to(dev), xs.to(dev), wq.to(dev), ws.to(dev)).float().cpu()
s127 = torch.full((M, K // 32), 127, dtype=torch.uint8)
# random codes with small magnitudes (exp field <= 9) so sums stay exact in fp16 output? use values: codes in {0x30..0x47} and sign
def rc(hi):
    c = torch.randint(0x28, hi, (M, K), dtype=torch.uint8)
    c |= (torch.randint(0, 2, (M, K), dtype=torch.uint8) << 7)
    return c
for hi in (0x40, 0x58, 0x7F):
    xq, wq = rc(hi), rc(hi)
    y = run(xq, s127, wq, s127)
    ref = dec(xq) @ dec(wq).t()
    err = (y - ref).abs()
    bad = (err > 2e-3 * ref.abs().max()).nonzero()
    print(f"random codes < {hi:#x}: max err {float(err.max()):.4g} / scale {float(ref.abs().max()):.4g}; bad entries {bad.shape[0]}; first bad {bad[:6].tolist()}", flush=True)
x = torch.randn(M, K)
xq, xs = ops.quantize_mx(x.half().to(dev))
xq0, xs0, xd = Q.quantize_mx(x.half())
print("quantizer codes equal:", bool(torch.equal(xq.cpu(), xq0)), "scales equal:", bool(torch.equal(xs.cpu(), xs0)))
wq = torch.full((N, K), 0x38, dtype=torch.uint8)
y = run(xq.cpu(), xs.cpu(), wq, s127)
ref = xd.sum(1, keepdim=True).expand(M, N)
err = (y - ref).abs()
print("signed x (device pack) x ones: max err", float(err.max()), "rows with error:", (err[:, 0] > 1e-2).nonzero().flatten().tolist()[:20])
r = int((err[:, 0]).argmax())
print("row", r, "scales", xs0[r].tolist(), "y", float(y[r, 0]), "ref", float(ref[r, 0]))
for kb in range(4):
    print("   block", kb, "sum", float(xd[r, kb*32:(kb+1)*32].sum()), "amax", float(xd[r, kb*32:(kb+1)*32].abs().max()))
