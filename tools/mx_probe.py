import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops
dev = "cuda"
M, N, K = 128, 128, 128
ONE = 0x38
def run(xq, xs, wq, ws):
    return ops.gemm_mx(xq.to(dev), xs.to(dev), wq.to(dev), ws.to(dev)).float().cpu()
xq = torch.full((M, K), ONE, dtype=torch.uint8); wq = torch.full((N, K), ONE, dtype=torch.uint8)
xs = torch.full((M, K // 32), 127, dtype=torch.uint8); ws = torch.full((N, K // 32), 127, dtype=torch.uint8)
m_ = torch.arange(M)[:, None]; kb = torch.arange(K // 32)[None, :]
xs6 = (127 + (m_ + kb) % 3).to(torch.uint8)
y = run(xq, xs6, wq, ws)
exp = (32.0 * (2.0 ** ((m_ + kb) % 3).float()).sum(1))
print("T6 x scale by (m+kb)%3: got col0 rows 0..7", y[:8, 0].tolist(), "expect", exp[:8].tolist(), "all match:", bool((y == exp[:, None]).all()))
ws7 = (127 + (m_ + kb) % 3).to(torch.uint8)
y = run(xq, xs, wq, ws7)
print("T7 w scale by (n+kb)%3: got row0 cols 0..7", y[0, :8].tolist(), "expect", exp[:8].tolist(), "all match:", bool((y == exp[None, :]).all()))
# negative codes
xq8 = xq.clone(); xq8[:, ::2] = ONE | 0x80
y = run(xq8, xs, wq, ws)
print("T8 alternating sign x: expect 0:", y.unique().tolist())
xq9 = xq.clone(); xq9[:, :64] = ONE | 0x80
y = run(xq9, xs, wq, ws)
print("T9 first half negative: expect 0:", y.unique().tolist())
# rows pattern: x row m has value code 0x38 + 8*(m%4)
xq10 = (0x38 + 8 * (torch.arange(M) % 4))[:, None].expand(M, K).contiguous().to(torch.uint8)
y = run(xq10, xs, wq, ws)
print("T10 x row value 2^(m%4): col0 rows 0..7:", y[:8, 0].tolist(), " rows 16..19", y[16:20, 0].tolist(), "rows 64..67", y[64:68,0].tolist())
# both scales vary
y = run(xq, xs6, wq, ws7)
exp2 = 32.0 * ((2.0 ** ((m_ + kb) % 3).float())[:, None, :] * (2.0 ** ((m_ + kb) % 3).float())[None, :, :]).sum(2)
print("T11 both vary: match:", bool((y == exp2).all()), " y[1,:4]", y[1, :4].tolist(), "exp", exp2[1, :4].tolist())
