import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops
for (c, hw) in [(320, 64), (960, 64), (1280, 16), (1280, 8)]:
    x = torch.randn(2, hw, hw, c, device="cuda").half(); g = torch.ones(c, device="cuda"); b = torch.zeros(c, device="cuda")
    for _ in range(5):
        ops.groupnorm_nhwc(x, g, b, 32, 1e-5, True)
    torch.cuda.synchronize()
