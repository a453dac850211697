"""Run the self-attention kernel repeatedly (for rocprofv3 --pmc): python tools/one_attn.py T d [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stablediffusioneo_amd import ops
T, d = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
c = 8 * d
q = torch.randn(2, T, c, device="cuda").half(); k = torch.randn(2, T, c, device="cuda").half(); vt = torch.randn(2, T, c, device="cuda").half()
for _ in range(reps):
    ops.attention(q, k, vt, 8)
torch.cuda.synchronize()
