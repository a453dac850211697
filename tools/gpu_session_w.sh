#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2w
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > $OUT/pytest_attn.log 2>&1; rc=$?
tail -3 $OUT/pytest_attn.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|mismatch" $OUT/pytest_attn.log | head -20; exit $rc; }
timeout -k 10 900 python -m pytest tests/test_fullsize_golden_gpu.py tests/test_nets_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -s > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log; grep "vae" $OUT/pytest.log | head
[ $rc -ne 0 ] && { grep -n "Error\|assert" $OUT/pytest.log | head -20; exit $rc; }
python - <<'PY'
import torch, time
from stablediffusioneo_amd import spec as S
from stablediffusioneo_amd.runtime import SdeoRuntime
rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15); rt.load_synthetic_device(0); rt.configure(2, 64, 64)
z = torch.randn(1, 4, 64, 64, device="cuda")
for _ in range(3): rt.vae_decode(z, want_u8=True)
torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): rt.vae_decode(z, want_u8=True)
b.record(); torch.cuda.synchronize(); print("vae decode ms", a.elapsed_time(b) / 10)
PY
