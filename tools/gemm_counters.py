"""Driver for the per-shape conv / GEMM counter passes: the five top shapes of a DDIM step / VAE decode,
each launched `reps` times at its production plan (the tuned table), one after the other, so that ONE rocprofv3 --pmc pass holds
all of them (dispatches are told apart by kernel name + grid):

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
              SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex sdeo -d <dir> -o p -- python3 tools/gemm_counters.py [reps]

tools/gemm_counters_summary.py turns the databases (one per counter set) into profiles/r03_gemm_counters.json."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from stablediffusioneo_amd import ops                  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1234)


def rn(*shape, s=1.0):
    return (torch.randn(*shape, device=dev, generator=g) * s).half()


# name, kind, (args)
SHAPES = [
    ("halo_M8192_N320_K2880", "conv", (2, 64, 64, 320, 320)),          # ResBlock conv3x3 320->320 @64x64 (halo kernel)
    ("geglu_M8192_N2560_K320", "geglu", (8192, 2560, 320)),            # ff.net.0.proj + GEGLU @64x64
    ("proj_M8192_N320_K320", "gemm", (8192, 320, 320)),                # K = C projection @64x64
    ("deep_M128_N1280_K11520", "conv", (2, 8, 8, 1280, 1280)),         # ResBlock conv3x3 1280->1280 @8x8 (weight stream, split-K)
    ("vae_M262144_N128_K1152", "conv", (1, 512, 512, 128, 128)),       # VAE ResnetBlock conv3x3 128->128 @512x512
    ("halo_M2048_N640_K5760", "conv", (2, 32, 32, 640, 640)),          # ResBlock conv3x3 640->640 @32x32
    ("proj_M2048_N640_K640", "gemm", (2048, 640, 640)),
]

fns = []
for name, kind, a in SHAPES:
    if kind == "conv":
        n, h, w, cin, cout = a
        x = rn(n, h, w, cin)
        wt = rn(cout, 3, 3, cin, s=0.02)
        fns.append((name, lambda x=x, wt=wt: ops.conv2d_nhwc(x, wt)))
    elif kind == "gemm":
        m, n, k = a
        x = rn(m, k)
        wt = rn(n, k, s=0.05)
        fns.append((name, lambda x=x, wt=wt: ops.gemm(x, wt)))
    else:
        m, n, k = a
        x = rn(m, k)
        wt = rn(n, k, s=0.05)
        fns.append((name, lambda x=x, wt=wt: ops.gemm_geglu(x, wt)))

for name, fn in fns:
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("ran", name, flush=True)
print("GEMM_COUNTERS done", flush=True)
