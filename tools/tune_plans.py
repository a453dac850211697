"""Measure GEMM plans on the MI355X for the configurations the benchmarks / tests use and write
stablediffusioneo_amd/tuned_plans_gfx950.json (copy it back from gpurun_out/ and commit it).

    python tools/tune_plans.py            # on the GPU box; writes gpurun_out/tuned_plans_gfx950.json
    SDEO_TUNED_PLANS=0 python tools/tune_plans.py     # ignore the committed table and re-measure every shape
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusioneo_amd import _lib, spec as S                 # noqa: E402
from stablediffusioneo_amd.runtime import SdeoRuntime             # noqa: E402

os.environ.setdefault("SDEO_AUTOTUNE", "1")      # measure every untabled shape on this device
import threading, time                                            # noqa: E402
_t0 = time.time()
def _beat():                                                      # a configure can measure for minutes without printing
    while True:
        time.sleep(60)
        print(f"... tuning, {time.time() - _t0:.0f} s", flush=True)
threading.Thread(target=_beat, daemon=True).start()
lib = _lib.load()
start = len(_lib.dump_tuned_plans(lib))
rt = SdeoRuntime(S.UNET_SD15, S.VAE_SD15)
for (n, h, w) in [(2, 64, 64), (2, 32, 32), (2, 96, 96), (4, 64, 64), (1, 64, 64), (2, 32, 48)]:
    rt.configure(n, h, w)
    print(f"configured n={n} {h}x{w}: {len(_lib.dump_tuned_plans(lib))} plans", flush=True)
# fp8-weight plans (their own keys: epilogue-class slot + 4) of the configurations BASELINE configs[4] runs
del rt
rt8 = SdeoRuntime(S.UNET_SD15, S.VAE_SD15, weight_bits=8)
rt8.load_synthetic_device(0)
for (n, h, w) in [(2, 64, 64), (4, 64, 64)]:
    rt8.configure(n, h, w)
    print(f"configured fp8 n={n} {h}x{w}: {len(_lib.dump_tuned_plans(lib))} plans", flush=True)
rows = sorted(_lib.dump_tuned_plans(lib))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = os.path.join(ROOT, "gpurun_out", "tuned_plans_gfx950.json")
with open(out, "w") as f:
    f.write("[\n" + ",\n".join(json.dumps(r) for r in rows) + "\n]\n")
print(f"{len(rows)} plans ({len(rows) - start} new) -> {out}")
