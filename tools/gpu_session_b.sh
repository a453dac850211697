#!/bin/bash
# quick GPU iteration: selected tests first (fail fast), then the whole GPU suite, then the bench with the per-shape profile
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2b}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
step() { echo "=== $1"; shift; "$@"; rc=$?; echo "=== rc=$rc"; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; return $rc; }
quick() { timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "${QUICK_K:-attention or layernorm or gemm}" > $OUT/quick.log 2>&1; rc=$?; tail -15 $OUT/quick.log; return $rc; }
full() { timeout -k 10 900 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log | cut -c1-300; grep -E "^(FAILED|ERROR)" $OUT/pytest.log | head -20; grep "parity-full" $OUT/pytest.log | grep -E "eps \(apply|final latent|image \(abs" ; return $rc; }
bench() { timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-profile $OUT/profile.json > $OUT/bench.json 2> $OUT/bench.err; rc=$?; python - <<PY
import json
try:
    b=json.load(open("$OUT/bench.json"))
    print({k:b.get(k) for k in ("value","ms_per_step","ms_per_unet_step","mfma_frac_whole_image","output_check")})
    r=b["roofline"]; print(r["kernel"], r["achieved"], r["frac"], "launches/image", r["launches_per_image_all_kernels"])
    for k,v in r["families"].items(): print(" ", k, v)
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench.err").read()[-2000:])
PY
return $rc; }
step quick quick || exit 1
step full full
step bench bench
