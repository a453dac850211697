#!/bin/bash
# fp8 iteration: fp8 tests first, then the whole suite, bench fp16 and fp8
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2e}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python -m pytest tests/test_fp8_gpu.py -m gpu -q -x -s > $OUT/fp8.log 2>&1; rc=$?; tail -12 $OUT/fp8.log | cut -c1-300; grep "parity-fp8" $OUT/fp8.log
[ $rc -ge 124 ] && exit $rc
[ $rc -ne 0 ] && exit 1
[ "$2" == "quick" ] && exit 0
bash tools/gpu_session_b.sh $TAG
echo "=== bench --fp8 (N=2)"; timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fp8 --dump-profile $OUT/profile_fp8.json > $OUT/bench_fp8.json 2> $OUT/bench_fp8.err; tail -c 600 $OUT/bench_fp8.err; python -c "
import json; b=json.load(open('$OUT/bench_fp8.json')); print({k:b.get(k) for k in ('value','ms_per_step','ms_per_unet_step','dtype')})"
echo "=== bench --fp8 --batch 2 (N=4, configs[4] shape per GPU)"; timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --fp8 --batch 2 > $OUT/bench_fp8_b2.json 2> $OUT/bench_fp8_b2.err; python -c "
import json; b=json.load(open('$OUT/bench_fp8_b2.json')); print({k:b.get(k) for k in ('value','ms_per_step','ms_per_unet_step','dtype')})"
echo "=== bench --batch 2 fp16"; timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --batch 2 > $OUT/bench_b2.json 2> $OUT/bench_b2.err; python -c "
import json; b=json.load(open('$OUT/bench_b2.json')); print({k:b.get(k) for k in ('value','ms_per_step','ms_per_unet_step','dtype')})"
