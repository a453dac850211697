#!/bin/bash
# quick check after a host-side change: sampler / engine / golden tests, then three bench lines on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2q
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_sampler_gpu.py tests/test_fullsize_golden_gpu.py tests/test_engine_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --fast-weights $BENCH_ARGS > $OUT/b.json 2> $OUT/b.err; rc=$?
  [ $rc -ge 124 ] && exit $rc
  python -c "
import json; b=json.load(open('$OUT/b.json')); print(b['value'], 'img/s', b['ms_per_unet_step'], 'ms/step', b['ms_per_step'], 'ms/image')"
done
