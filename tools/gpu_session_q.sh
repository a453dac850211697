#!/bin/bash
# quick check after a change: network / sampler / golden tests, then bench A/B of an env switch on the same box
# usage: gpu_session_q.sh [VAR "v1 v2"]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2q
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_sampler_gpu.py tests/test_fullsize_golden_gpu.py tests/test_nets_gpu.py tests/test_pair_gpu.py tests/test_fp8_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert" $OUT/pytest.log | head -20; exit $rc; }
VAR=${1:-SDEO_NONE}; VALS=${2:-x}
for round in 1 2; do
 for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --fast-weights > $OUT/b.json 2> $OUT/b.err; rc=$?
  [ $rc -ge 124 ] && exit $rc
  [ $rc -ne 0 ] && tail -5 $OUT/b.err
  python -c "
import json; b=json.load(open('$OUT/b.json')); print('$VAR=$v', b['value'], 'img/s', b['ms_per_unet_step'], 'ms/step', b['ms_per_step'], 'ms/image')"
 done
done
