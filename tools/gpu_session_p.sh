#!/bin/bash
# pair-launch validation: parity tests, then bench A/B (paired program vs two streams) on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2p
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_pair_gpu.py tests/test_nets_gpu.py tests/test_fullsize_golden_gpu.py tests/test_ops_gpu.py -x -q -m gpu -s > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
for cfg in "SDEO_PAIR=1" "SDEO_PAIR=0" "SDEO_PAIR=1" "SDEO_PAIR=0"; do
  env $cfg SDEO_PAIR_REPORT=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-weights > $OUT/b.json 2> $OUT/b.err; rc=$?
  [ $rc -ge 124 ] && exit $rc
  grep SDEO_PAIR: $OUT/b.err | tail -1
  python -c "
import json; b=json.load(open('$OUT/b.json')); print('$cfg', b['value'], 'img/s', b['ms_per_unet_step'], 'ms/step')"
done
for cfg in "SDEO_PAIR=1" "SDEO_PAIR=0"; do
  env $cfg timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-weights --fp8 > $OUT/b8.json 2> $OUT/b8.err; rc=$?
  [ $rc -ge 124 ] && exit $rc
  python -c "
import json; b=json.load(open('$OUT/b8.json')); print('fp8 $cfg', b['value'], 'img/s', b['ms_per_unet_step'], 'ms/step')"
done
