#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2t2
mkdir -p $OUT
cd $R
for cfg in "SDEO_GRAPH=1 SDEO_OVERLAP=1" "SDEO_GRAPH=1 SDEO_OVERLAP=0" "SDEO_GRAPH=0 SDEO_OVERLAP=1" "SDEO_GRAPH=0 SDEO_OVERLAP=0"; do
  env $cfg timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-weights > $OUT/b.json 2> $OUT/b.err; rc=$?
  [ $rc -ge 124 ] && exit $rc
  python -c "
import json; b=json.load(open('$OUT/b.json')); print('$cfg', b['value'], 'img/s', b['ms_per_unet_step'], 'ms/step')"
done
cd /tmp && export TMPDIR=/tmp
SDEO_GRAPH=0 timeout -k 10 500 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --fast-weights > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cd $R
echo "--- eager timeline"; python tools/timeline.py $(find $OUT/kt -name "*.db" | head -1) 9200 2>&1 | head -5
find $OUT -name "*.db" -delete
