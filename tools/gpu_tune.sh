#!/bin/bash
# (re-)measure the conv / GEMM plan table on the MI355X (tools/tune_plans.py; SDEO_TUNED_PLANS=0 = every shape), install it for the
# rest of this call, then two bench lines.  Copy gpurun_out/<tag>/tuned_plans_gfx950.json over the committed table afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-tune}
mkdir -p $OUT
cd $R
timeout -k 10 1080 python tools/tune_plans.py > $OUT/tune.log 2>&1; rc=$?; tail -3 $OUT/tune.log; echo "tune rc=$rc"
[ $rc -ne 0 ] && exit $rc
cp $R/gpurun_out/tuned_plans_gfx950.json $R/stablediffusioneo_amd/tuned_plans_gfx950.json && cp $R/gpurun_out/tuned_plans_gfx950.json $OUT/
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --fast-weights > $OUT/b.json 2> $OUT/b.err
  python -c "
import json; b=json.load(open('$OUT/b.json')); print(b['value'], 'img/s', b['ms_per_unet_step'], 'ms/step', b['ms_per_step'], 'ms/image')"
done
