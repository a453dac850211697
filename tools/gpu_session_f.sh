#!/bin/bash
# tune (fp16 + fp8 shapes), install the table, then fp8 tests + full suite + benches
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2f}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
echo "=== tune"; timeout -k 10 900 python tools/tune_plans.py > $OUT/tune.log 2>&1; rc=$?; tail -4 $OUT/tune.log; echo "=== rc=$rc"
[ $rc -ge 124 ] && exit $rc
[ $rc -eq 0 ] && cp $R/gpurun_out/tuned_plans_gfx950.json $R/stablediffusioneo_amd/tuned_plans_gfx950.json && cp $R/gpurun_out/tuned_plans_gfx950.json $OUT/
bash tools/gpu_session_e.sh $TAG
