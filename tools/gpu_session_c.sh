#!/bin/bash
# tune the plans of shapes missing from the table, install the merged table for the rest of the session, then tests + bench
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2c}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
echo "=== tune"; timeout -k 10 900 python tools/tune_plans.py > $OUT/tune.log 2>&1; rc=$?; tail -4 $OUT/tune.log; echo "=== rc=$rc"
[ $rc -ge 124 ] && exit $rc
[ $rc -eq 0 ] && cp $R/gpurun_out/tuned_plans_gfx950.json $R/stablediffusioneo_amd/tuned_plans_gfx950.json && cp $R/gpurun_out/tuned_plans_gfx950.json $OUT/
bash tools/gpu_session_b.sh $TAG
