#!/bin/bash
# tune the plans of shapes missing from the committed table (tools/tune_plans.py), install the merged table, then tests + bench
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2tune
mkdir -p $OUT
cd $R
timeout -k 10 1000 python tools/tune_plans.py > $OUT/tune.log 2>&1; rc=$?; tail -3 $OUT/tune.log; echo "tune rc=$rc"
[ $rc -ne 0 ] && exit $rc
cp $R/gpurun_out/tuned_plans_gfx950.json $R/stablediffusioneo_amd/tuned_plans_gfx950.json && cp $R/gpurun_out/tuned_plans_gfx950.json $OUT/
bash tools/gpu_session_q.sh SDEO_NONE "x"
