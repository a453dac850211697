#!/bin/bash
# selected test files (fail fast)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2g}
shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest "$@" -m gpu -q -x > $OUT/sel.log 2>&1; rc=$?; tail -25 $OUT/sel.log | cut -c1-300
exit $rc
