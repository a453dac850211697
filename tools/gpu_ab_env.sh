#!/bin/bash
# A/B of an environment switch on ONE box: bench (graph replay, fast weights) once per value, interleaved twice
# usage: gpu_ab_env.sh VAR "v1 v2 ..." [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; shift 2
cd $R
mkdir -p gpurun_out/ab
for round in 1 2; do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-weights "$@" > gpurun_out/ab/$VAR.$v.$round.json 2> gpurun_out/ab/$VAR.$v.$round.err
    rc=$?; [ $rc -ge 124 ] && exit $rc
    python - <<PY
import json
try:
    b=json.load(open("gpurun_out/ab/$VAR.$v.$round.json")); print("$VAR=$v round $round:", b["value"], "img/s", b["ms_per_unet_step"], "ms/step")
except Exception as e: print("$VAR=$v failed", e)
PY
  done
done
