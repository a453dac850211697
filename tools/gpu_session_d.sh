#!/bin/bash
# attention iteration: op tests, attention micro-bench (both kernels), then full tests + bench + kernel-trace timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2d}
OUT=$R/gpurun_out/$TAG
MODE0=$2
mkdir -p $OUT
cd $R
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention" > $OUT/quick.log 2>&1; rc=$?; tail -8 $OUT/quick.log | cut -c1-250
[ $rc -ge 124 ] && exit $rc
if [ $rc -eq 0 ]; then
  for m in "0 0" "1 0" "1 1"; do set -- $m; echo "SDEO_ATTN64=$1 SCHED=$2"; SDEO_ATTN64=$1 SDEO_ATTN64_SCHED=$2 timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep -E "T= 4096 Tk= 4096|T= 9216 Tk= 9216"; done
fi
[ "$MODE0" == "quick" ] && exit 0
bash tools/gpu_session_b.sh $TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --fast-weights > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cd $R
python tools/timeline.py $(find $OUT/kt -name "*.db" | head -1) 9000 > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
python tools/rocpd_summary.py stats $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
