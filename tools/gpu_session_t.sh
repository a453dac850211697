#!/bin/bash
# kernel trace of the bench under graph replay + timeline statistics
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2t}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --fast-weights > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
echo "rc=$?"
cd $R
python tools/timeline.py $(find $OUT/kt -name "*.db" | head -1) 9200 > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
python tools/rocpd_summary.py stats $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
find $OUT -name "*.db" -delete
