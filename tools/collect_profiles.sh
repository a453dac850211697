#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace statistics of the benchmark command, then separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a
# pass on gfx950) on single-kernel drivers.  Results land under gpurun_out/prof/; summaries are copied into profiles/ by hand.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
echo "kernel-trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/attn_$c -o p -- python3 $R/tools/one_attn.py 4096 40 5 > /dev/null 2> $OUT/attn_$c.err; echo "attn $c rc=$?"
  rocprofv3 --pmc $c -d $OUT/halo_$c -o p -- python3 $R/tools/one_conv.py 8192 320 2880 3 13 1 5 > /dev/null 2> $OUT/halo_$c.err; echo "halo $c rc=$?"
done
find $OUT -name "*.csv" | head -40
