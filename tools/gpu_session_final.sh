#!/bin/bash
# End-of-round evidence in one gpurun call: all GPU tests, bench lines (fp16 with the CPU baseline, fp8, batch 2), the rocprofv3 kernel
# trace of the bench command, the whole-step counter passes (eager, one stream) and the traffic of the dominant (kernel, shape).
# A step that is killed (rc >= 124) stops the session; a failing test (rc 1) does not.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2z}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
step() { echo "=== $1"; shift; "$@"; rc=$?; echo "=== rc=$rc"; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; return 0; }
run_pytest() { timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log; return $rc; }
run_bench() { timeout -k 10 600 python bench.py --steps 5 --warmup 1 --dump-profile $OUT/profile.json > $OUT/bench.json 2> $OUT/bench.err; rc=$?; head -c 400 $OUT/bench.json; echo; tail -2 $OUT/bench.err; return $rc; }
quick() { name=$1; shift; timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fast-weights "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; rc=$?; python -c "
import json; b=json.load(open('$OUT/bench_$name.json')); print('$name', b['value'], 'img/s', b.get('ms_per_unet_step'), 'ms/step')"; return $rc; }
step pytest run_pytest
step bench run_bench
step bench-fp8 quick fp8 --fp8 --dump-profile $OUT/profile_fp8.json
step bench-b2 quick b2 --batch 2 --no-roofline
step bench-fp8-b2 quick fp8_b2 --fp8 --batch 2 --no-roofline
step bench-768 quick 768 --res 768 --ddim-steps 50 --no-roofline
cd /tmp && export TMPDIR=/tmp
run_kt() { timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err; rc=$?; tail -2 $OUT/kt.err; return $rc; }
step kernel-trace run_kt
pmc() { n=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-include-regex sdeo -d $OUT/pmc_$n -o p -- python3 $R/tools/step_counters.py 512 vae > $OUT/pmc_$n.out 2> $OUT/pmc_$n.err; rc=$?; tail -1 $OUT/pmc_$n.out; tail -2 $OUT/pmc_$n.err; return $rc; }
step pmc-fetch pmc fetch FETCH_SIZE
step pmc-write pmc write WRITE_SIZE
step pmc-mfma pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
attn() { c=$1; timeout -k 10 200 rocprofv3 --pmc $c -d $OUT/attn_$c -o p -- python3 $R/tools/one_attn.py 4096 40 5 > /dev/null 2> $OUT/attn_$c.err; rc=$?; tail -1 $OUT/attn_$c.err; return $rc; }
step attn-fetch attn FETCH_SIZE
step attn-write attn WRITE_SIZE
cd $R
python tools/rocpd_summary.py stats $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
python tools/timeline.py $(find $OUT/kt -name "*.db" | head -1) 9200 > $OUT/timeline.txt 2>&1
python tools/step_counters_summary.py $OUT/step_counters.json fetch=$(find $OUT/pmc_fetch -name "*.db" | head -1) write=$(find $OUT/pmc_write -name "*.db" | head -1) mfma=$(find $OUT/pmc_mfma -name "*.db" | head -1) 2>&1 | tail -12
for c in FETCH_SIZE WRITE_SIZE; do python tools/rocpd_summary.py pmc $(find $OUT/attn_$c -name "*.db" | head -1) sdeo:: > $OUT/attn_$c.txt 2>&1; tail -3 $OUT/attn_$c.txt; done
# keep the merge-back small: the rocpd databases are large
find $OUT -name "*.db" -size +20M -delete
ls -la $OUT
