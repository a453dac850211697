#!/bin/bash
# One gpurun call: GPU tests, bench (+ full per-shape profile), rocprofv3 kernel trace of the bench, whole-step counter passes.
# A step that is killed (rc >= 124) stops the session; a failing test (rc 1) does not.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2a}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
step() { echo "=== $1"; shift; "$@"; rc=$?; echo "=== rc=$rc"; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; return 0; }
run_pytest() { timeout -k 10 900 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log; grep "parity-full" $OUT/pytest.log | head -60; return $rc; }
run_bench() { timeout -k 10 600 python bench.py --steps 3 --warmup 1 --dump-profile $OUT/profile.json > $OUT/bench.json 2> $OUT/bench.err; rc=$?; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err; return $rc; }
step pytest run_pytest
step bench run_bench
cd /tmp && export TMPDIR=/tmp
run_kt() { timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err; rc=$?; tail -2 $OUT/kt.err; return $rc; }
step kernel-trace run_kt
pmc() { n=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-include-regex sdeo -d $OUT/pmc_$n -o p -- python3 $R/tools/step_counters.py 512 vae > $OUT/pmc_$n.out 2> $OUT/pmc_$n.err; rc=$?; tail -1 $OUT/pmc_$n.out; tail -2 $OUT/pmc_$n.err; return $rc; }
step pmc-fetch pmc fetch FETCH_SIZE
step pmc-write pmc write WRITE_SIZE
step pmc-mfma pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
cd $R
python tools/rocpd_summary.py stats $(find $OUT/kt -name "*.db" | head -1) $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
python tools/step_counters_summary.py $OUT/step_counters.json fetch=$(find $OUT/pmc_fetch -name "*.db" | head -1) write=$(find $OUT/pmc_write -name "*.db" | head -1) mfma=$(find $OUT/pmc_mfma -name "*.db" | head -1) 2>&1 | tail -12
# keep the merge-back small: the rocpd databases are large
find $OUT -name "*.db" -size +20M -delete
ls -la $OUT
