#!/bin/bash
# One gpurun call = one session: `tools/gpu_session.sh <tag> <step> [<step> ...]`, outputs under gpurun_out/<tag>/.
# A step that is killed (rc >= 124) stops the session (no further GPU step after a hang); a failing test (rc 1) does not.
#
# steps:  pytest            all -m gpu tests                              -> pytest.log
#         pytest:<expr>     -m gpu tests selected with -k <expr>          -> pytest_<n>.log
#         bench             bench.py --steps 5 --warmup 1 with roofline + cpu baseline + per-shape dump -> bench.json, profile.json
#         quick[:flags]     bench.py --steps 3 --warmup 1 --no-cpu-baseline --fast-weights --no-roofline [flags, comma separated]
#         ab:<ENV=V>        the quick bench twice, without and with the environment setting (same box A/B)
#         profile[:flags]   per-(kernel, shape) table of one eagerly profiled image (in-library HIP events) -> profile_<n>.json
#         kt                rocprofv3 --kernel-trace --stats of the bench command -> kernel_stats.csv, timeline.txt
#         gemm_counters     per-shape SQ / FETCH / WRITE counter passes (tools/gemm_counters.py) -> gemm_counters.json
#         attn_counters     SQ counter passes on the T = 4096, d = 40 self-attention (tools/one_attn.py) -> attn_counters.json
#         step_counters     whole-step counter passes, eager + one stream (tools/step_counters.py) -> step_counters.json
#         pmc_graph_only / pmc_overlap_only   one --pmc pass of a 2-step sampler run with ONLY graph replay / ONLY the side stream on
#         py:<script>[,args] python <script> args                          -> py_<n>.log
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r3}
shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
N=0
step() { echo "=== $1"; shift; "$@"; rc=$?; echo "=== rc=$rc"; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi; return 0; }
s_pytest() { timeout -k 10 1100 python -m pytest tests -m gpu -q -s ${1:+-k "$1"} > $OUT/pytest${2}.log 2>&1; rc=$?; tail -5 $OUT/pytest${2}.log; return $rc; }
s_bench() { timeout -k 10 700 python bench.py --steps 5 --warmup 1 --dump-profile $OUT/profile.json > $OUT/bench.json 2> $OUT/bench.err; rc=$?; head -c 600 $OUT/bench.json; echo; tail -2 $OUT/bench.err; return $rc; }
s_quick() { name=$1; shift; timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fast-weights --no-roofline "$@" > $OUT/quick_$name.json 2> $OUT/quick_$name.err; rc=$?
  python -c "
import json; b=json.load(open('$OUT/quick_$name.json')); print('quick $name', b['value'], 'img/s', b.get('ms_per_unet_step'), 'ms/step')"; return $rc; }
s_profile() { timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fast-weights --dump-profile $OUT/profile_$1.json "${@:2}" > $OUT/profile_bench_$1.json 2> $OUT/profile_$1.err; rc=$?
  python -c "
import json; d=json.load(open('$OUT/profile_$1.json')); tot=sum(x['total_ms'] for x in d); n=sum(x['launches'] for x in d)
print('profile: %.1f ms eager per image, %d launches' % (tot, n))
for x in d[:40]: print('%-64s n=%5d ms=%7.2f us=%6.1f' % (x['kernel'][:64], x['launches'], x['total_ms'], 1e3*x['total_ms']/x['launches']))"; return $rc; }
s_kt() { cd /tmp; export TMPDIR=/tmp; timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err; rc=$?; cd $R
  tail -2 $OUT/kt.err; db=$(find $OUT/kt -name "*.db" | head -1)
  python tools/rocpd_summary.py stats $db $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1; python tools/timeline.py $db 9200 > $OUT/timeline.txt 2>&1
  find $OUT/kt -name "*.db" -size +20M -delete; return $rc; }
pmc_pass() { n=$1; script=$2; shift 2; cd /tmp; export TMPDIR=/tmp; timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-include-regex sdeo -d $OUT/pmc_$n -o p -- python3 $R/$script > $OUT/pmc_$n.out 2> $OUT/pmc_$n.err; rc=$?; cd $R; tail -1 $OUT/pmc_$n.out; tail -2 $OUT/pmc_$n.err; return $rc; }
s_gemm_counters() {
  pmc_pass gc_sq tools/gemm_counters.py SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES || return $?
  pmc_pass gc_sq2 tools/gemm_counters.py SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS || return $?
  pmc_pass gc_fetch tools/gemm_counters.py FETCH_SIZE || return $?
  pmc_pass gc_write tools/gemm_counters.py WRITE_SIZE || return $?
  python tools/gemm_counters_summary.py $OUT/gemm_counters.json $(find $OUT/pmc_gc_* -name "*.db") > $OUT/gemm_counters.txt 2>&1; cat $OUT/gemm_counters.txt
  find $OUT/pmc_gc_* -name "*.db" -size +20M -delete; return 0; }
s_attn_counters() {
  pmc_pass ac_sq "tools/one_attn.py 4096 40 6" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES || return $?
  pmc_pass ac_sq2 "tools/one_attn.py 4096 40 6" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS || return $?
  pmc_pass ac_sq3 "tools/one_attn.py 4096 40 6" SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL || return $?
  python tools/gemm_counters_summary.py $OUT/attn_counters.json $(find $OUT/pmc_ac_* -name "*.db") > $OUT/attn_counters.txt 2>&1; cat $OUT/attn_counters.txt
  find $OUT/pmc_ac_* -name "*.db" -size +20M -delete; return 0; }
s_step_counters() {
  pmc_pass fetch "tools/step_counters.py 512 vae" FETCH_SIZE || return $?
  pmc_pass write "tools/step_counters.py 512 vae" WRITE_SIZE || return $?
  pmc_pass mfma "tools/step_counters.py 512 vae" SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE || return $?
  python tools/step_counters_summary.py $OUT/step_counters.json fetch=$(find $OUT/pmc_fetch -name "*.db" | head -1) write=$(find $OUT/pmc_write -name "*.db" | head -1) mfma=$(find $OUT/pmc_mfma -name "*.db" | head -1) 2>&1 | tail -14
  find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma -name "*.db" -size +20M -delete; return 0; }
# the round-1 whole-bench --pmc crash (rocprofv3 died inside its dispatch interception under graph replay + the two-stream fork /
# join): ONE pass each with only one of the two switched on, stderr kept; never retried
s_pmc_mode() { mode=$1; cd /tmp; export TMPDIR=/tmp
  if [ $mode = graph_only ]; then export SDEO_STEPCTR_GRAPH=1 SDEO_STEPCTR_OVERLAP=0; elif [ $mode = both ]; then export SDEO_STEPCTR_GRAPH=1 SDEO_STEPCTR_OVERLAP=1; else export SDEO_STEPCTR_GRAPH=0 SDEO_STEPCTR_OVERLAP=1; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES --kernel-include-regex sdeo -d $OUT/pmc_$mode -o p -- python3 $R/tools/step_counters.py 512 short > $OUT/pmc_$mode.out 2> $OUT/pmc_$mode.err; rc=$?
  unset SDEO_STEPCTR_GRAPH SDEO_STEPCTR_OVERLAP; cd $R; echo "pmc $mode rc=$rc"; tail -1 $OUT/pmc_$mode.out; tail -3 $OUT/pmc_$mode.err
  find $OUT/pmc_$mode -name "*.db" -size +5M -delete; [ $rc -ge 124 ] && return $rc; return 0; }
for st in "$@"; do
  N=$((N + 1))
  case $st in
    pytest) step pytest s_pytest "" "" ;;
    pytest:*) step "$st" s_pytest "$(echo "${st#pytest:}" | tr "," " ")" _$N ;;
    bench) step bench s_bench ;;
    quick) step quick s_quick $N ;;
    quick:*) step "$st" s_quick $N $(echo "${st#quick:}" | tr ',' ' ') ;;
    ab:*) kv=${st#ab:}; step "ab off" s_quick ${N}_off; step "ab $kv" env $kv bash -c "$(declare -f s_quick); OUT=$OUT; s_quick ${N}_on" ;;
    profile) step profile s_profile $N ;;
    profile:*) step "$st" s_profile $N $(echo "${st#profile:}" | tr ',' ' ') ;;
    kt) step kt s_kt ;;
    gemm_counters) step gemm_counters s_gemm_counters ;;
    step_counters) step step_counters s_step_counters ;;
    attn_counters) step attn_counters s_attn_counters ;;
    pmc_graph_only) step pmc_graph_only s_pmc_mode graph_only ;;
    pmc_overlap_only) step pmc_overlap_only s_pmc_mode overlap_only ;;
    pmc_both) step pmc_both s_pmc_mode both ;;
    benchx:*) a=${st#benchx:}; nm=$(echo $a | tr -d ',-' | cut -c1-30); step "$st" bash -c "timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $(echo $a | tr ',' ' ') > $OUT/benchx_$nm.json 2> $OUT/benchx_$nm.err; rc=\$?; head -c 900 $OUT/benchx_$nm.json; echo; exit \$rc" ;;
    py:*) a=${st#py:}; step "$st" bash -c "timeout -k 10 600 python $(echo $a | tr ',' ' ') > $OUT/py_$N.log 2>&1; rc=\$?; tail -25 $OUT/py_$N.log; exit \$rc" ;;
    *) echo "unknown step $st" ;;
  esac
done
ls $OUT | head -60
