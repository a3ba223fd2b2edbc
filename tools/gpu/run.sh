#!/bin/bash
# One parameterised GPU-box runner (replaces the per-experiment r3_*.sh scripts).
#   usage: tools/gpu/run.sh <tag> <step> [<step> ...]        (from the repo root, inside gpurun)
# Every step logs to gpurun_out/<tag>/<step>.log / .err and a line in progress.log; a step that hits its timeout stops the
# plan (no further GPU step after a kill).  Steps:
#   test:<expr>      python -m pytest tests -m gpu -x -q -k '<expr>'        (test:all = the whole -m gpu suite)
#   bench            default bench line            bench_forced   the same under SPK_FORCE_REDUCER=1 (RCCL, one rank)
#   bench_ingest / bench_c4 / bench_c5 / bench_fast (no CPU legs)
#   smoke            __graft_entry__.smoke()
#   gpus2_gloo       `python bench.py --gpus 2` spawning its own two ranks, both on GPU 0 over gloo (control-flow rehearsal, not a measurement)
#   profile          tools/profile_bench.sh (kernel trace + FETCH / WRITE passes)      profile_sq   tools/profile_sq.sh
#   trace_forced     rocprofv3 --kernel-trace --stats of the forced-reducer bench (RCCL kernel names)
#   py:<script> [args are not supported: wrap them in a tools/ script]
set -o pipefail
TAG=$1; shift
D=gpurun_out/$TAG
mkdir -p $D
export TMPDIR=/tmp
R=$(pwd)
run() {
    local name=$1 to=$2; shift 2
    echo "=== $name $(date +%T)" | tee -a $D/progress.log
    timeout -k 10 "$to" "$@" > $D/$name.log 2> $D/$name.err
    local rc=$?
    echo "rc=$rc $name" | tee -a $D/progress.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $D/progress.log; exit 1; fi
    return 0
}
FAST="--no-cpu-baseline --no-eer --no-fp32-leg --no-f16-window"
for step in "$@"; do
    case "$step" in
        test:all)      run test_all 1100 python3 -m pytest tests -m gpu -x -q --durations=25 ;;
        test:*)        run "test_$(echo ${step#test:} | tr -c 'A-Za-z0-9\n' '_')" 900 python3 -m pytest tests -m gpu -x -q -s -k "${step#test:}" ;;
        bench)         run bench 500 python3 bench.py --steps 20 --warmup 5 ;;
        bench_fast)    run bench_fast 300 python3 bench.py --steps 20 --warmup 5 $FAST --no-extra ;;
        bench_forced)  SPK_FORCE_REDUCER=1 run bench_forced 300 python3 bench.py --steps 20 --warmup 5 $FAST --no-extra ;;
        bench_ingest)  run bench_ingest 300 python3 bench.py --steps 20 --warmup 5 $FAST --no-roofline --ingest --no-extra ;;
        bench_c4)      run bench_c4 400 python3 bench.py --config c4 --steps 16 --warmup 8 --no-cpu-baseline --no-eer ;;
        bench_c5)      run bench_c5 300 python3 bench.py --config c5 --steps 20 --warmup 5 ;;
        smoke)         run smoke 200 python3 __graft_entry__.py smoke ;;
        gpus2_gloo)    SPK_FORCE_DEVICE=0 SPK_DIST_BACKEND=gloo run gpus2_gloo 300 python3 bench.py --gpus 2 --batch 64 --steps 5 --warmup 2 $FAST --no-roofline --no-extra ;;
        profile)       run profile 600 bash tools/profile_bench.sh $D/prof ;;
        profile_sq)    run profile_sq 400 bash tools/profile_sq.sh $D/sq ;;
        profile_sq2)   run profile_sq2 500 bash tools/profile_sq2.sh $D/sq2 ;;
        bench_shapes)  SPK_LABEL_SHAPES=1 run bench_shapes 300 python3 bench.py --steps 10 --warmup 3 $FAST --no-extra ;;
        trace_forced)  mkdir -p $D/trace_forced
                       echo "=== trace_forced $(date +%T)" | tee -a $D/progress.log
                       ( cd /tmp && SPK_FORCE_REDUCER=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$D/trace_forced --output-format csv -- \
                           python3 $R/bench.py --steps 5 --warmup 2 $FAST --no-roofline --no-extra > $R/$D/trace_forced.log 2> $R/$D/trace_forced.err )
                       echo "rc=$? trace_forced" | tee -a $D/progress.log
                       find $D/trace_forced -name "*kernel_stats.csv" -exec cp {} $D/trace_forced_kernel_stats.csv \;
                       rm -rf $D/trace_forced ;;
        py:*)          run "py_$(basename ${step#py:} .py)" 900 python3 ${step#py:} ;;
        sh:*)          run "sh_$(basename ${step#sh:} .sh)" 900 bash ${step#sh:} $D ;;
        *)             echo "unknown step $step" | tee -a $D/progress.log; exit 2 ;;
    esac
done
rm -rf $D/prof/trace/*/*.db $D/prof/pmc_fetch $D/prof/pmc_write $D/sq/pmc_sq 2>/dev/null
du -sh $D; cat $D/progress.log
