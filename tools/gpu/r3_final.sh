#!/bin/bash
# round-3 evidence run on ONE box: the bench lines first (default with CPU baseline + EER, --ingest, c4, c5, two-rank rehearsal on
# one GPU, smoke), THEN the rocprofv3 passes of the default bench command (kernel trace + FETCH / WRITE passes + SQ counters).
# Order matters: bench lines taken right after the counter passes came out 2-4 % slower than on a fresh box (52.4-52.6 against
# 50.4-51.8 ms), so the profiler runs last.  bench.py takes `traffic` from the committed profiles/pmc_traffic.json (same csrc).
set -o pipefail
D=gpurun_out/r3final
mkdir -p $D
run() {
    local name=$1 to=$2; shift 2
    echo "=== $name $(date +%T)" | tee -a $D/progress.log
    timeout -k 10 "$to" "$@" > $D/$name.log 2> $D/$name.err
    local rc=$?
    echo "rc=$rc $name" | tee -a $D/progress.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $D/progress.log; exit 1; fi
    return 0
}
run bench_final 400 python3 bench.py --steps 20 --warmup 5
run bench_ingest 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eer --no-roofline --no-f16-window --ingest
run bench_c4 400 python3 bench.py --config c4 --steps 16 --warmup 8 --no-cpu-baseline --no-eer
run bench_c5 300 python3 bench.py --config c5 --steps 20 --warmup 5
SPK_FORCE_DEVICE=0 SPK_DIST_BACKEND=gloo run bench_gpus2_rehearsal 300 python3 bench.py --gpus 2 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline --no-eer --no-roofline --no-f16-window
run smoke 200 python3 __graft_entry__.py smoke
run profile_bench 500 bash tools/profile_bench.sh $D/prof
run profile_sq 300 bash tools/profile_sq.sh $D/sq
rm -rf $D/prof/trace/*/*.db $D/prof/pmc_fetch $D/prof/pmc_write $D/sq/pmc_sq 2>/dev/null
du -sh $D; cat $D/progress.log
