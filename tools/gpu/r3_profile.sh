#!/bin/bash
# rocprofv3 evidence of the default bench command: kernel trace + stats, FETCH / WRITE passes, SQ counters
set -o pipefail
D=gpurun_out/r3p
mkdir -p $D
echo "=== trace+pmc $(date +%T)" | tee -a $D/progress.log
timeout -k 10 700 bash tools/profile_bench.sh $D/prof > $D/profile_bench.log 2>&1; echo "rc=$? profile_bench" | tee -a $D/progress.log
echo "=== sq $(date +%T)" | tee -a $D/progress.log
timeout -k 10 400 bash tools/profile_sq.sh $D/sq > $D/profile_sq.log 2>&1; echo "rc=$? profile_sq" | tee -a $D/progress.log
rm -rf $D/prof/trace/*/*.db $D/prof/pmc_fetch/*/*.db 2>/dev/null
du -sh $D
cat $D/progress.log
